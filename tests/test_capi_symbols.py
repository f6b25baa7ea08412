"""The C-ABI library loads without a GPU and exports every symbol include/coulombgas.h declares; the package has no
CPU fallback and never touches oracle/ or tests/host_emul."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "coulombgas.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(cg_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported_and_bound():
    from coulombgas_amd import _lib
    from coulombgas_amd.build import build_hip
    build_hip()
    names = declared_symbols()
    assert len(names) >= 30
    L = C.CDLL(_lib.LIB_PATH)
    for nm in names:
        assert hasattr(L, nm), "libcoulombgas_hip.so does not export %s" % nm
        assert nm in _lib.PROTOTYPES, "no ctypes prototype for %s" % nm
    assert set(_lib.PROTOTYPES) <= set(names), set(_lib.PROTOTYPES) - set(names)
    _lib.lib()


def test_fails_loudly_without_gpu():
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present")
    import coulombgas_amd as cg
    from coulombgas_amd._lib import CoulombGasError
    with pytest.raises(CoulombGasError) as ei:
        cg.Engine(13, 2, 2, 16, 16, 6.39)
    assert ei.value.code == -2 and "HIP device" in str(ei.value)
    with pytest.raises(CoulombGasError):
        cg.FermiNet(2, 16, 16, 6.39).apply({}, None, __import__("numpy").zeros((13, 2)))


def test_missing_library_is_an_import_error(tmp_path, monkeypatch):
    from coulombgas_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError):
        _lib.lib()


def test_product_never_imports_oracle_or_emulation():
    pkg = os.path.join(ROOT, "coulombgas_amd")
    imp = re.compile(r"^\s*(from|import)\s+(oracle|tests)\b", re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if not f.endswith(".py"):
                continue
            src = open(os.path.join(dirpath, f)).read()
            assert not imp.search(src), "%s imports test infrastructure" % f
            if f not in ("_lib.py", "build.py"):
                assert "CDLL(" not in src, "%s loads a shared library itself" % f
    src = open(os.path.join(pkg, "_lib.py")).read()
    assert src.count("CDLL(") == 1 and "libcoulombgas_hip.so" in src
    # importing the package must not pull torch / the oracle in
    code = ("import sys; import coulombgas_amd; "
            "assert 'torch' not in sys.modules and not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules)")
    subprocess.check_call([sys.executable, "-c", code], cwd=ROOT)
