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


def test_production_kernels_carry_no_register_spill_scratch(tmp_path):
    """The kernels of the shipped configuration (d = 2, hs = ht = 16) in the built code objects: `.private_segment_fixed_size`
    <= 64 B per lane -- the bound the round-2 review set for the derivative kernels after 972 / 1172 B.  (Read from the library's
    gfx950 code-object metadata: no GPU needed.  A 208-byte pivot array slipped into all five derivative kernels once between two
    profile collections; this is the guard.)"""
    import shutil
    llvm = "/opt/rocm/lib/llvm/bin"
    if not (os.path.exists(os.path.join(llvm, "llvm-objdump")) and os.path.exists(os.path.join(llvm, "llvm-readelf"))):
        pytest.skip("ROCm's llvm-objdump / llvm-readelf not found")
    from coulombgas_amd import _lib
    from coulombgas_amd.build import build_hip
    build_hip()
    so = str(tmp_path / "lib.so")
    shutil.copy(_lib.LIB_PATH, so)
    subprocess.check_call([os.path.join(llvm, "llvm-objdump"), "--offloading", so], cwd=str(tmp_path), stdout=subprocess.DEVNULL)
    seen = {}
    for f in sorted(os.listdir(str(tmp_path))):
        if "amdgcn" not in f:
            continue
        notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", str(tmp_path / f)], capture_output=True, text=True).stdout
        name = None
        for line in notes.splitlines():
            m = re.match(r"\s+\.name:\s+(\S+)", line)
            if m:
                name = m.group(1)
            m = re.match(r"\s+\.private_segment_fixed_size:\s+(\d+)", line)
            if m and name:
                seen[name] = int(m.group(1))
    want = ("k_grad_lap2ILi2ELi16ELi16E", "k_scoresILi2ELi16ELi16E", "k_param_vjpILi2ELi16ELi16E", "k_mcmcILi2ELi16ELi16ELi64ELi13E",
            "k_mcmcILi2ELi16ELi16ELi256ELi29E", "k_mcmcILi2ELi16ELi16ELi512ELi49E", "k_mcmcILi2ELi16ELi16ELi512ELi57E", "k_chol_block", "k_chol_trsm", "k_chol_xinv", "k_chol_update", "k_fisher")
    for w in want:
        hits = {k: v for k, v in seen.items() if w in k}
        assert hits, "kernel %s not found in the library" % w
        for k, v in hits.items():
            # the one deliberate exception (round 4): k_param_vjp<..., 512, WPE = 4>, the instantiation whose registers are capped at 128 so
            # that two workgroups share a CU at N > 64 -- 224 B of spill scratch buy 20 % of the kernel (csrc/cg_k_derivs.inc)
            bound = 256 if ("k_param_vjp" in k and "ELi512ELi4EE" in k) else 64
            assert v <= bound, "%s: %d B of scratch per lane" % (k, v)
