"""Builds the in-tree native libraries (no JIT cache: the built .so travels with the source tree).

  coulombgas_amd/lib/libcoulombgas_hip.so   gfx950 kernels + C-ABI (include/coulombgas.h)   -- the product
  oracle/_build/libcg_oracle.so             plain-C CPU restatement (test/bench checker)    -- oracle/
  tests/host_emul/libcg_emul.so             device code compiled for the host, 1-thread shim -- tests only
"""
import os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "coulombgas_amd", "csrc")
LIBDIR = os.path.join(ROOT, "coulombgas_amd", "lib")
HIP_LIB = os.path.join(LIBDIR, "libcoulombgas_hip.so")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def hip_sources():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))] + [os.path.join(ROOT, "include", "coulombgas.h")]


HIP_UNITS = ("cg_k_sampler_a.hip", "cg_k_sampler_b.hip", "cg_k_derivs_a.hip", "cg_k_derivs_b.hip", "cg_k_big.hip", "cg_k_van.hip", "cg_hip.hip", "cg_k_generic.hip")
HIP_FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC"]


def _compile_units(objdir, flags, force=False):
    """hipcc -c of every translation unit, in parallel (one process per unit); returns the object files."""
    os.makedirs(objdir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hdrs = [f for f in hip_sources() if not f.endswith(".hip")]
    jobs, objs = [], []
    for u in HIP_UNITS:
        src, obj = os.path.join(CSRC, u), os.path.join(objdir, u[:-4] + ".o")
        objs.append(obj)
        if force or _newer(obj, hdrs + [src]):
            cmd = [hipcc] + HIP_FLAGS + list(flags) + ["-c", "-o", obj, src]
            print("+", " ".join(cmd), flush=True)
            jobs.append((cmd, subprocess.Popen(cmd)))
    failed = None
    for cmd, p in jobs:                                   # wait for every job (no orphan still writing .o files), then report
        if p.wait() != 0 and failed is None:
            failed = (p.returncode, cmd)
    if failed is not None:
        raise subprocess.CalledProcessError(*failed)
    return objs


def build_hip(force=False):
    os.makedirs(LIBDIR, exist_ok=True)
    if force or _newer(HIP_LIB, hip_sources()):
        objs = _compile_units(os.path.join(ROOT, "build", "hip"), [], force)
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", HIP_LIB] + objs + ["-ldl"])
    return HIP_LIB


def build_diag(name, flags):
    """Diagnostic A/B builds of the same ABI (tools/): lib/diag/lib<name>.so, selected with COULOMBGAS_HIP_LIB."""
    out = os.path.join(LIBDIR, "diag", "lib%s.so" % name)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    objs = _compile_units(os.path.join(ROOT, "build", "diag_" + name), flags, True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-ldl"])
    return out


SAN_FLAGS = ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]


def sanitize_requested():
    """CG_SANITIZE=1: the CPU-side native libraries (host emulation of the device headers, C oracle) are the AddressSanitizer +
    UndefinedBehaviorSanitizer builds (tools/sanitize_cpu.sh runs the CPU tests against them; GPU sanitizers are not available)."""
    return os.environ.get("CG_SANITIZE") == "1"


def build_emul(force=False, sanitize=None):
    sanitize = sanitize_requested() if sanitize is None else sanitize
    src = os.path.join(ROOT, "tests", "host_emul", "cg_emul.cpp")
    out = os.path.join(ROOT, "tests", "host_emul", "libcg_emul_asan.so" if sanitize else "libcg_emul.so")
    if force or _newer(out, hip_sources() + [src]):
        _run(["g++"] + (SAN_FLAGS if sanitize else ["-O2"]) + ["-std=c++17", "-shared", "-fPIC", "-o", out, src])
    return out


def build_oracle(force=False, sanitize=None):
    sanitize = sanitize_requested() if sanitize is None else sanitize
    src = os.path.join(ROOT, "oracle", "cg_oracle.c")
    outdir = os.path.join(ROOT, "oracle", "_build")
    out = os.path.join(outdir, "libcg_oracle_asan.so" if sanitize else "libcg_oracle.so")
    if not os.path.exists(src):
        return None
    os.makedirs(outdir, exist_ok=True)
    if force or _newer(out, [src]):
        _run(["gcc"] + (SAN_FLAGS if sanitize else ["-O3"]) + ["-fopenmp", "-shared", "-fPIC", "-o", out, src, "-lm"])
    return out


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--diag":       # python -m coulombgas_amd.build --diag NAME -DFLAG ...
        build_diag(sys.argv[2], sys.argv[3:]); sys.exit(0)
    if "--sanitize" in sys.argv:                            # the ASan + UBSan builds of the CPU-side libraries (tools/sanitize_cpu.sh)
        print(build_emul("--force" in sys.argv, True)); print(build_oracle("--force" in sys.argv, True)); sys.exit(0)
    f = "--force" in sys.argv
    build_hip(f); build_emul(f); build_oracle(f)
