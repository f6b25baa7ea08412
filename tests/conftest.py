import os, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# bench.py's N > 1 branch (torch.distributed rendezvous, RCCL communicator from the library, device-side accept-rate pmean) on a
# one-GPU box: launched ONCE, at session start -- before this process has touched the GPU -- as a child under torch.distributed.run
# with one rank and CG_FORCE_DIST=1; tests/test_gpu_rccl.py asserts on what it printed.
BENCH_DIST = {}


def pytest_sessionstart(session):
    expr = session.config.getoption("markexpr", default="") or ""
    if "gpu" not in expr or "not gpu" in expr:
        return
    try:
        import torch
        if torch.cuda.device_count() < 1:          # (counting devices does not initialise the GPU)
            return
    except ImportError:
        return
    import subprocess
    env = dict(os.environ, CG_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29700 + os.getpid() % 200
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-energy-check", "--no-update-extras"]
    try:
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        BENCH_DIST.update(rc=r.returncode, out=r.stdout, err=r.stderr[-4000:])
    except Exception as e:                          # noqa: BLE001 -- reported by the test
        BENCH_DIST.update(rc=-1, out="", err=repr(e))
    session.config._cg_bench_dist = BENCH_DIST      # (the test reads it from request.config: `tests.conftest` is another module object)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def _usable_cores():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


@pytest.fixture(scope="session", autouse=True)
def _torch_threads_within_cpu_quota():
    """the oracle (torch.func) defaults to one thread per host core; inside a cgroup CPU quota that only thrashes"""
    try:
        import torch
        torch.set_num_threads(_usable_cores())
    except ImportError:
        pass
    yield
