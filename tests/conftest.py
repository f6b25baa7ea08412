import os, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# bench.py's N > 1 branch (RANK / WORLD_SIZE / MASTER_* from the environment, the library's own RCCL communicator with its TCP id
# exchange, device-side accept-rate pmean, barrier and MAX through cg_allreduce_mean) on a one-GPU box: launched ONCE -- before this
# process has touched the GPU -- as a plain `python bench.py` child with world 1 and CG_FORCE_DIST=1, and only when the test that
# reads its output (tests/test_gpu_rccl.py::test_bench_distributed_branch_without_torch_distributed) has been collected.
BENCH_DIST = {}


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def pytest_collection_finish(session):
    if not any(it.name.startswith("test_bench_distributed_branch") for it in session.items) or session.config.getoption("collectonly", False):
        return
    if not os.path.exists("/dev/kfd"):              # no AMD GPU driver node: not a GPU box
        return
    import subprocess
    env = dict(os.environ, CG_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
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
