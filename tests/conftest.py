import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "voxelnet-pytorch_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


DDP_REHEARSAL = {}     # filled by pytest_sessionstart on a GPU run: {"procs": [...], "dir": ..., "error": ...}
BENCH_RUN = {}         # bench.py started beside them: {"proc": ..., "dir": ...}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_run(config):
    expr = (config.getoption("-m") or "").replace(" ", "")
    return "gpu" in expr and "notgpu" not in expr


def pytest_sessionstart(session):
    """tests/test_gpu_multiproc.py needs two FRESH processes on cuda:0.  They are started here, before this process has
    made any GPU call (a process that has initialised HIP must not fork + exec on the GPU boxes), run beside the other
    GPU tests and are collected by the test at the end."""
    import subprocess
    import tempfile
    if not _gpu_run(session.config) or os.environ.get("VN_NO_DDP_REHEARSAL") == "1":
        return
    try:
        import socket

        import torch
        if torch.cuda.device_count() < 1:          # (device_count() does not initialise the GPU)
            return
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
        s.close()
        d = tempfile.mkdtemp(prefix="vn_ddp_")
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs = []
        for r in range(2):
            log = open(os.path.join(d, f"rank{r}.log"), "w")
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ddp_worker.py"), str(r), "2", port, d],
                                          stdout=log, stderr=subprocess.STDOUT, env=env, cwd=ROOT))
        DDP_REHEARSAL.update(procs=procs, dir=d)
        # tests/test_gpu_bench_contract.py: bench.py's N = 1 path with the reducer attached, 2 steps, as its own process
        # (started here for the same reason as the ranks above)
        blog = open(os.path.join(d, "bench.err"), "w")
        bout = open(os.path.join(d, "bench.json"), "w")
        BENCH_RUN.update(dir=d, proc=subprocess.Popen(
            [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--windows", "1", "--timer-steps", "1",
             "--force-reducer", "--no-parity-mode"], stdout=bout, stderr=blog, env=env, cwd=ROOT))
    except Exception as e:  # noqa: BLE001 - reported by the test
        DDP_REHEARSAL["error"] = repr(e)


def pytest_collection_modifyitems(config, items):
    """the tests that collect the side processes started above go last: those processes (two ranks, one bench.py) then
    run BESIDE the other GPU tests instead of being waited for"""
    late = ("test_gpu_bench_contract.py", "test_gpu_multiproc.py")
    items.sort(key=lambda it: os.path.basename(str(it.fspath)) in late)          # (stable: the rest keep their order)


def pytest_sessionfinish(session, exitstatus):
    for p in DDP_REHEARSAL.get("procs", []) + ([BENCH_RUN["proc"]] if "proc" in BENCH_RUN else []):
        if p.poll() is None:
            p.kill()


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        return cache[name]
    return load
