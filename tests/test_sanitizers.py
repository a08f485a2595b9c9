"""AddressSanitizer + UBSan over the CPU side of the boundary (SURVEY.md section 5 "race detection / sanitizers"; CPU build
only: GPU ASan / xnack+ runs are not available on this pool).

`make -C voxelnet-pytorch_amd/csrc asan` instruments the HOST code of every entry point of the library (the host
voxelizer vn_voxelize_host_* — utils.py:10-100 as the DataLoader workers call it, dataset.py:58 — plus the argument
checks, plans and workspace carving of the device entry points); `make -C oracle asan` does the same for the C
restatement.  The existing host-side tests then run in a child interpreter with the sanitizer runtime preloaded and both
libraries swapped for the instrumented builds.  A canary (a deliberately undersized output buffer) proves the
instrumentation is live: a sanitizer that reports nothing on it would make the green run meaningless."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "voxelnet-pytorch_amd", "csrc")
ASAN_LIB = os.path.join(ROOT, "voxelnet-pytorch_amd", "voxelnet_amd", "lib", "libvoxelnet_hip_asan.so")
ORACLE_ASAN = os.path.join(ROOT, "oracle", "liboracle_asan.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def _runtime():
    if not os.path.exists(CLANG):
        return None
    rt = subprocess.run([CLANG, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    return rt if os.path.isabs(rt) and os.path.exists(rt) else None


@pytest.fixture(scope="module")
def asan_env():
    rt = _runtime()
    if rt is None:
        pytest.skip("clang's AddressSanitizer runtime is not installed")
    jobs = str(min(8, os.cpu_count() or 1))
    subprocess.run(["make", "-s", "-j", jobs, "-C", CSRC, "asan"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True)
    assert os.path.exists(ASAN_LIB) and os.path.exists(ORACLE_ASAN)
    env = dict(os.environ, LD_PRELOAD=rt, VN_LIB_PATH=ASAN_LIB, VN_ORACLE_LIB=ORACLE_ASAN,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",      # (CPython itself leaks by design)
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", VN_NO_DDP_REHEARSAL="1")
    return env


def test_host_side_tests_are_clean_under_asan_ubsan(asan_env):
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_host_boundary.py"), os.path.join(ROOT, "tests", "test_oracle_voxelize.py"),
                        os.path.join(ROOT, "tests", "test_abi.py")],
                       env=asan_env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    assert " passed" in out


CANARY = r"""
import ctypes, numpy as np, sys
sys.path.insert(0, {pkg!r}); sys.path.insert(0, {root!r})
from voxelnet_amd import _lib, synth
from voxelnet_amd.config import grid_config
L = _lib.load()
assert _lib.LIB_PATH.endswith("_asan.so")
g = grid_config("Car")
cloud = synth.synth_cloud("Car", 300, 3)
vg = _lib.VnGrid(g.D, g.H, g.W, g.vz, g.vy, g.vx, g.ox, g.oy, g.oz, g.T)
n = cloud.shape[0]
wsb = L.vn_voxelize_host_workspace_bytes(n, ctypes.byref(vg))
ws = np.empty(wsb, np.uint8)
k = ctypes.c_int64(0)
assert L.vn_voxelize_host_index(cloud.ctypes.data, n, ctypes.byref(vg), ws.ctypes.data, wsb, ctypes.byref(k)) == 0
K = k.value
feat = np.empty(((K - 1) * g.T * 7,), np.float32)          # one voxel row short: the last row's stores are out of bounds
coord = np.empty((K, 3), np.int64); num = np.empty((K,), np.int64)
print("canary: calling with an undersized feature buffer", flush=True)
L.vn_voxelize_host_gather(cloud.ctypes.data, n, ctypes.byref(vg), ws.ctypes.data, wsb, K, 0, 3, feat.ctypes.data,
                          coord.ctypes.data, num.ctypes.data)
print("canary: NOT detected", flush=True)
"""


def test_asan_is_live_canary_overflow_is_reported(asan_env):
    code = CANARY.format(pkg=os.path.join(ROOT, "voxelnet-pytorch_amd"), root=ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=asan_env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    assert "canary: calling" in out, out[-3000:]
    assert r.returncode != 0 and "NOT detected" not in out, out[-3000:]
    assert "AddressSanitizer: heap-buffer-overflow" in out, out[-3000:]
    assert "vn_voxelize_host_gather" in out, out[-3000:]       # the report names the instrumented frame
