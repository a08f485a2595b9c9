"""CPU: the HOST side of the native step executor under a stub launch layer.

Eight ranks of a data-parallel run are eight processes that each enqueue ~200 executor launches per 3.6-ms step
(train.py:148-155 through vn_net_prepare / vn_net_forward / vn_net_backward).  Nobody can run eight GPUs here, but the
part of that enqueue that is OUR code can be measured without one: tests/stub_hip/stub_hip.c replaces the 17 HIP runtime
entry points the library binds by no-ops (LD_PRELOAD), tests/stub_hip/drive_executor.cpp issues the per-step call
sequence of voxelnet_amd/model.py with fake device addresses.  Asserted:
  * the executor's own host work per step is small against the ~1 ms that ~270 real hipLaunchKernel calls cost
    (measured on the GPU boxes: 2.7-3.7 us each) — it is the runtime's launch path, not plan-making, that the step's
    host time consists of;
  * eight processes pinned to eight cores stay within 1.5x of one process: no shared state (locks, false sharing, a
    global allocator hot spot) couples the ranks' executors.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get("VN_LIB_PATH") or os.path.join(ROOT, "voxelnet-pytorch_amd", "voxelnet_amd", "lib", "libvoxelnet_hip.so")
SRC = os.path.join(ROOT, "tests", "stub_hip")


@pytest.fixture(scope="module")
def stub(tmp_path_factory):
    d = tmp_path_factory.mktemp("stub_hip")
    so, exe = str(d / "libstubhip.so"), str(d / "drive_executor")
    subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(SRC, "stub_hip.c")], check=True)
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(SRC, "drive_executor.cpp"), so, "-ldl"], check=True)
    return so, exe


def run(stub, core=None, steps=300, buckets=0, extra=()):
    so, exe = stub
    env = dict(os.environ, LD_PRELOAD=so)
    cmd = [exe, LIB, str(steps), str(buckets)] + [str(x) for x in extra]
    pre = None
    if core is not None:
        pre = lambda: os.sched_setaffinity(0, {core})     # noqa: E731 (runs in the child before exec: no GPU involved)
    return subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, preexec_fn=pre)


def result(p):
    out, err = p.communicate(timeout=300)
    assert p.returncode == 0, err
    launches, us = out.split()
    return float(launches), float(us)


def test_executor_host_work_per_step_is_small(stub):
    launches, us = result(run(stub))
    assert 150 <= launches <= 260, launches          # the executor's ~196 launches of a bf16 car step (VFE / loss / optimizer are other calls)
    # measured here: ~33 us per step = 0.17 us per launch; the real hipLaunchKernel costs 2.7-3.7 us each on the GPU boxes
    assert us < 250.0, us
    lb, usb = result(run(stub, buckets=1))           # the data-parallel form: per-group unpacks + bucket events
    assert lb >= launches and usb < 300.0, (lb, usb)


def test_one_call_step_host_work(stub):
    """vn_net_step (voxel feature encoder + network + loss + backward + clip/SGD as one call) under the stub: every argument
    check passes for the three modes, it issues the executor's launches plus the encoder's, the loss's and the optimizer's,
    and its own host work stays tens of microseconds"""
    base, _ = result(run(stub))
    for mode in (0, 1, 2):
        launches, us = result(run(stub, buckets=2, extra=(mode,)))
        assert launches >= 150 and us < 300.0, (mode, launches, us)
        if mode == 0:
            assert base + 15 <= launches <= base + 45, (base, launches)      # + VFE 3+1 fwd, 3+ bwd, finalizes, loss 2, cast, tick, cat, clip 2


@pytest.mark.parametrize("mode", [0, 1, 2], ids=["bf16", "fp32", "fp32x3"])
@pytest.mark.parametrize("grid", [(400, 352, 12345), (16, 24, 300), (200, 240, 9000), (400, 352, 160000)],
                         ids=["car", "tiny", "ped-sized", "dense"])
def test_every_mode_plans_and_issues_a_step_on_every_grid(stub, mode, grid):
    """the whole call sequence of a train step returns VN_OK under the stub for every precision mode on the full-size, tiny,
    pedestrian-sized and dense (K = 160k: the first layer's dense backward route) grids — kernel selection, workspace
    carving and every host-side argument check, without a GPU (a geometry the selected kernel refuses shows up HERE: round 5's
    in-place weight-gradient passes were first wired for the 400 x 352 images only and failed the 16 x 24 fixture on the box)"""
    H, W, K = grid
    launches, us = result(run(stub, steps=3, extra=(mode, H, W, K)))
    assert launches > 100


def test_eight_pinned_processes_stay_within_1p5x_of_one(stub):
    """eight executors on eight cores against one: within 1.5x — or, on a host whose eight "cores" are SMT siblings / a shared
    box (this build container: a private-memory control workload itself loses 1.3-1.6x), within 1.4x of what that control
    loses: the ranks' executors share no state"""
    cores = sorted(os.sched_getaffinity(0))
    if len(cores) < 8:
        pytest.skip(f"{len(cores)} cores")

    def ratio(buckets, steps):
        _, one = result(run(stub, core=cores[0], steps=steps, buckets=buckets))
        procs = [run(stub, core=c, steps=steps, buckets=buckets) for c in cores[:8]]
        return max(result(p)[1] for p in procs) / one

    # a shared CI host: up to eight attempts, the first one inside the bound counts (a lost time slice on ONE of the eight
    # cores during a 50-ms measurement is enough to miss it); a host whose private-memory control itself loses more than 2x
    # in the median is too noisy to say anything: skip rather than fail
    notes, controls = [], []
    for attempt in range(8):
        control = ratio(-1, 20000)
        r = ratio(1 if attempt % 3 == 2 else 0, 3000)
        controls.append(control)
        notes.append(f"executor {r:.2f}x / control {control:.2f}x")
        if r <= max(1.5, 1.4 * control):      # (a lock or a shared hot spot shows as 3-8x, not as 1.4x)
            return
    if sorted(controls)[len(controls) // 2] > 2.0:
        pytest.skip("host too noisy for a scaling measurement: " + "; ".join(notes))
    pytest.fail("eight pinned processes never inside the bound in eight attempts: " + "; ".join(notes))
