"""GPU: 20 iterations of the reference's train loop (train.py:148-155) through the HIP path against
tests/golden/trajectory_tiny.npz (the IMPORTED reference's loop, tools/gen_golden.py trajectory): a collate-format batch
with voxel buffers from the device voxelizer, the batch's label lines -> device target generation (model.py:309) ->
RPN3D.forward (loss included) -> loss.backward() -> ClipSGD (clip_grad_norm_(5) + SGD(0.01)) -> zero_grad(), with the
BatchNorm running statistics and num_batches_tracked the forward updates on the way.

Tolerances: tests/test_oracle_trajectory.py explains why the loop cannot be compared step by step at 1e-3 — the
reference's own fp32 and fp64 runs of it (both in the fixture) part by 1e-3 ... 3e-2 from the third iteration on.
fp32 mode (the parity mode): iteration 0 within 1e-4 (nothing has fed back yet: this is the <= 1e-3 bar of BASELINE.json
with margin), every iteration within max(1e-3, 3 x the running maximum of that band), final parameters / running
statistics within 2 x (worst) and 3 x (median) of the reference's own fp32-vs-fp64 final-state distance.
bf16 mode (the benchmarked mode): the same trajectory — iteration 0 within 2e-2 (the bf16 forward's map error), every
iteration within max(3e-2, 3 x band), the SAME final-state rule.  Measured: the bf16 loop stays 1e-5 ... 2.5e-2 from the
reference's fp32 loop over the 19 all-negative iterations, final state median 7.7e-4 / worst 0.12 — inside the band the
reference's own fp32 and fp64 runs span (1e-3 ... 3e-2, 7.9e-4 / 0.11): training in the benchmarked mode follows the
reference's loss curve as closely as the reference follows itself."""
from dataclasses import replace

import numpy as np
import pytest
import torch

from oracle import torch_ref as tr
from trajectory_common import batch_clouds, check_final_state, check_targets, check_trajectory, fixture_labels, traj_grid

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("mode", ["fp32", "fp32x3", "bf16"])
def test_train_trajectory(golden, mode):
    from voxelnet_amd import model as M
    from voxelnet_amd.optim import ClipSGD
    from voxelnet_amd.targets import TargetGenerator
    from voxelnet_amd.voxelize import voxelize_device
    g = golden("trajectory_tiny")
    steps = int(g["steps"])
    grid = traj_grid(g)
    try:
        M.set_precision(mode)
        m = M.RPN3D("Car")
        m.load_state_dict(tr.make_state_dict("Car"))
        m.feature_net._grid = replace(m.feature_net._grid, H=grid.H, W=grid.W)
        m = m.to(DEV)
        gen = TargetGenerator("Car", DEV, anchors=g["anchors"])
        m.target_fn = lambda label, shape: gen(label)          # (the fixture's 24 x 24 anchor grid instead of the class default)
        batches = []
        for j in range(4):
            feats, coords, nums = [], [], []
            for i, cloud in enumerate(batch_clouds(g, j)):
                f, c, n = voxelize_device(torch.from_numpy(cloud).to(DEV), grid, i, coord_cols=4)
                feats.append(f); coords.append(c); nums.append(n)
            assert [f.shape[0] for f in feats] == list(g[f"K{j}"])
            labels = fixture_labels(g, j)
            check_targets(g, j, *[t.cpu().numpy() for t in gen(labels)], tgt_rtol=1e-6)
            batches.append(([f"b{j}s0", f"b{j}s1"], labels, feats, nums, coords, None, None))
        opt = ClipSGD(m.parameters(), lr=float(g["lr"]), max_norm=float(g["clip"]))
        losses = []
        for it in range(steps):
            m.train(True)
            out = m(batches[int(g["order"][it])], DEV)
            out[2].backward()
            total = opt.step()
            opt.zero_grad()
            scal = [float(v) for v in out[2:]]
            if it == 0:
                tol0 = {"fp32": 1e-4, "fp32x3": 5e-4}.get(mode, 2e-2)   # (fp32x3, round 4: three bf16 MFMAs per product, ~2^-16 each)
                np.testing.assert_allclose(scal, g["scalars"][0], rtol=tol0, atol=1e-6)
                # (the total gradient norm is a chained quantity: ReLU-mask flips move it by ~1e-3 in fp32, tests/test_gpu_model.py)
                assert abs(float(total) - g["grad_norm"][0]) <= (0.2 if mode == "bf16" else 5e-3) * g["grad_norm"][0]
            losses.append(scal[0])
        torch.cuda.synchronize()
        assert np.isfinite(losses).all()
        floor, wm, mm = (3e-2, 2.0, 3.0) if mode == "bf16" else (1e-3, 2.0, 3.0)
        check_trajectory(g, losses, floor, 3.0, f"HIP {mode}")
        check_final_state(g, {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}, steps, f"HIP {mode}", wm, mm)
    finally:
        M.set_precision("bf16")        # (also on a failing assertion: the precision is process-global)


@pytest.mark.parametrize("mode", ["fp32", "fp32x3", "bf16"])
def test_overfit_200_steps_with_positives(golden, mode):
    """tests/golden/overfit_tiny.npz: 200 iterations over four frames whose labels give positive anchors in every batch
    from the first step (regression loss and its quirk, loss.py:9, active throughout), against the imported reference's
    fp32 loop — per iteration within max(floor, 3 x the reference's own fp32-vs-fp64 band), the last 20 iterations' mean
    loss within 5 % (the band the reference itself stays in on these settings), and the loss falls below a third of its
    start.  The targets come from the labels INSIDE RPN3D.forward on the target stream (no target_fn: the early-targets
    path of the benchmarked step), with the fixture's 24 x 24 anchor grid installed as the module's generator."""
    from trajectory_common import check_overfit, overfit_clouds, overfit_state_dict
    from voxelnet_amd import model as M
    from voxelnet_amd.optim import ClipSGD
    from voxelnet_amd.targets import TargetGenerator
    from voxelnet_amd.voxelize import voxelize_device
    g = golden("overfit_tiny")
    steps = int(g["steps"])
    grid = traj_grid(g)
    try:
        M.set_precision(mode)
        m = M.RPN3D("Car")
        m.load_state_dict(overfit_state_dict(g, tr.make_state_dict))
        m.feature_net._grid = replace(m.feature_net._grid, H=grid.H, W=grid.W)
        m = m.to(DEV)
        gen = TargetGenerator("Car", torch.device(DEV), anchors=g["anchors"])
        m.__dict__["_targets"] = gen                      # what RPN3D._target_generator caches: forward() finds it and
        m.anchors = gen.anchors                           # generates the targets itself, on its target stream
        assert m.target_fn is None
        batches = []
        for j in range(2):
            feats, coords, nums = [], [], []
            for i, cloud in enumerate(overfit_clouds(g, j)):
                f, c, n = voxelize_device(torch.from_numpy(cloud).to(DEV), grid, i, coord_cols=4)
                feats.append(f); coords.append(c); nums.append(n)
            assert [f.shape[0] for f in feats] == list(g[f"K{j}"])
            labels = fixture_labels(g, j)
            check_targets(g, j, *[t.cpu().numpy() for t in gen(labels)], tgt_rtol=1e-6)
            batches.append(([f"b{j}s0", f"b{j}s1"], labels, feats, nums, coords, None, None))
        opt = ClipSGD(m.parameters(), lr=float(g["lr"]), max_norm=float(g["clip"]))
        losses = []
        for it in range(steps):
            m.train(True)
            out = m(batches[int(g["order"][it])], DEV)
            out[2].backward()
            opt.step()
            opt.zero_grad()
            if it == 0:
                np.testing.assert_allclose([float(v) for v in out[2:]], g["scalars"][0], rtol={"fp32": 1e-4, "fp32x3": 5e-4}.get(mode, 2e-2), atol=1e-5)
            losses.append(float(out[2]))
        torch.cuda.synchronize()
        assert m.__dict__.get("_tgt_stream") is not None          # the early-targets path ran
        assert np.isfinite(losses).all()
        check_overfit(g, losses, 3e-2 if mode == "bf16" else 1e-3, 3.0, f"HIP {mode}")
        check_final_state(g, {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}, steps, f"HIP {mode}", 2.0, 3.0)
    finally:
        M.set_precision("bf16")
