"""Device RPN target generation (csrc/targets.hip through vn_rpn_targets / voxelnet_amd.targets) against the oracle
(oracle/targets.py, itself pinned to the reference by tests/golden/targets_car.npz) and against that fixture directly.
Bar: which anchors are positive / negative — bit-exact; regression targets — the float64 oracle values rounded to
float32, within 1 float32 ulp (the device's float64 log may differ from glibc's in the last float64 bit)."""
import os

import numpy as np
import pytest
import torch

from oracle import targets as ot

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(__file__), "golden", "targets_car.npz")


def _check(pos, neg, tgt, ref_pos, ref_neg, ref_tgt):
    assert pos.dtype == torch.float32 and tgt.dtype == torch.float32
    assert np.array_equal(pos.cpu().numpy(), ref_pos.astype(np.float32))
    assert np.array_equal(neg.cpu().numpy(), ref_neg.astype(np.float32))
    t, r = tgt.cpu().numpy(), ref_tgt.astype(np.float32)
    assert np.array_equal(t != 0, r != 0)
    np.testing.assert_allclose(t, r, rtol=1.2e-7, atol=0)


def test_targets_match_reference_fixture():
    from voxelnet_amd import targets as T
    g = np.load(GOLD, allow_pickle=False)
    n = int(g["n_samples"])
    labels = [[str(s) for s in g[f"labels{b}"]] for b in range(n)]
    shape = tuple(int(v) for v in g["shape"])
    gen = T.TargetGenerator("Car", DEV)
    assert np.array_equal(gen.anchors, ot.generate_anchors("Car"))           # bit-identical anchors
    pos, neg, tgt = gen(labels, shape)
    assert pos.shape == (n, *shape, 2) and neg.shape == (n, *shape, 2) and tgt.shape == (n, *shape, 14)
    for b in range(n):
        assert np.array_equal(np.flatnonzero(pos[b].cpu().numpy()).astype(np.int32), g[f"pos_idx{b}"]), b
        assert np.array_equal(np.packbits(neg[b].cpu().numpy().reshape(-1).astype(np.uint8)), g[f"neg_bits{b}"]), b
        t = tgt[b].cpu().numpy().reshape(-1)
        nz = np.flatnonzero(t)
        assert np.array_equal(nz.astype(np.int32), g[f"tgt_idx{b}"]), b
        np.testing.assert_allclose(t[nz], g[f"tgt_val{b}"].astype(np.float32), rtol=1.2e-7, atol=0)
    # module-level function with the reference's signature
    p2, n2, t2 = T.generate_targets(labels, shape, gen.anchors, "Car", "lidar", DEV)
    assert torch.equal(p2, pos) and torch.equal(n2, neg) and torch.equal(t2, tgt)


@pytest.mark.parametrize("seed,counts", [(1, [3, 0, 40]), (2, [1]), (3, [128, 7])])
def test_targets_match_oracle_on_random_boxes(seed, counts):
    """boxes straight in lidar coordinates: dense scenes, duplicates (ties between boxes), boxes outside the range"""
    from voxelnet_amd import targets as T
    rng = np.random.default_rng(seed)
    boxes = []
    for c in counts:
        b = np.stack([rng.uniform(-5, 75, c), rng.uniform(-45, 45, c), rng.uniform(-2, -1, c), rng.uniform(1.3, 1.8, c),
                      rng.uniform(1.4, 1.9, c), rng.uniform(3.2, 4.6, c), rng.uniform(-1.57, 1.57, c)], axis=1)
        if c >= 3:
            b[1] = b[0]                      # identical boxes: the first one must win every tie
        boxes.append(b)
    anchors = ot.generate_anchors("Car")
    ref = ot.generate_targets_from_boxes(boxes, (200, 176), anchors, "Car")
    gen = T.TargetGenerator("Car", DEV)
    _check(*gen.from_boxes(boxes), *ref)
    with pytest.raises(T._lib.VoxelnetHipError):
        gen.from_boxes([np.zeros((129, 7))])


def test_host_helpers_match_oracle():
    from voxelnet_amd import targets as T
    g = np.load(GOLD, allow_pickle=False)
    labels = [[str(s) for s in g[f"labels{b}"]] for b in range(int(g["n_samples"]))]
    for a, b in zip(T.label_to_gt_box_3d(labels, "Car"), ot.label_to_gt_box_3d(labels, "Car")):
        assert np.array_equal(a, b)
        assert np.array_equal(T.gt_standup_boxes(a), ot.gt_standup_2d(b))
    for cls in ("Car", "Pedestrian", "Cyclist"):
        assert np.array_equal(T.generate_anchors(cls), ot.generate_anchors(cls))


def test_rpn3d_forward_generates_targets_from_labels():
    """RPN3D.forward(batch, device) with label lines in x[1] == the same call with the oracle's targets passed in"""
    from voxelnet_amd import model as M
    from voxelnet_amd import synth
    from voxelnet_amd.config import grid_config
    from voxelnet_amd.voxelize import voxelize_device
    g = np.load(GOLD, allow_pickle=False)
    labels = [[str(s) for s in g["labels0"]]]
    M.set_precision("bf16")
    torch.manual_seed(3)
    model = M.RPN3D("Car").to(DEV).train(True)
    grid = grid_config("Car")
    f, c, _ = voxelize_device(torch.from_numpy(synth.workload_frames(1, batch=1)[0]).to(DEV), grid, 0, coord_cols=4)
    batch = (None, labels, [f], None, [c], None, None)
    out = model(batch, DEV)
    ref_t = ot.generate_targets(labels, (200, 176), ot.generate_anchors("Car"))
    # (train-mode BatchNorm updates running stats only: the two forwards see the same weights)
    out2 = model(batch, DEV, targets=tuple(torch.from_numpy(a.astype(np.float32)).to(DEV) for a in ref_t))
    for a, b in zip(out[2:], out2[2:]):
        assert abs(a.item() - b.item()) <= 1e-6 * max(1.0, abs(b.item()))
    assert model.anchors.shape == (200, 176, 2, 7)
