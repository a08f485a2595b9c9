"""GPU: HIP voxelizer + scatter through the C ABI vs the oracle and the golden
vectors of the imported reference (utils.py:10-100, model.py:102-106).
Bar: coordinates/counts bit-exact (int64), features bit-exact (fp32 bits)."""
import hashlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def hip_voxelize(points, target, T=None, **kw):
    from voxelnet_amd.config import grid_config
    from voxelnet_amd.voxelize import voxelize_device
    g = grid_config(target, T=T, **kw)
    f, c, n = voxelize_device(torch.from_numpy(points).cuda(), g, 0, coord_cols=3)
    return {"feature_buffer": f.cpu().numpy(), "coordinate_buffer": c.cpu().numpy(), "number_buffer": n.cpu().numpy()}


@pytest.mark.parametrize("tag,target", [("car", "Car"), ("ped", "Pedestrian")])
def test_golden_small_bit_exact(golden, tag, target):
    g = golden(f"voxelize_{tag}_small")
    out = hip_voxelize(g["points"], target)
    assert np.array_equal(out["coordinate_buffer"], g["coordinate_buffer"])
    assert np.array_equal(out["number_buffer"], g["number_buffer"])
    assert np.array_equal(out["feature_buffer"].view(np.uint32), g["feature_buffer"].view(np.uint32))


def test_degenerate(golden):
    g = golden("voxelize_degenerate")
    out = hip_voxelize(g["far_points"], "Car")
    assert out["coordinate_buffer"].shape == (0, 3) and out["feature_buffer"].shape == (0, 35, 7)
    one = hip_voxelize(g["one_points"], "Car")
    assert np.array_equal(one["feature_buffer"], g["one_feature"])
    assert np.array_equal(one["coordinate_buffer"], g["one_coord"])
    empty = hip_voxelize(np.zeros((0, 4), np.float32), "Car")
    assert empty["number_buffer"].shape == (0,)


@pytest.mark.parametrize("cfg_id,target", [(2, "Car"), (3, "Pedestrian")])
def test_full_size_digest(golden, cfg_id, target):
    from voxelnet_amd import synth
    g = golden("voxelize_full_digest")
    w = synth.WORKLOADS[cfg_id]
    cloud = synth.synth_cloud(target, w["k0"], synth.frame_seed(cfg_id, 0), w["mean_extra"], w["T"])
    np.random.seed(7)
    np.random.shuffle(cloud)
    out = hip_voxelize(cloud, target)
    assert out["coordinate_buffer"].shape[0] == int(g[f"cfg{cfg_id}_K"])
    assert sha(out["coordinate_buffer"]) == str(g[f"cfg{cfg_id}_coord_sha"])
    assert sha(out["number_buffer"]) == str(g[f"cfg{cfg_id}_number_sha"])
    assert sha(out["feature_buffer"]) == str(g[f"cfg{cfg_id}_feature_sha"])


def test_pcl_to_voxels_dropin_shuffles_in_place(golden):
    """Same call as the reference: shuffles the caller's array (utils.py:35) and
    returns the dict of numpy buffers; replaying the shuffle through the oracle
    gives identical buffers."""
    from oracle import voxelize as ov
    from voxelnet_amd import synth
    from voxelnet_amd.voxelize import pcl_to_voxels, prepare_voxel
    cloud = synth.synth_cloud("Car", 500, 3)
    twin = cloud.copy()
    np.random.seed(123)
    out = pcl_to_voxels(cloud, "Car")
    np.random.seed(123)
    np.random.shuffle(twin)
    assert np.array_equal(cloud, twin)
    ref = ov.voxelize(twin, "Car")
    for k in ref:
        assert np.array_equal(out[k], ref[k]) and out[k].dtype == ref[k].dtype
    f, n, c = prepare_voxel([out, out])
    assert c[1].shape[1] == 4 and (c[1][:, 0] == 1).all()


def test_dense_config5_vs_oracle():
    """BASELINE config 5 shape: 300k points, 40k voxels, T=64 (one frame)."""
    from oracle import voxelize as ov
    from voxelnet_amd import synth
    w = synth.WORKLOADS[5]
    cloud = synth.synth_cloud("Car", w["k0"], synth.frame_seed(5, 0), w["mean_extra"], w["T"])
    assert cloud.shape[0] > 250000
    ref = ov.voxelize(cloud, "Car", T=64)
    out = hip_voxelize(cloud, "Car", T=64)
    assert ref["coordinate_buffer"].shape[0] == 40000
    for k in ref:
        assert np.array_equal(out[k], ref[k])


def test_hot_voxel_many_points():
    """>64 points in one voxel exercises the long-segment path; first T in input order."""
    from oracle import voxelize as ov
    rng = np.random.default_rng(0)
    hot = np.concatenate([rng.uniform([10.0, 1.0, -1.0, 0], [10.19, 1.19, -0.61, 1], (1000, 4)),
                          rng.uniform([0, -40, -3, 0], [70, 40, 1, 1], (500, 4))]).astype(np.float32)
    rng.shuffle(hot)
    ref = ov.voxelize(hot, "Car")
    out = hip_voxelize(hot, "Car")
    for k in ref:
        assert np.array_equal(out[k], ref[k])
    assert ref["number_buffer"].max() == 35


def test_scatter_fwd_bwd_vs_oracle():
    from oracle import torch_ref as tr
    from voxelnet_amd import ops
    rng = np.random.default_rng(1)
    B, D, H, W, C = 2, 10, 16, 24, 128
    K = 300
    lin = rng.choice(B * D * H * W, K, replace=False)
    coord = np.stack([lin // (D * H * W), (lin // (H * W)) % D, (lin // W) % H, lin % W], 1).astype(np.int64)
    vw = rng.standard_normal((K, C)).astype(np.float32)
    ref = tr.scatter_dense(torch.from_numpy(vw), torch.from_numpy(coord), (B, D, H, W))
    vw_d = torch.from_numpy(vw).cuda().requires_grad_(True)
    dense = ops.scatter_dense(vw_d, torch.from_numpy(coord).cuda(), (B, D, H, W))
    assert torch.equal(dense.detach().cpu(), ref)           # pure data movement: bit-exact
    up = torch.from_numpy(rng.standard_normal(ref.shape).astype(np.float32))
    dense.backward(up.cuda())
    want = up[coord[:, 0], coord[:, 1], coord[:, 2], coord[:, 3]]
    assert torch.equal(vw_d.grad.cpu(), want)
