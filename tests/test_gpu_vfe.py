"""Voxel feature encoder (csrc/vfe.hip through vn_vfe_fwd / vn_vfe_bwd) on inputs built to hit every path of its
effective-row packing — voxels with 1..T points in all three packing classes (<= 8, <= 16, <= 64 rows), voxels whose
slots are all identical, all-zero padded slots (mask 0), a point that equals the padding pattern, features with no
identical suffix at all — against the oracle restatement of model.py:74-100 evaluated in float64.
Tolerance: 1e-4 of the output's max on the voxel features, 1e-3 of each gradient's max on the parameter gradients
and 1e-4 on the BatchNorm running statistics (fp32 arithmetic, different summation order)."""
import numpy as np
import pytest
import torch

from oracle import torch_ref as tr

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _voxel(rng, n, T, zero_pad=False):
    f = np.zeros((T, 7), dtype=np.float32)
    pts = rng.uniform(-2.0, 2.0, size=(n, 3)).astype(np.float32)
    f[:n, :3] = pts
    f[:n, 3] = np.round(rng.uniform(0, 1, size=n), 2)
    c = np.zeros(3, np.float32) if zero_pad else (pts.sum(0) / np.float32(max(n, 1))).astype(np.float32)
    f[:, 4:7] = f[:, :3] - c            # utils.py:87-88: every slot, the padded ones included
    return f


def _features(T, seed, dense=False):
    rng = np.random.default_rng(seed)
    if dense:
        return rng.standard_normal((150, T, 7)).astype(np.float32)       # no two slots equal: r = T everywhere
    counts = list(range(1, T + 1)) * 3 + [1, 2, 3, 4] * 30 + [7, 8, 9, 15, 16, 17, T - 1, T] * 4
    counts = [min(c, T) for c in counts]
    vox = [_voxel(rng, n, T) for n in counts]
    vox.append(_voxel(rng, min(3, T), T, zero_pad=True))         # padded slots all zero: masked out (model.py:95-96)
    allsame = np.tile(rng.standard_normal(7).astype(np.float32), (T, 1))
    vox.append(allsame)                                   # one effective row
    v = _voxel(rng, min(5, T), T)
    v[1] = v[T - 1]                                        # a point that equals the padding pattern mid-voxel
    vox.append(v)
    order = rng.permutation(len(vox))
    return np.stack([vox[i] for i in order])


def rel_err(a, b):
    a = a.detach().double().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = b.detach().double().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("T,dense", [(35, False), (45, False), (64, False), (5, False), (35, True), (2, False)])
def test_vfe_effective_rows_match_oracle(T, dense):
    from voxelnet_amd import model as M
    feat = torch.from_numpy(_features(T, 100 + T, dense))
    K = feat.shape[0]
    sd = tr.make_state_dict("Car")
    keys = M.VFE_KEYS
    bufk = ["feature_net.vfe_1.bn.running_mean", "feature_net.vfe_1.bn.running_var",
            "feature_net.vfe_2.bn.running_mean", "feature_net.vfe_2.bn.running_var"]
    # oracle in float64
    sd64 = {k: (v.double().clone() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    leaves = {k: sd64[k].requires_grad_(True) for k in keys}
    work = dict(sd64)
    work.update(leaves)
    ref = tr.voxel_features(feat.double(), work, True)
    up = torch.from_numpy(np.random.default_rng(9).standard_normal((K, 128)).astype(np.float32))
    ref.backward(up.double())
    # HIP path
    params = [sd[k].clone().to(DEV) for k in keys]
    bufs = [sd[k].clone().to(DEV) for k in bufk]
    fd = feat.to(DEV)
    vw, stats, wst = M.featnet_forward(fd, params, bufs, True)
    assert rel_err(vw, ref) < 1e-4
    for b, k in zip(bufs, bufk):
        assert rel_err(b, work[k]) < 1e-4, k
    grads = M.featnet_backward(fd, wst, stats, up.to(DEV), params)
    for g, k in zip(grads, keys):
        assert rel_err(g, leaves[k].grad) < 1e-3, k
    # eval mode (running statistics) through the same packing
    sd_eval = {k: v.double() for k, v in sd.items() if v.is_floating_point()}
    ref_eval = tr.voxel_features(feat.double(), {**sd, **sd_eval}, False)
    bufs2 = [sd[k].clone().to(DEV) for k in bufk]
    vw_eval, _, _ = M.featnet_forward(fd, params, bufs2, False)
    assert rel_err(vw_eval, ref_eval) < 1e-4
    # run-to-run bit reproducibility (fixed work list, fixed summation order)
    vw2, _, wst2 = M.featnet_forward(fd, params, [sd[k].clone().to(DEV) for k in bufk], True)
    assert torch.equal(vw, vw2)
    grads2 = M.featnet_backward(fd, wst2, stats, up.to(DEV), params)
    for a, b in zip(grads, grads2):
        assert torch.equal(a, b)


@pytest.mark.parametrize("K", [1, 17, 4095, 4097, 8200])
def test_vfe_partition_chunk_boundaries(K):
    """k_vfe_partition hands 4096 voxels to a workgroup, 16 per thread (one 16-byte load of row counts): voxel counts
    around those boundaries, all three packing classes mixed, forward + backward against the float64 oracle."""
    from voxelnet_amd import model as M
    T = 35
    pool = _features(T, 7)
    rng = np.random.default_rng(K)
    feat = torch.from_numpy(pool[rng.integers(0, pool.shape[0], size=K)])
    sd = tr.make_state_dict("Car")
    keys = M.VFE_KEYS
    bufk = ["feature_net.vfe_1.bn.running_mean", "feature_net.vfe_1.bn.running_var",
            "feature_net.vfe_2.bn.running_mean", "feature_net.vfe_2.bn.running_var"]
    sd64 = {k: (v.double().clone() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    leaves = {k: sd64[k].requires_grad_(True) for k in keys}
    work = dict(sd64)
    work.update(leaves)
    ref = tr.voxel_features(feat.double(), work, True)
    up = torch.from_numpy(np.random.default_rng(9).standard_normal((K, 128)).astype(np.float32))
    ref.backward(up.double())
    params = [sd[k].clone().to(DEV) for k in keys]
    bufs = [sd[k].clone().to(DEV) for k in bufk]
    fd = feat.to(DEV)
    vw, stats, wst = M.featnet_forward(fd, params, bufs, True)
    assert rel_err(vw, ref) < 1e-4
    grads = M.featnet_backward(fd, wst, stats, up.to(DEV), params)
    for g, k in zip(grads, keys):
        assert rel_err(g, leaves[k].grad) < 1e-3, k


def test_vfe_backward_at_the_benchmarked_size_vs_float64():
    """The VFE forward + backward on the voxel buffers of the BENCHMARKED batch (BASELINE configs[1]: two car frames,
    K ~ 12.4k voxels, T = 35, the effective-row packing classes as the real workload mixes them) against the float64
    restatement of model.py:74-100 with a seeded upstream gradient — the frozen, VFE-only counterpart of
    test_gpu_bf16_parity.py::test_bf16_step_vs_fp32_step, whose VFE bars (rel-L2 2.5 / cosine 0.4) only bound the chaos of
    the bf16 network in between and say nothing about these kernels (round-3 advisor).  Here: voxel features 1e-4, every
    parameter gradient 1e-3 of its maximum AND 1e-3 in relative L2."""
    from voxelnet_amd import model as M
    from voxelnet_amd import synth
    from voxelnet_amd.config import grid_config
    from voxelnet_amd.voxelize import voxelize_device
    grid = grid_config("Car")
    feats = []
    for b, cloud in enumerate(synth.workload_frames(1, batch=2)):
        f, _, _ = voxelize_device(torch.from_numpy(cloud).to(DEV), grid, b, coord_cols=4)
        feats.append(f)
    fd = torch.cat(feats).contiguous()
    K = fd.shape[0]
    assert 10000 < K < 16000 and fd.shape[1:] == (35, 7)
    sd = tr.make_state_dict("Car")
    keys = M.VFE_KEYS
    bufk = ["feature_net.vfe_1.bn.running_mean", "feature_net.vfe_1.bn.running_var",
            "feature_net.vfe_2.bn.running_mean", "feature_net.vfe_2.bn.running_var"]
    sd64 = {k: (v.double().clone() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    leaves = {k: sd64[k].requires_grad_(True) for k in keys}
    work = dict(sd64)
    work.update(leaves)
    ref = tr.voxel_features(fd.cpu().double(), work, True)
    up = torch.from_numpy(np.random.default_rng(19).standard_normal((K, 128)).astype(np.float32) * 1e-2)
    ref.backward(up.double())
    params = [sd[k].clone().to(DEV) for k in keys]
    bufs = [sd[k].clone().to(DEV) for k in bufk]
    vw, stats, wst = M.featnet_forward(fd, params, bufs, True)
    assert rel_err(vw, ref) < 1e-4
    grads = M.featnet_backward(fd, wst, stats, up.to(DEV), params)
    for g, k in zip(grads, keys):
        r = leaves[k].grad
        l2 = float((g.double().cpu() - r).norm() / r.norm().clamp(min=1e-30))
        assert rel_err(g, r) < 1e-3 and l2 < 1e-3, (k, rel_err(g, r), l2)


@pytest.mark.parametrize("T", [35, 64])
def test_vfe_fwd_rows_are_the_cast_of_the_voxel_features(T):
    """vn_vfe_fwd_rows (vn_net_step in the bf16 mode: the encoder's last pass also writes the bf16 rows the first Conv3d's
    rulebook GEMM reads) against vn_vfe_fwd + vn_cast_rows: the fp32 voxel features, the statistics and the bf16 rows
    bit-identical, on the mixed-class voxels of this file"""
    import ctypes

    from voxelnet_amd import _lib
    from voxelnet_amd import engine as E
    from voxelnet_amd import model as M
    feat = torch.from_numpy(_features(T, 300 + T)).to(DEV)
    K = feat.shape[0]
    sd = tr.make_state_dict("Car")
    bufk = ["feature_net.vfe_1.bn.running_mean", "feature_net.vfe_1.bn.running_var",
            "feature_net.vfe_2.bn.running_mean", "feature_net.vfe_2.bn.running_var"]
    params = [sd[k].clone().to(DEV) for k in M.VFE_KEYS]
    vw_a, stats_a, _ = M.featnet_forward(feat, params, [sd[k].clone().to(DEV) for k in bufk], True)
    rows_a = torch.empty((K, 128), dtype=torch.bfloat16, device=DEV)
    _lib.call("vn_cast_rows", vw_a.data_ptr(), _lib.VN_F32, 128, K, 128, rows_a.data_ptr(), _lib.VN_BF16, 128, 0, E.stream())
    bufs = [sd[k].clone().to(DEV) for k in bufk]
    w = _lib.VnVfeWeights(params[0].data_ptr(), params[1].data_ptr(), params[2].data_ptr(), params[3].data_ptr(),
                          bufs[0].data_ptr(), bufs[1].data_ptr(), params[4].data_ptr(), params[5].data_ptr(),
                          params[6].data_ptr(), params[7].data_ptr(), bufs[2].data_ptr(), bufs[3].data_ptr())
    wsb = _lib.load().vn_vfe_workspace_bytes(K, T)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    vw_b = torch.empty((K, 128), dtype=torch.float32, device=DEV)
    rows_b = torch.full((K, 128), float("nan"), dtype=torch.bfloat16, device=DEV)
    stats_b = torch.empty(320, dtype=torch.float32, device=DEV)
    _lib.call("vn_vfe_fwd_rows", feat.data_ptr(), K, T, ctypes.byref(w), 1, E.BN_MOMENTUM, E.BN_EPS, vw_b.data_ptr(),
              rows_b.data_ptr(), stats_b.data_ptr(), ws.data_ptr(), wsb, E.stream())
    torch.cuda.synchronize()
    assert torch.equal(vw_a, vw_b) and torch.equal(stats_a, stats_b)
    assert torch.equal(rows_a.view(torch.int16), rows_b.view(torch.int16))
