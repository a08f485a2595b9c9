"""GPU: the nn.Module surface (voxelnet_amd.model) end to end through the C ABI against the
golden vectors of the imported reference (tools/gen_golden.py) and the oracle.

  FeatureLearningNet   model.py:85-108   featnet_tiny.npz      (fp32 VALU kernels: 1e-4)
  RPN3D.detect         model.py:305-306  middle_tiny_{car,ped}.npz  fwd maps + all 104 parameter grads
  RPN3D.forward/loss   model.py:298-362  rpn3d_tiny.npz
  full-size car frame  BASELINE config 1 car_full.npz (lattice of the maps)
Parity bar (BASELINE.json north_star): fp32, <= 1e-3 relative for voxel features and RPN
maps — asserted in 'fp32' mode (exact fp32 MFMA products); 'bf16x3' and 'bf16' are checked
separately with their own stated tolerances (the 23-layer Conv+BN+ReLU stack amplifies any
per-layer rounding ~20x; fp32 vs fp64 runs of the reference's own torch ops differ by ~2e-4)."""
from dataclasses import replace

import numpy as np
import pytest
import torch

from oracle import torch_ref as tr

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel_err(a, b):
    a = a.detach().float().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))


def digest(t):
    f = t.detach().reshape(-1).double().cpu()
    stride = max(1, f.numel() // 256)
    return np.array([f.norm().item(), f.sum().item(), f.abs().sum().item()]), t.detach().reshape(-1)[::stride].cpu().numpy()


def split(g):
    feats = torch.from_numpy(g["features"])
    coords = torch.from_numpy(g["coords"])
    lens = [int(x) for x in g["feat_lens"]]
    return list(torch.split(feats, lens)), list(torch.split(coords, lens))


def make_model(cls, H=None, W=None, mode="fp32"):
    from voxelnet_amd import model as M
    M.set_precision(mode)
    m = M.RPN3D(cls)
    m.load_state_dict(tr.make_state_dict(cls))
    if H is not None:
        m.feature_net._grid = replace(m.feature_net._grid, H=H, W=W)
    return m.to(DEV)


def test_feature_net(golden):
    from voxelnet_amd import model as M
    g = golden("featnet_tiny")
    feats, coords = split(g)
    m = make_model("Car", 16, 24)
    fn = m.feature_net
    fd, cd = [f.to(DEV) for f in feats], [c.to(DEV) for c in coords]
    fn.eval()
    with torch.no_grad():
        dense = fn(fd, cd)
    c = torch.cat(coords)
    rows = dense[c[:, 0], c[:, 1], c[:, 2], c[:, 3]]
    assert rel_err(rows, g["eval_rows"]) < 1e-4
    fn.train()
    dense = fn(fd, cd)
    assert dense.shape == (2, 10, 16, 24, 128) and dense.dtype == torch.float32
    rows = dense[c[:, 0], c[:, 1], c[:, 2], c[:, 3]]
    assert rel_err(rows, g["train_rows"]) < 1e-4
    assert abs(dense.double().sum().item() - float(g["train_dense_sum"])) < 1e-2
    assert int((dense.abs().sum(-1) != 0).sum()) <= c.shape[0]
    up = torch.from_numpy((np.random.default_rng(31).standard_normal(tuple(dense.shape)) * 1e-2).astype(np.float32))
    dense.backward(up.to(DEV))
    for k, p in fn.named_parameters():
        ref = g["grad." + k]
        assert rel_err(p.grad, ref) < 2e-3, k
    for k, b in fn.named_buffers():
        if "running" in k:
            assert rel_err(b, g["buf." + k]) < 1e-4, k
    assert int(fn.vfe_1.bn.num_batches_tracked) == 1
    with pytest.raises(M._lib.VoxelnetHipError):
        fn([f for f in feats], [c_ for c_ in coords])       # CPU tensors: no fallback


@pytest.mark.parametrize("cls,tag", [("Car", "car"), ("Pedestrian", "ped")])
def test_detect_fwd_bwd_fp32(golden, cls, tag):
    g = golden(f"middle_tiny_{tag}")
    feats, coords = split(g)
    m = make_model(cls, 16, 24, "fp32")
    m.train()
    prob, reg = m.detect([f.to(DEV) for f in feats], [c.to(DEV) for c in coords])
    assert rel_err(prob, g["prob"]) < 1e-3
    assert rel_err(reg, g["reg"]) < 1e-3
    dp = torch.from_numpy((np.random.default_rng(41).standard_normal(g["prob"].shape) * 1e-1).astype(np.float32))
    dr = torch.from_numpy((np.random.default_rng(42).standard_normal(g["reg"].shape) * 1e-1).astype(np.float32))
    torch.autograd.backward([prob, reg], [dp.to(DEV), dr.to(DEV)])
    worst = 0.0
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        d, smp = digest(p.grad)
        ref_d, ref_s = g["gdig." + k], g["gsmp." + k]
        if k.endswith("conv.bias") and "prob_conv" not in k and "reg_conv" not in k or k.endswith("deconv.bias"):
            # bias in front of a train-mode BatchNorm: true gradient 0, the reference holds rounding noise
            assert d[0] < 1e-3 * (1.0 + ref_d[0]) + 1e-2, k
            continue
        e = float(np.abs(smp - ref_s).max() / max(np.abs(ref_s).max(), 1e-12))
        worst = max(worst, e)
        assert e < 5e-3, (k, e)
        assert abs(d[0] - ref_d[0]) < 5e-3 * ref_d[0] + 1e-9, (k, d, ref_d)
    for k, b in m.named_buffers():
        if "running" in k:
            assert rel_err(b, g["buf." + k]) < 2e-3, k
    print("worst sampled-gradient error", worst)


def test_middle_module_boundary(golden):
    """MiddleConvNet.forward on an fp32 (B,D,H,W,128) tensor (predict.py:59-60 call pattern)."""
    g = golden("middle_tiny_car")
    feats, coords = split(g)
    m = make_model("Car", 16, 24, "fp32")
    m.train()
    dense = m.feature_net([f.to(DEV) for f in feats], [c.to(DEV) for c in coords])
    prob, reg = m.middle_rpn(dense)
    assert rel_err(prob, g["prob"]) < 1e-3 and rel_err(reg, g["reg"]) < 1e-3
    (prob.sum() + reg.sum()).backward()
    assert m.feature_net.vfe_1.fcn[0].weight.grad is not None


@pytest.mark.parametrize("mode,tol", [("bf16x3", 5e-3), ("bf16", 0.25)])
def test_detect_reduced_precision_modes(golden, mode, tol):
    g = golden("middle_tiny_car")
    feats, coords = split(g)
    m = make_model("Car", 16, 24, mode)
    m.train()
    prob, reg = m.detect([f.to(DEV) for f in feats], [c.to(DEV) for c in coords])
    # stated tolerances relative to the map maximum (not the parity bar): bf16x3 5e-3, bf16 0.25
    print(mode, "prob err", rel_err(prob, g["prob"]), "reg err", rel_err(reg, g["reg"]))
    assert rel_err(prob, g["prob"]) < tol
    assert rel_err(reg, g["reg"]) < tol
    torch.autograd.backward([prob, reg], [torch.ones_like(prob) * 0.1, torch.ones_like(reg) * 0.1])
    for k, p in m.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k


def test_rpn3d_forward_loss(golden):
    g = golden("rpn3d_tiny")
    feats, coords = split(golden("middle_tiny_car"))
    m = make_model("Car", 16, 24, "fp32")
    m.train()
    batch = (["a", "b"], None, feats, None, coords, None, None)
    out = m(batch, DEV, targets=(g["pos"], g["neg"], g["targets"]))
    assert len(out) == 7
    prob, delta, loss, cls_loss, reg_loss, cpos, cneg = out
    got = np.array([loss.item(), cls_loss.item(), reg_loss.item(), cpos.item(), cneg.item()])
    np.testing.assert_allclose(got, g["scalars"], rtol=2e-3)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(m.parameters(), 5)          # train.py:153
    torch.optim.SGD(m.parameters(), lr=0.01).step()            # train.py:130,154
    for k, p in m.named_parameters():
        d, _ = digest(p.grad)
        assert np.isfinite(d).all(), k


def test_car_full_forward(golden):
    """BASELINE config 1: one full-size synthetic car frame, B=1, train-mode forward, exact mode."""
    from oracle import voxelize as ov
    from voxelnet_amd import synth
    from voxelnet_amd.config import grid_config
    from voxelnet_amd.voxelize import voxelize_device
    g = golden("car_full")
    w = synth.WORKLOADS[1]
    cloud = synth.synth_cloud("Car", w["k0"], synth.frame_seed(1, 0), w["mean_extra"], w["T"])
    np.random.seed(7); np.random.shuffle(cloud)
    f, c, n = voxelize_device(torch.from_numpy(cloud).to(DEV), grid_config("Car"), 0, coord_cols=4)
    assert f.shape[0] == int(g["K"])
    m = make_model("Car", mode="fp32")
    m.train()
    with torch.no_grad():
        prob, reg = m.detect([f], [c])
    assert prob.shape == (1, 2, 200, 176) and reg.shape == (1, 14, 200, 176)
    assert rel_err(prob[:, :, ::8, ::8], g["prob_lattice"]) < 1e-3
    assert rel_err(reg[:, :, ::8, ::8], g["reg_lattice"]) < 1e-3
