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
    # Gradient checker: the oracle evaluated in float64 on the same inputs.  The fp32 golden
    # gradients are NOT usable as the bar for the layers upstream of block3.3: on this fixture the
    # reference's own fp32 run flips one ReLU mask element there relative to exact arithmetic
    # (fp32 vs fp64 runs of the same torch ops differ by 1e-1 at block3.3 and 1-4e-2 in every layer
    # upstream, 4e-5 downstream), and this implementation lands on the fp64 side of that flip.
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in tr.make_state_dict(cls).items()}
    _, _, ref = tr.forward_backward([f.double() for f in feats], coords, sd64, (10, 16, 24), cls, dp.double(),
                                    dr.double())
    report = []
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        r = ref[k].numpy()
        if k.endswith("conv.bias") and "prob_conv" not in k and "reg_conv" not in k or k.endswith("deconv.bias"):
            # bias in front of a train-mode BatchNorm: the true gradient is exactly 0
            assert float(p.grad.abs().max()) < 1e-3, k
            continue
        gq = p.grad.detach().double().cpu().numpy()
        report.append((k, rel_err(p.grad, r.astype(np.float32)), float(np.linalg.norm(gq - r) / np.linalg.norm(r))))
    for k, e, l2 in report:
        print(f"{k:50s} vs fp64 oracle: max err {e:.2e}  rel L2 {l2:.2e}")
    # On this 16x24 fixture block3 has 12 (car) / 48 (ped) sites per channel, so ONE ReLU-mask flip (an fp32
    # rounding event at z ~ 0, which also separates the reference's own fp32 and fp64 runs) moves every
    # upstream gradient by 1e-2..1e-1 of its maximum.  The chained bar here is therefore a relative L2 of
    # 5e-2 (any wiring/tap/transposition bug gives O(1)); tight per-layer gradient parity is asserted in
    # test_gpu_layers.py and the tight chained check runs at full size (test_car_full_backward), where a
    # flip is diluted over ~1e5 sites per channel.
    for k, e, l2 in report:
        assert l2 < 5e-2, (k, e, l2)
    tail = [e for k, e, l2 in report if k.split(".")[1] in ("prob_conv", "reg_conv")]
    assert max(tail) < 1e-3
    # the heads' gradients do not pass through any ReLU: the fp32 golden of the reference run agrees tightly
    for k in ("middle_rpn.prob_conv.conv.weight", "middle_rpn.reg_conv.conv.weight"):
        d, smp = digest(dict(m.named_parameters())[k].grad)
        assert float(np.abs(smp - g["gsmp." + k]).max() / np.abs(g["gsmp." + k]).max()) < 1e-3, k
    for k, b in m.named_buffers():
        if "running" in k:
            assert rel_err(b, g["buf." + k]) < 2e-3, k


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


@pytest.mark.parametrize("mode,tol", [("fp32x3", 1e-3), ("bf16", 0.25)])
def test_detect_reduced_precision_modes(golden, mode, tol):
    g = golden("middle_tiny_car")
    feats, coords = split(g)
    m = make_model("Car", 16, 24, mode)
    m.train()
    prob, reg = m.detect([f.to(DEV) for f in feats], [c.to(DEV) for c in coords])
    # stated tolerances relative to the map maximum: bf16x3 5e-3, bf16 0.25; fp32x3 (round 4: fp32 storage, three bf16 MFMAs
    # per product, native executor) meets the PARITY bar of 1e-3
    print(mode, "prob err", rel_err(prob, g["prob"]), "reg err", rel_err(reg, g["reg"]))
    assert rel_err(prob, g["prob"]) < tol
    assert rel_err(reg, g["reg"]) < tol
    torch.autograd.backward([prob, reg], [torch.ones_like(prob) * 0.1, torch.ones_like(reg) * 0.1])
    for k, p in m.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
    from voxelnet_amd import model as M
    M.set_precision("bf16")


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


def test_car_full_backward():
    """Full-size car frame (B=1), fp32 mode: maps AND all 104 parameter gradients against the oracle run on
    this box's CPU cores (fp32, same inputs).  Bar for the maps: 1e-3 (BASELINE.json).  Bar for the chained
    gradients: relative L2 <= 0.1 — the reference's own gradients are not defined more tightly than that:
    its fp32 and fp64 runs (same torch ops, this frame) differ by 1.5e-3 / 1.1e-3 on the prob / reg maps and by
    0.08-0.22 relative L2 on the gradients upstream of the heads (the first Conv3d's BatchNorm normalises
    channels that are constant on 99 % of the sites, which amplifies fp32 rounding; DESIGN.md "Parity").
    Measured here: <= 0.06.  Tight gradient parity is asserted per layer in test_gpu_layers.py."""
    from oracle import voxelize as ov
    from voxelnet_amd import synth
    w = synth.WORKLOADS[1]
    cloud = synth.synth_cloud("Car", w["k0"], synth.frame_seed(1, 1), w["mean_extra"], w["T"])
    v = ov.voxelize(cloud, "Car")
    f, _, c = ov.prepare_voxel([v])
    feats, coords = [torch.from_numpy(f[0])], [torch.from_numpy(c[0])]
    rng = np.random.default_rng(77)
    dp = torch.from_numpy((rng.standard_normal((1, 2, 200, 176)) * 1e-2).astype(np.float32))
    dr = torch.from_numpy((rng.standard_normal((1, 14, 200, 176)) * 1e-2).astype(np.float32))
    rp, rr, ref = tr.forward_backward(feats, coords, tr.make_state_dict("Car"), (10, 400, 352), "Car", dp, dr)
    m = make_model("Car", mode="fp32")
    m.train()
    prob, reg = m.detect([feats[0].to(DEV)], [coords[0].to(DEV)])
    assert rel_err(prob, rp.numpy()) < 1e-3 and rel_err(reg, rr.numpy()) < 1e-3
    torch.autograd.backward([prob, reg], [dp.to(DEV), dr.to(DEV)])
    worst = (None, 0.0, 0.0)
    for k, p in m.named_parameters():
        r = ref[k].numpy()
        if k.endswith("conv.bias") and "prob_conv" not in k and "reg_conv" not in k or k.endswith("deconv.bias"):
            continue
        e = rel_err(p.grad, r)
        l2 = float(np.linalg.norm(p.grad.cpu().numpy().astype(np.float64) - r) / (np.linalg.norm(r) + 1e-30))
        if l2 > worst[2]:
            worst = (k, e, l2)
        print(f"{k:50s} max err {e:.2e}  rel L2 {l2:.2e}")
    print("worst gradient", worst)
    assert worst[2] < 0.1, worst


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


def test_sparse_first_layer_has_no_state_between_steps(golden):
    """The rulebook first layer works from per-step buffers in the workspace (voxel index grid, P rows): a different
    cloud in between must leave no trace: detect(A), detect(B), detect(A) gives bit-identical maps for A (train
    mode: batch statistics only)."""
    feats, coords = split(golden("middle_tiny_car"))
    feats, coords = [f.to(DEV) for f in feats], [c.to(DEV) for c in coords]
    m = make_model("Car", 16, 24, "bf16")
    m.train()
    with torch.no_grad():
        p1, r1 = m.detect(feats, coords)
        # B: the same voxels moved by one cell along x, and fewer of them
        feats_b = [f[: max(1, f.shape[0] // 2)] for f in feats]
        coords_b = [c[: max(1, c.shape[0] // 2)].clone() for c in coords]
        for c in coords_b:
            c[:, 3] = (c[:, 3] + 1) % 24
        m.detect(feats_b, coords_b)
        p3, r3 = m.detect(feats, coords)
    assert torch.equal(p1, p3) and torch.equal(r1, r3)


def test_reducer_path_gives_the_same_gradients(golden):
    """Bucketed path (parallel.GradAllReducer attached: gradients written into the flat buckets, per-group events,
    comm stream) == plain path, on one GPU (world size 1: no collective, same plumbing)."""
    from voxelnet_amd import parallel
    g = golden("rpn3d_tiny")
    feats, coords = split(golden("middle_tiny_car"))
    batch = (["a", "b"], None, feats, None, coords, None, None)
    grads = []
    for use_reducer in (False, True):
        m = make_model("Car", 16, 24, "fp32")
        m.train()
        named = list(m.named_parameters())
        if use_reducer:
            m.grad_reducer = parallel.GradAllReducer(named)
        out = m(batch, DEV, targets=(g["pos"], g["neg"], g["targets"]))
        out[2].backward()
        if use_reducer:
            m.grad_reducer.finish(named)
        torch.cuda.synchronize()
        grads.append({k: p.grad.detach().clone() for k, p in named})
    for k in grads[0]:
        a, b = grads[0][k], grads[1][k]
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7), (k, float((a - b).abs().max()))


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_train_step_on_a_grid_the_native_executor_does_not_take(mode):
    """A model with `native_executor = False` (or bf16x3, or a depth the executor does not plan): detect() must route the
    whole step — forward AND backward — through the per-layer orchestration with the parameters as autograd inputs (round 2
    picked the one-tensor 'anchor' call from a different predicate than the forward's path choice and the backward raised).
    On a D = 9 grid (the three Conv3d layers still fold to depth 2: model.py:207-209, 262).  fp32: maps against the oracle
    at 1e-3; both modes: every parameter receives a finite gradient through autograd (torch.autograd.grad works here)."""
    from oracle import voxelize as ov
    from voxelnet_amd import model as M
    from voxelnet_amd import synth
    from voxelnet_amd.config import grid_config
    g9 = grid_config("Car", D=9, H=16, W=24, oy=1.6)
    feats, coords = [], []
    for i in range(2):
        cloud = synth.synth_cloud("Car", k0=120 + 30 * i, seed=700 + i, grid=g9, overflow_frac=0.03)
        v = ov.voxelize(cloud, "Car", D=9, H=16, W=24, oy=1.6)
        feats.append(torch.from_numpy(v["feature_buffer"]))
        coords.append(torch.from_numpy(np.pad(v["coordinate_buffer"], ((0, 0), (1, 0)), constant_values=i)))
    M.set_precision(mode)
    m = M.RPN3D("Car")
    m.load_state_dict(tr.make_state_dict("Car"))
    m.feature_net._grid = g9
    m = m.to(DEV).train()
    m.native_executor = False                  # (round 4: the executor plans D = 9 ... 12 itself — test_native_executor_other_depths)
    assert not m._native_ok(mode)
    prob, reg = m.detect([f.to(DEV) for f in feats], [c.to(DEV) for c in coords])
    assert prob.shape == (2, 2, 8, 12)
    if mode == "fp32":
        with torch.no_grad():
            sd = tr.make_state_dict("Car")
            rp, rr = tr.middle_rpn(tr.feature_net(feats, coords, sd, (9, 16, 24), True), sd, "Car", True)
        assert rel_err(prob, rp.numpy()) < 1e-3 and rel_err(reg, rr.numpy()) < 1e-3
    params = list(m.parameters())
    grads = torch.autograd.grad(prob.square().mean() + reg.square().mean(), params)
    assert len(grads) == 104 and all(g is not None and torch.isfinite(g).all() for g in grads)
    assert sum(float(g.abs().sum()) for g in grads) > 0
    M.set_precision("bf16")


@pytest.mark.parametrize("D", [9, 11, 12])
def test_native_executor_other_depths(D):
    """The native executor on grids of depth 9, 11 and 12 (round 4: the D == 10 restriction of csrc/runtime.hip is gone;
    model.py:207-209 only needs the depth to fold to 2): fp32-mode maps against the CPU oracle at 1e-3, and the whole
    backward against the per-layer orchestration of the SAME kernels (native_executor = False) — maps within 1e-4, every
    parameter gradient within 5e-2 relative L2 (measured 2e-2 at the VFE end of the chain) (the two paths differ in the first layer's summation order — rulebook vs
    row-list — which the 23-layer BatchNorm / ReLU stack amplifies; the reference's own fp32 / fp64 gradients differ more)."""
    from oracle import voxelize as ov
    from voxelnet_amd import model as M
    from voxelnet_amd import synth
    from voxelnet_amd.config import grid_config
    g = grid_config("Car", D=D, H=16, W=24, oy=1.6)
    feats, coords = [], []
    for i in range(2):
        cloud = synth.synth_cloud("Car", k0=120 + 30 * i, seed=900 + i + D, grid=g, overflow_frac=0.03)
        v = ov.voxelize(cloud, "Car", D=D, H=16, W=24, oy=1.6)
        feats.append(torch.from_numpy(v["feature_buffer"]))
        coords.append(torch.from_numpy(np.pad(v["coordinate_buffer"], ((0, 0), (1, 0)), constant_values=i)))
    with torch.no_grad():
        sd = tr.make_state_dict("Car")
        rp, rr = tr.middle_rpn(tr.feature_net(feats, coords, sd, (D, 16, 24), True), sd, "Car", True)
    fd, cd = [f.to(DEV) for f in feats], [c.to(DEV) for c in coords]
    up = (torch.full((2, 2, 8, 12), 0.05, device=DEV), torch.full((2, 14, 8, 12), -0.03, device=DEV))
    try:
        M.set_precision("fp32")
        out = {}
        for native in (True, False):
            m = M.RPN3D("Car")
            m.load_state_dict(tr.make_state_dict("Car"))
            m.feature_net._grid = g
            m = m.to(DEV).train()
            m.native_executor = native
            assert m._native_ok("fp32") == native
            prob, reg = m.detect(fd, cd)
            torch.autograd.backward([prob, reg], list(up))
            out[native] = (prob.detach(), reg.detach(), {k: p.grad.detach().clone() for k, p in m.named_parameters()})
        pn, rn, gn = out[True]
        pf, rf, gf = out[False]
        assert rel_err(pn, rp.numpy()) < 1e-3 and rel_err(rn, rr.numpy()) < 1e-3
        em = (rel_err(pn, pf.cpu().numpy()), rel_err(rn, rf.cpu().numpy()))
        assert em[0] < 1e-4 and em[1] < 1e-4, em
        worst = max((float((gn[k] - gf[k]).norm() / gf[k].norm().clamp(min=1e-30)), k) for k in gf
                    if not (k.endswith("conv.bias") and "prob_conv" not in k and "reg_conv" not in k) and not k.endswith("deconv.bias"))
        print(f"D = {D}: native vs per-layer: maps {em[0]:.1e} / {em[1]:.1e}, gradients worst rel-L2 {worst}")
        assert worst[0] < 5e-2, worst        # (the chained-gradient bound of test_detect_fwd_bwd_fp32 on these tiny grids)
    finally:
        M.set_precision("bf16")


@pytest.mark.parametrize("B,empty", [(1, None), (3, 1), (5, 4)])
def test_odd_batches_and_an_empty_frame(B, empty):
    """Batch sizes the fixtures do not hold (1, 3, 5) with one frame of the batch EMPTY (every point outside the range
    crop: utils.py:63-88 then returns K = 0 buffers, and model.py:102-106 leaves that sample's dense grid at zero): the
    device voxelizer returns the same empty buffers as the oracle's, the fp32-mode maps meet the CPU oracle at 1e-3, and
    a backward through every precision mode gives finite gradients for all 104 parameters."""
    from oracle import voxelize as ov
    from voxelnet_amd import model as M
    from voxelnet_amd import synth
    from voxelnet_amd.config import grid_config
    from voxelnet_amd.voxelize import voxelize_device
    g = grid_config("Car", D=10, H=16, W=24, oy=1.6)
    feats, coords = [], []
    for i in range(B):
        cloud = synth.synth_cloud("Car", k0=90 + 25 * i, seed=1300 + 7 * B + i, grid=g, overflow_frac=0.03)
        if i == empty:
            cloud = cloud.copy()
            cloud[:, 0] = -500.0                                   # all behind the sensor: nothing survives the crop
        v = ov.voxelize(cloud, "Car", D=10, H=16, W=24, oy=1.6)
        fb, cb, _ = voxelize_device(torch.from_numpy(cloud).to(DEV), g, i, coord_cols=4)
        assert fb.shape[0] == v["feature_buffer"].shape[0] and (i != empty or fb.shape[0] == 0)
        feats.append(torch.from_numpy(v["feature_buffer"]))
        coords.append(torch.from_numpy(np.pad(v["coordinate_buffer"], ((0, 0), (1, 0)), constant_values=i)))
    with torch.no_grad():
        sd = tr.make_state_dict("Car")
        rp, rr = tr.middle_rpn(tr.feature_net(feats, coords, sd, (10, 16, 24), True), sd, "Car", True)
    assert rp.shape[0] == B
    fd, cd = [f.to(DEV) for f in feats], [c.to(DEV) for c in coords]
    up = (torch.full((B, 2, 8, 12), 0.05, device=DEV), torch.full((B, 14, 8, 12), -0.03, device=DEV))
    try:
        for mode, tol in (("fp32", 1e-3), ("fp32x3", 1e-3), ("bf16", 0.25)):
            M.set_precision(mode)
            m = M.RPN3D("Car")
            m.load_state_dict(tr.make_state_dict("Car"))
            m.feature_net._grid = g
            m = m.to(DEV).train()
            assert m._native_ok(mode)
            prob, reg = m.detect(fd, cd)
            torch.autograd.backward([prob, reg], list(up))
            e = (rel_err(prob, rp.numpy()), rel_err(reg, rr.numpy()))
            print(f"B = {B}, empty frame {empty}, {mode}: maps vs CPU oracle {e[0]:.1e} / {e[1]:.1e}")
            assert e[0] < tol and e[1] < tol, (mode, e)
            grads = [p.grad for p in m.parameters()]
            assert len(grads) == 104 and all(g_ is not None and torch.isfinite(g_).all() for g_ in grads)
            assert sum(float(g_.abs().sum()) for g_ in grads) > 0
    finally:
        M.set_precision("bf16")


def test_a_batch_without_any_voxel_is_refused():
    """K = 0 over the whole batch: the reference cannot run it either (train-mode BatchNorm1d over zero rows raises
    "Expected more than 1 value per channel", model.py:76); here the executor refuses the call with an error, it does not
    launch over empty row sets."""
    from voxelnet_amd import _lib
    from voxelnet_amd import model as M
    from voxelnet_amd.config import grid_config
    m = M.RPN3D("Car")
    m.load_state_dict(tr.make_state_dict("Car"))
    m.feature_net._grid = grid_config("Car", D=10, H=16, W=24, oy=1.6)
    m = m.to(DEV).train()
    with pytest.raises(_lib.VoxelnetHipError):
        m.detect([torch.zeros((0, 35, 7), device=DEV)], [torch.zeros((0, 4), dtype=torch.int64, device=DEV)])
    torch.cuda.synchronize()


def _hip_masks(st):
    """ReLU decisions of the HIP forward, per layer, in the oracle's NC(D)HW output shapes: mask = (a > 0) of the stored
    activation (per-layer orchestration state, voxelnet_amd/net.py)"""
    out = {}
    for name, s in st.layers.items():
        if name == "heads":
            continue
        a = s.a.t[..., :s.a.C].float()
        B = a.shape[0]
        if name == "middle_layer.2":                 # BEV rows (B,1,H,W,128), channel d*64 + c  ->  (B,64,2,H,W)
            H, W = a.shape[2], a.shape[3]
            m = a.reshape(B, H, W, 2, 64).permute(0, 4, 3, 1, 2)
        elif s.spec.dim == 3:
            m = a.permute(0, 4, 1, 2, 3)
        else:
            m = a[:, 0].permute(0, 3, 1, 2)
        out[name] = (m > 0).double().cpu().contiguous()
    return out


@pytest.mark.parametrize("cls,tag", [("Car", "car"), ("Pedestrian", "ped")])
def test_fp32_backward_chain_with_frozen_masks(golden, cls, tag):
    """The fp32 (parity-mode) backward as a CHAIN — heads -> 23 layers -> VFE — against the float64 oracle evaluated with
    the ReLU decisions of the HIP forward (oracle/torch_ref.middle_rpn(masks=...)).  test_detect_fwd_bwd_fp32 above has to
    allow 5e-2 on the chained gradients because a single ReLU-mask flip (an fp32 rounding event at z ~ 0, which also
    separates the reference's own fp32 and fp64 runs) moves every upstream gradient by 1e-2..1e-1; with the masks frozen
    that discontinuity is gone and what is left is fp32 accumulation noise through 23 BatchNorm layers.
    Bar: every one of the 104 gradients within 5e-4 relative L2 (measured 5.7e-5 car / 2.5e-5 ped), maps 1e-4."""
    from voxelnet_amd import model as M
    from voxelnet_amd import net as N
    g = golden(f"middle_tiny_{tag}")
    feats, coords = split(g)
    m = make_model(cls, 16, 24, "fp32")
    m.train()
    fn, mid = m.feature_net, m.middle_rpn
    feature = torch.cat([f.to(DEV) for f in feats], 0).contiguous()
    coord = torch.cat([c.to(DEV) for c in coords], 0).contiguous()
    vparams = [p.detach() for p in M._vfe_weights(fn)]
    names, P, Bf, flat = M._collect_middle(mid)
    P = M._detached(P)
    P["heads"] = M._heads_params([f.detach() for f in flat[-4:]])
    vw, stats, wst = M.featnet_forward(feature, vparams, fn._bufs(), True)
    dense = M.scatter_rows(vw, coord, len(feats), fn._grid.dims, "fp32")
    prob, reg, st = N.middle_forward(dense, P, Bf, mid._block1_stride, True, "fp32", sparse=(coord, vw))
    dp = torch.from_numpy((np.random.default_rng(41).standard_normal(g["prob"].shape) * 1e-1).astype(np.float32))
    dr = torch.from_numpy((np.random.default_rng(42).standard_normal(g["reg"].shape) * 1e-1).astype(np.float32))
    G, d_vw = N.middle_backward(st, dp.to(DEV), dr.to(DEV), P)
    vg = M.featnet_backward(feature, wst, stats, d_vw, vparams)
    torch.cuda.synchronize()
    masks = _hip_masks(st)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in tr.make_state_dict(cls).items()}
    rp, rr, ref = tr.forward_backward([f.double() for f in feats], coords, sd64, (10, 16, 24), cls, dp.double(), dr.double(),
                                      masks=masks)
    assert rel_err(prob, rp.float().numpy()) < 1e-4 and rel_err(reg, rr.float().numpy()) < 1e-4
    got = {}
    for n in names:
        cv = "deconv" if n.startswith("deconv") else "conv"
        got[f"middle_rpn.{n}.{cv}.weight"] = G[n]["weight"]
        got[f"middle_rpn.{n}.batch_norm.weight"] = G[n]["gamma"]
        got[f"middle_rpn.{n}.batch_norm.bias"] = G[n]["beta"]
    got["middle_rpn.prob_conv.conv.weight"], got["middle_rpn.reg_conv.conv.weight"] = G["heads"]["weight"][:2], G["heads"]["weight"][2:]
    got["middle_rpn.prob_conv.conv.bias"], got["middle_rpn.reg_conv.conv.bias"] = G["heads"]["bias"][:2], G["heads"]["bias"][2:]
    for key, t in zip(M.VFE_KEYS, vg):
        got[key] = t
    worst = ("", 0.0)
    for k, t in got.items():
        r = ref[k].double()
        l2 = float((t.double().cpu() - r).norm() / (r.norm() + 1e-30))
        if l2 > worst[1]:
            worst = (k, l2)
    print(f"frozen masks, fp32 chain ({cls}): worst gradient rel-L2", worst)
    assert worst[1] < 5e-4, worst
    M.set_precision("bf16")


def test_car_full_backward_frozen_masks():
    """The same frozen-mask chain at FULL size (one synthetic car frame, 10 x 400 x 352 grid, fp32 mode, per-layer
    orchestration with the sparse first layer) against the oracle in float32 on this box's CPU cores with the HIP
    forward's ReLU decisions: test_car_full_backward has to allow 0.1 on the chained gradients (measured <= 0.06: mask
    flips); with the masks frozen the fp32 chain agrees to the printed figure (both sides fp32: accumulation noise of two
    different summation orders through 23 BatchNorm layers).  Measured 2.7e-4; bar 2e-3."""
    from oracle import voxelize as ov
    from voxelnet_amd import model as M
    from voxelnet_amd import net as N
    from voxelnet_amd import synth
    w = synth.WORKLOADS[1]
    cloud = synth.synth_cloud("Car", w["k0"], synth.frame_seed(1, 1), w["mean_extra"], w["T"])
    v = ov.voxelize(cloud, "Car")
    f, _, c = ov.prepare_voxel([v])
    feats, coords = [torch.from_numpy(f[0])], [torch.from_numpy(c[0])]
    rng = np.random.default_rng(77)
    dp = torch.from_numpy((rng.standard_normal((1, 2, 200, 176)) * 1e-2).astype(np.float32))
    dr = torch.from_numpy((rng.standard_normal((1, 14, 200, 176)) * 1e-2).astype(np.float32))
    m = make_model("Car", mode="fp32")
    m.train()
    fn, mid = m.feature_net, m.middle_rpn
    feature, coord = feats[0].to(DEV).contiguous(), coords[0].to(DEV).contiguous()
    vparams = [p.detach() for p in M._vfe_weights(fn)]
    names, P, Bf, flat = M._collect_middle(mid)
    P = M._detached(P)
    P["heads"] = M._heads_params([t.detach() for t in flat[-4:]])
    vw, stats, wst = M.featnet_forward(feature, vparams, fn._bufs(), True)
    dense = M.scatter_rows(vw, coord, 1, fn._grid.dims, "fp32")
    prob, reg, st = N.middle_forward(dense, P, Bf, mid._block1_stride, True, "fp32", sparse=(coord, vw))
    G, d_vw = N.middle_backward(st, dp.to(DEV), dr.to(DEV), P)
    vg = M.featnet_backward(feature, wst, stats, d_vw, vparams)
    torch.cuda.synchronize()
    masks = {k: t.float() for k, t in _hip_masks(st).items()}
    del st, dense
    torch.cuda.empty_cache()
    rp, rr, ref = tr.forward_backward(feats, coords, tr.make_state_dict("Car"), (10, 400, 352), "Car", dp, dr, masks=masks)
    assert rel_err(prob, rp.numpy()) < 1e-3 and rel_err(reg, rr.numpy()) < 1e-3
    got = {}
    for n in names:
        cv = "deconv" if n.startswith("deconv") else "conv"
        got[f"middle_rpn.{n}.{cv}.weight"] = G[n]["weight"]
        got[f"middle_rpn.{n}.batch_norm.weight"] = G[n]["gamma"]
        got[f"middle_rpn.{n}.batch_norm.bias"] = G[n]["beta"]
    got["middle_rpn.prob_conv.conv.weight"], got["middle_rpn.reg_conv.conv.weight"] = G["heads"]["weight"][:2], G["heads"]["weight"][2:]
    for key, t in zip(M.VFE_KEYS, vg):
        got[key] = t
    worst = ("", 0.0)
    for k, t in got.items():
        r = ref[k].double()
        l2 = float((t.double().cpu() - r).norm() / (r.norm() + 1e-30))
        if l2 > worst[1]:
            worst = (k, l2)
    print("frozen masks, fp32 chain at full size: worst gradient rel-L2", worst)
    assert worst[1] < 2e-3, worst
    M.set_precision("bf16")
