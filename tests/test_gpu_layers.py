"""GPU: every ConvMD / DeConv2d variant of the reference (model.py:206-254) through the
C ABI (gather-GEMM, wgrad, BatchNorm kernels) against the golden vectors produced by the
imported reference modules and against the oracle (oracle/torch_ref.py, CPU fp32).

Tolerances (relative to the tensor's maximum): "fp32" (exact fp32 MFMA products; the parity
mode for BASELINE.json's <=1e-3 bar) 2e-4 per layer, "bf16x3" 1e-3.  The "bf16" mode (the benchmarked
one) is checked stage by stage against a bf16-operand oracle with half-ulp bars in
tests/test_gpu_bf16_parity.py (these same fixtures + the production tile sizes)."""
import numpy as np
import pytest
import torch

from layer_cases import LAYER_CASES, build_layer_state, layer_input, layer_upstream

pytestmark = pytest.mark.gpu

TOL = {"fp32": 2e-4, "bf16x3": 1e-3, "fp32x3": 1e-3, "bf16": 8e-3}   # bf16: heads only (fp32 output, bf16 operands vs the fp32 reference)


def rel_err(a, b):
    a = a.detach().float().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = max(np.abs(b).max(), 1e-6)
    return float(np.abs(a - b).max() / scale)


def robust_err(a, b, q=95.0):
    """q-th percentile of |a-b| over max|b|: the backward of ReLU(BN(.)) is discontinuous at z = 0, so a
    pre-activation within rounding distance of 0 may flip one mask element and move a handful of dx values
    (one flipped element reaches up to taps x Cin of them, ~1.4 % here) by a whole term; the percentile ignores
    those and still catches any systematic error."""
    a = a.detach().float().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.percentile(np.abs(a - b), q) / max(np.abs(b).max(), 1e-6))


def act_nchw(a, dim):
    """activation Rows -> NC(D)HW fp32 (hi + lo in split layout)"""
    from voxelnet_amd import engine as E
    C = a.C
    y = E.rows_to_nchw(E.Rows(a.t[..., :C], C), dim)
    if a.lo_off:
        y = y + E.rows_to_nchw(E.Rows(a.t[..., a.lo_off:a.lo_off + C], C), dim)
    return y


def _spec(case):
    from voxelnet_amd.engine import LayerSpec
    name, kind, dim, cin, cout, k, s, p, sp = case
    if dim == 3:
        return LayerSpec(name, 3, cin, cout, (k, k, k), tuple(s), tuple(p))
    return LayerSpec(name, 2, cin, cout, (1, k, k), (1,) + tuple(s), (0,) + tuple(p), transposed=(kind == "deconv"),
                     bn=(kind != "head"), relu=(kind != "head"))


# (the two heads, 2 and 14 output channels, run fused as one N = 16 product: test_heads_fused, not through this per-layer path)
PER_LAYER_CASES = [c for c in LAYER_CASES if not (c[1] == "head" and c[4] % 4)]


@pytest.fixture
def x3_flag():
    """"fp32x3" is "fp32" storage plus the operand dtype VN_F32X3 in every conv geometry (engine.X3, set by set_precision)"""
    from voxelnet_amd import engine as E
    yield E.X3
    E.X3["on"] = False


@pytest.mark.parametrize("mode", ["fp32", "fp32x3"])
@pytest.mark.parametrize("case", PER_LAYER_CASES, ids=lambda c: c[0])
def test_layer_fwd_bwd(golden, case, mode, x3_flag):
    from voxelnet_amd import engine as E
    x3_flag["on"] = mode == "fp32x3"
    g = golden("layers_tiny")
    name, kind, dim, cin, cout, k, s, p, sp = case
    idx = [c[0] for c in LAYER_CASES].index(name)
    tol = TOL[mode]
    spec = _spec(case)
    sd = build_layer_state(idx, case)
    wkey = "deconv" if kind == "deconv" else "conv"
    dev = torch.device("cuda:0")
    P = {"weight": sd[f"L.{wkey}.weight"].to(dev), "bias": sd[f"L.{wkey}.bias"].to(dev)}
    Bf = None
    if kind != "head":
        P["gamma"], P["beta"] = sd["L.batch_norm.weight"].to(dev), sd["L.batch_norm.bias"].to(dev)
        Bf = {"running_mean": sd["L.batch_norm.running_mean"].to(dev),
              "running_var": sd["L.batch_norm.running_var"].to(dev)}
    x = layer_input(idx, case).to(dev)
    xr = E.nchw_to_rows(x, mode)
    a, st = E.layer_forward(spec, xr, P, Bf, True, mode, y_dtype=torch.float32 if kind == "head" else None)
    if kind == "head":
        y = E.rows_to_nchw(a, dim)
    else:
        y = act_nchw(a, dim)
    assert rel_err(y, g[name + ".y"]) < tol
    up = layer_upstream(idx, case, tuple(y.shape)).to(dev)
    if kind == "head":
        da = E.nchw_to_rows(up, mode)                       # conv-output gradient in the mode's row format
    else:
        da = E.nchw_to_plain_rows(up, E.plain_dtype_of(mode))
    grads, dx = E.layer_backward(st, da, P, mode)
    # backward checker: the oracle on CPU with the SAME ReLU mask as the kernels used (see torch_ref._relu)
    from oracle import torch_ref as tr
    mask = (y > 0).cpu() if kind != "head" else None
    leaves = {k_: v.clone().requires_grad_(True) for k_, v in sd.items() if "running" not in k_ and "num_batches" not in k_}
    work = dict(sd); work.update(leaves)
    xc = layer_input(idx, case).requires_grad_(True)
    if kind == "deconv":
        yo = tr.deconv2d(xc, work, "L", s, p, True, relu_mask=mask)
    else:
        yo = tr.conv_md(xc, work, "L", dim, s, p, bn=(kind == "conv"), act=(kind == "conv"), training=True,
                        relu_mask=mask)
    yo.backward(layer_upstream(idx, case, tuple(yo.shape)))
    assert rel_err(E.rows_to_nchw(dx, dim), xc.grad.numpy()) < tol
    if mode == "fp32":   # and the golden dx of the imported reference (robust to an isolated mask flip)
        assert robust_err(E.rows_to_nchw(dx, dim), g[name + ".dx"]) < tol
    ref_w = leaves[f"L.{wkey}.weight"].grad.numpy()
    assert rel_err(grads["weight"], ref_w) < tol
    ref_b = leaves[f"L.{wkey}.bias"].grad.numpy()
    if kind == "head":
        assert rel_err(grads["bias"], ref_b) < tol
    else:
        # conv bias feeding a train-mode BatchNorm: the true gradient is 0 (rounding noise in the reference)
        assert float(grads["bias"].abs().max()) < 1e-2 * max(1.0, float(np.abs(up.cpu().numpy()).sum()) * 1e-3)
        for key, ref_k in (("gamma", "L.batch_norm.weight"), ("beta", "L.batch_norm.bias")):
            ref_g = leaves[ref_k].grad.numpy()
            assert rel_err(grads[key], ref_g) < max(tol, 1e-3)
        for k_ in ("running_mean", "running_var"):
            assert rel_err(Bf[k_], g[f"{name}.buf.batch_norm.{k_}"]) < 1e-3


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_heads_fused(mode):
    """prob_conv + reg_conv (model.py:253-254,276-281) as one N=16 GEMM with the sigmoid epilogue."""
    from oracle import torch_ref as tr
    from voxelnet_amd import engine as E
    from voxelnet_amd.net import HEADS
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.standard_normal((2, 768, 8, 12)).astype(np.float32))
    w = tr._fill((16, 768, 1, 1), 77, 1.0 / np.sqrt(768))
    b = tr._fill((16,), 78, 0.1)
    P = {"weight": w.to(dev), "bias": b.to(dev)}
    y, st = E.layer_forward(HEADS, E.nchw_to_rows(x.to(dev), mode), P, None, True, mode, y_dtype=torch.float32)
    prob = E.rows_to_nchw(E.Rows(y.t[..., 0:2], 2), 2, sigmoid_first_n=2)
    reg = E.rows_to_nchw(E.Rows(y.t[..., 2:16], 14), 2)
    ref = torch.nn.functional.conv2d(x, w, b)
    assert rel_err(prob, torch.sigmoid(ref[:, :2]).numpy()) < TOL[mode]
    assert rel_err(reg, ref[:, 2:].numpy()) < TOL[mode]


@pytest.mark.parametrize("mode", ["fp32"])
def test_bev_fold_and_strided_views(mode):
    """middle_layer.2 -> BEV reshape (model.py:262) -> block1.0, against the oracle."""
    from oracle import torch_ref as tr
    from voxelnet_amd import engine as E
    from voxelnet_amd.net import layer_table
    dev = torch.device("cuda:0")
    specs = dict(layer_table(2))
    sd = tr.make_state_dict("Car")
    rng = np.random.default_rng(11)
    x = torch.from_numpy(rng.standard_normal((2, 64, 3, 8, 12)).astype(np.float32))

    def params(name):
        pre = "middle_rpn." + name
        return ({"weight": sd[pre + ".conv.weight"].to(dev), "bias": sd[pre + ".conv.bias"].to(dev),
                 "gamma": sd[pre + ".batch_norm.weight"].to(dev), "beta": sd[pre + ".batch_norm.bias"].to(dev)},
                {"running_mean": sd[pre + ".batch_norm.running_mean"].clone().to(dev),
                 "running_var": sd[pre + ".batch_norm.running_var"].clone().to(dev)})

    P3, B3 = params("middle_layer.2")
    P1, B1 = params("block1.0")
    a3, st3 = E.layer_forward(specs["middle_layer.2"], E.nchw_to_rows(x.to(dev), mode), P3, B3, True, mode, bev_out=True)
    a1, st1 = E.layer_forward(specs["block1.0"], a3, P1, B1, True, mode)
    out = act_nchw(a1, 2)
    xs = x.clone().requires_grad_(True)
    lv = {k: v.clone().requires_grad_(True) for k, v in sd.items()
          if (k.startswith("middle_rpn.middle_layer.2") or k.startswith("middle_rpn.block1.0"))
          and "running" not in k and "num_batches" not in k}
    work = dict(tr.make_state_dict("Car")); work.update(lv)
    r3 = tr.conv_md(xs, work, "middle_rpn.middle_layer.2", 3, (2, 1, 1), (1, 1, 1))
    r1 = tr.conv_md(r3.reshape(2, -1, 8, 12), work, "middle_rpn.block1.0", 2, (2, 2), (1, 1))
    assert rel_err(out, r1.detach().numpy()) < 1e-3
    up = torch.from_numpy(rng.standard_normal(tuple(r1.shape)).astype(np.float32))
    r1.backward(up)
    g1, d_bev = E.layer_backward(st1, E.nchw_to_plain_rows(up.to(dev), torch.float32), P1, mode)
    g3, dx = E.layer_backward(st3, d_bev, P3, mode, bev_da=True)
    assert rel_err(g1["weight"], lv["middle_rpn.block1.0.conv.weight"].grad.numpy()) < 1e-3
    assert rel_err(g3["weight"], lv["middle_rpn.middle_layer.2.conv.weight"].grad.numpy()) < 1e-3
    assert rel_err(g3["gamma"], lv["middle_rpn.middle_layer.2.batch_norm.weight"].grad.numpy()) < 1e-3
    assert rel_err(E.rows_to_nchw(dx, 3), xs.grad.numpy()) < 1e-3


def test_batched_pack_unpack_matches_single_calls():
    """vn_pack_weights_batch / vn_unpack_wgrads_batch == the per-layer calls, bit for bit (all four modes, BEV fold)"""
    from voxelnet_amd import _lib, engine as E
    DEV = "cuda:0"
    torch.manual_seed(5)
    cases = [(64, 128, 27, 0, 1), (64, 128, 27, 1, 1), (256, 128, 4, 2, 1), (256, 128, 4, 3, 1), (128, 128, 9, 0, 2),
             (128, 128, 9, 1, 2), (16, 768, 1, 0, 1),
             # shapes that do not divide into the batched kernel's LDS tiles (2 rows; 32 x 8), the fold in the row-tile modes
             (50, 36, 9, 1, 1), (70, 20, 27, 2, 1), (3, 8, 16, 3, 1), (130, 128, 9, 3, 2), (5, 40, 1, 0, 1), (16, 768, 1, 1, 1)]
    ws, singles, jobs = [], [], (_lib.VnPackJob * len(cases))()
    batch = []
    for i, (co, ci, taps, mode, fold) in enumerate(cases):
        w = torch.randn(co * ci * taps, device=DEV)
        a = torch.zeros(co * ci * taps, dtype=torch.bfloat16, device=DEV)
        b = torch.zeros_like(a)
        _lib.call("vn_pack_weight", w.data_ptr(), co, ci, taps, mode, 0, fold, a.data_ptr(), _lib.VN_BF16, E.stream())
        jobs[i] = _lib.VnPackJob(w.data_ptr(), b.data_ptr(), co, ci, taps, mode, 0, fold, _lib.VN_BF16, 0)
        ws.append(w); singles.append(a); batch.append(b)
    _lib.call("vn_pack_weights_batch", jobs, len(cases), E.stream())
    for a, b in zip(singles, batch):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    ucases = [c for c in cases if c[3] in (0, 2)]
    ujobs, us, ub = (_lib.VnUnpackJob * len(ucases))(), [], []
    keep = []
    for i, (co, ci, taps, mode, fold) in enumerate(ucases):
        dwp = torch.randn(co * ci * taps, device=DEV)
        a, b = torch.zeros_like(dwp), torch.zeros_like(dwp)
        _lib.call("vn_unpack_wgrad", dwp.data_ptr(), co, ci, taps, mode, fold, a.data_ptr(), E.stream())
        ujobs[i] = _lib.VnUnpackJob(dwp.data_ptr(), b.data_ptr(), co, ci, taps, mode, fold, 0)
        keep.append(dwp); us.append(a); ub.append(b)
    _lib.call("vn_unpack_wgrads_batch", ujobs, len(ucases), E.stream())
    for a, b in zip(us, ub):
        assert torch.equal(a, b)


@pytest.mark.parametrize("co,ci,taps,mode,fold,chunks", [
    (50, 36, 9, 0, 1, 1), (70, 20, 27, 2, 1, 1), (3, 8, 16, 0, 1, 3), (17, 4, 1, 2, 1, 2), (128, 256, 16, 0, 1, 11),
    (64, 64, 27, 2, 1, 20), (130, 128, 9, 0, 2, 9), (16, 768, 1, 0, 1, 17)])
def test_unpack_tiles_edge_shapes(co, ci, taps, mode, fold, chunks):
    """k_unpack_wgrads_batch works on tiles of the torch layout (whole rows in mode 0, 16 x 16 x taps in mode 2): shapes that do
    not divide into tiles, one to twenty row chunks (the eight-in-flight sum and its remainder loop), the BEV fold — against
    the index map written out in numpy (csrc/layout.hip torch_index), bit for bit (the chunk sum in the same order)."""
    from voxelnet_amd import _lib, engine as E
    DEV = "cuda:0"
    rng = np.random.default_rng(co * 1000 + ci + taps)
    n = co * ci * taps
    stride = n + 64
    parts = rng.standard_normal((chunks, stride)).astype(np.float32)
    acc = parts[0, :n].copy()
    for c in range(1, chunks):
        acc = acc + parts[c, :n]                        # fp32, in order
    packed = acc.reshape(taps, co, ci)                  # [tap][n][k]
    k = np.arange(ci)
    cif = (k % (ci // fold)) * fold + k // (ci // fold) if fold > 1 else k
    want = np.zeros(n, dtype=np.float32)
    if mode == 0:                                       # torch[(n * ci + fold(k)) * taps + tap]
        w3 = want.reshape(co, ci, taps)
        w3[:, cif, :] = packed.transpose(1, 2, 0)
    else:                                               # torch[(k * co + n) * taps + tap]
        w3 = want.reshape(ci, co, taps)
        w3[:, :, :] = packed.transpose(2, 1, 0)
    dwp = torch.from_numpy(parts.reshape(-1)).to(DEV)
    out = torch.full((n,), float("nan"), device=DEV)
    jobs = (_lib.VnUnpackJob * 1)()
    jobs[0] = _lib.VnUnpackJob(dwp.data_ptr(), out.data_ptr(), co, ci, taps, mode, fold, chunks, stride)
    _lib.call("vn_unpack_wgrads_batch", jobs, 1, E.stream())
    assert np.array_equal(out.cpu().numpy(), want)


def test_unpack_sums_chunk_partials():
    """vnUnpackJob.chunks: dw = sum over the row-chunk partials (vn_conv_wgrad_partials' layout), fixed order"""
    from voxelnet_amd import _lib, engine as E
    DEV = "cuda:0"
    torch.manual_seed(6)
    co, ci, taps, chunks = 128, 64, 9, 5
    n = co * ci * taps
    part = torch.randn(chunks, n, device=DEV)
    out = torch.zeros(n, device=DEV)
    jobs = (_lib.VnUnpackJob * 1)()
    jobs[0] = _lib.VnUnpackJob(part.data_ptr(), out.data_ptr(), co, ci, taps, 0, 1, chunks, n)
    _lib.call("vn_unpack_wgrads_batch", jobs, 1, E.stream())
    ref = torch.zeros(n, device=DEV)
    acc = part[0].clone()
    for c in range(1, chunks):
        acc = acc + part[c]
    _lib.call("vn_unpack_wgrad", acc.data_ptr(), co, ci, taps, 0, 1, ref.data_ptr(), E.stream())
    assert torch.equal(out, ref)


def test_scatter_update_and_clear_keep_the_grid_zero():
    """vn_scatter_dense_update == vn_scatter_dense_fwd on an all-zero grid, and update(NULL) restores all-zero"""
    from voxelnet_amd import _lib, engine as E
    DEV = "cuda:0"
    torch.manual_seed(7)
    B, D, H, W, C, K = 2, 3, 8, 8, 128, 40
    cells = torch.randperm(B * D * H * W)[:K]
    coord = torch.stack([cells // (D * H * W), (cells // (H * W)) % D, (cells // W) % H, cells % W], 1).to(DEV)
    vw = torch.randn(K, C, device=DEV)
    ref = torch.full((B, D, H, W, C), 7.0, dtype=torch.bfloat16, device=DEV)          # fwd overwrites everything
    _lib.call("vn_scatter_dense_fwd", vw.data_ptr(), coord.data_ptr(), K, C, B, D, H, W, ref.data_ptr(), _lib.VN_BF16, C, 0,
              E.stream())
    grid = torch.zeros((B, D, H, W, C), dtype=torch.bfloat16, device=DEV)
    _lib.call("vn_scatter_dense_update", vw.data_ptr(), coord.data_ptr(), K, C, B, D, H, W, grid.data_ptr(), _lib.VN_BF16, C, 0,
              E.stream())
    assert torch.equal(grid.view(torch.int16), ref.view(torch.int16))
    _lib.call("vn_scatter_dense_update", None, coord.data_ptr(), K, C, B, D, H, W, grid.data_ptr(), _lib.VN_BF16, C, 0,
              E.stream())
    assert int(grid.view(torch.int16).ne(0).sum()) == 0


def test_bn_bwd_apply_flagged_skips_unflagged_rows():
    """rows with flag 0 are left untouched, flagged rows equal vn_bn_bwd_apply"""
    from voxelnet_amd import _lib, engine as E
    DEV = "cuda:0"
    torch.manual_seed(8)
    M, C = 1000, 64
    da = torch.randn(M, C, device=DEV).to(torch.bfloat16)
    y = torch.randn(M, C, device=DEV).to(torch.bfloat16)
    stats = torch.randn(4 * C, device=DEV)
    stats[C:2 * C].abs_()
    coef = torch.randn(3 * C, device=DEV)
    flags = (torch.rand(M, device=DEV) < 0.3).to(torch.uint8)
    full = torch.empty(M, C, dtype=torch.bfloat16, device=DEV)
    _lib.call("vn_bn_bwd_apply", da.data_ptr(), _lib.VN_BF16, C, y.data_ptr(), _lib.VN_BF16, C, M, C, stats.data_ptr(),
              coef.data_ptr(), 1, full.data_ptr(), _lib.VN_BF16, C, 0, E.stream())
    part = torch.full((M, C), 5.0, dtype=torch.bfloat16, device=DEV)
    _lib.call("vn_bn_bwd_apply_flagged", da.data_ptr(), _lib.VN_BF16, C, y.data_ptr(), _lib.VN_BF16, C, M, C,
              stats.data_ptr(), coef.data_ptr(), 1, part.data_ptr(), _lib.VN_BF16, C, flags.data_ptr(), E.stream())
    f = flags.bool()
    assert torch.equal(part[f].view(torch.int16), full[f].view(torch.int16))
    assert bool((part[~f] == 5.0).all())


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_first_layer_sparse_batchnorm_passes_equal_the_dense_ones(dt):
    """vn_bn_apply_flagged / vn_bn_bwd_reduce_slab_flagged / vn_bn_bwd_apply_list (first middle layer: rows without an
    occupied voxel in reach hold the conv bias and are not read) == the dense passes, bit for bit"""
    from voxelnet_amd import _lib, engine as E
    DEV = "cuda:0"
    lib = _lib.load()
    torch.manual_seed(9)
    B, D, H, W, C = 2, 3, 20, 24, 64
    M = B * D * H * W
    vdt = _lib.VN_BF16 if dt == torch.bfloat16 else _lib.VN_F32
    bias = torch.randn(C, device=DEV) * 0.3
    flags = (torch.rand(M, device=DEV) < 0.15).to(torch.uint8)
    y = (torch.randn(M, C, device=DEV) * 1.5).to(dt)
    y[flags == 0] = bias.to(dt)                      # what the first layer's bias fill leaves at the inactive sites
    da = torch.randn(M, C, device=DEV).to(dt)
    stats = torch.cat([torch.randn(C) * 0.2, torch.rand(C) + 0.5, torch.rand(C) + 0.5, torch.randn(C) * 0.2]).to(DEV)
    coef = torch.randn(3 * C, device=DEV)
    a0, a1 = torch.empty_like(y), torch.empty_like(y)
    _lib.call("vn_bn_apply", y.data_ptr(), vdt, C, M, C, stats.data_ptr(), 1, a0.data_ptr(), vdt, C, 0, E.stream())
    _lib.call("vn_bn_apply_flagged", y.data_ptr(), vdt, C, M, C, stats.data_ptr(), 1, a1.data_ptr(), vdt, C, flags.data_ptr(),
              bias.data_ptr(), E.stream())
    assert torch.equal(a0, a1)
    rows = lib.vn_bn_bwd_slab_rows(M, C)
    s0 = torch.empty((rows, 2, C), device=DEV)
    s1 = torch.empty_like(s0)
    _lib.call("vn_bn_bwd_reduce_slab", da.data_ptr(), vdt, C, y.data_ptr(), vdt, C, M, C, stats.data_ptr(), 1, s0.data_ptr(),
              E.stream())
    _lib.call("vn_bn_bwd_reduce_slab_flagged", da.data_ptr(), vdt, C, y.data_ptr(), vdt, C, M, C, stats.data_ptr(), 1,
              s1.data_ptr(), flags.data_ptr(), bias.data_ptr(), E.stream())
    assert torch.equal(s0, s1)
    full = torch.empty_like(y)
    _lib.call("vn_bn_bwd_apply", da.data_ptr(), vdt, C, y.data_ptr(), vdt, C, M, C, stats.data_ptr(), coef.data_ptr(), 1,
              full.data_ptr(), vdt, C, 0, E.stream())
    idx = torch.nonzero(flags).flatten()
    lst = torch.stack([idx // (D * H * W), (idx // (H * W)) % D, (idx // W) % H, idx % W], 1).contiguous()
    cap = int(lst.shape[0]) + 37                     # capacity launch: entries past *count are ignored
    lst = torch.cat([lst, torch.zeros((37, 4), dtype=torch.int64, device=DEV)])
    cnt = torch.tensor([int(idx.numel())], dtype=torch.int32, device=DEV)
    part = torch.full_like(y, 5.0)
    _lib.call("vn_bn_bwd_apply_list", da.data_ptr(), vdt, y.data_ptr(), vdt, C, D, H, W, stats.data_ptr(), coef.data_ptr(), 1,
              part.data_ptr(), vdt, lst.data_ptr(), cnt.data_ptr(), cap, E.stream())
    f = flags.bool()
    assert torch.equal(part[f], full[f]) and bool((part[~f] == 5.0).all())


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_first_layer_batchnorm_backward_from_the_active_sites_only(dt):
    """vn_dgrad_total + vn_conv_gather_gemm_rows(out_linear) + vn_bn_bwd_reduce_list / _finalize_list / _apply_list_rows: the
    first middle layer's BatchNorm backward from middle_layer.1's data gradient at the ACTIVE sites only and box sums of
    middle_layer.1's dy, against the dense route (dense 3x3x3 data gradient -> dense reduce -> finalize -> apply)."""
    import ctypes
    import torch.nn.functional as F
    from voxelnet_amd import _lib, engine as E, net as N
    DEV = "cuda:0"
    lib = _lib.load()
    torch.manual_seed(11)
    mode = "bf16" if dt == torch.bfloat16 else "fp32"
    vdt = _lib.VN_BF16 if dt == torch.bfloat16 else _lib.VN_F32
    B, D0, H, W, C = 2, 5, 20, 24, 64                # a0 grid (middle_layer.0's output); middle_layer.1: D 5 -> 3
    D1 = D0 - 2
    M0 = B * D0 * H * W
    sp = dict(N.layer_table(2))["middle_layer.1"]
    assert sp.k == (3, 3, 3) and sp.stride == (1, 1, 1) and sp.pad == (0, 1, 1) and sp.cin == 64 and sp.cout == 64
    w = torch.randn(64, 64, 3, 3, 3, device=DEV) * 0.05
    wq = w.to(dt).float()                            # what the data-gradient kernels read
    dy1 = (torch.randn(B, D1, H, W, 64, device=DEV) * 0.5).to(dt)
    # ---- total over all sites of the data gradient, against conv_transpose3d in float64
    ws_bytes = lib.vn_dgrad_total_workspace_bytes(64)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=DEV)
    tot = torch.empty(64, device=DEV)
    _lib.call("vn_dgrad_total", dy1.data_ptr(), vdt, B, D1, H, W, 64, 64, 3, w.data_ptr(), 0, ws.data_ptr(), ws_bytes,
              tot.data_ptr(), E.stream())
    da_ref = F.conv_transpose3d(dy1.double().permute(0, 4, 1, 2, 3), wq.double(), stride=1, padding=(0, 1, 1))   # (B,64,5,H,W)
    assert da_ref.shape == (B, 64, D0, H, W)
    tot_ref = da_ref.sum(dim=(0, 2, 3, 4))
    scale = float(da_ref.abs().sum(dim=(0, 2, 3, 4)).max())
    assert float((tot.double() - tot_ref).abs().max()) < 2e-6 * scale
    # dy_sums_to_zero: the same with the per-channel mean of dy removed first (what a BatchNorm backward's output looks like)
    dyz = (dy1.float() - dy1.float().mean(dim=(0, 1, 2, 3))).to(dt)
    totz = torch.empty(64, device=DEV)
    _lib.call("vn_dgrad_total", dyz.data_ptr(), vdt, B, D1, H, W, 64, 64, 3, w.data_ptr(), 1, ws.data_ptr(), ws_bytes,
              totz.data_ptr(), E.stream())
    daz = F.conv_transpose3d(dyz.double().permute(0, 4, 1, 2, 3), wq.double(), stride=1, padding=(0, 1, 1))
    # (the rounded dyz sums to rounding noise, not to zero: that noise is what the flag drops; bound = its size through the taps)
    noise = float(dyz.double().sum(dim=(0, 1, 2, 3)).abs().max()) * 27 * 64 * float(wq.abs().max())
    assert float((totz.double() - daz.sum(dim=(0, 2, 3, 4))).abs().max()) < 2e-6 * scale + noise
    # ---- the dense route
    x_dy = E.Rows(dy1, 64)
    wpd = E.pack_weight(w, sp, 1, mode)
    dense = E.Rows(torch.empty(B, D0, H, W, 64, dtype=dt, device=DEV), 64)
    b = ((1, 1, 1), (-1, -1, -1), tuple(-p for p in sp.pad), sp.stride)
    E.gather_gemm(x_dy, wpd, None, dense, sp.k, 64, 64, *b, (D0, H, W))
    da = dense.t.reshape(M0, 64)
    assert float((da.double() - da_ref.permute(0, 2, 3, 4, 1).reshape(M0, 64)).abs().max()) < (3e-2 if dt == torch.bfloat16 else 1e-4)
    bias = torch.randn(C, device=DEV) * 0.3
    flags = (torch.rand(M0, device=DEV) < 0.12).to(torch.uint8)
    y = (torch.randn(M0, C, device=DEV) * 1.5).to(dt)
    y[flags == 0] = bias.to(dt)
    stats = torch.cat([torch.randn(C) * 0.2, torch.rand(C) + 0.5, torch.rand(C) + 0.5, torch.randn(C) * 0.2]).to(DEV)
    gamma = torch.rand(C, device=DEV) + 0.5
    rows = lib.vn_bn_bwd_slab_rows(M0, C)
    s0 = torch.empty((rows, 2, C), device=DEV)
    _lib.call("vn_bn_bwd_reduce_slab", da.data_ptr(), vdt, C, y.data_ptr(), vdt, C, M0, C, stats.data_ptr(), 1, s0.data_ptr(),
              E.stream())
    coef0, dg0, db0 = torch.empty(3 * C, device=DEV), torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    _lib.call("vn_bn_bwd_finalize_slab", s0.data_ptr(), rows, M0, C, gamma.data_ptr(), stats.data_ptr(), coef0.data_ptr(),
              dg0.data_ptr(), db0.data_ptr(), E.stream())
    full = torch.empty_like(y)
    _lib.call("vn_bn_bwd_apply", da.data_ptr(), vdt, C, y.data_ptr(), vdt, C, M0, C, stats.data_ptr(), coef0.data_ptr(), 1,
              full.data_ptr(), vdt, C, 0, E.stream())
    # ---- the list route
    idx = torch.nonzero(flags).flatten()
    n = int(idx.numel())
    lst = torch.stack([idx // (D0 * H * W), (idx // (H * W)) % D0, (idx // W) % H, idx % W], 1).contiguous()
    cap = n + 53
    lst = torch.cat([lst, torch.zeros((53, 4), dtype=torch.int64, device=DEV)])
    cnt = torch.tensor([n], dtype=torch.int32, device=DEV)
    dac = torch.full((cap, 64), 7.0, dtype=dt, device=DEV)
    g = E._geom(B, x_dy, (D0, H, W), 64, 0, 64, sp.k, *b, (0, 0, 0, 64))
    _lib.call("vn_conv_gather_gemm_rows", dy1.data_ptr(), wpd.data_ptr(), None, dac.data_ptr(), vdt, ctypes.byref(g), lst.data_ptr(),
              cap, cnt.data_ptr(), 1, None, E.stream())
    assert torch.equal(dac[:n], da[idx])             # same kernel arithmetic on the same rows
    lrows = lib.vn_bn_bwd_list_slab_rows(cap, C)
    s1 = torch.empty((lrows, 3, C), device=DEV)
    _lib.call("vn_bn_bwd_reduce_list", dac.data_ptr(), vdt, y.data_ptr(), vdt, C, D0, H, W, stats.data_ptr(), 1, s1.data_ptr(),
              lst.data_ptr(), cnt.data_ptr(), cap, E.stream())
    coef1, dg1, db1 = torch.empty(3 * C, device=DEV), torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    # (total of the kernel-produced dense gradient for the comparison with the dense route: isolates the list kernels from the
    #  rounding of the dense gradient's elements; the box-sum total was checked against float64 above)
    tot_k = da.double().sum(0).float()
    _lib.call("vn_bn_bwd_finalize_list", s1.data_ptr(), lrows, M0, C, gamma.data_ptr(), stats.data_ptr(), tot_k.data_ptr(),
              bias.data_ptr(), vdt, 1, coef1.data_ptr(), dg1.data_ptr(), db1.data_ptr(), E.stream())
    sc = float(da.float().abs().sum(0).max())
    assert float((dg1 - dg0).abs().max()) < 2e-5 * sc * 3 and float((db1 - db0).abs().max()) < 2e-5 * sc
    assert torch.allclose(coef1, coef0, rtol=1e-4, atol=2e-5 * sc / M0)
    part = torch.full_like(y, 5.0)
    _lib.call("vn_bn_bwd_apply_list_rows", dac.data_ptr(), vdt, y.data_ptr(), vdt, C, D0, H, W, stats.data_ptr(), coef0.data_ptr(), 1,
              part.data_ptr(), vdt, lst.data_ptr(), cnt.data_ptr(), cap, E.stream())
    f = flags.bool()
    assert torch.equal(part[f], full[f]) and bool((part[~f] == 5.0).all())


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_middle_layer_1_weight_gradient_from_the_active_sites(dt):
    """a0 = cvec outside the first layer's active sites: dW(middle_layer.1) = [row-list weight gradient of the rows a0 - cvec at
    the active sites: vn_act_delta_rows + vn_conv_wgrad_partials_counted + unpack]  +  cvec (x) box sums of dy
    [vn_box_col_sums + vn_wgrad_const_add], against torch's conv weight gradient in float64 on the same operands."""
    import ctypes
    from voxelnet_amd import _lib, engine as E, net as N
    DEV = "cuda:0"
    lib = _lib.load()
    torch.manual_seed(21)
    vdt = _lib.VN_BF16 if dt == torch.bfloat16 else _lib.VN_F32
    B, D0, H, W, C = 2, 5, 20, 24, 64
    D1 = D0 - 2
    M0 = B * D0 * H * W
    sp = dict(N.layer_table(2))["middle_layer.1"]
    # the first layer's BatchNorm state, its conv bias and an activation tensor that is relu(BN(bias)) outside the active sites
    stats = torch.cat([torch.randn(C) * 0.2, torch.rand(C) + 0.5, torch.rand(C) + 0.5, torch.randn(C) * 0.2]).to(DEV)
    bias = torch.randn(C, device=DEV) * 0.3
    mean, S, be = stats[:C], stats[2 * C:3 * C], stats[3 * C:]
    cvec = torch.relu(S * (bias.to(dt).float() - mean) + be).to(dt)           # as k_bn_apply<flagged> writes it
    flags = torch.rand(M0, device=DEV) < 0.12
    a0 = cvec.repeat(M0, 1)
    a0[flags] = (torch.rand(int(flags.sum()), C, device=DEV) * 2.0).to(dt)
    dy1 = (torch.randn(B, D1, H, W, 64, device=DEV) * 0.5).to(dt)
    # ---- reference: dW of conv3d(a0, w, stride 1, padding (0,1,1)) in float64
    x64 = a0.double().reshape(B, D0, H, W, C).permute(0, 4, 1, 2, 3).contiguous().requires_grad_(False)
    w64 = torch.zeros(64, 64, 3, 3, 3, dtype=torch.float64, device=DEV, requires_grad=True)
    y64 = torch.nn.functional.conv3d(x64, w64, stride=1, padding=(0, 1, 1))
    y64.backward(dy1.double().permute(0, 4, 1, 2, 3).contiguous())
    ref = w64.grad
    # ---- the sparse route
    idx = torch.nonzero(flags).flatten()
    n = int(idx.numel())
    lst = torch.stack([idx // (D0 * H * W), (idx // (H * W)) % D0, (idx // W) % H, idx % W], 1).contiguous()
    cap = n + 100
    lst = torch.cat([lst, torch.zeros((100, 4), dtype=torch.int64, device=DEV)])
    cnt = torch.tensor([n], dtype=torch.int32, device=DEV)
    drows = torch.full((cap, C), 3.0, dtype=dt, device=DEV)                   # rows past the count are never read
    _lib.call("vn_act_delta_rows", a0.data_ptr(), vdt, C, D0, H, W, stats.data_ptr(), bias.data_ptr(), vdt, 1, lst.data_ptr(),
              cnt.data_ptr(), cap, drows.data_ptr(), vdt, E.stream())
    # (the kernel forms cvec with one fma, this test with a multiply and an add: an ulp of fp32)
    assert torch.allclose(drows[:n].float(), (a0[idx].float() - cvec.float()).to(dt).float(), rtol=0, atol=2e-2 if dt == torch.bfloat16 else 1e-6)
    x_dy = E.Rows(dy1, 64)
    b = ((1, 1, 1), (-1, -1, -1), tuple(-p for p in sp.pad), sp.stride)
    g = E._geom(B, x_dy, (D0, H, W), 64, 0, 64, sp.k, *b, (0, 0, 0, 64))
    ws, ws_bytes = E.wgrad_workspace(g, 0, cap, DEV)
    chunks = ctypes.c_int32(0)
    _lib.call("vn_conv_wgrad_partials_counted", dy1.data_ptr(), drows.data_ptr(), ctypes.byref(g), lst.data_ptr(), cap,
              cnt.data_ptr(), ws.data_ptr(), ws_bytes, ctypes.byref(chunks), E.stream())
    dw = torch.zeros(64, 64, 3, 3, 3, device=DEV)
    job = (_lib.VnUnpackJob * 1)()
    job[0].dw_packed, job[0].dw = ws.data_ptr(), dw.data_ptr()
    job[0].c_out, job[0].c_in, job[0].taps, job[0].mode, job[0].cin_fold = 64, 64, 27, 2, 1
    job[0].chunks, job[0].chunk_stride = chunks.value, 27 * 64 * 64
    _lib.call("vn_unpack_wgrads_batch", job, 1, E.stream())
    bs_bytes = lib.vn_dgrad_total_workspace_bytes(64)
    bws = torch.empty(bs_bytes, dtype=torch.uint8, device=DEV)
    _lib.call("vn_box_col_sums", dy1.data_ptr(), vdt, B, D1, H, W, 64, bws.data_ptr(), bs_bytes, E.stream())
    _lib.call("vn_wgrad_const_add", dw.data_ptr(), bws.data_ptr(), bs_bytes, 64, 64, 3, stats.data_ptr(), bias.data_ptr(), vdt, vdt,
              1, E.stream())
    # forward-error bound of an fp32-accumulated sum (plus, in bf16, the rounding of the rows a0 - cvec to bf16: 2^-9 of each)
    xa = a0.double().abs().reshape(B, D0, H, W, C).permute(0, 4, 1, 2, 3).contiguous()
    wa = torch.zeros(64, 64, 3, 3, 3, dtype=torch.float64, device=DEV, requires_grad=True)
    torch.nn.functional.conv3d(xa, wa, stride=1, padding=(0, 1, 1)).backward(dy1.double().abs().permute(0, 4, 1, 2, 3).contiguous())
    terms = wa.grad
    err = (dw.double() - ref).abs()
    tol = (2e-5 if dt == torch.float32 else 6e-4) * terms + 1e-9      # bf16: + half a bf16 ulp of the rows a0 - cvec (12 % of the sites)
    assert bool((err <= tol).all()), float((err / tol).max())
    l2 = float((dw.double() - ref).norm() / ref.norm())
    assert l2 < (1e-5 if dt == torch.float32 else 2e-3), l2


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("shape", [(100, 88, 128), (50, 44, 256)], ids=["block2", "block3"])
def test_data_gradient_with_fused_batchnorm_backward_sums(dt, shape):
    """vn_conv_dgrad_bn_bwd (the small-image 3x3 data gradient with the BatchNorm-backward sums of the layer below in its
    epilogue) == vn_conv_gather_gemm followed by vn_bn_bwd_reduce_slab: same dx bit for bit, same sums after the finalize"""
    import ctypes
    from voxelnet_amd import _lib, engine as E, net as N
    DEV = "cuda:0"
    lib = _lib.load()
    torch.manual_seed(31)
    mode = "bf16" if dt == torch.bfloat16 else "fp32"
    vdt = _lib.VN_BF16 if dt == torch.bfloat16 else _lib.VN_F32
    H, W, C = shape
    B = 2
    M = B * H * W
    sp = dict(N.layer_table(2))["block2.1" if C == 128 else "block3.1"]
    assert sp.cin == C and sp.cout == C and sp.k == (1, 3, 3)
    w = torch.randn(C, C, 3, 3, device=DEV) * 0.03
    wpd = E.pack_weight(w, sp, 1, mode)
    dy = E.Rows((torch.randn(B, 1, H, W, C, device=DEV) * 0.5).to(dt), C)
    b = ((1, 1, 1), (-1, -1, -1), tuple(-p for p in sp.pad), sp.stride)
    dx0 = E.Rows(torch.empty(B, 1, H, W, C, dtype=dt, device=DEV), C)
    dx1 = E.Rows(torch.empty(B, 1, H, W, C, dtype=dt, device=DEV), C)
    g = E.gather_geometry(dy, dx0, sp.k, C, C, *b, (1, H, W))
    assert lib.vn_conv_plan_id(ctypes.byref(g)) == 123
    y = (torch.randn(M, C, device=DEV) * 1.2).to(dt)                       # the conv output of the layer below
    stats = torch.cat([torch.randn(C) * 0.2, torch.rand(C) + 0.5, torch.rand(C) + 0.5, torch.randn(C) * 0.2]).to(DEV)
    gamma = torch.rand(C, device=DEV) + 0.5
    # ---- two launches
    _lib.call("vn_conv_gather_gemm", dy.ptr(), wpd.data_ptr(), None, dx0.ptr(), vdt, ctypes.byref(g), 0, None, E.stream())
    rows0 = lib.vn_bn_bwd_slab_rows(M, C)
    s0 = torch.empty((rows0, 2, C), device=DEV)
    _lib.call("vn_bn_bwd_reduce_slab", dx0.ptr(), vdt, C, y.data_ptr(), vdt, C, M, C, stats.data_ptr(), 1, s0.data_ptr(), E.stream())
    out0 = [torch.empty(3 * C, device=DEV), torch.empty(C, device=DEV), torch.empty(C, device=DEV)]
    _lib.call("vn_bn_bwd_finalize_slab", s0.data_ptr(), rows0, M, C, gamma.data_ptr(), stats.data_ptr(), out0[0].data_ptr(),
              out0[1].data_ptr(), out0[2].data_ptr(), E.stream())
    # ---- one launch
    rows1 = lib.vn_conv_stats_slab_rows(ctypes.byref(g))
    s1 = torch.full((rows1, 2, C), float("nan"), device=DEV)
    _lib.call("vn_conv_dgrad_bn_bwd", dy.ptr(), wpd.data_ptr(), dx1.ptr(), vdt, ctypes.byref(g), y.data_ptr(), vdt, stats.data_ptr(),
              s1.data_ptr(), E.stream())
    assert torch.equal(dx0.t, dx1.t)
    out1 = [torch.empty(3 * C, device=DEV), torch.empty(C, device=DEV), torch.empty(C, device=DEV)]
    _lib.call("vn_bn_bwd_finalize_slab", s1.data_ptr(), rows1, M, C, gamma.data_ptr(), stats.data_ptr(), out1[0].data_ptr(),
              out1[1].data_ptr(), out1[2].data_ptr(), E.stream())
    sc = float(dx0.t.float().abs().sum(dim=(0, 1, 2, 3)).max())
    for a, bb in zip(out0[1:], out1[1:]):                                  # d_gamma, d_beta: fp32 partials in another grouping
        assert float((a - bb).abs().max()) < 1e-5 * sc * 3
    assert torch.allclose(out0[0], out1[0], rtol=1e-4, atol=1e-5 * sc / M)


def test_fp32x3_conv3d_weight_gradient_through_hi_lo_copies():
    """fp32x3 (vnNetConfig.mode 2), middle_layer.2's weight gradient (model.py:209 backward) as the executor computes it: both
    operands cast once to [hi|lo] bf16 rows (vn_cast_rows with a residual offset), then THREE launches of the bf16 weight
    gradient over them — hi.hi, lo(src).hi, hi(src).lo(rows) — into consecutive partial slabs that one unpack sums.  At the
    production size (2 x 3 x 400 x 352 -> 2 x 2 x 400 x 352, 64 -> 64 channels, the patch kernel: plan 200) against the exact
    fp32 weight gradient of the same operands: relative L2 <= 2e-5 (three bf16 products carry ~2^-16 each)."""
    import ctypes
    from voxelnet_amd import _lib
    from voxelnet_amd import engine as E
    dev = torch.device("cuda:0")
    lib = _lib.load()
    B, din, dout, H, W, C = 2, 3, 2, 400, 352, 64
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn((B, din, H, W, C), generator=g).to(dev)
    dy = (torch.randn((B, dout, H, W, C), generator=g) * 1e-2).to(dev)
    taps, k, s, p = 27, (3, 3, 3), (2, 1, 1), (1, 1, 1)
    one = (1, 1, 1)

    def partials(src, rows, src_view, row_view, out, out_bytes):
        gq = E._geom(B, src_view, (dout, H, W), C, 0, C, k, s, one, p, one, row_view.strides)
        ch = ctypes.c_int32(0)
        _lib.call("vn_conv_wgrad_partials", src, rows, ctypes.byref(gq), 0, None, 0, out, out_bytes, ctypes.byref(ch), E.stream())
        return ch.value, gq
    # exact fp32 reference: the same entry point on the fp32 operands
    xr, dr = E.Rows(x, C), E.Rows(dy, C)
    g32 = E._geom(B, xr, (dout, H, W), C, 0, C, k, s, one, p, one, dr.strides)
    ws32, wsb32 = E.wgrad_workspace(g32, 0, 0, dev)
    ch32, _ = partials(xr.ptr(), dr.ptr(), xr, dr, ws32.data_ptr(), wsb32)
    dw32 = torch.empty((C, C, 3, 3, 3), device=dev)
    jobs = (_lib.VnUnpackJob * 1)()
    jobs[0] = _lib.VnUnpackJob(ws32.data_ptr(), dw32.data_ptr(), C, C, taps, 0, 1, ch32, taps * C * C)
    _lib.call("vn_unpack_wgrads_batch", jobs, 1, E.stream())
    # the executor's route
    xh = torch.empty((B, din, H, W, 2 * C), dtype=torch.bfloat16, device=dev)
    dh = torch.empty((B, dout, H, W, 2 * C), dtype=torch.bfloat16, device=dev)
    _lib.call("vn_cast_rows", x.data_ptr(), _lib.VN_F32, C, x.numel() // C, C, xh.data_ptr(), _lib.VN_BF16, 2 * C, C, E.stream())
    _lib.call("vn_cast_rows", dy.data_ptr(), _lib.VN_F32, C, dy.numel() // C, C, dh.data_ptr(), _lib.VN_BF16, 2 * C, C, E.stream())
    hi = x.bfloat16().float()
    assert torch.equal(xh[..., :C].float(), hi) and torch.equal(xh[..., C:].float(), (x - hi).bfloat16().float())
    xv, dv = E.Rows(xh[..., :C], C), E.Rows(dh[..., :C], C)           # views: row width 2 C, C channels
    gb = E._geom(B, xv, (dout, H, W), C, 0, C, k, s, one, p, one, dv.strides)
    assert lib.vn_conv_wgrad_plan_id(ctypes.byref(gb), 0, 0) == 200           # the bf16 patch kernel
    per = lib.vn_conv_wgrad_workspace_bytes(ctypes.byref(gb), 0, 0)
    ws = torch.empty(3 * per, dtype=torch.uint8, device=dev)
    total = 0
    for src_lo, rows_lo in ((0, 0), (1, 0), (0, 1)):
        ch, _ = partials(xh.data_ptr() + 2 * C * src_lo, dh.data_ptr() + 2 * C * rows_lo, xv, dv,
                         ws.data_ptr() + total * taps * C * C * 4, per)
        total += ch
    dw = torch.empty_like(dw32)
    jobs[0] = _lib.VnUnpackJob(ws.data_ptr(), dw.data_ptr(), C, C, taps, 0, 1, total, taps * C * C)
    _lib.call("vn_unpack_wgrads_batch", jobs, 1, E.stream())
    torch.cuda.synchronize()
    l2 = float((dw.double() - dw32.double()).norm() / dw32.double().norm())
    print(f"fp32x3 Conv3d weight gradient through [hi|lo] copies vs exact fp32: rel-L2 {l2:.2e} ({total} partial slabs)")
    assert l2 < 2e-5, l2


def split_fp32_storage(t):
    """fp32 tensor (..., C), C % 8 == 0 -> its VN_F32X3S bytes as bf16 (..., 2 C): per 8 channels eight hi parts, then eight
    lo parts (hi = bf16(x), lo = bf16(x - hi))"""
    g = t.reshape(t.shape[:-1] + (t.shape[-1] // 8, 8))
    hi = g.to(torch.bfloat16)
    lo = (g - hi.float()).to(torch.bfloat16)
    return torch.cat([hi, lo], dim=-1).reshape(t.shape[:-1] + (t.shape[-1] * 2,))


def test_fp32x3_weights_are_split_once_by_the_pack(x3_flag):
    """fp32x3: a convolution launched with VN_F32X3 reads its weights as hi / lo bf16 granules that
    vn_pack_weight(packed_dtype VN_F32X3) wrote — include/voxelnet_hip.h documents the order (round 5: the "split fp32"
    format of VN_F32X3S): every group of 8 input channels is 32 B = its eight hi parts, then its eight lo parts.  (a) the
    packed bytes against that description, bit for bit, all four packing modes and the BEV fold; K = 16: plain fp32 (split in
    registers).  (b) conv outputs and data gradients of one layer per kernel family in fp32x3 against the exact fp32 mode:
    rel-L2 < 2e-5 (three bf16 products: ~2^-16 each) — a misplaced hi / lo value would cost ~1e-3."""
    from voxelnet_amd import _lib, engine as E
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    for (co, ci, taps, mode, fold) in [(64, 128, 27, 0, 1), (64, 128, 27, 1, 1), (256, 128, 4, 2, 1), (256, 128, 4, 3, 1),
                                       (128, 128, 9, 0, 2), (16, 768, 1, 0, 1), (16, 768, 1, 1, 1)]:
        w = torch.from_numpy(rng.standard_normal((co, ci, taps) if mode < 2 else (ci, co, taps)).astype(np.float32)).to(dev)
        N, K = (co, ci) if mode in (0, 2) else (ci, co)
        plain = torch.empty((taps, N, K), dtype=torch.float32, device=dev)
        split = torch.empty((taps, N, K), dtype=torch.float32, device=dev)
        _lib.call("vn_pack_weight", w.data_ptr(), co, ci, taps, mode, 0, fold, plain.data_ptr(), _lib.VN_F32, E.stream())
        _lib.call("vn_pack_weight", w.data_ptr(), co, ci, taps, mode, 0, fold, split.data_ptr(), E.VN_F32X3, E.stream())
        jobs = (_lib.VnPackJob * 1)()
        batch = torch.empty_like(split)
        jobs[0] = _lib.VnPackJob(w.data_ptr(), batch.data_ptr(), co, ci, taps, mode, 0, fold, E.VN_F32X3, 0)
        _lib.call("vn_pack_weights_batch", jobs, 1, E.stream())
        torch.cuda.synchronize()
        assert torch.equal(split.view(torch.int32), batch.view(torch.int32))
        if K % 32:
            assert torch.equal(split, plain)
            continue
        want = split_fp32_storage(plain)
        assert torch.equal(split.view(torch.bfloat16).view(taps, N, K * 2).view(torch.int16), want.view(torch.int16))
    cases = {c[0]: c for c in LAYER_CASES}
    for name in ("c3_s111_p011", "c3_s211_p111", "c2_s1", "c2_s2", "c2_s1_256", "d_k2s2", "d_k3s1"):
        case = cases[name]
        idx = [c[0] for c in LAYER_CASES].index(name)
        kind, dim = case[1], case[2]
        spec = _spec(case)
        sd = build_layer_state(idx, case)
        wkey = "deconv" if kind == "deconv" else "conv"
        res = {}
        for mode in ("fp32", "fp32x3"):
            x3_flag["on"] = mode == "fp32x3"
            P = {"weight": sd[f"L.{wkey}.weight"].to(dev), "bias": sd[f"L.{wkey}.bias"].to(dev),
                 "gamma": sd["L.batch_norm.weight"].to(dev), "beta": sd["L.batch_norm.bias"].to(dev)}
            Bf = {"running_mean": sd["L.batch_norm.running_mean"].to(dev).clone(),
                  "running_var": sd["L.batch_norm.running_var"].to(dev).clone()}
            xr = E.nchw_to_rows(layer_input(idx, case).to(dev), "fp32")
            a, st = E.layer_forward(spec, xr, P, Bf, True, "fp32")
            up = layer_upstream(idx, case, tuple(E.rows_to_nchw(a, dim).shape)).to(dev)
            grads, dx = E.layer_backward(st, E.nchw_to_plain_rows(up, torch.float32), P, "fp32")
            res[mode] = (st.y.t.float().clone(), dx.t.float().clone())
        x3_flag["on"] = False
        (y32, dx32), (y3, dx3) = res["fp32"], res["fp32x3"]
        l2 = float((y3 - y32).norm() / y32.norm())
        assert 0.0 < l2 < 2e-5, (name, "y", l2)
        # (dx passes the ReLU mask of BN(y): a pre-activation within 1e-5 of zero flips between the two modes and moves a few
        #  dx values by a whole term — robust_err looks at the 95th percentile, which a misplaced weight part would still move)
        e95 = robust_err(dx3, dx32.cpu().numpy())
        assert 0.0 < e95 < 5e-5, (name, "dx", e95)
