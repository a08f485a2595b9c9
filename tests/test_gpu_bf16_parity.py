"""GPU: the bf16 (benchmarked) kernels, stage by stage, against a bf16-OPERAND oracle — at the production tile
selections (k_conv_patch 6x32 / 10x16 / 4x16 tiles, k_conv_patch2d (4x16, three weight stages), k_gather_gemm 128x128 / 160x128 / 64x128 / 256x64, k_wgrad_patch,
k_wgrad<4,4>, the rulebook first layer) of ConvMD / DeConv2d (model.py:111-199, layer table model.py:206-254).

What "equal" means in bf16 mode.  The kernels read bf16 operands, accumulate in fp32 and store bf16 (activations, data
gradients) or fp32 (weight gradients, statistics).  The oracle here is PyTorch-CPU in float64 on EXACTLY the operands
the kernel under test reads (inputs are pre-rounded to bf16 on the host; for the later stages the oracle takes the
kernel-produced tensors of the earlier stage), so the only differences left are
  (a) the order of the fp32 accumulation (delta <= ~1e-6 of the sum of |terms|) and
  (b) the final round-to-nearest-even to bf16 (half a bf16 ulp = up to 2^-9 of the value).
Bars, written out below where they are asserted:
  * bf16 outputs (y, a, dy, dx): |got - ref| <= 0.5 ulp_bf16(value) + 1e-4 * rms(ref)  for EVERY element, i.e. the
    output is the correctly rounded exact result up to accumulation noise.  (A max-error bar "relative to the tensor
    maximum" cannot be below 2^-8 = 3.9e-3 for a bf16-stored tensor: that is half an ulp at the top of a binade.)
  * fp32 outputs (dW, dgamma, dbeta, BatchNorm mean / invstd): forward-error bound 2e-5 * sum|terms| per element and
    relative L2 <= 1e-4.
Any wrong tap, shifted halo row, swapped tile or stale LDS stage is an O(1) error on the affected elements.
Checked by breaking k_wgrad_patch's tap shift (tw -> tw+1 for one tap): m1/m2 fail with rel-L2 ~0.3."""
import ctypes
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import torch_ref as tr

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
EPS = 1e-5
MAP_MAX, MAP_L2 = 0.15, 0.1             # test_bf16_step_vs_fp32_step, maps (measured: 0.087 / 0.058)
GRAD_L2, GRAD_COS = 0.8, 0.7            # ... conv / BatchNorm gradients (measured: <= 0.71, >= 0.77)
# ... VFE gradients: a realisation of the forward-induced chaos, not a tolerance of the VFE kernels (those are held to the
# fp64 oracle in test_gpu_vfe.py).  Measured <= 1.16 / >= 0.48 with the round-2 VFE; the round-3 VFE, whose output differs
# from it by 1e-7 relative (BatchNorm sums over 1024 instead of 512 slabs; tools/vfe_dump.py), gives 1.72 / 0.80 for
# vfe_1.fcn.0.bias: that last-bit change of the input of the bf16 network is enough to move this distance by 50 %.
# Round 5 (the first BatchNorm's sums in another fixed order, k_vfe_rows_p1: again a last-bit change of the encoder's
# output) gives 1.22 / 0.395 for vfe_1.bn.weight.  Three realisations: cosines 0.48, 0.80, 0.395 — these two numbers are a
# guard against garbage (a wrong sign, a missing term: cosine <= 0, distance >> 1), nothing finer.
VFE_GRAD_L2, VFE_GRAD_COS = 3.0, 0.2
FROZEN_L2, FROZEN_COS = 0.05, 0.999    # test_bf16_backward_chain_on_frozen_forward (measured: <= 0.036 / >= 0.9994; convs <= 0.019)


def bf16r(t):
    """round an fp32 tensor to the nearest bf16 (ties to even), keep it as fp32 values"""
    return t.float().bfloat16().float()


def seeded(shape, seed, scale=1.0):
    return torch.from_numpy((np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32))


def ulp_bf16(v):
    """bf16 ulp (8 significand bits) at |v|, float64 numpy"""
    v = np.abs(np.asarray(v, dtype=np.float64))
    out = np.zeros_like(v)
    nz = v > 0
    out[nz] = np.exp2(np.floor(np.log2(v[nz])) - 7)
    return out


def assert_rounded(got, ref, what, extra=None):
    """got: bf16-stored values; ref: float64 exact-operand oracle.  Every element within half a bf16 ulp plus
    accumulation noise (1e-4 of the tensor's rms) [plus an optional per-element slack `extra`]."""
    got = got.detach().double().cpu().numpy()
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    rms = float(np.sqrt(np.mean(ref * ref))) + 1e-30
    bound = 0.5 * ulp_bf16(np.maximum(np.abs(got), np.abs(ref))) + 1e-4 * rms
    if extra is not None:
        bound = bound + extra
    err = np.abs(got - ref)
    bad = err > bound
    l2 = float(np.linalg.norm(got - ref) / (np.linalg.norm(ref) + 1e-30))
    assert not bad.any(), (f"{what}: {int(bad.sum())} of {bad.size} elements beyond half a bf16 ulp; worst "
                           f"{float((err / np.maximum(bound, 1e-30)).max()):.2f}x the bound, rel-L2 {l2:.2e}")
    assert l2 < 3e-3, (what, l2)           # bf16 rounding noise alone: ~1e-3
    return float(err.max() / (np.abs(ref).max() + 1e-30)), l2


def assert_fp32_sum(got, ref, abs_terms, what, l2_tol=1e-4):
    """fp32-accumulated sums against float64: forward-error bound 2e-5 * sum|terms| (+ tiny absolute) for every element,
    relative L2 <= l2_tol (1e-3 for weight gradients: sums over 2e4-1.4e5 sites with heavy cancellation)"""
    got = got.detach().double().cpu().numpy()
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    bound = 2e-5 * np.asarray(abs_terms, dtype=np.float64) + 1e-12
    err = np.abs(got - ref)
    assert (err <= bound).all(), (what, float((err / bound).max()))
    l2 = float(np.linalg.norm(got - ref) / (np.linalg.norm(ref) + 1e-30))
    assert l2 < l2_tol, (what, l2)
    return l2


def rows_to_nchw64(t, dim):
    """(B,D,H,W,C) rows tensor (any dtype) -> NC(D)HW float64 on the CPU"""
    x = t.detach().double().cpu()
    x = x.permute(0, 4, 1, 2, 3).contiguous()
    return x if dim == 3 else x[:, :, 0]


def dense_geom(_lib, dtype, B, src_dims, row_dims, Cs, Cr, k, mul, tmul, pad, div):
    """vnConv of a launch over contiguous tensors (only used to ask the library which kernel it would pick)"""
    g = _lib.VnConv()
    g.dtype = dtype
    g.B = B
    g.Ds, g.Hs, g.Ws = src_dims
    g.Dr, g.Hr, g.Wr = row_dims
    g.Cs, g.src_wrap, g.Cr = Cs, 0, Cr
    g.kD, g.kH, g.kW = k
    g.mulD, g.mulH, g.mulW = mul
    g.tmulD, g.tmulH, g.tmulW = tmul
    g.padD, g.padH, g.padW = pad
    g.divD, g.divH, g.divW = div
    g.src_sW = Cs; g.src_sH = src_dims[2] * Cs; g.src_sD = src_dims[1] * g.src_sH; g.src_sB = src_dims[0] * g.src_sD
    g.out_sW = Cr; g.out_sH = row_dims[2] * Cr; g.out_sD = row_dims[1] * g.out_sH; g.out_sB = row_dims[0] * g.out_sD
    return g


def plan_ids(_lib, spec, B, in_dims):
    """(forward, data-gradient, weight-gradient) kernel ids the library picks for this layer at this size"""
    lib = _lib.load()
    od = spec.out_dims(in_dims)
    neg = tuple(-p for p in spec.pad)
    one, mone = (1, 1, 1), (-1, -1, -1)
    if spec.transposed:
        gf = dense_geom(_lib, _lib.VN_BF16, B, in_dims, od, spec.cin, spec.cout, spec.k, one, mone, neg, spec.stride)
        gd = dense_geom(_lib, _lib.VN_BF16, B, od, in_dims, spec.cout, spec.cin, spec.k, spec.stride, one, spec.pad, one)
        gw = dense_geom(_lib, _lib.VN_BF16, B, od, in_dims, spec.cout, spec.cin, spec.k, spec.stride, one, spec.pad, one)
    else:
        gf = dense_geom(_lib, _lib.VN_BF16, B, in_dims, od, spec.cin, spec.cout, spec.k, spec.stride, one, spec.pad, one)
        gd = dense_geom(_lib, _lib.VN_BF16, B, od, in_dims, spec.cout, spec.cin, spec.k, one, mone, neg, spec.stride)
        gw = dense_geom(_lib, _lib.VN_BF16, B, in_dims, od, spec.cin, spec.cout, spec.k, spec.stride, one, spec.pad, one)
    return (lib.vn_conv_plan_id(ctypes.byref(gf)), lib.vn_conv_plan_id(ctypes.byref(gd)),
            lib.vn_conv_wgrad_plan_id(ctypes.byref(gw), 0, 0))


# weight-gradient kernel of the wide stride-1 3x3 layers: 44 = the single-tap 128 x 128 row form (round 4's nine-tap 128 x 64
# patch tile, plan 202, left the library in round 5)
WG9 = 44

# name, kind, dim, cin, cout, k, stride, pad | test (B, input spatial) | production (B, input spatial) it stands for |
# expected (forward, data-gradient, weight-gradient) kernel ids (vn_conv_plan_id / vn_conv_wgrad_plan_id)
CASES = [
    ("middle_layer.1", "conv", 3, 64, 64, 3, (1, 1, 1), (0, 1, 1), (1, (4, 134, 140)), (2, (5, 400, 352)), (103, 103, 200)),
    ("middle_layer.2", "conv", 3, 64, 64, 3, (2, 1, 1), (1, 1, 1), (1, (3, 134, 140)), (2, (3, 400, 352)), (103, 103, 200)),
    ("block1.0", "conv", 2, 128, 128, 3, (2, 2), (1, 1), (2, (400, 352)), (2, (400, 352)), (4, 1, 44)),
    ("block1.1", "conv", 2, 128, 128, 3, (1, 1), (1, 1), (1, (134, 140)), (2, (200, 176)), (100, 100, WG9)),
    ("deconv1", "deconv", 2, 128, 256, 3, (1, 1), (1, 1), (1, (134, 140)), (2, (200, 176)), (100, 100, WG9)),
    ("block2.0", "conv", 2, 128, 128, 3, (2, 2), (1, 1), (2, (200, 176)), (2, (200, 176)), (1, 4, 44)),
    ("block2.1", "conv", 2, 128, 128, 3, (1, 1), (1, 1), (2, (100, 88)), (2, (100, 88)), (123, 123, WG9)),
    ("deconv2", "deconv", 2, 128, 256, 2, (2, 2), (0, 0), (2, (100, 88)), (2, (100, 88)), (4, 1, 44)),
    ("block3.0", "conv", 2, 128, 256, 3, (2, 2), (1, 1), (2, (100, 88)), (2, (100, 88)), (2, 1, 44)),
    ("block3.1", "conv", 2, 256, 256, 3, (1, 1), (1, 1), (2, (50, 44)), (2, (50, 44)), (123, 123, WG9)),
    ("deconv3", "deconv", 2, 256, 256, 4, (4, 4), (0, 0), (2, (50, 44)), (2, (50, 44)), (4, 2, 44)),
]


def make_spec(case):
    from voxelnet_amd.engine import LayerSpec
    name, kind, dim, cin, cout, k, s, p = case[:8]
    if dim == 3:
        return LayerSpec(name, 3, cin, cout, (k, k, k), tuple(s), tuple(p))
    return LayerSpec(name, 2, cin, cout, (1, k, k), (1,) + tuple(s), (0,) + tuple(p), transposed=(kind == "deconv"))


def oracle_conv64(x, w, spec, kind):
    """float64 conv / conv_transpose without bias on NC(D)HW tensors"""
    if kind == "deconv":
        return F.conv_transpose2d(x, w, None, spec.stride[1:], spec.pad[1:])
    if spec.dim == 3:
        return F.conv3d(x, w, None, spec.stride, spec.pad)
    return F.conv2d(x, w, None, spec.stride[1:], spec.pad[1:])


def emulate_fp32_bn(y, mean, S, beta):
    """z = fmaf(S, y - mean, beta) as the kernels evaluate it in fp32 (y - mean rounded to fp32, then one fused
    multiply-add): float64 holds the product exactly, so rounding the float64 result once more reproduces fmaf up to
    double rounding (~1e-9 of the elements)."""
    d0 = (y.double() - mean.double()).float()
    z = (S.double() * d0.double() + beta.double()).float()
    return d0, z


@pytest.mark.parametrize("case", CASES, ids=lambda c: c[0])
def test_bf16_layer_stages(case):
    """production tile selections (asserted: the kernels this size runs are the kernels the full-size step runs)"""
    run_stages(case, [c[0] for c in CASES].index(case[0]))


def _tiny_cases():
    from layer_cases import LAYER_CASES
    return [c[:8] + ((2, c[8]), None, None) for c in LAYER_CASES if c[1] != "head"]


@pytest.mark.parametrize("case", _tiny_cases(), ids=lambda c: c[0])
def test_bf16_layer_stages_tiny(case):
    """the 8 x 12 ... 2 x 3 layer fixtures of test_gpu_layers.py (ragged single tiles, every residue class) in bf16 mode,
    same bars — replaces the 1e-1 / cosine bars that mode had against the fp32 golden vectors"""
    run_stages(case, 50 + [c[0] for c in _tiny_cases()].index(case[0]))


def run_stages(case, idx):
    from voxelnet_amd import _lib, engine as E
    name, kind, dim, cin, cout, k, s, p = case[:8]
    (B, sp), prod, expect = case[8], case[9], case[10]
    spec = make_spec(case)
    in_dims = (1,) + tuple(sp) if dim == 2 else tuple(sp)
    ids = plan_ids(_lib, spec, B, in_dims)
    if prod is not None:
        # ---- the kernels this size runs are the kernels production runs
        Bp, spp = prod
        prod_dims = (1,) + tuple(spp) if dim == 2 else tuple(spp)
        ids_prod = plan_ids(_lib, spec, Bp, prod_dims)
        assert ids == ids_prod, (name, ids, ids_prod)
        for got, want in zip(ids, expect):
            assert want is None or got == want, (name, ids, expect)
    dev = torch.device(DEV)
    taps = spec.taps
    fan = cin if kind == "deconv" else cin * taps
    wshape = (cin, cout, k, k) if kind == "deconv" else (cout, cin) + (k,) * dim
    w = bf16r(tr._fill(wshape, 900 + idx, 1.0 / np.sqrt(fan)))
    bias = tr._fill((cout,), 910 + idx, 0.1)
    gamma = 1.0 + tr._fill((cout,), 920 + idx, 0.2)
    beta = tr._fill((cout,), 930 + idx, 0.1)
    x = bf16r(seeded((B, cin) + tuple(sp), 940 + idx))
    P = {"weight": w.to(dev), "bias": bias.to(dev), "gamma": gamma.to(dev), "beta": beta.to(dev)}
    Bf = {"running_mean": torch.zeros(cout, device=dev), "running_var": torch.ones(cout, device=dev)}
    xr = E.nchw_to_rows(x.to(dev), "bf16")
    assert torch.equal(rows_to_nchw64(xr.t, dim).float(), x)      # exact: x is bf16-representable

    # ================= stage 1: convolution forward + fused statistics + BatchNorm apply =================
    a, st = E.layer_forward(spec, xr, P, Bf, True, "bf16")
    xd = x.double().requires_grad_(True)
    wd = w.double().requires_grad_(True)
    y64 = oracle_conv64(xd, wd, spec, kind)                       # without bias, float64
    red = (0, 2, 3, 4) if dim == 3 else (0, 2, 3)
    shp = (1, cout, 1, 1, 1) if dim == 3 else (1, cout, 1, 1)
    y_k = rows_to_nchw64(st.y.t, dim)
    e_y = assert_rounded(st.y.t, (y64.detach() + bias.double().view(shp)).permute(*((0, 2, 3, 4, 1) if dim == 3 else (0, 2, 3, 1)))
                         .reshape(st.y.t.shape).numpy(), name + " y")
    n = y64[:, 0].numel()
    m64 = y64.detach().mean(dim=red)
    v64 = y64.detach().var(dim=red, unbiased=False)
    stats = st.stats.detach().cpu().double().view(4, cout)          # mean | invstd | S | beta
    std = v64.sqrt()
    assert float(((stats[0] - (m64 + bias.double())).abs() / std).max()) < 1e-4, name + " batch mean"
    assert float((stats[1] * (v64 + EPS).sqrt() - 1).abs().max()) < 1e-4, name + " batch invstd"
    assert torch.allclose(stats[2], gamma.double() * stats[1], rtol=1e-6), name
    rm = 0.1 * (m64 + bias.double())
    rv = 0.9 + 0.1 * v64 * n / (n - 1)
    assert float((Bf["running_mean"].cpu().double() - rm).abs().max()) < 1e-4 * float(std.max()), name
    assert float((Bf["running_var"].cpu().double() / rv - 1).abs().max()) < 1e-4, name
    # BatchNorm + ReLU of the kernel's own y with the kernel's own statistics: the same fp32 formula
    sf = [stats[i].float().view(shp) for i in range(4)]
    d0, z = emulate_fp32_bn(y_k.float(), sf[0], sf[2], sf[3])
    a_k = rows_to_nchw64(a.t, dim)
    a_ref = torch.relu(z).double()
    assert_rounded(a_k.float(), a_ref.numpy(), name + " a = relu(bn(y)) from the stored y")
    exact = float((a_k.float().bfloat16() == a_ref.float().bfloat16()).double().mean())
    assert exact > 0.9995, (name, exact)
    # ... and the whole layer against the pure operand oracle (y rounded once more in between: + |S| * half an ulp of y)
    z64 = gamma.double().view(shp) * (y64.detach() - m64.view(shp)) / (v64.view(shp) + EPS).sqrt() + beta.double().view(shp)
    slack = (sf[2].double().abs() * 0.5 * torch.from_numpy(ulp_bf16(y_k.numpy()))).numpy() + 2e-4 * np.abs(z64.numpy())
    e_a = assert_rounded(a_k.float(), torch.relu(z64).numpy(), name + " layer output", extra=slack)

    # ================= stage 2: BatchNorm backward (slab reduction, finalize, apply) =================
    lib = _lib.load()
    od = st.out_dims
    M, C = st.y.M, cout
    da = bf16r(seeded(tuple(a_k.shape), 950 + idx))
    dar = E.nchw_to_plain_rows(da.to(dev), torch.bfloat16)
    rows = lib.vn_bn_bwd_slab_rows(M, C)
    slab = torch.empty((rows, 2, C), dtype=torch.float32, device=dev)
    coef = torch.empty(3 * C, dtype=torch.float32, device=dev)
    dgam = torch.empty(C, dtype=torch.float32, device=dev)
    dbet = torch.empty(C, dtype=torch.float32, device=dev)
    dy = E.new_rows(B, od, C, torch.bfloat16, False, dev)
    _lib.call("vn_bn_bwd_reduce_slab", dar.ptr(), _lib.VN_BF16, dar.row_stride(), st.y.ptr(), _lib.VN_BF16,
              st.y.row_stride(), M, C, st.stats.data_ptr(), 1, slab.data_ptr(), E.stream())
    _lib.call("vn_bn_bwd_finalize_slab", slab.data_ptr(), rows, M, C, P["gamma"].data_ptr(), st.stats.data_ptr(),
              coef.data_ptr(), dgam.data_ptr(), dbet.data_ptr(), E.stream())
    _lib.call("vn_bn_bwd_apply", dar.ptr(), _lib.VN_BF16, dar.row_stride(), st.y.ptr(), _lib.VN_BF16, st.y.row_stride(),
              M, C, st.stats.data_ptr(), coef.data_ptr(), 1, dy.ptr(), _lib.VN_BF16, dy.row_stride(), 0, E.stream())
    dz = (da * (z > 0).float()).double()                                 # the kernels' mask: z > 0 in fp32
    xh = d0.double() * sf[1].double()
    s1, s2 = dz.sum(dim=red), (dz * xh).sum(dim=red)
    assert_fp32_sum(dbet, s1.numpy(), dz.abs().sum(dim=red).numpy(), name + " dbeta")
    assert_fp32_sum(dgam, s2.numpy(), (dz * xh).abs().sum(dim=red).numpy(), name + " dgamma")
    S64, inv64 = sf[2].double(), sf[1].double()
    dy_ref = S64 * dz - (S64 * inv64 * (s2 / M).view(shp)) * d0.double() - S64 * (s1 / M).view(shp)
    dy_k = rows_to_nchw64(dy.t, dim)
    assert_rounded(dy_k.float(), dy_ref.numpy(), name + " dy")

    # ================= stages 3 + 4: weight gradient and data gradient from the kernel's dy =================
    y64.backward(dy_k)                                                   # float64 autograd on the same operands
    one = (1, 1, 1)
    dw = torch.empty_like(P["weight"])
    chunks = ctypes.c_int32(0)
    if spec.transposed:
        g = E._geom(B, dy, st.in_dims, cout, 0, cin, spec.k, spec.stride, one, spec.pad, one, xr.strides)
        srcp, rowp, un = dy.ptr(), xr.ptr(), (cin, cout)
    else:
        g = E._geom(B, xr, od, cin, 0, cout, spec.k, spec.stride, one, spec.pad, one, dy.strides)
        srcp, rowp, un = xr.ptr(), dy.ptr(), (cout, cin)
    assert lib.vn_conv_wgrad_plan_id(ctypes.byref(g), 0, 0) == ids[2]
    ws, ws_bytes = E.wgrad_workspace(g, 0, 0, dev)
    _lib.call("vn_conv_wgrad_partials", srcp, rowp, ctypes.byref(g), 0, None, 0, ws.data_ptr(), ws_bytes,
              ctypes.byref(chunks), E.stream())
    jobs = (_lib.VnUnpackJob * 1)()
    jobs[0] = _lib.VnUnpackJob(ws.data_ptr(), dw.data_ptr(), un[0], un[1], taps, 0, 1, chunks.value, taps * cin * cout)
    _lib.call("vn_unpack_wgrads_batch", jobs, 1, E.stream())
    # sum|terms| of a weight-gradient element ~ sum_m |x||dy| <= sqrt(sum x^2 sum dy^2): use the Cauchy-Schwarz bound
    xs, ds = float(np.sqrt((x.double() ** 2).sum() / cin)), float(np.sqrt((dy_k ** 2).sum() / cout))
    l2w = assert_fp32_sum(dw, wd.grad.numpy(), np.full(tuple(wd.grad.shape), xs * ds), name + " dW", l2_tol=1e-3)
    dx = E.Rows(torch.empty((B,) + tuple(st.in_dims) + (cin,), dtype=torch.bfloat16, device=dev), cin)
    wp = E.pack_weight(P["weight"], spec, 3 if spec.transposed else 1, "bf16")
    neg = tuple(-q for q in spec.pad)
    if spec.transposed:
        E.gather_gemm(dy, wp, None, dx, spec.k, cout, cin, spec.stride, one, spec.pad, one, st.in_dims)
    else:
        E.gather_gemm(dy, wp, None, dx, spec.k, cout, cin, one, (-1, -1, -1), neg, spec.stride, st.in_dims)
    e_dx = assert_rounded(rows_to_nchw64(dx.t, dim).float(), xd.grad.numpy(), name + " dx")
    print(f"{name:16s} kernels {ids}: y {e_y[0]:.1e}/{e_y[1]:.1e}  a {e_a[0]:.1e}/{e_a[1]:.1e}  dx {e_dx[0]:.1e}/{e_dx[1]:.1e} "
          f"(max err / max, rel-L2)  dW rel-L2 {l2w:.1e}")


def test_bf16_heads_and_loss_grad_rows():
    """prob_conv + reg_conv (model.py:253-254, 276-281) as the production N=16 GEMM over the 768-channel concat at the
    full 200 x 176 map, B=2 (k_gather_gemm 256x64 tile), data gradient (16 -> 768) and weight gradient, bf16 operands."""
    from voxelnet_amd import _lib, engine as E
    from voxelnet_amd.net import HEADS
    dev = torch.device(DEV)
    B, H, W = 2, 200, 176
    x = bf16r(seeded((B, 768, H, W), 7001))
    w = bf16r(tr._fill((16, 768, 1, 1), 7002, 1.0 / np.sqrt(768)))
    b = tr._fill((16,), 7003, 0.1)
    P = {"weight": w.to(dev), "bias": b.to(dev)}
    xr = E.nchw_to_rows(x.to(dev), "bf16")
    y, st = E.layer_forward(HEADS, xr, P, None, True, "bf16", y_dtype=torch.float32)
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y64 = F.conv2d(xd, wd, b.double())
    got = rows_to_nchw64(y.t, 2)
    sabs = F.conv2d(x.double().abs(), w.double().abs(), None)
    assert float(((got - y64.detach()).abs() / (2e-5 * sabs + 1e-9)).max()) <= 1.0
    dyv = bf16r(seeded((B, 16, H, W), 7004, 1e-2))
    dyr = E.nchw_to_rows(dyv.to(dev), "bf16")
    grads, dx = E.layer_backward(st, dyr, P, "bf16")
    y64.backward(dyv.double())
    assert_rounded(rows_to_nchw64(dx.t, 2).float(), xd.grad.numpy(), "heads dx")
    xs, ds = float(np.sqrt((x.double() ** 2).sum() / 768)), float(np.sqrt((dyv.double() ** 2).sum() / 16))
    assert_fp32_sum(grads["weight"], wd.grad.numpy(), np.full((16, 768, 1, 1), xs * ds), "heads dW", l2_tol=1e-3)
    assert_fp32_sum(grads["bias"], dyv.double().sum(dim=(0, 2, 3)).numpy(), dyv.double().abs().sum(dim=(0, 2, 3)).numpy(),
                    "heads db")


def test_bf16_rulebook_first_layer():
    """middle_layer.0 (model.py:207) as production runs it in bf16: voxel rows x packed weights as one GEMM, then the
    rulebook gather-sum per active site (no dense grid), against conv3d of the scattered grid in float64 on the same
    bf16 operands; its backward (flagged BatchNorm apply, row-list weight / data gradient) through the executor is
    covered by test_bf16_step_vs_fp32_step and the fp32 full-frame tests."""
    from voxelnet_amd import _lib, engine as E
    from voxelnet_amd.net import layer_table
    dev = torch.device(DEV)
    lib = _lib.load()
    spec = dict(layer_table(2))["middle_layer.0"]
    B, D, H, W, K = 2, 10, 134, 140, 3000
    rng = np.random.default_rng(8101)
    cells = rng.choice(B * D * H * W, size=K, replace=False)
    cells.sort()
    coord = torch.from_numpy(np.stack([cells // (D * H * W), (cells // (H * W)) % D, (cells // W) % H, cells % W], 1).astype(np.int64))
    vw = bf16r(seeded((K, 128), 8102))
    w = bf16r(tr._fill((64, 128, 3, 3, 3), 8103, 1.0 / np.sqrt(128 * 27)))
    bias = tr._fill((64,), 8104, 0.1)
    od = spec.out_dims((D, H, W))
    dense = torch.zeros((B, D, H, W, 128), dtype=torch.float64)
    dense[coord[:, 0], coord[:, 1], coord[:, 2], coord[:, 3]] = vw.double()
    y64 = F.conv3d(dense.permute(0, 4, 1, 2, 3), w.double(), bias.double(), spec.stride, spec.pad)
    # --- the executor's call sequence (csrc/runtime.hip net_prepare + vn_net_forward, layer 0)
    coord_d, vw_d = coord.to(dev), vw.to(dev).bfloat16()
    wp = E.pack_weight(w.to(dev), spec, 0, "bf16")
    y = E.Rows(torch.empty((B,) + od + (64,), dtype=torch.bfloat16, device=dev), 64)
    M = y.M
    bias_d = bias.to(dev)
    _lib.call("vn_fill_rows", y.ptr(), _lib.VN_BF16, M, 64, 64, bias_d.data_ptr(), E.stream())
    g = _lib.VnConv()
    g.dtype = _lib.VN_BF16
    g.B = B
    g.Ds, g.Hs, g.Ws = D, H, W
    g.Dr, g.Hr, g.Wr = od
    g.Cs, g.src_wrap, g.Cr = 128, 0, 64
    g.kD = g.kH = g.kW = 3
    g.mulD, g.mulH, g.mulW = spec.stride
    g.tmulD = g.tmulH = g.tmulW = 1
    g.padD, g.padH, g.padW = spec.pad
    g.divD = g.divH = g.divW = 1
    g.src_sW = 128; g.src_sH = W * 128; g.src_sD = H * W * 128; g.src_sB = D * H * W * 128
    g.out_sB, g.out_sD, g.out_sH, g.out_sW = y.strides
    cap = min(M, K * 2 * 3 * 3)
    aws_bytes = lib.vn_active_sites_workspace_bytes(ctypes.byref(g))
    aws = torch.empty(aws_bytes, dtype=torch.uint8, device=dev)
    alist = torch.empty((cap, 4), dtype=torch.int64, device=dev)
    acount = torch.zeros(64, dtype=torch.int32, device=dev)
    _lib.call("vn_active_sites", coord_d.data_ptr(), K, ctypes.byref(g), aws.data_ptr(), aws_bytes, alist.data_ptr(), cap,
              acount.data_ptr(), E.stream())
    igrid = torch.empty(B * D * H * W, dtype=torch.int32, device=dev)
    _lib.call("vn_voxel_index_grid", coord_d.data_ptr(), K, B, D, H, W, igrid.data_ptr(), E.stream())
    rbP = torch.empty((K, 27 * 64), dtype=torch.float32, device=dev)
    q = _lib.VnConv()
    q.dtype = _lib.VN_BF16
    q.B = 1
    q.Ds = q.Hs = 1; q.Ws = K
    q.Dr = q.Hr = 1; q.Wr = K
    q.Cs, q.src_wrap, q.Cr = 128, 0, 27 * 64
    q.kD = q.kH = q.kW = 1
    q.mulD = q.mulH = q.mulW = 1
    q.tmulD = q.tmulH = q.tmulW = 1
    q.divD = q.divH = q.divW = 1
    q.src_sB = q.src_sD = q.src_sH = K * 128; q.src_sW = 128
    q.out_sB = q.out_sD = q.out_sH = K * 27 * 64; q.out_sW = 27 * 64
    _lib.call("vn_conv_gather_gemm", vw_d.data_ptr(), wp.data_ptr(), None, rbP.data_ptr(), _lib.VN_F32, ctypes.byref(q), 0,
              None, E.stream())
    srows = lib.vn_rulebook_slab_rows(cap)
    slab = torch.empty((srows, 2, 64), dtype=torch.float32, device=dev)
    _lib.call("vn_rulebook_combine", rbP.data_ptr(), igrid.data_ptr(), alist.data_ptr(), cap, acount.data_ptr(),
              ctypes.byref(g), bias_d.data_ptr(), y.ptr(), _lib.VN_BF16, slab.data_ptr(), E.stream())
    ref = y64.permute(0, 2, 3, 4, 1).reshape(y.t.shape).numpy()
    e = assert_rounded(y.t, ref, "rulebook y")
    # fused statistics: sum / sum of squares of (y - bias) over the active rows == over all rows
    s = slab.double().sum(0).cpu()
    yc = (y64 - bias.double().view(1, 64, 1, 1, 1))
    assert_fp32_sum(s[0].float(), yc.sum(dim=(0, 2, 3, 4)).numpy(), yc.abs().sum(dim=(0, 2, 3, 4)).numpy(), "rulebook sum")
    assert_fp32_sum(s[1].float(), (yc * yc).sum(dim=(0, 2, 3, 4)).numpy(), (yc * yc).sum(dim=(0, 2, 3, 4)).numpy(), "rulebook sumsq")
    print("rulebook first layer: y max err / max %.1e, rel-L2 %.1e, active sites %d of %d" % (e[0], e[1], int(acount[0]), M))


def test_bf16_step_vs_fp32_step():
    """One full-size car step at B=2 (BASELINE configs[1]) in bf16 — the benchmarked configuration, through the native
    executor — against the same step in the fp32 parity mode (which test_gpu_model.py pins to the oracle at <= 1e-3).

    What the distances below are (round 3, tools/grad_attribution.py + tools/forward_error_profile.py): NOT
    backward arithmetic.  (i) Promoting any or all of the bf16 mode's gradient tensors (concat gradient, every data
    gradient, the logit gradient) to fp32 storage changes no gradient distance in the third digit; (ii) the exact fp32
    kernels produce the same distances once only the conv weights (0.31-0.43 below the deconvs), only the activations
    (0.33-0.45) or both (0.40-0.52) are rounded to bf16 VALUES; (iii) with the forward frozen the bf16 backward chain is
    within 0.004-0.036 of an exact backward (test_bf16_backward_chain_on_frozen_forward).  The cause is the FORWARD: this
    random-weight BatchNorm/ReLU stack amplifies any perturbation ~1.2x per layer (the fp32 mode's own summation-order
    difference grows 8.6e-8 -> 1.2e-5 over the 23 layers), the bf16 activations are 0.4 % (first layer) ... 10 % (block3)
    away from the fp32 ones, 0.007 % ... 4.2 % of the ReLU masks differ, and a flipped mask element changes its gradient
    by 100 %.  The reference's own fp32 and fp64 gradients differ by 0.08-0.22 on this frame for the same reason.
    Bars: what was measured plus margin; that training in this mode FOLLOWS the reference is tests/test_gpu_trajectory.py."""
    from voxelnet_amd import model as M
    from voxelnet_amd import synth
    from voxelnet_amd.config import grid_config
    from voxelnet_amd.voxelize import voxelize_device
    grid = grid_config("Car")
    frames = synth.workload_frames(2, batch=2)
    feats, coords = [], []
    for b, f in enumerate(frames):
        fb, cb, _ = voxelize_device(torch.from_numpy(f).to(DEV), grid, b, coord_cols=4)
        feats.append(fb)
        coords.append(cb)
    # the benchmarked step's own backward signal: the RPN loss (model.py:310-352) on seeded targets (bench.py's)
    rng = np.random.default_rng(99)
    pos = (rng.random((2, 200, 176, 2)) < 0.002).astype(np.float32)
    neg = ((rng.random((2, 200, 176, 2)) < 0.98) & (pos == 0)).astype(np.float32)
    tgt = (rng.standard_normal((2, 200, 176, 14)) * 0.1).astype(np.float32)
    targets = tuple(torch.from_numpy(a).to(DEV) for a in (pos, neg, tgt))
    out = {}
    for mode in ("fp32", "bf16"):
        M.set_precision(mode)
        m = M.RPN3D("Car")
        m.load_state_dict(tr.make_state_dict("Car"))
        m = m.to(DEV).train()
        res = m((None, None, feats, None, coords, None, None), DEV, targets=targets)
        res[2].backward()
        torch.cuda.synchronize()
        out[mode] = (res[0].detach().double().cpu(), res[1].detach().double().cpu(),
                     {k: p.grad.detach().double().cpu().clone() for k, p in m.named_parameters()}, float(res[2]))
        del m
    print(f"bf16 vs fp32 step, loss: {out['bf16'][3]:.6f} vs {out['fp32'][3]:.6f}")
    assert abs(out["bf16"][3] - out["fp32"][3]) < 2.5e-2 * abs(out["fp32"][3])      # (measured 1.9e-2: -log(1 - p + 1e-6) at p ~ 1)
    M.set_precision("bf16")
    report = {}
    for i, nm in enumerate(("prob", "reg")):
        a, b = out["bf16"][i], out["fp32"][i]
        report[nm] = (float((a - b).abs().max() / b.abs().max()), float((a - b).norm() / b.norm()))
        print(f"bf16 vs fp32 step, {nm} map: max err / max {report[nm][0]:.2e}, rel-L2 {report[nm][1]:.2e}")
    worst = ("", 0.0)
    table = []
    for k, gb in out["bf16"][2].items():
        gf = out["fp32"][2][k]
        if k.endswith("conv.bias") and "prob_conv" not in k and "reg_conv" not in k or k.endswith("deconv.bias"):
            assert float(gb.abs().max()) == 0.0, k      # bias in front of a train-mode BatchNorm: exactly 0
            continue
        assert torch.isfinite(gb).all(), k
        l2 = float((gb - gf).norm() / (gf.norm() + 1e-30))
        cos = float((gb * gf).sum() / (gb.norm() * gf.norm() + 1e-30))
        table.append((k, l2, cos, float(gf.norm())))
        if l2 > worst[1]:
            worst = (k, l2)
    for k, l2, cos, nrm in table:
        print(f"   {k:52s} rel-L2 {l2:.3f}  cos {cos:.3f}  |g_fp32| {nrm:.3e}")
    print("bf16 vs fp32 step, worst parameter gradient rel-L2:", worst)
    for nm, (emax, l2) in report.items():
        assert emax < MAP_MAX and l2 < MAP_L2, (nm, emax, l2)
    # Gradients: see the docstring — the distance is forward-induced (ReLU-mask flips), the bars are measured + margin.
    for k, l2, cos, nrm in table:
        heads, vfe = ("prob_conv" in k or "reg_conv" in k), k.startswith("feature_net")
        assert l2 < (0.1 if heads else VFE_GRAD_L2 if vfe else GRAD_L2), (k, l2, cos)
        assert cos > (0.99 if heads else VFE_GRAD_COS if vfe else GRAD_COS), (k, l2, cos)


def _car_step_inputs():
    """two full-size car frames voxelized on the device + the seeded RPN targets of bench.py"""
    from voxelnet_amd import synth
    from voxelnet_amd.config import grid_config
    from voxelnet_amd.voxelize import voxelize_device
    grid = grid_config("Car")
    feats, coords = [], []
    for b, f in enumerate(synth.workload_frames(2, batch=2)):
        fb, cb, _ = voxelize_device(torch.from_numpy(f).to(DEV), grid, b, coord_cols=4)
        feats.append(fb)
        coords.append(cb)
    rng = np.random.default_rng(99)
    pos = (rng.random((2, 200, 176, 2)) < 0.002).astype(np.float32)
    neg = ((rng.random((2, 200, 176, 2)) < 0.98) & (pos == 0)).astype(np.float32)
    tgt = (rng.standard_normal((2, 200, 176, 14)) * 0.1).astype(np.float32)
    return feats, coords, tuple(torch.from_numpy(a).to(DEV) for a in (pos, neg, tgt))


def _bf16_valued(sd):
    """conv / deconv / head weights rounded to bf16 VALUES: what the bf16 mode's weight packing reads"""
    return {k: (v.bfloat16().float() if k.endswith("conv.weight") or k.endswith("deconv.weight") else v) for k, v in sd.items()}


def _dead_bias(k):
    return (k.endswith("conv.bias") and "prob_conv" not in k and "reg_conv" not in k) or k.endswith("deconv.bias")


def test_bf16_backward_chain_on_frozen_forward():
    """The bf16 backward as a CHAIN (heads -> 23 layers -> VFE, full-size car step, B=2): the gradients of the bf16 mode
    against an exact-fp32 backward (fp32 MFMA kernels, fp32 gradient storage) through THE SAME saved forward — the bf16
    forward's y / a / statistics cast to fp32, so both backwards see identical ReLU masks and normalised values and
    differ only in what the bf16 backward rounds: dy (the MFMA operand), the stored data gradients, the logit gradient.

    Why this and not "bf16 step vs fp32 step" for the chain: tools/grad_attribution.py shows that the step-vs-step
    gradient distance (0.4-0.6 relative L2 below the deconvs) is a property of this network FUNCTION, not of any
    backward arithmetic: the exact fp32 kernels give the same distance as soon as only the conv weights (0.31-0.43), only
    the activations (0.33-0.45) or both (0.40-0.52) are rounded to bf16 values, and promoting every gradient tensor of
    the bf16 mode to fp32 storage changes nothing.  With the forward frozen the chain is measurable."""
    from voxelnet_amd import engine as E
    from voxelnet_amd import model as M
    from voxelnet_amd import net as N
    from voxelnet_amd.engine import Rows
    feats, coords, targets = _car_step_inputs()
    sd = _bf16_valued(tr.make_state_dict("Car"))
    M.set_precision("bf16")
    m = M.RPN3D("Car")
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    fn, mid = m.feature_net, m.middle_rpn
    feature = torch.cat(feats, 0).contiguous()
    coord = torch.cat(coords, 0).contiguous()
    B, K = 2, feature.shape[0]
    vparams = [p.detach() for p in M._vfe_weights(fn)]
    names, P, Bf, flat = M._collect_middle(mid)
    P = M._detached(P)
    P["heads"] = M._heads_params([f.detach() for f in flat[-4:]])
    vw, stats, wst = M.featnet_forward(feature, vparams, fn._bufs(), True)
    dense = M.scatter_rows(vw, coord, B, fn._grid.dims, "bf16")
    vw_rows = vw.bfloat16()
    prob, reg, st = N.middle_forward(dense, P, Bf, mid._block1_stride, True, "bf16", sparse=(coord, vw_rows))
    pl, rl = prob.detach().requires_grad_(), reg.detach().requires_grad_()
    m.loss(pl, rl, *targets)[0].backward()
    d_prob, d_reg = pl.grad, rl.grad
    # ---- the bf16 chain
    Gb, dvw_b = N.middle_backward(st, d_prob, d_reg, P)
    vg_b = M.featnet_backward(feature, wst, stats, dvw_b, vparams)
    # ---- the exact chain through the same saved forward
    st32 = N.MiddleState()
    st32.layers, st32.block1_stride, st32.mode, st32.prob, st32.fmap = {}, st.block1_stride, "fp32", st.prob, st.fmap
    st32.sparse = (coord, vw_rows.float())
    for name, s in st.layers.items():
        t = E.LayerState()
        t.spec, t.in_dims, t.out_dims, t.stats = s.spec, s.in_dims, s.out_dims, s.stats
        t.x = s.x if name == "middle_layer.0" else Rows(s.x.t.float(), s.x.C)     # (the dense grid is not read by the sparse backward)
        t.y = Rows(s.y.t.float(), s.y.C)
        t.a = Rows(s.a.t.float(), s.a.C) if s.a is not None else None
        st32.layers[name] = t
    Gf, dvw_f = N.middle_backward(st32, d_prob, d_reg, P)
    vg_f = M.featnet_backward(feature, wst, stats, dvw_f, vparams)
    torch.cuda.synchronize()

    def dist(a, b):
        a, b = a.double(), b.double()
        return float((a - b).norm() / (b.norm() + 1e-30)), float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
    rows = [("d_vw (K,128)",) + dist(dvw_b, dvw_f)]
    for n in names + ["heads"]:
        for k in ("weight", "gamma", "beta") if n != "heads" else ("weight", "bias"):
            rows.append((f"{n}.{k}",) + dist(Gb[n][k], Gf[n][k]))
    for key, a, b in zip(M.VFE_KEYS, vg_b, vg_f):
        rows.append((key,) + dist(a, b))
    for nm, l2, cos in rows:
        print(f"   frozen forward, bf16 vs exact backward: {nm:44s} rel-L2 {l2:.4f}  cos {cos:.5f}")
    worst = max(rows, key=lambda r: r[1])
    print("frozen forward: worst", worst)
    # ---- the native executor's bf16 step (fused epilogues, first-layer shortcuts, two streams) against the same chain
    res = m((None, None, feats, None, coords, None, None), DEV, targets=targets)
    res[2].backward()
    torch.cuda.synchronize()
    nat = {k: p.grad.detach() for k, p in m.named_parameters()}
    chain = {}
    for n in names:
        cv = "deconv" if n.startswith("deconv") else "conv"
        chain[f"middle_rpn.{n}.{cv}.weight"] = Gb[n]["weight"]
        chain[f"middle_rpn.{n}.batch_norm.weight"] = Gb[n]["gamma"]
        chain[f"middle_rpn.{n}.batch_norm.bias"] = Gb[n]["beta"]
    chain["middle_rpn.prob_conv.conv.weight"], chain["middle_rpn.reg_conv.conv.weight"] = Gb["heads"]["weight"][:2], Gb["heads"]["weight"][2:]
    chain["middle_rpn.prob_conv.conv.bias"], chain["middle_rpn.reg_conv.conv.bias"] = Gb["heads"]["bias"][:2], Gb["heads"]["bias"][2:]
    for key, g in zip(M.VFE_KEYS, vg_b):
        chain[key] = g
    # (printed, not asserted beyond the maps: the two bf16 forwards differ only in the summation order of the first layer —
    #  rulebook P rows vs row-list gather — and this random-weight BatchNorm/ReLU stack amplifies every perturbation ~1.2x
    #  per layer (tools/forward_error_profile.py; the fp32 mode's own order-only difference grows 8.6e-8 -> 1.2e-5): the
    #  maps come out 0.6 % / 3 % apart and the gradients 0.1-0.4, exactly like bf16 vs fp32 — a property of the function)
    mp, mr = dist(res[0].detach(), prob), dist(res[1].detach(), reg)
    print("native maps vs per-layer maps (rel-L2, cos):", mp, mr)
    nworst = ("", 0.0, 1.0)
    for k, g in chain.items():
        l2, cos = dist(nat[k], g)
        if l2 > nworst[1]:
            nworst = (k, l2, cos)
    print("native executor vs per-layer orchestration (both bf16): worst gradient", nworst)
    assert worst[1] < FROZEN_L2 and min(r[2] for r in rows) > FROZEN_COS, worst
    assert mp[0] < MAP_L2 and mr[0] < MAP_L2, (mp, mr)


def test_heads_streaming_kernels():
    """vn_heads_fwd / vn_heads_dgrad (the two 1x1 heads of model.py:276-281 as streaming kernels over the 768-channel concat)
    against float64 on the bf16 operands the kernels read: fp32 outputs within the forward-error bound of an fp32 sum, the
    bf16 data gradient correctly rounded (half a bf16 ulp + accumulation noise), at the full 2 x 200 x 176 map and on a
    site count that is not a multiple of 16"""
    from voxelnet_amd import _lib, engine as E
    for B, S in ((2, 200 * 176), (1, 1003)):
        M = B * S
        cat = bf16r(seeded((M, 768), 9100 + S, 0.7))
        w = bf16r(seeded((16, 768), 9200 + S, 0.05))
        bias = seeded((16,), 9300 + S, 0.1)
        cat_d = cat.to(torch.bfloat16).to(DEV)
        wf = w.to(torch.bfloat16).to(DEV).contiguous()                      # [16][768]: vn_pack_weight mode 0 of the (16,768,1,1) weight
        wd = w.t().contiguous().to(torch.bfloat16).to(DEV)                  # [768][16]: mode 1
        prob = torch.full((B, 2, S), float("nan"), device=DEV)
        reg = torch.full((B, 14, S), float("nan"), device=DEV)
        _lib.call("vn_heads_fwd", cat_d.data_ptr(), 768, wf.data_ptr(), bias.to(DEV).data_ptr(), B, S, prob.data_ptr(), reg.data_ptr(),
                  E.stream())
        lin = cat.double() @ w.double().t() + bias.double()                  # (M,16)
        terms = cat.double().abs() @ w.double().abs().t() + bias.double().abs()
        lin_b = lin.reshape(B, S, 16).permute(0, 2, 1)
        terms_b = terms.reshape(B, S, 16).permute(0, 2, 1)
        assert_fp32_sum(reg, lin_b[:, 2:], terms_b[:, 2:], "heads reg")
        p_ref = torch.sigmoid(lin_b[:, :2])
        assert float((prob.double().cpu() - p_ref).abs().max()) < 1e-6 + 2e-5 * float(terms_b[:, :2].max()) * 0.25
        # ---- data gradient
        g = bf16r(seeded((M, 16), 9400 + S, 0.3))
        g_d = g.to(torch.bfloat16).to(DEV)
        dcat = torch.full((M, 768), float("nan"), dtype=torch.bfloat16, device=DEV)
        _lib.call("vn_heads_dgrad", g_d.data_ptr(), 16, wd.data_ptr(), dcat.data_ptr(), 768, M, E.stream())
        assert_rounded(dcat.float(), (g.double() @ w.double()).numpy(), "heads data gradient")
