"""GPU: drop-in boundary behaviour beyond the fused train path.

  * VFELayer.forward(inputs, mask) on its own (model.py:74-82) — forward and backward vs the float64 oracle;
    two stand-alone layers + max == the fused FeatureLearningNet kernels
  * gradient accumulation semantics of the native backward (zero_grad(set_to_none=False), two backward() calls)
  * backward through an eval-mode forward: the eval-mode BatchNorm backward of the reference's autograd (round 3; it raised before)
  * two RPN3D instances in one process do not share executor state (vn_net_create contexts)
  * pcl_to_voxels: device path == host entry, bit for bit
  * vn_comm_* / vn_allreduce_bucket (RCCL bound at run time) on a one-rank communicator, and the reducer's direct path"""
import ctypes
from dataclasses import replace

import numpy as np
import pytest
import torch

from oracle import torch_ref as tr

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-12))


def tiny_batch(golden):
    g = golden("middle_tiny_car")
    feats = torch.from_numpy(g["features"])
    coords = torch.from_numpy(g["coords"])
    lens = [int(x) for x in g["feat_lens"]]
    return [f.to(DEV) for f in torch.split(feats, lens)], [c.to(DEV) for c in torch.split(coords, lens)]


def make_model(mode="fp32", cls="Car"):
    from voxelnet_amd import model as M
    M.set_precision(mode)
    m = M.RPN3D(cls)
    m.load_state_dict(tr.make_state_dict(cls))
    m.feature_net._grid = replace(m.feature_net._grid, H=16, W=24)
    return m.to(DEV)


@pytest.mark.parametrize("cin,cout,T,training", [(7, 32, 35, True), (32, 128, 35, True), (7, 32, 5, False), (32, 128, 45, False)])
def test_vfe_layer_standalone_vs_oracle(cin, cout, T, training):
    from voxelnet_amd import model as M
    rng = np.random.default_rng(1000 + cin + T)
    K = 257
    x = torch.from_numpy(rng.standard_normal((K, T, cin)).astype(np.float32))
    npts = rng.integers(1, T + 1, size=K)
    mask = torch.from_numpy((np.arange(T)[None, :] < npts[:, None])[..., None])          # (K,T,1) bool, model.py:95-96
    x = x * mask                                                                        # padded slots are zero rows
    layer = M.VFELayer(cin, cout)
    sd = {"L.fcn.0.weight": tr._fill((cout // 2, cin), 11, 1.0 / np.sqrt(cin)), "L.fcn.0.bias": tr._fill((cout // 2,), 12, 0.1),
          "L.bn.weight": 1.0 + tr._fill((cout // 2,), 13, 0.2), "L.bn.bias": tr._fill((cout // 2,), 14, 0.1),
          "L.bn.running_mean": tr._fill((cout // 2,), 15, 0.3), "L.bn.running_var": 1.0 + tr._fill((cout // 2,), 16, 0.5).abs(),
          "L.bn.num_batches_tracked": torch.zeros((), dtype=torch.long)}
    layer.load_state_dict({k[2:]: v.clone() for k, v in sd.items()})
    layer = layer.to(DEV).train(training)
    xg = x.to(DEV).requires_grad_(True)
    out = layer(xg, mask.to(DEV))
    assert out.shape == (K, T, cout)
    up = torch.from_numpy(rng.standard_normal((K, T, cout)).astype(np.float32))
    out.backward(up.to(DEV))
    sd64 = {k: (v.double().clone() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    leaves = {k: sd64[k].requires_grad_(True) for k in ("L.fcn.0.weight", "L.fcn.0.bias", "L.bn.weight", "L.bn.bias")}
    x64 = x.double().requires_grad_(True)
    ref = tr.vfe_layer(x64, mask, sd64, "L", training)
    ref.backward(up.double())
    assert rel(out, ref) < 1e-5
    assert rel(xg.grad, x64.grad) < 1e-4
    assert rel(layer.fcn[0].weight.grad, leaves["L.fcn.0.weight"].grad) < 1e-4
    assert rel(layer.fcn[0].bias.grad, leaves["L.fcn.0.bias"].grad) < 1e-4
    assert rel(layer.bn.weight.grad, leaves["L.bn.weight"].grad) < 1e-4
    assert rel(layer.bn.bias.grad, leaves["L.bn.bias"].grad) < 1e-4
    if training:
        assert rel(layer.bn.running_mean, sd64["L.bn.running_mean"]) < 1e-5      # updated in place by F.batch_norm
        assert rel(layer.bn.running_var, sd64["L.bn.running_var"]) < 1e-5
        assert int(layer.bn.num_batches_tracked) == 1
    with pytest.raises(M._lib.VoxelnetHipError):
        layer(x, mask)                                                           # CPU tensors: no fallback


def test_two_standalone_layers_equal_the_fused_feature_net(golden):
    """model.py:93-100 composed by hand from two VFELayer calls == the fused kernels of FeatureLearningNet"""
    feats, coords = tiny_batch(golden)
    m = make_model("fp32").train()
    fn = m.feature_net
    feature = torch.cat(feats)
    mask = feature.max(dim=2, keepdim=True)[0] != 0
    import copy
    fn2 = copy.deepcopy(fn)
    x = fn2.vfe_2(fn2.vfe_1(feature, mask), mask)
    voxelwise = x.max(dim=1)[0]
    dense = fn(feats, coords)
    c = torch.cat(coords)
    rows = dense[c[:, 0], c[:, 1], c[:, 2], c[:, 3]]
    assert rel(voxelwise, rows) < 1e-5
    assert rel(fn2.vfe_2.bn.running_var, fn.vfe_2.bn.running_var) < 1e-5


def test_gradient_accumulation_and_zero_grad_in_place(golden):
    """p.grad present at backward time (zero_grad(set_to_none=False), or a second backward before zero_grad): the native
    path must ADD, like autograd does — not overwrite, not double."""
    feats, coords = tiny_batch(golden)
    rng = np.random.default_rng(5)
    ups = [(torch.from_numpy((rng.standard_normal((2, 2, 8, 12)) * 0.1).astype(np.float32)).to(DEV),
            torch.from_numpy((rng.standard_normal((2, 14, 8, 12)) * 0.1).astype(np.float32)).to(DEV)) for _ in range(2)]

    def run(m, up):
        prob, reg = m.detect(feats, coords)
        torch.autograd.backward([prob, reg], list(up))

    singles = []
    for up in ups:
        m = make_model("fp32").train()
        run(m, up)
        singles.append({k: p.grad.detach().clone() for k, p in m.named_parameters()})
    m = make_model("fp32").train()
    run(m, ups[0])
    run(m, ups[1])                              # no zero_grad in between: sums
    # (BatchNorm running statistics differ between the runs, batch statistics — which the gradients use — do not)
    for k, p in m.named_parameters():
        want = singles[0][k] + singles[1][k]
        assert torch.allclose(p.grad, want, rtol=1e-5, atol=1e-6 * float(want.abs().max()) + 1e-12), k
    for p in m.parameters():
        p.grad.zero_()                          # == optimizer.zero_grad(set_to_none=False)
    run(m, ups[0])
    for k, p in m.named_parameters():
        assert torch.allclose(p.grad, singles[0][k], rtol=1e-5, atol=1e-6 * float(singles[0][k].abs().max()) + 1e-12), k
    # and the usual path afterwards (set_to_none=True) is unchanged
    for p in m.parameters():
        p.grad = None
    run(m, ups[1])
    for k, p in m.named_parameters():
        assert torch.equal(p.grad, singles[1][k]), k


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_backward_through_eval_forward(golden, mode):
    """model.eval(); forward; backward — the reference's autograd gives gradients through ConvMD / DeConv2d / VFELayer with
    BatchNorm on its RUNNING statistics (model.py:76, 158-167, 195-199 with self.training False): mean and invstd are
    constants, dy = gamma * invstd * dz, and the conv biases in front of a BatchNorm get a real gradient.  One train-mode
    forward first, so that the running statistics are not their initial 0 / 1.  fp32: maps 1e-3, every gradient (the conv
    biases included) within 1e-3 relative L2 of the oracle (measured 4e-6: no batch statistics, so no chaotic coupling
    between sites); bf16: finite, maps within the bf16 band, gradients cos > 0.9.  The inference call pattern under
    no_grad stays on the native executor."""
    from voxelnet_amd import model as M
    feats, coords = tiny_batch(golden)
    m = make_model(mode).train()
    sd = tr.make_state_dict("Car")
    cf, cc = [f.cpu() for f in feats], [c.cpu() for c in coords]
    with torch.no_grad():
        m.detect(feats, coords)                                        # train-mode forward: running statistics move
        tr.middle_rpn(tr.feature_net(cf, cc, sd, (10, 16, 24), True), sd, "Car", True)
    m.eval()
    rng = np.random.default_rng(12)
    dp = torch.from_numpy((rng.standard_normal((2, 2, 8, 12)) * 1e-2).astype(np.float32))
    dr = torch.from_numpy((rng.standard_normal((2, 14, 8, 12)) * 1e-2).astype(np.float32))
    rp, rr, ref = tr.forward_backward(cf, cc, sd, (10, 16, 24), "Car", dp, dr, training=False)
    prob, reg = m.detect(feats, coords)
    tol = 1e-3 if mode == "fp32" else 0.15
    assert rel(prob, rp) < tol and rel(reg, rr) < tol, (rel(prob, rp), rel(reg, rr))
    torch.autograd.backward([prob, reg], [dp.to(DEV), dr.to(DEV)])
    torch.cuda.synchronize()
    worst = ("", 0.0)
    for k, p in m.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
        r = ref[k].double()
        g = p.grad.double().cpu()
        l2 = float((g - r).norm() / (r.norm() + 1e-30))
        cos = float((g * r).sum() / (g.norm() * r.norm() + 1e-30))
        if mode == "fp32":
            assert l2 < 1e-3, (k, l2)
        else:
            assert cos > 0.9, (k, l2, cos)
        if l2 > worst[1]:
            worst = (k, l2)
    # the conv biases in front of a BatchNorm are NOT zero in eval mode
    assert float(m.middle_rpn.block1[1].conv.bias.grad.abs().max()) > 0
    print(f"eval-mode backward, {mode}: worst gradient rel-L2", worst)
    with torch.no_grad():                       # the inference call pattern (predict.py:58-60): native executor, unaffected
        m.detect(feats, coords)
    M.set_precision("bf16")


def test_two_models_do_not_share_executor_state(golden):
    """interleaved steps of two RPN3D instances (own vn_net_create contexts, arenas, gradient buffers) == separate runs"""
    feats, coords = tiny_batch(golden)
    up = (torch.full((2, 2, 8, 12), 0.1, device=DEV), torch.full((2, 14, 8, 12), 0.1, device=DEV))

    def grads(m):
        return {k: p.grad.detach().clone() for k, p in m.named_parameters()}

    a = make_model("bf16").train()
    pa, ra = a.detect(feats, coords)
    torch.autograd.backward([pa, ra], list(up))
    ref = grads(a)
    a2, b2 = make_model("bf16").train(), make_model("bf16").train()
    p1, r1 = a2.detect(feats, coords)
    p2, r2 = b2.detect([f[:50] for f in feats], [c[:50] for c in coords])      # another instance, other inputs, in between
    torch.autograd.backward([p2, r2], list(up))
    torch.autograd.backward([p1, r1], list(up))
    assert torch.equal(p1, pa) and torch.equal(r1, ra)
    got = grads(a2)
    assert all(torch.equal(got[k], ref[k]) for k in ref)
    assert a2._net_handle(DEV).value != b2._net_handle(DEV).value


def test_pcl_to_voxels_device_equals_host():
    from voxelnet_amd import synth
    from voxelnet_amd.voxelize import pcl_to_voxels
    cloud = synth.synth_cloud("Car", 2500, 77)
    a, b = cloud.copy(), cloud.copy()
    np.random.seed(3)
    da = pcl_to_voxels(a, "Car", device=DEV)
    np.random.seed(3)
    db = pcl_to_voxels(b, "Car", device="cpu")
    assert np.array_equal(a, b)                                              # both shuffled their argument the same way
    for k in ("feature_buffer", "coordinate_buffer", "number_buffer"):
        assert da[k].dtype == db[k].dtype and np.array_equal(da[k].view(np.uint8), db[k].view(np.uint8)), k


def test_rccl_wrapper_one_rank_and_reducer_direct_path(golden):
    from voxelnet_amd import _lib, parallel
    lib = _lib.load()
    ident = (ctypes.c_ubyte * 128)()
    rc = lib.vn_comm_unique_id(ident)
    if rc == -2:
        pytest.skip("no librccl on this box")
    assert rc == 0
    comm = ctypes.c_void_p()
    torch.cuda.set_device(0)
    _lib.call("vn_comm_create", ctypes.byref(comm), ident, 1, 0)
    x = torch.arange(1000003, dtype=torch.float32, device=DEV)
    want = x * 0.5
    st = torch.cuda.current_stream()
    _lib.call("vn_allreduce_bucket", comm, x.data_ptr(), x.numel(), 0.5, ctypes.c_void_p(st.cuda_stream))
    torch.cuda.synchronize()
    assert torch.equal(x, want)
    assert lib.vn_allreduce_bucket(comm, x.data_ptr() + 4, 10, 1.0, None) == -1      # misaligned bucket
    assert lib.vn_comm_destroy(comm) == 0
    # the reducer through the same wrapper (world size 1: mean == identity), native bucket-event path
    feats, coords = tiny_batch(golden)
    g = golden("rpn3d_tiny")
    batch = (["a", "b"], None, [f.cpu() for f in feats], None, [c.cpu() for c in coords], None, None)
    res = []
    for direct in (False, True):
        m = make_model("fp32").train()
        named = list(m.named_parameters())
        if direct:
            m.grad_reducer = parallel.GradAllReducer(named, direct_rccl=True)
            assert m.grad_reducer.comm is not None
        out = m(batch, DEV, targets=(g["pos"], g["neg"], g["targets"]))
        out[2].backward()
        if direct:
            m.grad_reducer.finish(named)
        torch.cuda.synchronize()
        res.append({k: p.grad.detach().clone() for k, p in named})
        if direct:
            m.grad_reducer.close()
    for k in res[0]:
        assert torch.allclose(res[0][k], res[1][k], rtol=1e-5, atol=1e-7), k


def test_prepare_protocol_is_explicit_and_checked(golden):
    """vn_net_prepare's phases are named by cfg->prepared (0 whole / 1 first layer's needs / 2 the rest) and checked: a
    phase 2 without (or for another step than) its phase 1, and a vn_net_forward that claims `prepared` without a prepare
    for ITS workspace / K, return VN_EINVAL instead of running on stale packed weights (round-3 advisor finding); an
    error leaves no half-issued state behind, and the normal sequence still works afterwards."""
    from voxelnet_amd import _lib
    from voxelnet_amd import engine as E
    from voxelnet_amd import model as M
    feats, coords = tiny_batch(golden)
    m = make_model("bf16").train()
    lib = _lib.load()
    mid = m.middle_rpn
    arr, _ = M._native_layer_arrays(mid)
    coord = torch.cat(coords).contiguous()
    K = coord.shape[0]
    cfg = _lib.VnNetConfig(2, 10, 16, 24, mid._block1_stride, 0, 1, 1, 0, 0)
    nb = lib.vn_net_workspace_bytes(ctypes.byref(cfg), K)
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    ws2 = torch.empty(nb, dtype=torch.uint8, device=DEV)
    net = m._net_handle(torch.device(DEV))
    heads = M._heads_params([p.detach() for p in m._flat_params()[-4:]])
    hw, hb = heads["weight"], heads["bias"]
    side = torch.cuda.Stream()
    st, sd = E.stream(), ctypes.c_void_p(side.cuda_stream)
    EINVAL = -1

    def prepare(phase, w=ws, k=K, heads_ptr=hw.data_ptr()):
        cfg.prepared = phase
        return lib.vn_net_prepare(net, ctypes.byref(cfg), arr, heads_ptr if phase != 1 else None, coord.data_ptr(), k, w.data_ptr(), nb, sd)

    vw = torch.zeros((K, 128), dtype=torch.bfloat16, device=DEV)
    prob = torch.empty((2, 2, 8, 12), device=DEV)
    reg = torch.empty((2, 14, 8, 12), device=DEV)

    def forward(w=ws, prepared=1):
        cfg.prepared = prepared
        return lib.vn_net_forward(net, ctypes.byref(cfg), arr, hw.data_ptr(), hb.data_ptr(), None, coord.data_ptr(), vw.data_ptr(), K,
                                  w.data_ptr(), nb, prob.data_ptr(), reg.data_ptr(), st, sd)
    torch.cuda.synchronize()
    assert prepare(2) == EINVAL                                   # phase 2 without a phase 1
    assert forward() == EINVAL                                    # "prepared" without any prepare
    assert prepare(1) == 0 and prepare(2, w=ws2) == EINVAL        # phase 2 for another workspace
    assert prepare(2) == EINVAL                                   # ... and the failed call cleared the first phase
    assert prepare(1) == 0 and prepare(2, k=K - 1) == EINVAL      # phase 2 for another K
    assert prepare(0, heads_ptr=None) == EINVAL                   # the one-call form needs the heads' weights
    assert prepare(1) == 0 and prepare(2) == 0
    assert forward(w=ws2) == EINVAL                               # prepared, but for another arena
    assert forward() == EINVAL                                    # ... which also consumed / dropped the prepare
    assert prepare(1) == 0 and prepare(2) == 0 and forward() == 0
    assert forward() == EINVAL                                    # one forward per prepare
    assert prepare(0) == 0 and forward() == 0                     # the one-call form
    assert forward(prepared=0) == 0                               # and the forward that prepares for itself
    torch.cuda.synchronize()
    assert torch.isfinite(prob).all() and torch.isfinite(reg).all()
    # the module path (two-phase prepare on its side stream) is unaffected
    p1, r1 = m.detect(feats, coords)
    torch.cuda.synchronize()
    assert torch.isfinite(p1).all()
