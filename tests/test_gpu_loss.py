"""Fused RPN loss (csrc/loss.hip through vn_rpn_loss_fwd / vn_rpn_loss_bwd) against the oracle restatement of
voxelnet/model.py:309-352 + voxelnet/loss.py run in fp64 on the CPU.  Tolerance: 1e-5 relative on the five
scalars, 1e-5 of the gradient's max on the gradients (fp32 arithmetic, fixed summation order)."""
import numpy as np
import pytest
import torch

from oracle import torch_ref as tr

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _case(B, H, W, seed, empty_sample=False):
    g = torch.Generator().manual_seed(seed)
    prob = torch.rand((B, 2, H, W), generator=g) * 0.98 + 0.01
    prob[0, 0, 0, :4] = torch.tensor([0.0, 1.0, 1e-7, 1 - 1e-7])[: min(4, W)]      # the 1e-6 guards (model.py:340-341)
    delta = torch.randn((B, 14, H, W), generator=g) * 0.2
    pos = (torch.rand((B, H, W, 2), generator=g) < 0.05).float()
    neg = ((torch.rand((B, H, W, 2), generator=g) < 0.8).float() * (1 - pos))
    if empty_sample:
        pos[B - 1] = 0          # no positive anchor in the last sample: normaliser clips to 1 (model.py:313-317)
    tgt = torch.randn((B, H, W, 14), generator=g) * 0.2
    tgt[..., 3] += 0.5          # some |diff| beyond 1/sigma^2: both smooth-L1 branches (loss.py:7-11)
    return prob, delta, pos, neg, tgt


@pytest.mark.parametrize("B,H,W,empty", [(2, 16, 24, False), (3, 7, 5, True), (2, 200, 176, False)])
def test_loss_values_and_gradients(B, H, W, empty):
    from voxelnet_amd import model as M
    prob, delta, pos, neg, tgt = _case(B, H, W, 7 + B, empty)
    gw = torch.tensor([1.0, 0.3, -0.7, 0.11, 2.0])                  # upstream gradients of the five outputs
    # oracle, fp64
    p64, d64 = prob.double().requires_grad_(True), delta.double().requires_grad_(True)
    ref = torch.stack(tr.rpn_loss(p64, d64, pos.double(), neg.double(), tgt.double(), 1.5, 1.0, 3.0))
    (ref * gw.double()).sum().backward()
    # HIP path
    pg, dg = prob.to(DEV).requires_grad_(True), delta.to(DEV).requires_grad_(True)
    out = M._LossFn.apply(pg, dg, pos.to(DEV), neg.to(DEV), tgt.to(DEV), 1.5, 1.0, 3.0)     # five scalars
    assert len(out) == 5 and all(o.dim() == 0 for o in out)
    sum(o * w for o, w in zip(out, gw.to(DEV))).backward()
    np.testing.assert_allclose([o.item() for o in out], ref.detach().numpy(), rtol=1e-5, atol=1e-6)
    for got, want in ((pg.grad, p64.grad), (dg.grad, d64.grad)):
        got, want = got.cpu().double(), want
        assert (got - want).abs().max().item() <= 1e-5 * want.abs().max().item() + 1e-12
    # the usual case: only the total loss is differentiated (the other four upstream gradients arrive as None)
    p64b, d64b = prob.double().requires_grad_(True), delta.double().requires_grad_(True)
    tr.rpn_loss(p64b, d64b, pos.double(), neg.double(), tgt.double(), 1.5, 1.0, 3.0)[0].backward()
    pg2, dg2 = prob.to(DEV).requires_grad_(True), delta.to(DEV).requires_grad_(True)
    M._LossFn.apply(pg2, dg2, pos.to(DEV), neg.to(DEV), tgt.to(DEV), 1.5, 1.0, 3.0)[0].backward()
    for got, want in ((pg2.grad, p64b.grad), (dg2.grad, d64b.grad)):
        assert (got.cpu().double() - want).abs().max().item() <= 1e-5 * want.abs().max().item() + 1e-12


def test_loss_module_api(golden):
    """RPN3D.loss keeps the reference's 5-tuple and accepts the numpy arrays generate_targets returns"""
    from voxelnet_amd import model as M
    m = M.RPN3D("Car").to(DEV)
    prob, delta, pos, neg, tgt = _case(2, 8, 8, 3)
    out = m.loss(prob.to(DEV), delta.to(DEV), pos.numpy(), neg.numpy(), tgt.numpy())
    assert len(out) == 5 and all(o.dim() == 0 for o in out)
    ref = tr.rpn_loss(prob.double(), delta.double(), pos.double(), neg.double(), tgt.double(), m.alpha, m.beta, m.sigma)
    np.testing.assert_allclose([o.item() for o in out], [r.item() for r in ref], rtol=1e-5)


def test_loss_rejects_bad_shapes():
    from voxelnet_amd import model as M
    prob, delta, pos, neg, tgt = [t.to(DEV) for t in _case(2, 8, 8, 3)]
    with pytest.raises(ValueError):
        M._LossFn.apply(prob, delta[:, :7], pos, neg, tgt, 1.5, 1.0, 3.0)


@pytest.mark.parametrize("B,H,W,empty", [(2, 16, 24, False), (3, 7, 5, True), (2, 200, 176, False)])
def test_loss_in_three_pieces_equals_the_two_passes_bit_for_bit(B, H, W, empty):
    """vn_rpn_loss_norm + vn_rpn_loss_fwd_bwd (ONE pass: sums and gradients) + vn_rpn_loss_finalize — the form vn_net_step
    schedules — against vn_rpn_loss_fwd + vn_rpn_loss_bwd: the five scalars and both gradients bit-identical, with all five
    upstream gradients set and with only g_loss (NULL for the others)"""
    from voxelnet_amd import _lib
    prob, delta, pos, neg, tgt = (t.to(DEV).contiguous() for t in _case(B, H, W, 21 + B, empty))
    lib = _lib.load()
    wsb = lib.vn_rpn_loss_workspace_bytes(B, H, W)
    st = _lib.raw_stream()
    gw = [torch.tensor([v], device=DEV) for v in (1.0, 0.3, -0.7, 0.11, 2.0)]
    for ups in (gw, [gw[0], None, None, None, None]):
        gp = [None if g is None else g.data_ptr() for g in ups]
        ws_a = torch.zeros(wsb, dtype=torch.uint8, device=DEV)
        out_a = torch.empty(5, device=DEV)
        dp_a, dd_a = torch.empty_like(prob), torch.empty_like(delta)
        _lib.call("vn_rpn_loss_fwd", prob.data_ptr(), delta.data_ptr(), pos.data_ptr(), neg.data_ptr(), tgt.data_ptr(), B, H, W,
                  1.5, 1.0, 3.0, ws_a.data_ptr(), wsb, out_a.data_ptr(), st)
        _lib.call("vn_rpn_loss_bwd", prob.data_ptr(), delta.data_ptr(), pos.data_ptr(), neg.data_ptr(), tgt.data_ptr(), B, H, W,
                  1.5, 1.0, 3.0, ws_a.data_ptr(), *gp, dp_a.data_ptr(), dd_a.data_ptr(), st)
        ws_b = torch.full((wsb,), 255, dtype=torch.uint8, device=DEV)
        out_b = torch.empty(5, device=DEV)
        dp_b, dd_b = torch.empty_like(prob), torch.empty_like(delta)
        _lib.call("vn_rpn_loss_norm", pos.data_ptr(), neg.data_ptr(), B, H, W, ws_b.data_ptr(), wsb, st)
        _lib.call("vn_rpn_loss_fwd_bwd", prob.data_ptr(), delta.data_ptr(), pos.data_ptr(), neg.data_ptr(), tgt.data_ptr(), B, H, W,
                  1.5, 1.0, 3.0, ws_b.data_ptr(), wsb, *gp, dp_b.data_ptr(), dd_b.data_ptr(), st)
        _lib.call("vn_rpn_loss_finalize", ws_b.data_ptr(), wsb, B, H, W, 1.5, 1.0, out_b.data_ptr(), st)
        torch.cuda.synchronize()
        assert torch.equal(out_a, out_b), (out_a, out_b)
        assert torch.equal(dp_a, dp_b) and torch.equal(dd_a, dd_b)
        assert torch.isfinite(out_a).all()


@pytest.mark.parametrize("B,H,W,empty,f32", [(2, 16, 24, False, False), (3, 7, 5, True, True), (2, 200, 176, False, False),
                                              (2, 200, 176, False, True)])
def test_loss_pass_writes_the_heads_gradient_rows_bit_for_bit(B, H, W, empty, f32):
    """vn_rpn_loss_fwd_bwd_rows (what vn_net_step launches between the heads and their backward) against
    vn_rpn_loss_fwd_bwd followed by vn_heads_bwd: d_prob, d_delta, the partial sums and the (B*H*W, 16) gradient rows of the
    heads' backward — d_logit = d_prob * p * (1 - p), then the 14 regression gradients (model.py:303-304 backward) —
    bit-identical, as bf16 rows (the bf16 mode) and as fp32 rows (the fp32 / fp32x3 modes)"""
    from voxelnet_amd import _lib
    prob, delta, pos, neg, tgt = (t.to(DEV).contiguous() for t in _case(B, H, W, 33 + B, empty))
    lib = _lib.load()
    wsb = lib.vn_rpn_loss_workspace_bytes(B, H, W)
    st = _lib.raw_stream()
    g = torch.tensor([0.7], device=DEV)
    S = H * W
    rdt, cdt = (torch.float32, _lib.VN_F32) if f32 else (torch.bfloat16, _lib.VN_BF16)
    ws_a = torch.zeros(wsb, dtype=torch.uint8, device=DEV)
    dp_a, dd_a = torch.empty_like(prob), torch.empty_like(delta)
    rows_a = torch.full((B * S, 16), float("nan"), dtype=rdt, device=DEV)
    _lib.call("vn_rpn_loss_norm", pos.data_ptr(), neg.data_ptr(), B, H, W, ws_a.data_ptr(), wsb, st)
    _lib.call("vn_rpn_loss_fwd_bwd", prob.data_ptr(), delta.data_ptr(), pos.data_ptr(), neg.data_ptr(), tgt.data_ptr(), B, H, W,
              1.5, 1.0, 3.0, ws_a.data_ptr(), wsb, g.data_ptr(), None, None, None, None, dp_a.data_ptr(), dd_a.data_ptr(), st)
    _lib.call("vn_heads_bwd", dp_a.data_ptr(), dd_a.data_ptr(), prob.data_ptr(), B, S, rows_a.data_ptr(), cdt, 16, 0, st)
    ws_b = torch.zeros(wsb, dtype=torch.uint8, device=DEV)
    dp_b, dd_b = torch.empty_like(prob), torch.empty_like(delta)
    rows_b = torch.full((B * S, 16), float("nan"), dtype=rdt, device=DEV)
    _lib.call("vn_rpn_loss_norm", pos.data_ptr(), neg.data_ptr(), B, H, W, ws_b.data_ptr(), wsb, st)
    _lib.call("vn_rpn_loss_fwd_bwd_rows", prob.data_ptr(), delta.data_ptr(), pos.data_ptr(), neg.data_ptr(), tgt.data_ptr(), B, H, W,
              1.5, 1.0, 3.0, ws_b.data_ptr(), wsb, g.data_ptr(), dp_b.data_ptr(), dd_b.data_ptr(), rows_b.data_ptr(), cdt, 16, 0, st)
    torch.cuda.synchronize()
    assert torch.equal(dp_a, dp_b) and torch.equal(dd_a, dd_b)
    assert torch.equal(ws_a, ws_b)
    assert torch.isfinite(rows_a.float()).all()
    assert torch.equal(rows_a.view(torch.int16 if not f32 else torch.int32), rows_b.view(torch.int16 if not f32 else torch.int32))
