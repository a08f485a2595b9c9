"""Optimizer tail (csrc/optim.hip through vn_clip_sgd / voxelnet_amd.optim.ClipSGD) against the oracle restatement of
voxelnet/train.py:153-154 (clip_grad_norm_(params, 5) + SGD(lr=0.01).step()).  fp32 arithmetic with a different
summation order for the norm: tolerance 1e-6 relative on the norm, 2 ulp-ish (1e-6 relative + 1e-9) on parameters."""
import pytest
import torch

from oracle import torch_ref as tr

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# odd sizes: chunk tails, tensors that start off a 16-byte boundary inside a flat buffer, a tensor of one element
SHAPES = [(16, 7), (16,), (3,), (1,), (64, 128, 3, 3, 3), (4097,), (8191,), (2, 768, 1, 1), (5, 3)]


def _tensors(scale, seed, flat):
    g = torch.Generator().manual_seed(seed)
    ps = [torch.randn(s, generator=g) for s in SHAPES]
    gs = [torch.randn(s, generator=g) * scale for s in SHAPES]
    if not flat:
        return ps, gs, [p.to(DEV) for p in ps], [x.to(DEV) for x in gs]
    n = sum(p.numel() for p in ps)
    fp, fg = torch.empty(n, device=DEV), torch.empty(n, device=DEV)
    dp, dg, off = [], [], 0
    for p, x in zip(ps, gs):
        dp.append(fp[off:off + p.numel()].view_as(p).copy_(p))
        dg.append(fg[off:off + p.numel()].view_as(p).copy_(x))
        off += p.numel()
    return ps, gs, dp, dg


@pytest.mark.parametrize("scale,flat,scale_grads", [(1.0, False, True), (1e-4, True, False), (1.0, True, True)])
def test_clip_sgd_matches_oracle(scale, flat, scale_grads):
    from voxelnet_amd.optim import ClipSGD
    ps, gs, dp, dg = _tensors(scale, 11, flat)
    params = [torch.nn.Parameter(p) for p in dp]
    for p, g in zip(params, dg):
        p.grad = g
    g_before = [g.clone() for g in dg]
    opt = ClipSGD(params, lr=0.01, max_norm=5.0, scale_grads=scale_grads)
    total = opt.step()
    ref_p, ref_g, ref_total = tr.clip_sgd_step(ps, gs, 0.01, 5.0)
    assert abs(total.item() - ref_total.item()) <= 1e-6 * ref_total.item()
    clipped = ref_total.item() > 5.0
    assert clipped == (scale == 1.0)                      # the cases cover both sides of the clamp
    for p, rp in zip(params, ref_p):
        torch.testing.assert_close(p.detach().cpu(), rp, rtol=1e-6, atol=1e-9)
    for g, g0, rg in zip(dg, g_before, ref_g):
        if scale_grads:
            torch.testing.assert_close(g.cpu(), rg, rtol=1e-6, atol=1e-12)
        else:
            assert torch.equal(g, g0)                      # untouched
    # second step on the same tensors reuses the chunk table; a parameter without a gradient is skipped
    params[2].grad = None
    kept = params[2].detach().clone()
    opt.step()
    assert torch.equal(params[2].detach(), kept)


def test_clip_sgd_refuses_cpu_tensors():
    from voxelnet_amd import _lib
    from voxelnet_amd.optim import ClipSGD
    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = torch.ones(4)
    with pytest.raises(_lib.VoxelnetHipError):
        ClipSGD([p], 0.01, 5.0).step()


def test_train_step_with_fused_tail_matches_torch_tail():
    """Whole detector: one train step with ClipSGD == the same step with torch's clip_grad_norm_ + SGD."""
    from voxelnet_amd import model as M
    from voxelnet_amd import synth
    from voxelnet_amd.config import grid_config
    from voxelnet_amd.optim import ClipSGD
    from voxelnet_amd.voxelize import voxelize_device
    M.set_precision("bf16")
    grid = grid_config("Car")
    frames = synth.workload_frames(1, batch=1, frame0=0)
    feats, coords = [], []
    for b, f in enumerate(frames):
        fb, cb, _ = voxelize_device(torch.from_numpy(f).to(DEV), grid, b, coord_cols=4)
        feats.append(fb)
        coords.append(cb)
    results = []
    for fused in (False, True):
        torch.manual_seed(5)
        model = M.RPN3D("Car").to(DEV).train(True)
        params = list(model.parameters())
        h, w = model.rpn_output_shape
        g = torch.Generator().manual_seed(3)
        pos = (torch.rand((1, h, w, 2), generator=g) < 0.02).float().to(DEV)
        neg = (1 - pos) * (torch.rand((1, h, w, 2), generator=g) < 0.9).float().to(DEV)
        tgt = (torch.randn((1, h, w, 14), generator=g) * 0.3).to(DEV)
        out = model((None, None, feats, None, coords, None, None), DEV, targets=(pos, neg, tgt))
        out[2].backward()
        if fused:
            norm = ClipSGD(params, 0.01, 5.0).step()
        else:
            norm = torch.nn.utils.clip_grad_norm_(params, 5.0)
            torch.optim.SGD(params, lr=0.01).step()
        torch.cuda.synchronize()
        results.append((norm.item(), [p.detach().clone() for p in params]))
    (n0, p0), (n1, p1) = results
    assert abs(n0 - n1) <= 1e-5 * n0
    for a, b in zip(p0, p1):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-8)
