"""GPU: RPN3D.train_step — the whole train step (model.py:298-362 + train.py:151-154) as ONE library call (vn_net_step,
csrc/runtime.hip) — against the same step issued as separate calls (forward, loss.backward(), ClipSGD.step()): the same
kernels in the same order on the same streams, so everything must be BIT-identical: outputs, losses, every parameter
after the update, running statistics, counters."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _inputs(B=2):
    from voxelnet_amd import synth
    from voxelnet_amd.config import grid_config
    from voxelnet_amd.voxelize import voxelize_device
    grid = grid_config("Car")
    batches = []
    for step in range(3):
        frames = synth.workload_frames(2, batch=B, frame0=step * B)
        feats, coords = [], []
        for b, f in enumerate(frames):
            fb, cb, _ = voxelize_device(torch.from_numpy(f).to(DEV), grid, b, coord_cols=4)
            feats.append(fb)
            coords.append(cb)
        labels = np.empty(B, dtype=object)
        for b in range(B):
            labels[b] = synth.synth_labels("Car", 6, seed=8100 + step * B + b)
        batches.append((None, labels, feats, None, coords, None, None))
    return batches


def _model(mode, reducer):
    from voxelnet_amd import model as M
    from voxelnet_amd import parallel
    from voxelnet_amd.config import GRADIENT_CLIP, LR
    from voxelnet_amd.optim import ClipSGD
    M.set_precision(mode)
    torch.manual_seed(4321)
    m = M.RPN3D("Car").to(DEV).train()
    if reducer:
        m.grad_reducer = parallel.GradAllReducer(list(m.named_parameters()))
    return m, ClipSGD(list(m.parameters()), LR, GRADIENT_CLIP)


def _run(mode, reducer, fused, batches, given_targets=None):
    m, opt = _model(mode, reducer)
    outs = []
    for x in batches:
        if fused:
            assert m._step_fused_ok(mode, opt)
            out = m.train_step(x, DEV, opt, targets=given_targets)
        else:
            out = m(x, DEV, targets=given_targets)
            out[2].backward()
            if reducer:
                m.grad_reducer.finish(list(m.named_parameters()))
            opt.step()
        outs.append([o.detach().clone() for o in out])
        grads = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
        opt.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    state = {k: v.detach().clone() for k, v in m.state_dict().items()}
    return outs, grads, state, float(opt._norm[0])


@pytest.mark.parametrize("mode,reducer", [("bf16", False), ("fp32", False), ("fp32x3", False), ("bf16", True)])
def test_one_call_step_is_bit_identical_to_the_separate_calls(mode, reducer):
    from voxelnet_amd import model as M
    batches = _inputs()
    try:
        a = _run(mode, reducer, False, batches)
        b = _run(mode, reducer, True, batches)
    finally:
        M.set_precision("bf16")
    for step, (oa, ob) in enumerate(zip(a[0], b[0])):
        for i, (ta, tb) in enumerate(zip(oa, ob)):
            assert torch.equal(ta, tb), (mode, "output", i, "of step", step)
    assert a[3] == b[3] and np.isfinite(a[3]) and a[3] > 0
    assert a[1].keys() == b[1].keys() and all(torch.equal(a[1][k], b[1][k]) for k in a[1]), "last step's gradients"
    assert a[2].keys() == b[2].keys()
    for k in a[2]:
        assert torch.equal(a[2][k], b[2][k]), (mode, k)
    assert int(a[2]["feature_net.vfe_1.bn.num_batches_tracked"]) == 3
    assert float(a[0][0][2]) != float(a[0][2][2])        # (the three steps saw different batches and moving weights)


def test_one_call_step_with_given_targets_and_without_optimizer():
    """targets=(pos, neg, targets) handed in (the reference's generate_targets arrays) and optimizer=None: the step leaves
    the gradients on the parameters and does not move them"""
    from voxelnet_amd import model as M
    batches = _inputs()[:1]
    h, w = 200, 176
    rng = np.random.default_rng(77)
    pos = (rng.random((2, h, w, 2)) < 0.002).astype(np.float32)
    neg = ((rng.random((2, h, w, 2)) < 0.98) & (pos == 0)).astype(np.float32)
    tgt = (rng.standard_normal((2, h, w, 14)) * 0.1).astype(np.float32)
    given = (pos, neg, tgt)
    m, _ = _model("bf16", False)
    before = {k: v.detach().clone() for k, v in m.named_parameters()}
    out = m.train_step(batches[0], DEV, None, targets=given)
    torch.cuda.synchronize()
    assert all(torch.equal(before[k], p.detach()) for k, p in m.named_parameters())
    g1 = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    m2, _ = _model("bf16", False)
    out2 = m2(batches[0], DEV, targets=given)
    out2[2].backward()
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(out, out2))
    assert all(torch.equal(g1[k], p.grad) for k, p in m2.named_parameters())


def test_what_the_one_call_step_does_not_cover_takes_the_separate_calls():
    """eval mode, the per-layer orchestration and existing .grad tensors (accumulation) are not the fused call's: train_step
    then runs forward / backward / step itself — same results as calling them by hand"""
    from voxelnet_amd import model as M
    batches = _inputs()[:1]
    m, opt = _model("bf16", False)
    m.native_executor = False
    assert not m._step_fused_ok("bf16", opt)
    M.set_precision("fp32")
    try:
        m, opt = _model("fp32", False)
        m.native_executor = False
        out = m.train_step(batches[0], DEV, opt)
        torch.cuda.synchronize()
        assert torch.isfinite(out[2]).item() and all(p.grad is not None for p in m.parameters())
        m2, opt2 = _model("fp32", False)
        m2.native_executor = False
        out2 = m2(batches[0], DEV)
        out2[2].backward()
        opt2.step()
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(out, out2))
        assert all(torch.equal(p, q) for p, q in zip(m.parameters(), m2.parameters()))
    finally:
        M.set_precision("bf16")


def test_one_call_step_on_another_grid_with_ragged_batches():
    """a 10 x 16 x 24 grid (8 x 12 map: not the class default the scratch buffers could be mis-sized for), batch 3 with an
    EMPTY frame, K changing from step to step (the scratch grows), fp32 mode: bit-identical to the separate calls"""
    from voxelnet_amd import model as M
    from voxelnet_amd import synth
    from voxelnet_amd.config import GRADIENT_CLIP, LR, grid_config
    from voxelnet_amd.optim import ClipSGD
    from voxelnet_amd.voxelize import voxelize_device
    g = grid_config("Car", D=10, H=16, W=24, oy=1.6)
    rng = np.random.default_rng(11)
    batches = []
    for step, k0 in enumerate((60, 400, 150)):
        feats, coords = [], []
        for i in range(3):
            cloud = synth.synth_cloud("Car", k0=k0 + 20 * i, seed=2100 + 10 * step + i, grid=g, overflow_frac=0.03)
            if i == step % 3:
                cloud = cloud.copy()
                cloud[:, 0] = -500.0
            fb, cb, _ = voxelize_device(torch.from_numpy(cloud).to(DEV), g, i, coord_cols=4)
            feats.append(fb)
            coords.append(cb)
        pos = (rng.random((3, 8, 12, 2)) < 0.05).astype(np.float32)
        neg = ((rng.random((3, 8, 12, 2)) < 0.9) & (pos == 0)).astype(np.float32)
        tgt = (rng.standard_normal((3, 8, 12, 14)) * 0.1).astype(np.float32)
        batches.append(((None, None, feats, None, coords, None, None), (pos, neg, tgt)))
    assert len({sum(f.shape[0] for f in b[0][2]) for b in batches}) == 3
    res = []
    try:
        M.set_precision("fp32")
        for fused in (False, True):
            torch.manual_seed(99)
            m = M.RPN3D("Car")
            m.feature_net._grid = g
            m = m.to(DEV).train()
            opt = ClipSGD(list(m.parameters()), LR, GRADIENT_CLIP)
            outs = []
            for x, t in batches:
                if fused:
                    assert m._step_fused_ok("fp32", opt)
                    out = m.train_step(x, DEV, opt, targets=t)
                else:
                    out = m(x, DEV, targets=t)
                    out[2].backward()
                    opt.step()
                outs.append([o.detach().clone() for o in out])
                opt.zero_grad(set_to_none=True)
            torch.cuda.synchronize()
            res.append((outs, {k: v.detach().clone() for k, v in m.state_dict().items()}))
    finally:
        M.set_precision("bf16")
    (oa, sa), (ob, sb) = res
    assert oa[0][0].shape == (3, 2, 8, 12)
    for step in range(3):
        assert all(torch.equal(a, b) for a, b in zip(oa[step], ob[step])), step
    assert all(torch.equal(sa[k], sb[k]) for k in sa)
