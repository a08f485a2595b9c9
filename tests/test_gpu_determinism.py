"""Run-to-run bit reproducibility of the whole train step core (VFE -> executor forward -> loss -> executor backward on
two streams -> VFE backward): the same model and inputs, repeated; every output and every one of the 104 gradients must
be bit-identical to the first repetition.  This is the regression test for two defects it found (tools/
debug_determinism.py): a write-after-read race on an LDS stage buffer in the convolution kernels (a bare s_barrier
without lgkmcnt(0): one wrong 4x16 tile in ~10 % of the steps when a second kernel shared the CUs) and float atomics in
the heads' bias gradient."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("reducer", [False, True])
def test_train_step_is_bit_reproducible(reducer):
    """reducer: the data-parallel path (gradients written into the all-reduce buckets, per-bucket events, comm stream)
    with world size 1"""
    from voxelnet_amd import model as M
    from voxelnet_amd import synth
    from voxelnet_amd.config import grid_config
    from voxelnet_amd.voxelize import voxelize_device
    M.set_precision("bf16")
    grid = grid_config("Car")
    feats, coords = [], []
    for b, f in enumerate(synth.workload_frames(1, batch=2, frame0=0)):
        fb, cb, _ = voxelize_device(torch.from_numpy(f).to(DEV), grid, b, coord_cols=4)
        feats.append(fb)
        coords.append(cb)
    torch.manual_seed(5)
    model = M.RPN3D("Car").to(DEV).train(True)
    if reducer:
        from voxelnet_amd import parallel
        model.grad_reducer = parallel.GradAllReducer(list(model.named_parameters()))
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    h, w = model.rpn_output_shape
    g = torch.Generator().manual_seed(3)
    pos = (torch.rand((2, h, w, 2), generator=g) < 0.02).float().to(DEV)
    neg = (1 - pos) * (torch.rand((2, h, w, 2), generator=g) < 0.9).float().to(DEV)
    tgt = (torch.randn((2, h, w, 14), generator=g) * 0.3).to(DEV)
    names = [n for n, _ in model.named_parameters()]
    ref = None
    for rep in range(30):
        model.load_state_dict(sd0)                         # same weights and running statistics every time
        model.zero_grad(set_to_none=True)
        out = model((None, None, feats, None, coords, None, None), DEV, targets=(pos, neg, tgt))
        out[2].backward()
        if reducer:
            model.grad_reducer.finish(list(model.named_parameters()))
        cur = [out[0].detach().clone(), out[1].detach().clone(), out[2].detach().clone()] + \
              [p.grad.detach().clone() for p in model.parameters()]
        torch.cuda.synchronize()
        if ref is None:
            ref = cur
            continue
        bad = [n for n, a, b in zip(["prob", "delta", "loss"] + names, ref, cur) if not torch.equal(a, b)]
        assert not bad, f"repetition {rep} differs from the first in {len(bad)} tensors: {bad[:6]}"
