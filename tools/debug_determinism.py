"""Repeat the same train step (same seed, inputs, fresh model) and compare loss / every gradient tensor bit-wise with
the first repetition; alternates the torch and the fused optimizer tail like tests/test_gpu_optim.py."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "voxelnet-pytorch_amd")]
import torch
from voxelnet_amd import model as M, synth
from voxelnet_amd.config import grid_config
from voxelnet_amd.optim import ClipSGD
from voxelnet_amd.voxelize import voxelize_device
DEV = "cuda:0"
M.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16")
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 40
grid = grid_config("Car")
frames = synth.workload_frames(1, batch=B, frame0=0)
feats, coords = [], []
for b, f in enumerate(frames):
    fb, cb, _ = voxelize_device(torch.from_numpy(f).to(DEV), grid, b, coord_cols=4)
    feats.append(fb); coords.append(cb)
ref = None
bad_reps = 0
for rep in range(REPS):
    torch.manual_seed(5)
    model = M.RPN3D("Car").to(DEV).train(True)
    for kv in os.environ.get("VN_DBG_SET", "").split(","):
        if kv:
            k_, v_ = kv.split("=")
            setattr(model, k_, bool(int(v_)))
    params = list(model.parameters())
    names = [n for n, _ in model.named_parameters()]
    if os.environ.get("VN_DBG_REDUCER") == "1":          # the DDP path (bucket events, comm stream), world size 1
        from voxelnet_amd import parallel
        model.grad_reducer = parallel.GradAllReducer(list(model.named_parameters()))
    h, w = model.rpn_output_shape
    g = torch.Generator().manual_seed(3)
    pos = (torch.rand((B, h, w, 2), generator=g) < 0.02).float().to(DEV)
    neg = (1 - pos) * (torch.rand((B, h, w, 2), generator=g) < 0.9).float().to(DEV)
    tgt = (torch.randn((B, h, w, 14), generator=g) * 0.3).to(DEV)
    fcat = torch.cat(feats)
    vparams = [p.detach() for p in M._vfe_weights(model.feature_net)]
    vw_probe, st_probe, _ = M.featnet_forward(fcat, vparams, [b.clone() for b in model.feature_net._bufs()], True)
    vw_probe, st_probe = vw_probe.clone(), st_probe.clone()
    out = model((None, None, feats, None, coords, None, None), DEV, targets=(pos, neg, tgt))
    out[2].backward()
    if model.grad_reducer is not None:
        model.grad_reducer.finish(list(model.named_parameters()))
    early = torch.nn.utils.clip_grad_norm_(params, 1e9)          # stream order, no sync (as the test does)
    grads = [p.grad.detach().clone() for p in params]
    prob = out[0].detach().clone()
    if rep % 2:
        ClipSGD(params, 0.01, 5.0).step()
    else:
        torch.optim.SGD(params, lr=0.01).step()
    torch.cuda.synchronize()
    bufs_now = {n: b.detach().clone() for n, b in model.named_buffers() if n.endswith("running_mean") or n.endswith("running_var")}
    cur = (out[2].item(), early.item(), prob, grads, vw_probe, st_probe, bufs_now)
    if rep == 0:
        first = cur
        continue
    if rep == 1:
        print("rep 1 vs rep 0: prob max |diff|", (prob - first[2]).abs().max().item(), "loss", first[0], cur[0], "norm", first[1], cur[1])
    if ref is None:
        ref_reg = out[1].detach().clone()
        ref = cur
        print(f"rep 0: loss {cur[0]:.9f} norm {cur[1]:.6f}")
        continue
    diff = [n for n, a, b in zip(names, ref[3], grads) if not torch.equal(a, b)]
    if not torch.equal(prob, ref[2]) or diff:
        bad_reps += 1
        if torch.equal(prob, ref[2]):
            print(f"rep {rep}: forward identical, gradients differ: {len(diff)}: {diff[:10]}")
            continue
        dl = [n for n in bufs_now if not torch.equal(bufs_now[n], ref[6][n])]
        print("   BatchNorm running statistics that differ:", dl[:12], "..." if len(dl) > 12 else "")
        dm = (prob - ref[2]).abs().amax(1)[0]          # (h, w) of sample 0
        ys, xs = dm.nonzero(as_tuple=True)
        print(f"   prob differs at {ys.numel()} of {dm.numel()} pixels; rows {ys.min().item()}..{ys.max().item()} cols "
              f"{xs.min().item()}..{xs.max().item()}; max {dm.max().item():.4f}; reg equal {torch.equal(out[1].detach(), ref_reg)}")
        print(f"rep {rep}: VFE out equal {torch.equal(vw_probe, ref[4])} VFE stats equal {torch.equal(st_probe, ref[5])} "
              f"loss {cur[0]:.9f} norm {cur[1]:.6f} prob equal {torch.equal(prob, ref[2])} "
              f"differing grads {len(diff)}: {diff[:8]}")
print("repetitions that differ from the first:", bad_reps, "of", REPS - 1)
