"""Alone vs in-step duration of every launch of the native executor ("Next round" item 2).

Both columns come from the executor's own HIP events (vn_net_timing_begin/_read: every launch bracketed on ITS stream):
  alone    the step with overlap_wgrad = False — ONE stream, every launch runs by itself, in dependency order;
  in-step  the product schedule — main stream (the dependency chain) + side stream (weight gradients, deconv branches,
           packing, unpack) sharing the CUs.
The ratio is what a launch pays for its neighbours; the sum of the main-chain launches' ALONE times plus one launch
boundary each is the floor of this two-stream design (nothing on the chain can run before its producer).

usage: python tools/inflation.py [bf16|fp32x3|fp32] [steps]
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "voxelnet-pytorch_amd")]
import bench  # noqa: E402
from voxelnet_amd import _lib, net as N, synth  # noqa: E402
from voxelnet_amd import model as M  # noqa: E402
from voxelnet_amd.config import grid_config  # noqa: E402
from voxelnet_amd.optim import ClipSGD  # noqa: E402
from voxelnet_amd.voxelize import voxelize_device  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
M.set_precision(prec)
torch.manual_seed(0)
model = M.RPN3D("Car").to(dev).train()
opt = ClipSGD(list(model.parameters()), 0.01, 5.0)
grid = grid_config("Car")
frames = [torch.from_numpy(f).to(dev) for f in synth.workload_frames(2, batch=2)]
targets = bench.synthetic_targets(2, 200, 176, 99, dev)
fc = [voxelize_device(p, grid, b, coord_cols=4) for b, p in enumerate(frames)]
feats, coords = [x[0] for x in fc], [x[1] for x in fc]


def step():
    out = model((None, None, feats, None, coords, None, None), dev, targets=targets)
    out[2].backward()
    opt.step()
    opt.zero_grad(set_to_none=True)


def timed(overlap):
    model.overlap_wgrad = overlap
    for _ in range(6):
        step()
    torch.cuda.synchronize()
    # whole-step time without the timing events
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        step()
    e1.record()
    torch.cuda.synchronize()
    step_ms = e0.elapsed_time(e1) / 20
    h = model._net_handle(dev)
    buf = (_lib.VnTimingRecord * 4096)()
    n = ctypes.c_int32(0)
    acc, order = {}, []
    for _ in range(nsteps):
        _lib.call("vn_net_timing_begin", h, 4096)
        step()
        torch.cuda.synchronize()
        _lib.call("vn_net_timing_read", h, buf, 4096, ctypes.byref(n))
        seen = {}
        for r in buf[:n.value]:
            k = (r.kind, r.layer)
            i = seen.get(k, 0)
            seen[k] = i + 1
            key = (r.kind, r.layer, i)          # i-th launch of that (family, layer) in issue order
            if key not in acc:
                acc[key] = 0.0
                order.append(key)
            acc[key] += r.ms * 1e3 / nsteps
    return step_ms, acc, order


names = bench.KIND_NAMES
table = [n for n, _ in N.layer_table(2)] + ["heads"]
step_alone, alone, order_a = timed(False)
step_in, instep, order = timed(True)
SIDE_KINDS = {2, 7, 8}          # weight gradients, unpack, pack
SIDE_LAYERS = {8, 15}           # deconv1 / deconv2 run on the side stream in both directions
print(f"# {prec}, car batch 2: alone = one stream (overlap_wgrad = False), in-step = the two-stream product schedule; "
      f"mean of {nsteps} timed steps, HIP events on the launch's own stream")
print(f"# whole step (no timing events, 20 steps): one stream {step_alone:.3f} ms, two streams {step_in:.3f} ms")
print("| stream | family | layer | # | alone us | in-step us | ratio |")
print("|---|---|---|---|---|---|---|")
tot = {"main": [0.0, 0.0, 0], "side": [0.0, 0.0, 0]}
fam = {}
for key in order:
    kind, layer, i = key
    a, s = alone.get(key), instep[key]
    st = "side" if (kind in SIDE_KINDS or layer in SIDE_LAYERS or (kind == 9 and layer == 0 and i < 3) or (kind == 10 and layer == 23 and i >= 1)) else "main"
    lname = table[layer] if 0 <= layer < len(table) else "-"
    if a is None:
        print(f"| {st} | {names[kind]} | {lname} | {i} | - | {s:.1f} | - |")
        continue
    print(f"| {st} | {names[kind]} | {lname} | {i} | {a:.1f} | {s:.1f} | {s / a if a > 0 else 0:.2f} |")
    tot[st][0] += a
    tot[st][1] += s
    tot[st][2] += 1
    f = fam.setdefault((st, names[kind]), [0.0, 0.0, 0])
    f[0] += a
    f[1] += s
    f[2] += 1
print()
print("| stream | family | launches | alone ms | in-step ms | ratio |")
print("|---|---|---|---|---|---|")
for (st, nm), (a, s, n) in sorted(fam.items()):
    print(f"| {st} | {nm} | {n} | {a / 1e3:.3f} | {s / 1e3:.3f} | {s / a if a > 0 else 0:.2f} |")
for st in ("main", "side"):
    a, s, n = tot[st]
    print(f"| {st} | ALL | {n} | {a / 1e3:.3f} | {s / 1e3:.3f} | {s / a if a > 0 else 0:.2f} |")
a, s, n = tot["main"]
print()
print(f"main chain (executor launches only; VFE / loss / optimizer / voxelizer are outside the executor): {n} launches, "
      f"alone {a / 1e3:.3f} ms, in-step {s / 1e3:.3f} ms (+{(s - a) / 1e3:.3f} ms paid for the side stream's presence); "
      f"with 1.1-2.0 us per dependent launch boundary the chain's floor is {a / 1e3 + n * 1.1e-3:.3f}-{a / 1e3 + n * 2.0e-3:.3f} ms")
