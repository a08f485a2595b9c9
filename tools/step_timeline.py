"""Two-stream timeline of ONE native-executor train step from the executor's own HIP events (vn_net_timing_begin/_read:
every launch bracketed on its stream; start offsets relative to the first launch): which launches of the main chain
run beside which weight gradients, where a stream sits idle, what the tail looks like.
usage: python tools/step_timeline.py [--csv out.csv] [--dense]   (--dense: BASELINE configs[4], batch 4, T = 64)"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "voxelnet-pytorch_amd")]
import bench  # noqa: E402
from voxelnet_amd import _lib, synth  # noqa: E402
from voxelnet_amd import model as M  # noqa: E402
from voxelnet_amd.config import grid_config  # noqa: E402
from voxelnet_amd.optim import ClipSGD  # noqa: E402
from voxelnet_amd.voxelize import voxelize_device  # noqa: E402

dev = torch.device("cuda:0")
M.set_precision("bf16")
torch.manual_seed(0)
model = M.RPN3D("Car").to(dev).train()
opt = ClipSGD(list(model.parameters()), 0.01, 5.0)
DENSE = "--dense" in sys.argv
NB = 4 if DENSE else 2
grid = grid_config("Car", T=64) if DENSE else grid_config("Car")
frames = [torch.from_numpy(f).to(dev) for f in synth.workload_frames(5 if DENSE else 2, batch=NB)]
targets = bench.synthetic_targets(NB, 200, 176, 99, dev)
fc = [voxelize_device(p, grid, b, coord_cols=4) for b, p in enumerate(frames)]
feats, coords = [x[0] for x in fc], [x[1] for x in fc]


def step():
    out = model((None, None, feats, None, coords, None, None), dev, targets=targets)
    out[2].backward()
    opt.step()
    opt.zero_grad(set_to_none=True)


for _ in range(10):
    step()
torch.cuda.synchronize()
h = model._net_handle(dev)
buf = (_lib.VnTimingRecord * 4096)()
n = ctypes.c_int32(0)
_lib.call("vn_net_timing_begin", h, 4096)
step()
torch.cuda.synchronize()
_lib.call("vn_net_timing_read", h, buf, 4096, ctypes.byref(n))
recs = [(r.kind, r.layer, r.start_ms * 1e3, r.ms * 1e3) for r in buf[:n.value]]
names = bench.KIND_NAMES
side_layers_fwd = {8, 15}
rows = []
for i, (kind, layer, t0, dur) in enumerate(recs):
    # side stream: weight gradients, unpack, pack + first-layer preparation, deconv1/deconv2 (both directions)
    side = kind in (2, 7, 8) or (kind == 9 and dur > 0 and i < 6) or layer in side_layers_fwd
    rows.append((t0, t0 + dur, "side" if side else "main", names[kind], layer))
rows.sort()
print(f"{len(rows)} launches; step spans {max(r[1] for r in rows):.0f} us (timing mode: every launch bracketed by two events)")
last = {"main": 0.0, "side": 0.0}
for t0, t1, st, nm, layer in rows:
    gap = t0 - last[st]
    flag = f"   <-- {gap:.0f} us idle" if gap > 12 else ""
    pad = "" if st == "main" else " " * 46
    print(f"{pad}{t0:8.0f} {t1 - t0:6.0f}  {nm:14s} L{layer:<3d}{flag}")
    last[st] = t1
if "--csv" in sys.argv:
    with open(sys.argv[sys.argv.index("--csv") + 1], "w") as fh:
        for r in rows:
            fh.write(",".join(str(x) for x in r) + "\n")
