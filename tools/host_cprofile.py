"""cProfile of the host side of bench.py's step (un-throttled: the queue is drained every 10 steps): where the ~1.8 ms of
host time per step go that are not hipLaunchKernel itself.
usage: python tools/host_cprofile.py [steps] [separate]   (default: the one-call step, RPN3D.train_step; "separate": forward /
backward / optimizer.step as separate calls — there the backward runs on autograd's device thread, which cProfile does not see:
its whole time shows up as run_backward)"""
import cProfile
import io
import os
import pstats
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "voxelnet-pytorch_amd")]
from voxelnet_amd import model as M, synth  # noqa: E402
from voxelnet_amd.config import grid_config  # noqa: E402
from voxelnet_amd.optim import ClipSGD  # noqa: E402
from voxelnet_amd.voxelize import VoxelBatch, VoxelBuffers, pipeline_stream, voxelize_device_async  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
SEPARATE = "separate" in sys.argv[2:]
dev = torch.device("cuda:0")
M.set_precision("bf16")
torch.manual_seed(0)
model = M.RPN3D("Car").to(dev).train()
opt = ClipSGD(list(model.parameters()), 0.01, 5.0)
grid = grid_config("Car")
frames = [torch.from_numpy(f).to(dev) for f in synth.workload_frames(2, batch=2)]
labels = np.empty(2, dtype=object)
for b in range(2):
    labels[b] = synth.synth_labels("Car", 6, seed=7000 + b)
vs = pipeline_stream(dev)
slots = [[VoxelBuffers(p.shape[0], grid, 4, dev) for p in frames] for _ in range(3)]
state = {"i": 0}


def step():
    bufs = slots[state["i"] % 3]
    state["i"] += 1
    with torch.cuda.stream(vs):
        hs = [voxelize_device_async(p, grid, b, coord_cols=4, buffers=bufs[b]) for b, p in enumerate(frames)]
    fc = [h.result() for h in hs]
    torch.cuda.current_stream().wait_event(hs[-1].event)
    feats = VoxelBatch.ahead([x[0] for x in fc], vs, torch.float32)
    coords = VoxelBatch.ahead([x[1] for x in fc], vs, torch.int64)
    if SEPARATE:
        out = model((None, labels, feats, None, coords, None, None), dev)
        out[2].backward()
        opt.step()
    else:
        model.train_step((None, labels, feats, None, coords, None, None), dev, opt)
    opt.zero_grad(set_to_none=True)


for _ in range(10):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
for i in range(N):
    if i % 10 == 0:
        torch.cuda.synchronize()
    pr.enable()
    step()
    pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
st = pstats.Stats(pr, stream=s)
st.sort_stats("tottime").print_stats(45)
txt = s.getvalue()
print(f"(all times are totals over {N} steps: divide by {N} for per-step seconds)")
print(txt[:9000])
