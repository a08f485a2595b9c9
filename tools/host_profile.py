"""cProfile of the host side of one train step (enqueue only: static voxel buffers, no device sync inside the steps).
usage: python tools/host_profile.py [nsteps]"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "voxelnet-pytorch_amd")]
import torch
from voxelnet_amd import model as M, synth
from voxelnet_amd.config import grid_config
from voxelnet_amd.optim import ClipSGD
from voxelnet_amd.voxelize import voxelize_device
import bench
dev = torch.device("cuda:0")
M.set_precision("bf16")
torch.manual_seed(0)
model = M.RPN3D("Car").to(dev).train()
params = list(model.parameters())
opt = ClipSGD(params, 0.01, 5.0)
grid = grid_config("Car")
frames = [torch.from_numpy(f).to(dev) for f in synth.workload_frames(2, batch=2)]
targets = bench.synthetic_targets(2, 200, 176, 99, dev)
import numpy as np
labels = np.empty(2, dtype=object)
for b in range(2):
    labels[b] = synth.synth_labels("Car", 6, seed=7000 + b)
USE_LABELS = os.environ.get("HOST_PROFILE_LABELS", "1") == "1"     # targets generated from the label lines inside the step (bench.py's default)
fc = [voxelize_device(p, grid, b, coord_cols=4) for b, p in enumerate(frames)]
feats, coords = [x[0] for x in fc], [x[1] for x in fc]


def step():
    if USE_LABELS:
        out = model((None, labels, feats, None, coords, None, None), dev)
    else:
        out = model((None, None, feats, None, coords, None, None), dev, targets=targets)
    out[2].backward()
    opt.step()
    opt.zero_grad(set_to_none=True)


n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"enqueue {1e3 * (t1 - t0) / n:.3f} ms/step over {n} steps (un-profiled; paced by the GPU once its queue is full)")
# the host's own cost: a few steps into an EMPTY queue (nothing to wait for), repeated
best = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step(); step()
    best.append((time.perf_counter() - t0) / 2)
    torch.cuda.synchronize()
best.sort()
print(f"enqueue into an empty queue: median {1e3 * best[len(best) // 2]:.3f} ms/step, min {1e3 * best[0]:.3f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(30)
