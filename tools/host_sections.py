"""Host time per section of one train step, un-throttled (10 steps enqueued without a device sync)."""
import os, sys, time
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
fc = [voxelize_device(p, grid, b, coord_cols=4) for b, p in enumerate(frames)]
feats, coords = [x[0] for x in fc], [x[1] for x in fc]
acc = {}
def tick(name, t0):
    t = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + t - t0
    return t
def step(record):
    t = time.perf_counter()
    prob, delta = model.detect(feats, coords)
    if record: t = tick("detect (cat + VFE + executor forward)", t)
    out = model.loss(prob, delta, *targets)
    if record: t = tick("loss forward", t)
    out[0].backward()
    if record: t = tick("backward (loss, executor, VFE)", t)
    opt.step(); opt.zero_grad(set_to_none=True)
    if record: t = tick("ClipSGD + zero_grad", t)
for _ in range(5): step(False)
torch.cuda.synchronize()
N = 10
t0 = time.perf_counter()
for _ in range(N): step(True)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
for k, v in acc.items():
    print(f"{k:45s} {1e3 * v / N:6.2f} ms/step")
print(f"enqueue {1e3*(t1-t0)/N:.2f} ms/step, total {1e3*(t2-t0)/N:.2f} ms/step")
