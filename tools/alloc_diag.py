"""Does the steady-state step call hipMalloc/hipFree?  Prints caching-allocator counters around 20 steps."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "voxelnet-pytorch_amd")]
import torch
from voxelnet_amd import model as M, synth
from voxelnet_amd.config import grid_config
from voxelnet_amd.voxelize import voxelize_device
import bench
dev = torch.device("cuda:0")
M.set_precision("bf16")
model = M.RPN3D("Car").to(dev).train()
params = list(model.parameters())
opt = torch.optim.SGD(params, lr=0.01)
grid = grid_config("Car")
frames = [torch.from_numpy(f).to(dev) for f in synth.workload_frames(2, batch=2)]
targets = bench.synthetic_targets(2, 200, 176, 99, dev)
fc = [voxelize_device(p, grid, b, coord_cols=4) for b, p in enumerate(frames)]
feats, coords = [x[0] for x in fc], [x[1] for x in fc]
def step():
    out = model((None, None, feats, None, coords, None, None), dev, targets=targets)
    out[2].backward()
    torch.nn.utils.clip_grad_norm_(params, 5.0)
    opt.step(); opt.zero_grad(set_to_none=True)
for _ in range(5): step()
torch.cuda.synchronize()
keys = ["num_device_alloc", "num_device_free", "num_alloc_retries", "reserved_bytes.all.current", "allocated_bytes.all.peak"]
s0 = torch.cuda.memory_stats()
t0 = time.perf_counter()
for _ in range(20): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
s1 = torch.cuda.memory_stats()
for k in keys: print(k, s0.get(k), "->", s1.get(k))
print(f"enqueue {1e3*(t1-t0)/20:.2f} ms/step total {1e3*(t2-t0)/20:.2f} ms/step")
import torch.utils.benchmark
t0 = time.perf_counter()
for _ in range(20):
    x = torch.empty((2, 10, 400, 352, 128), dtype=torch.bfloat16, device=dev)
t1 = time.perf_counter()
print(f"torch.empty(721MB) alone: {1e3*(t1-t0)/20:.3f} ms")
torch.cuda.synchronize()
ts = []
for _ in range(12):
    t0 = time.perf_counter(); step(); ts.append(1e3 * (time.perf_counter() - t0))
torch.cuda.synchronize()
print("per-step enqueue ms after a sync:", " ".join(f"{t:.2f}" for t in ts))
# sections of one step, host time only (queue drained before each)
def sect():
    torch.cuda.synchronize(); t = [time.perf_counter()]
    out = model((None, None, feats, None, coords, None, None), dev, targets=targets); t.append(time.perf_counter())
    out[2].backward(); t.append(time.perf_counter())
    torch.nn.utils.clip_grad_norm_(params, 5.0); t.append(time.perf_counter())
    opt.step(); opt.zero_grad(set_to_none=True); t.append(time.perf_counter())
    return [1e3 * (b - a) for a, b in zip(t, t[1:])]
for _ in range(3): r = sect()
print("host ms: forward+loss %.2f backward %.2f clip %.2f sgd %.2f" % tuple(r))
