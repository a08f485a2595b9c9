"""How long does the host need to ENQUEUE one train step (no device sync inside)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "voxelnet-pytorch_amd")]
import numpy as np, torch
from voxelnet_amd import model as M, synth
from voxelnet_amd.config import grid_config
from voxelnet_amd.voxelize import voxelize_device
import bench
dev = torch.device("cuda:0")
M.set_precision("bf16")
torch.manual_seed(0)
model = M.RPN3D("Car").to(dev).train()
params = list(model.parameters())
from voxelnet_amd.optim import ClipSGD
opt = ClipSGD(params, 0.01, 5.0)
grid = grid_config("Car")
frames = [torch.from_numpy(f).to(dev) for f in synth.workload_frames(2, batch=2)]
targets = bench.synthetic_targets(2, 200, 176, 99, dev)
fc = [voxelize_device(p, grid, b, coord_cols=4) for b, p in enumerate(frames)]
feats, coords = [x[0] for x in fc], [x[1] for x in fc]
def step(vox):
    global feats, coords
    if vox:
        fc = [voxelize_device(p, grid, b, coord_cols=4) for b, p in enumerate(frames)]
        feats, coords = [x[0] for x in fc], [x[1] for x in fc]
    out = model((None, None, feats, None, coords, None, None), dev, targets=targets)
    out[2].backward()
    opt.step(); opt.zero_grad(set_to_none=True)
for vox in (False, True):
    for _ in range(3): step(vox)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): step(vox)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"voxelize_in_step={vox}: enqueue {1e3*(t1-t0)/10:.2f} ms/step, total {1e3*(t2-t0)/10:.2f} ms/step")
