"""Digest of one fp32x3 train step's RPN maps and gradients (car config, batch 2, synthetic frames): run it under
VN_X3_PRESPLIT=0 and =1 — weights split in registers / once by the pack launch — and compare the lines (they must be equal:
the operands are the same hi / lo values either way).  usage: python tools/x3_digest.py [mode]"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "voxelnet-pytorch_amd")]
import numpy as np
import torch
from voxelnet_amd import model as M, synth
from voxelnet_amd.config import grid_config
from voxelnet_amd.voxelize import voxelize_device

mode = sys.argv[1] if len(sys.argv) > 1 else "fp32x3"
dev = torch.device("cuda:0")
M.set_precision(mode)
torch.manual_seed(0)
model = M.RPN3D("Car").to(dev).train()
grid = grid_config("Car")
frames = [torch.from_numpy(f).to(dev) for f in synth.workload_frames(2, batch=2)]
labels = np.empty(2, dtype=object)
for b in range(2):
    labels[b] = synth.synth_labels("Car", 6, seed=7000 + b)
fc = [voxelize_device(p, grid, b, coord_cols=4) for b, p in enumerate(frames)]
out = model((None, labels, [x[0] for x in fc], None, [x[1] for x in fc], None, None), dev)
out[2].backward()
torch.cuda.synchronize()
h = hashlib.sha256()
for t in (out[0], out[1]):
    h.update(t.detach().float().cpu().numpy().tobytes())
g = hashlib.sha256()
for p in model.parameters():
    g.update(p.grad.detach().float().cpu().numpy().tobytes())
print(f"{mode} VN_X3_PRESPLIT={os.environ.get('VN_X3_PRESPLIT', '1')} loss {float(out[2]):.9g} maps {h.hexdigest()[:16]} grads {g.hexdigest()[:16]}")
