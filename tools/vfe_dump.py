"""Dump the VFE forward output and parameter gradients of the car workload (seeded d_voxelwise) to an .npz — run once per
library build (VN_LIB_PATH) and compare: python tools/vfe_dump.py out.npz ; python tools/vfe_dump.py a.npz b.npz (compare)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "voxelnet-pytorch_amd")]
import numpy as np  # noqa: E402

if len(sys.argv) == 3:
    a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
    for k in a.files:
        x, y = a[k].astype(np.float64), b[k].astype(np.float64)
        print(f"{k:12s} rel-L2 {np.linalg.norm(x - y) / (np.linalg.norm(y) + 1e-300):.3e}  max|a-b| {np.abs(x - y).max():.3e}  |b| {np.linalg.norm(y):.3e}")
    sys.exit(0)

import torch  # noqa: E402

from voxelnet_amd import model as M, synth  # noqa: E402
from voxelnet_amd.config import grid_config  # noqa: E402
from voxelnet_amd.voxelize import voxelize_device  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)
grid = grid_config("Car")
feat = torch.cat([voxelize_device(torch.from_numpy(f).to(dev), grid, b, coord_cols=4)[0] for b, f in enumerate(synth.workload_frames(2))])
m = M.RPN3D("Car").to(dev).train()
params = [p.detach() for p in M._vfe_weights(m.feature_net)]
vw, stats, wst = M.featnet_forward(feat, params, m.feature_net._bufs(), True)
dvw = torch.randn_like(vw)
grads = M.featnet_backward(feat, wst, stats, dvw, params)
torch.cuda.synchronize()
out = {"voxelwise": vw.cpu().numpy(), "stats": stats.cpu().numpy()}
for i, g in enumerate(grads):
    out[f"grad{i}"] = g.cpu().numpy()
np.savez(sys.argv[1], **out)
print("saved", sys.argv[1], {k: v.shape for k, v in out.items()})
