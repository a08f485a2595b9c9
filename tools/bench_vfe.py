import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "voxelnet-pytorch_amd")]
import torch
from voxelnet_amd import model as M
dev = "cuda:0"
from voxelnet_amd import synth
from voxelnet_amd.config import grid_config
from voxelnet_amd.voxelize import voxelize_device
torch.manual_seed(0)
if len(sys.argv) > 1 and sys.argv[1] == "random":
    K, T = 12000, 35
    feat = torch.randn(K, T, 7, device=dev)          # every slot differs: the r = T worst case
else:
    grid = grid_config("Car")
    feat = torch.cat([voxelize_device(torch.from_numpy(f).to(dev), grid, b, coord_cols=4)[0]
                      for b, f in enumerate(synth.workload_frames(2))])
    K, T = feat.shape[0], feat.shape[1]
print("K", K, "T", T)
m = M.RPN3D("Car").to(dev).train()
params = [p.detach() for p in M._vfe_weights(m.feature_net)]
bufs = m.feature_net._bufs()
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
vw, stats, wst = M.featnet_forward(feat, params, bufs, True)
dvw = torch.randn_like(vw)
print("vfe fwd ms", timeit(lambda: M.featnet_forward(feat, params, bufs, True)))
print("vfe bwd ms", timeit(lambda: M.featnet_backward(feat, wst, stats, dvw, params)))
