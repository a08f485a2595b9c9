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
    dense = len(sys.argv) > 1 and sys.argv[1] == "dense"      # BASELINE configs[4] (T = 64, batch 4) instead of configs[1]
    grid = grid_config("Car", T=64) if dense else grid_config("Car")
    feat = torch.cat([voxelize_device(torch.from_numpy(f).to(dev), grid, b, coord_cols=4)[0]
                      for b, f in enumerate(synth.workload_frames(5 if dense else 2))])
    K, T = feat.shape[0], feat.shape[1]
print("K", K, "T", T)
# effective rows per voxel (csrc/vfe.hip: 1 + last slot that differs bit-wise from slot T-1) and the wave items they make
bits = feat.view(torch.int32)
diff = (bits != bits[:, T - 1:T, :]).any(-1)                   # (K, T)
r = 1 + torch.where(diff.any(1), (diff.int() * torch.arange(1, T + 1, device=dev)).max(1).values, torch.zeros(K, dtype=torch.int64, device=dev))
nA, nB, nC, nD = int((r <= 8).sum()), int(((r > 8) & (r <= 16)).sum()), int(((r > 16) & (r <= 32)).sum()), int((r > 32).sum())
print(f"rows: mean {float(r.float().mean()):.2f}  classes r<=8 {nA}  r<=16 {nB}  r<=32 {nC}  r<=64 {nD}  -> wave items {(nA + 7) // 8} + {(nB + 3) // 4}"
      f" + {(nC + 1) // 2} + {nD} = {(nA + 7) // 8 + (nB + 3) // 4 + (nC + 1) // 2 + nD}; effective rows {int(r.sum())} of {K * T}")
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
