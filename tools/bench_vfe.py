import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "voxelnet-pytorch_amd")]
import torch
from voxelnet_amd import model as M
dev = "cuda:0"
K, T = 12000, 35
torch.manual_seed(0)
feat = torch.randn(K, T, 7, device=dev)
feat[:, 20:, :4] = 0
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
