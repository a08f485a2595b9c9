"""gather-GEMM time vs number of workgroups (block1.1 shape: 3x3, 128->128, and block3.1: 256->256)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "voxelnet-pytorch_amd")]
import torch
from voxelnet_amd import engine as E, net as N
dev = "cuda:0"
specs = dict(N.layer_table(2))
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for name in sys.argv[1:] or ["block1.1", "block3.1"]:
    sp = specs[name]
    for nb in (32, 64, 128, 256, 384, 512, 550, 768, 1024, 2048):
        dims = (1, nb, 128)
        x = E.Rows(torch.randn((1,) + dims + (sp.cin,), device=dev).to(torch.bfloat16), sp.cin)
        w = torch.randn((sp.cout, sp.cin) + sp.k[3 - sp.dim:], device=dev) * 0.05
        bias = torch.zeros(sp.cout, device=dev)
        y = E.Rows(torch.empty((1,) + dims + (sp.cout,), dtype=torch.bfloat16, device=dev), sp.cout)
        wp = E.pack_weight(w, sp, 0, "bf16")
        a = (sp.stride, (1, 1, 1), sp.pad, (1, 1, 1))
        t = timeit(lambda: E.gather_gemm(x, wp, bias, y, sp.k, sp.cin, sp.cout, *a, dims))
        fl = 2.0 * y.M * sp.cout * sp.cin * sp.taps
        print(f"{name} rows {y.M:7d} (128-row tiles {nb:5d}) {t*1e3:7.1f} us {fl/t/1e9:6.0f} TF/s")
