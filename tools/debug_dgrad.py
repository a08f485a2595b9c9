import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "voxelnet-pytorch_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import torch.nn.functional as F
from voxelnet_amd import engine as E
from oracle import torch_ref as tr
dev = "cuda:0"
cin, cout, k, s, p, sp = 128, 64, 3, (2, 1, 1), (1, 1, 1), (10, 8, 12)
w = tr._fill((cout, cin, k, k, k), 200, 1.0 / np.sqrt(cin * 27))
spec = E.LayerSpec("t", 3, cin, cout, (k, k, k), s, p)
odims = spec.out_dims(sp)
rng = np.random.default_rng(3)
dy = torch.from_numpy(rng.standard_normal((2, cout) + odims).astype(np.float32))
x = torch.zeros((2, cin) + sp, dtype=torch.float64, requires_grad=True)
y = F.conv3d(x, w.double(), None, s, p)
y.backward(dy.double())
ref = x.grad
for mode in ["fp32", "bf16x3", "bf16"]:
    dyr = E.nchw_to_rows(dy.to(dev), mode)
    wp = E.pack_weight(w.to(dev), spec, 1, mode)
    dx = E.Rows(torch.full((2,) + sp + (cin,), 7.0, dtype=E.plain_dtype_of(mode), device=dev), cin)
    E.gather_gemm(dyr, wp, None, dx, spec.k, cout, cin, (1, 1, 1), (-1, -1, -1), tuple(-q for q in p), s, sp)
    got = E.rows_to_nchw(dx, 3).cpu().double()
    d = (got - ref).abs()
    print(mode, "dgrad err", (d.max() / ref.abs().max()).item())
    bad = (d > 1e-3 * ref.abs().max()).nonzero()
    print("  bad", bad.shape[0], "of", d.numel(), bad[:6].tolist(), bad[-3:].tolist())
    if bad.shape[0]:
        print("   channels", sorted(set(bad[:, 1].tolist()))[:20], "d", sorted(set(bad[:, 2].tolist())))
