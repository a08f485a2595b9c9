"""Per-step cycle trace of one gather-GEMM workgroup (instrumented copy of conv.hip built by hand).
usage: python tools/trace_conv.py <lib.so> <layer>"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "voxelnet-pytorch_amd")]
import numpy as np, torch
from voxelnet_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from voxelnet_amd import engine as E, net as N
dev = "cuda:0"
name = sys.argv[2]
specs = dict(N.layer_table(2))
IN = {"middle_layer.1": (5, 400, 352), "middle_layer.2": (3, 400, 352), "block1.1": (1, 200, 176), "block2.1": (1, 100, 88),
      "block3.1": (1, 50, 44), "deconv1": (1, 200, 176)}
sp, dims, B = specs[name], IN[name], 2
x = E.Rows(torch.randn((B,) + dims + (sp.cin,), device=dev).to(torch.bfloat16), sp.cin)
w = torch.randn((sp.cin, sp.cout) + sp.k[3 - sp.dim:] if sp.transposed else (sp.cout, sp.cin) + sp.k[3 - sp.dim:], device=dev) * 0.05
bias = torch.zeros(sp.cout, device=dev)
od = sp.out_dims(dims)
y = E.Rows(torch.empty((B,) + od + (sp.cout,), dtype=torch.bfloat16, device=dev), sp.cout)
wp = E.pack_weight(w, sp, 2 if sp.transposed else 0, "bf16")
a = ((1, 1, 1), (-1, -1, -1), tuple(-p for p in sp.pad), sp.stride) if sp.transposed else (sp.stride, (1, 1, 1), sp.pad, (1, 1, 1))
for _ in range(3):
    E.gather_gemm(x, wp, bias, y, sp.k, sp.cin, sp.cout, *a, od)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 4096)()
lib = _lib.load()
lib.vn_debug_trace.restype = ctypes.c_int
rc = lib.vn_debug_trace(buf, 4096)
t = np.array(buf[:], dtype=np.int64)
n = int(t[0]); n = min(n, 500)
r = t[8:8 + n * 8].reshape(n, 8)
print(f"{name}: nsteps {n}; clocks per step (s_memtime): wait-vmcnt | barrier | prep+issue (burst build only) | rest (MFMA phase) | total")
for s in range(min(n, 10)):
    t0, t1, t2, t3, t4 = r[s, :5]
    nxt = r[s + 1, 0] if s + 1 < n else t4
    iss = (t3 - t2) if t3 else 0
    print(f"  step {s:3d}: {t1-t0:6d} | {t2-t1:6d} | {iss:6d} | {t4-(t3 if t3 else t2):6d} | {nxt-t0:6d}")
tot = r[1:, 0] - r[:-1, 0]
print(f"  mean step {tot.mean():.0f} clk; wait {np.mean(r[1:,1]-r[1:,0]):.0f}  barrier {np.mean(r[1:,2]-r[1:,1]):.0f}  "
      f"issue {np.mean(np.where(r[1:,3]>0, r[1:,3]-r[1:,2], 0)):.0f}  rest {np.mean(r[1:,4]-np.where(r[1:,3]>0, r[1:,3], r[1:,2])):.0f}")
