"""Per-stage cycle trace of one k_wgrad_patch workgroup (instrumented copy of wgrad.hip built by hand)."""
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
IN = {"middle_layer.1": (5, 400, 352), "middle_layer.2": (3, 400, 352), "block1.1": (1, 200, 176), "block2.1": (1, 100, 88), "block3.1": (1, 50, 44)}
sp, dims, B = specs[name], IN[name], 2
x = E.Rows(torch.randn((B,) + dims + (sp.cin,), device=dev).to(torch.bfloat16), sp.cin)
od = sp.out_dims(dims)
dy = E.Rows(torch.randn((B,) + od + (sp.cout,), device=dev).to(torch.bfloat16), sp.cout)
dwp = torch.zeros((sp.taps, sp.cout, sp.cin), device=dev)
g = E._geom(B, x, od, sp.cin, 0, sp.cout, sp.k, sp.stride, (1, 1, 1), sp.pad, (1, 1, 1), dy.strides)
ws, wsb = E.wgrad_workspace(g, 0, 0, dev)
for _ in range(3):
    _lib.call("vn_conv_wgrad", x.ptr(), dy.ptr(), dwp.data_ptr(), ctypes.byref(g), 0, ws.data_ptr(), wsb, E.stream())
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 4096)()
lib = _lib.load(); lib.vn_debug_trace.restype = ctypes.c_int
lib.vn_debug_trace(buf, 4096)
t = np.array(buf[:], dtype=np.int64)
n = min(int(t[0]), 400)
r = t[8:8 + n * 8].reshape(n, 8)
print(f"{name}: stages {n}; wait | barrier | issue next stage | transposed reads + MFMA | total")
for s in range(2, min(n, 8)):
    t0, t1, t2, t3, t4, t5 = r[s, :6]
    print(f"  {t1-t0:6d} | {t2-t1:6d} | {t3-t2:6d} | {t4-t3:6d} (table {t5-t3 if t5 else 0}) | {r[s+1,0]-t0 if s+1<n else t4-t0:6d}")
tot = r[1:, 0] - r[:-1, 0]
print(f"  mean stage {tot.mean():.0f} clk")
