"""Micro-benchmark of the two MFMA kernels on the network's layer shapes (bf16), HIP-event timed.
usage: python tools/bench_conv.py [layer ...]   (default: a representative set)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "voxelnet-pytorch_amd")]
import torch
from voxelnet_amd import engine as E, net as N
if os.environ.get("VN_LIB"):        # an experimental build of the library (tools/ubench/bin/*.so)
    from voxelnet_amd import _lib as _l
    _l.LIB_PATH = os.path.abspath(os.environ["VN_LIB"])

dev = "cuda:0"
specs = dict(N.layer_table(2))
specs["heads"] = N.HEADS
B = 2
IN = {"middle_layer.1": (5, 400, 352), "middle_layer.2": (3, 400, 352), "block1.0": (1, 400, 352), "block1.1": (1, 200, 176),
      "deconv1": (1, 200, 176), "block2.0": (1, 200, 176), "block2.1": (1, 100, 88), "deconv2": (1, 100, 88),
      "block3.0": (1, 100, 88), "heads": (1, 200, 176), "block3.1": (1, 50, 44), "deconv3": (1, 50, 44), "middle_layer.0": (10, 400, 352)}
names = sys.argv[1:] or ["middle_layer.1", "middle_layer.2", "block1.0", "block1.1", "block2.1", "block3.1", "deconv1", "deconv3"]
mode = "bf16"

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

for name in names:
    sp = specs[name]
    dims = IN[name]
    x = E.Rows(torch.randn((B,) + dims + (sp.cin,), device=dev).to(torch.bfloat16), sp.cin)
    w = torch.randn((sp.cin, sp.cout) + sp.k[3 - sp.dim:] if sp.transposed else (sp.cout, sp.cin) + sp.k[3 - sp.dim:], device=dev) * 0.05
    bias = torch.zeros(sp.cout, device=dev)
    od = sp.out_dims(dims)
    y = E.Rows(torch.empty((B,) + od + (sp.cout,), dtype=torch.bfloat16, device=dev), sp.cout)
    wp = E.pack_weight(w, sp, 2 if sp.transposed else 0, mode)
    wpd = E.pack_weight(w, sp, 3 if sp.transposed else 1, mode)
    taps = sp.taps
    flops = 2.0 * y.M * sp.cout * sp.cin * taps if not sp.transposed else 2.0 * x.M * sp.cout * sp.cin * taps
    if sp.transposed:
        a = ((1, 1, 1), (-1, -1, -1), tuple(-p for p in sp.pad), sp.stride)
        b = (sp.stride, (1, 1, 1), sp.pad, (1, 1, 1))
    else:
        a = (sp.stride, (1, 1, 1), sp.pad, (1, 1, 1))
        b = ((1, 1, 1), (-1, -1, -1), tuple(-p for p in sp.pad), sp.stride)
    slab = torch.empty((-(-y.M // 32), 2, sp.cout), device=dev) if not sp.transposed and name != "heads" else None
    t_f = timeit(lambda: E.gather_gemm(x, wp, bias, y, sp.k, sp.cin, sp.cout, *a, od, stats=slab))
    dx = E.Rows(torch.empty((B,) + dims + (sp.cin,), dtype=torch.bfloat16, device=dev), sp.cin)
    dy = E.Rows(torch.randn((B,) + od + (sp.cout,), device=dev).to(torch.bfloat16), sp.cout)
    t_d = timeit(lambda: E.gather_gemm(dy, wpd, None, dx, sp.k, sp.cout, sp.cin, *b, dims))
    import ctypes
    from voxelnet_amd import _lib
    if sp.transposed:
        dwp = torch.zeros((taps, sp.cin, sp.cout), device=dev)
        g = E._geom(B, dy, dims, sp.cout, 0, sp.cin, sp.k, sp.stride, (1, 1, 1), sp.pad, (1, 1, 1), x.strides)
        ws, wsb = E.wgrad_workspace(g, 0, 0, dev)
        t_w = timeit(lambda: _lib.call("vn_conv_wgrad", dy.ptr(), x.ptr(), dwp.data_ptr(), ctypes.byref(g), 0, ws.data_ptr(), wsb, E.stream()))
    else:
        dwp = torch.zeros((taps, sp.cout, sp.cin), device=dev)
        g = E._geom(B, x, od, sp.cin, 0, sp.cout, sp.k, sp.stride, (1, 1, 1), sp.pad, (1, 1, 1), dy.strides)
        ws, wsb = E.wgrad_workspace(g, 0, 0, dev)
        t_w = timeit(lambda: _lib.call("vn_conv_wgrad", x.ptr(), dy.ptr(), dwp.data_ptr(), ctypes.byref(g), 0, ws.data_ptr(), wsb, E.stream()))
    print(f"{name:16s} {flops/1e9:7.1f} GF  fwd {t_f*1e3:7.1f} us {flops/t_f/1e9:6.0f} TF | dgrad {t_d*1e3:7.1f} us {flops/t_d/1e9:6.0f} TF | wgrad {t_w*1e3:7.1f} us {flops/t_w/1e9:6.0f} TF")
