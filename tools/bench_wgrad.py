"""Weight-gradient kernels alone (bf16), per layer shape of the car network: vn_conv_wgrad_partials (the launch the native
executor makes; the partial slabs stay in the workspace) timed with HIP events, plus the kernel the library picked and
its row chunks.  Knobs (VN_WGRAD_PATCH, VN_WGP2_BLOCKS, ...) are read once per process: one run per setting.
usage: python tools/bench_wgrad.py [layer ...]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "voxelnet-pytorch_amd")]
import torch
from voxelnet_amd import _lib, engine as E, net as N

dev = "cuda:0"
specs = dict(N.layer_table(2))
B = 2
IN = {"middle_layer.1": (5, 400, 352), "middle_layer.2": (3, 400, 352), "block1.0": (1, 400, 352), "block1.1": (1, 200, 176),
      "deconv1": (1, 200, 176), "block2.0": (1, 200, 176), "block2.1": (1, 100, 88), "deconv2": (1, 100, 88),
      "block3.0": (1, 100, 88), "block3.1": (1, 50, 44), "deconv3": (1, 50, 44)}
names = sys.argv[1:] or ["middle_layer.2", "block1.1", "deconv1", "block2.1", "block3.1", "block1.0", "block2.0", "block3.0", "deconv2", "deconv3"]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


lib = _lib.load()
print(lib.vn_build_info().decode())
tot = 0.0
for name in names:
    sp = specs[name]
    dims = IN[name]
    od = sp.out_dims(dims)
    x = E.Rows(torch.randn((B,) + dims + (sp.cin,), device=dev).to(torch.bfloat16), sp.cin)
    dy = E.Rows(torch.randn((B,) + od + (sp.cout,), device=dev).to(torch.bfloat16), sp.cout)
    taps = sp.taps
    if sp.transposed:
        g = E._geom(B, dy, dims, sp.cout, 0, sp.cin, sp.k, sp.stride, (1, 1, 1), sp.pad, (1, 1, 1), x.strides)
        src, rows = dy, x
        flops = 2.0 * x.M * sp.cout * sp.cin * taps
    else:
        g = E._geom(B, x, od, sp.cin, 0, sp.cout, sp.k, sp.stride, (1, 1, 1), sp.pad, (1, 1, 1), dy.strides)
        src, rows = x, dy
        flops = 2.0 * dy.M * sp.cout * sp.cin * taps
    ws, wsb = E.wgrad_workspace(g, 0, 0, dev)
    chunks = ctypes.c_int32(0)
    pid = lib.vn_conv_wgrad_plan_id(ctypes.byref(g), 0, 0)
    t = timeit(lambda: _lib.call("vn_conv_wgrad_partials", src.ptr(), rows.ptr(), ctypes.byref(g), 0, None, 0, ws.data_ptr(), wsb,
                                 ctypes.byref(chunks), E.stream()))
    tot += t
    part_mb = chunks.value * taps * sp.cin * sp.cout * 4 / 1e6
    print(f"{name:16s} plan {pid:5d} chunks {chunks.value:3d} partials {part_mb:6.1f} MB  {flops/1e9:7.1f} GF  {t*1e3:7.1f} us {flops/t/1e9:6.0f} TF")
print(f"sum {tot*1e3:.1f} us")
