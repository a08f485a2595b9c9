"""Shader-clock phases of a tap step of k_conv_patch2d (small-image 3x3 convolutions), from a -DVN_P2D_TRACE build:
    make -C voxelnet-pytorch_amd/csrc OUT=../../tools/ubench/bin/libp2dtrace.so BUILD=build/p2dtrace EXTRA=-DVN_P2D_TRACE
    python tools/trace_patch2d.py tools/ubench/bin/libp2dtrace.so block3.1 [block2.1]
wave 0 of one workgroup stamps s_memtime before its wait, after it, after the barrier, after the LDS-DMA issue and after
the step's MFMAs; printed per step and as means (the stamps themselves cost ~10 %)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "voxelnet-pytorch_amd")]
import numpy as np, torch
from voxelnet_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from voxelnet_amd import engine as E, net as N
dev = "cuda:0"
specs = dict(N.layer_table(2))
IN = {"block2.1": (1, 100, 88), "block3.1": (1, 50, 44)}
lib = _lib.load()
fn = ctypes.CDLL(_lib.LIB_PATH).vn_debug_p2d_trace
for name in sys.argv[2:] or ["block3.1", "block2.1"]:
    sp, dims, B = specs[name], IN[name], 2
    x = E.Rows(torch.randn((B,) + dims + (sp.cin,), device=dev).to(torch.bfloat16), sp.cin)
    w = torch.randn((sp.cout, sp.cin) + sp.k[3 - sp.dim:], device=dev) * 0.05
    bias = torch.zeros(sp.cout, device=dev)
    od = sp.out_dims(dims)
    y = E.Rows(torch.empty((B,) + od + (sp.cout,), dtype=torch.bfloat16, device=dev), sp.cout)
    wp = E.pack_weight(w, sp, 0, "bf16")
    a = (sp.stride, (1, 1, 1), sp.pad, (1, 1, 1))
    slab = torch.empty((-(-y.M // 32), 2, sp.cout), device=dev)
    big = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    for cold in (False, True):
        for _ in range(3):
            if cold:
                big.zero_()            # push the weights and the input out of the L2s / a good part of the Infinity Cache
            E.gather_gemm(x, wp, bias, y, sp.k, sp.cin, sp.cout, *a, od, stats=slab)
        torch.cuda.synchronize()
        buf = (ctypes.c_longlong * 512)()
        assert fn(buf) == 0
        t = np.array(buf[:], dtype=np.int64).reshape(64, 8)
        n = 9 * ((sp.cin + 63) // 64)
        t = t[:n]
        print(f"{name} ({'cold' if cold else 'warm'} caches): {n} tap steps; clocks: wait | barrier | DMA issue | reads+MFMA | step")
        for s in range(n):
            nxt = t[s + 1, 0] if s + 1 < n else t[s, 4]
            print(f"  step {s:2d}: {t[s,1]-t[s,0]:6d} | {t[s,2]-t[s,1]:6d} | {t[s,3]-t[s,2]:6d} | {t[s,4]-t[s,3]:6d} | {nxt-t[s,0]:6d}")
        d = t[1:, 0] - t[:-1, 0]
        print(f"  mean step {d.mean():.0f} clk: wait {np.mean(t[1:,1]-t[1:,0]):.0f}, barrier {np.mean(t[1:,2]-t[1:,1]):.0f}, "
              f"issue {np.mean(t[1:,3]-t[1:,2]):.0f}, reads+MFMA {np.mean(t[1:,4]-t[1:,3]):.0f}; first stamp to last {t[n-1,4]-t[0,0]} clk")
