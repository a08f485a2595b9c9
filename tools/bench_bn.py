"""Micro-benchmark of the BatchNorm passes alone (no side stream beside them) at the first layers' sizes, bf16 rows:
achieved HBM rate per pass.  usage: python tools/bench_bn.py [batch]   (default 4: the dense config's batch)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "voxelnet-pytorch_amd")]
import torch
from voxelnet_amd import _lib
from voxelnet_amd import engine as E

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
lib = _lib.load()


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


print(f"batch {B}; us and TB/s per pass (algorithmic bytes: reduce 2 tensors, apply 3, forward apply 2)")
for name, D, H, W, C in (("middle_layer.0", 5, 400, 352, 64), ("middle_layer.1", 3, 400, 352, 64), ("middle_layer.2", 2, 400, 352, 64),
                         ("block1.0", 1, 200, 176, 128), ("block2.0", 1, 100, 88, 128), ("block3.0", 1, 50, 44, 256)):
    M = B * D * H * W
    y = torch.randn((M, C), device=dev).to(torch.bfloat16)
    da = torch.randn((M, C), device=dev).to(torch.bfloat16)
    dy = torch.empty_like(y)
    a = torch.empty_like(y)
    stats = torch.rand(4 * C, device=dev) + 0.5
    gamma = torch.rand(C, device=dev) + 0.5
    coef = torch.empty(3 * C, device=dev)
    dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
    rows = lib.vn_bn_bwd_slab_rows(M, C)
    slab = torch.empty((rows, 2, C), device=dev)
    st = E.stream()
    t_r = timeit(lambda: _lib.call("vn_bn_bwd_reduce_slab", da.data_ptr(), _lib.VN_BF16, C, y.data_ptr(), _lib.VN_BF16, C, M, C,
                                   stats.data_ptr(), 1, slab.data_ptr(), st))
    t_f = timeit(lambda: _lib.call("vn_bn_bwd_finalize_slab", slab.data_ptr(), rows, M, C, gamma.data_ptr(), stats.data_ptr(),
                                   coef.data_ptr(), dg.data_ptr(), db.data_ptr(), st))
    t_a = timeit(lambda: _lib.call("vn_bn_bwd_apply", da.data_ptr(), _lib.VN_BF16, C, y.data_ptr(), _lib.VN_BF16, C, M, C,
                                   stats.data_ptr(), coef.data_ptr(), 1, dy.data_ptr(), _lib.VN_BF16, C, 0, st))
    t_p = timeit(lambda: _lib.call("vn_bn_apply", y.data_ptr(), _lib.VN_BF16, C, M, C, stats.data_ptr(), 1, a.data_ptr(),
                                   _lib.VN_BF16, C, 0, st))
    nb = M * C * 2
    print(f"{name:16s} M {M:8d} C {C:3d} slab rows {rows:5d} | bwd reduce {t_r:7.1f} us {2 * nb / t_r / 1e6:5.2f} | finalize {t_f:6.1f} us | "
          f"bwd apply {t_a:7.1f} us {3 * nb / t_a / 1e6:5.2f} | fwd apply {t_p:7.1f} us {2 * nb / t_p / 1e6:5.2f}")
