"""vn_bn_finalize_slab + vn_bn_apply against vn_bn_finalize_apply_slab (and the backward pair) alone, per layer shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "voxelnet-pytorch_amd")]
import torch
from voxelnet_amd import _lib, engine as E
dev = "cuda:0"
vdt, dt = _lib.VN_BF16, torch.bfloat16


def timeit(fn, n=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for name, M, C, rows in [("block3", 4400, 256, 69), ("block2", 17600, 128, 275), ("block1", 70400, 128, 550), ("deconv", 70400, 256, 640)]:
    y = torch.randn((M, C), device=dev).to(dt)
    da = torch.randn((M, C), device=dev).to(dt)
    a = torch.empty_like(y)
    dy = torch.empty_like(y)
    slab = torch.rand((rows, 2, C), device=dev) + 1.0
    shift, gamma, beta = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.zeros(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    stats, coef = torch.empty(4 * C, device=dev), torch.empty(3 * C, device=dev)
    dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
    st = E.stream()
    fin = lambda: _lib.call("vn_bn_finalize_slab", slab.data_ptr(), rows, M, C, shift.data_ptr(), gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0.1, 1e-5, stats.data_ptr(), st)
    app = lambda: _lib.call("vn_bn_apply", y.data_ptr(), vdt, C, M, C, stats.data_ptr(), 1, a.data_ptr(), vdt, C, 0, st)
    fus = lambda: _lib.call("vn_bn_finalize_apply_slab", slab.data_ptr(), rows, M, C, shift.data_ptr(), gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0.1, 1e-5, stats.data_ptr(), y.data_ptr(), vdt, C, 1, a.data_ptr(), vdt, C, st)
    bfin = lambda: _lib.call("vn_bn_bwd_finalize_slab", slab.data_ptr(), rows, M, C, gamma.data_ptr(), stats.data_ptr(), coef.data_ptr(), dg.data_ptr(), db.data_ptr(), st)
    bapp = lambda: _lib.call("vn_bn_bwd_apply", da.data_ptr(), vdt, C, y.data_ptr(), vdt, C, M, C, stats.data_ptr(), coef.data_ptr(), 1, dy.data_ptr(), vdt, C, 0, st)
    bfus = lambda: _lib.call("vn_bn_bwd_finalize_apply_slab", slab.data_ptr(), rows, M, C, gamma.data_ptr(), stats.data_ptr(), coef.data_ptr(), dg.data_ptr(), db.data_ptr(), da.data_ptr(), vdt, C, y.data_ptr(), vdt, C, 1, dy.data_ptr(), vdt, C, st)
    both = lambda: (fin(), app())
    bboth = lambda: (bfin(), bapp())
    print(f"{name:8s} M {M:6d} C {C:3d}: finalize {timeit(fin):5.1f}  apply {timeit(app):5.1f}  both {timeit(both):5.1f}  fused {timeit(fus):5.1f} us | "
          f"bwd finalize {timeit(bfin):5.1f}  apply {timeit(bapp):5.1f}  both {timeit(bboth):5.1f}  fused {timeit(bfus):5.1f} us")
