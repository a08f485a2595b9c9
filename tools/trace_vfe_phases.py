"""Phase timings inside the VFE kernels from a -DVFE_TRACE build of the library (csrc/vfe.hip: lane 0 of every wave stores
the shader clock at the phase boundaries of its items):

    cd voxelnet-pytorch_amd/csrc && make OUT=../../tools/ubench/bin/libtrace.so BUILD=/tmp/trace_build EXTRA=-DVFE_TRACE
    VN_LIB_PATH=$PWD/tools/ubench/bin/libtrace.so python tools/trace_vfe_phases.py [dense]

prints, per kernel (p2, p3, b1, b2) and per packing class (G = 8 / 4 / 1 voxels per wave item), the mean clocks between
consecutive trace points and the items per wave."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "voxelnet-pytorch_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from voxelnet_amd import _lib, model as M, synth  # noqa: E402
from voxelnet_amd.config import grid_config  # noqa: E402
from voxelnet_amd.voxelize import voxelize_device  # noqa: E402

dev = "cuda:0"
dense = len(sys.argv) > 1 and sys.argv[1] == "dense"
grid = grid_config("Car", T=64) if dense else grid_config("Car")
feat = torch.cat([voxelize_device(torch.from_numpy(f).to(dev), grid, b, coord_cols=4)[0]
                  for b, f in enumerate(synth.workload_frames(5 if dense else 2))])
m = M.RPN3D("Car").to(dev).train()
params = [p.detach() for p in M._vfe_weights(m.feature_net)]
bufs = m.feature_net._bufs()
vw, stats, wst = M.featnet_forward(feat, params, bufs, True)
dvw = torch.randn_like(vw)
lib = ctypes.CDLL(_lib.LIB_PATH)
lib.vn_debug_vfe_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
NWAVES, NITEMS, NP = 1024 * 8, 60, 16   # VFE_BLOCKS_MAX workgroups x 8 wave slots (csrc/vfe.hip VFE_TR_SLOT)
NAMES = {1: "inputs", 2: "layer1+bn1", 3: "agg1", 4: "u", 5: "h2 (mfma)", 6: "impulses+d_pre2", 7: "d_agg1", 8: "dW2+d_p1 (mfma)",
         14: "rest of item"}
for kid, kname in ((2, "p2"), (3, "p3"), (11, "b1"), (12, "b2")):
    buf = torch.zeros(NWAVES * NITEMS * NP, dtype=torch.int64, device=dev)
    assert lib.vn_debug_vfe_trace(buf.data_ptr(), kid) == 0
    if kid < 10:
        M.featnet_forward(feat, params, bufs, True)
    else:
        M.featnet_backward(feat, wst, stats, dvw, params)
    torch.cuda.synchronize()
    assert lib.vn_debug_vfe_trace(None, 0) == 0
    t = buf.cpu().numpy().reshape(NWAVES, NITEMS, NP)
    used = t[:, :, 0] != 0
    per_wave = used.sum(1)
    span = np.where(per_wave > 0, t[:, :, 14].max(1) - np.where(used, t[:, :, 0], np.iinfo(np.int64).max).min(1), 0)
    print(f"== k_vfe_{kname}: waves with items {int((per_wave > 0).sum())}, items per wave mean {per_wave[per_wave > 0].mean():.2f} max {per_wave.max()}, "
          f"first-item-start to last-item-end per wave: mean {span[per_wave > 0].mean():.0f} max {span.max()} ticks")
    for g in (8, 4, 1):
        sel = used & (t[:, :, 15] == g)
        if not sel.any():
            continue
        rows = t[sel]
        line, prev = [], 0
        for k in sorted(NAMES):
            if (rows[:, k] == 0).all():
                continue
            d = rows[:, k] - rows[:, prev]
            line.append(f"{NAMES[k]} {d.mean():.0f}")
            prev = k
        print(f"   G={g}: {sel.sum()} items, total {np.mean(rows[:, 14] - rows[:, 0]):.0f} | " + " | ".join(line))
