"""Where does the bf16 forward leave the fp32 forward?  Per layer of the full-size car step (B=2), per-layer orchestration
(voxelnet_amd/net.py) in both modes: relative L2 distance of the conv output y and of the activation a, and what makes a
train-mode BatchNorm amplify the rounding of a stored y: max and median over the channels of |mean| / std of y (a bf16
value carries 2^-9 of ITSELF as rounding noise, the BatchNorm divides what is left after the mean by std).

    python tools/forward_error_profile.py > gpurun_out/fwd_profile.log
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "voxelnet-pytorch_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import torch_ref as tr  # noqa: E402
from voxelnet_amd import model as M  # noqa: E402
from voxelnet_amd import net as N  # noqa: E402
from voxelnet_amd import synth  # noqa: E402
from voxelnet_amd.config import grid_config  # noqa: E402
from voxelnet_amd.voxelize import voxelize_device  # noqa: E402

DEV = "cuda:0"


def forward(mode, feats, coords, sd, sparse=True):
    M.set_precision(mode)
    m = M.RPN3D("Car")
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    fn, mid = m.feature_net, m.middle_rpn
    feature = torch.cat(feats, 0).contiguous()
    coord = torch.cat(coords, 0).contiguous()
    vparams = [p.detach() for p in M._vfe_weights(fn)]
    names, P, Bf, flat = M._collect_middle(mid)
    P = M._detached(P)
    P["heads"] = M._heads_params([f.detach() for f in flat[-4:]])
    vw, _, _ = M.featnet_forward(feature, vparams, fn._bufs(), True)
    dense = M.scatter_rows(vw, coord, 2, fn._grid.dims, mode)
    vw_rows = vw if mode == "fp32" else vw.bfloat16()
    prob, reg, st = N.middle_forward(dense, P, Bf, mid._block1_stride, True, mode, sparse=(coord, vw_rows) if sparse else None)
    torch.cuda.synchronize()
    return names, prob, reg, st, P


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def main():
    grid = grid_config("Car")
    feats, coords = [], []
    for b, f in enumerate(synth.workload_frames(2, batch=2)):
        fb, cb, _ = voxelize_device(torch.from_numpy(f).to(DEV), grid, b, coord_cols=4)
        feats.append(fb)
        coords.append(cb)
    sd = tr.make_state_dict("Car")
    names, p32, r32, st32, P = forward("fp32", feats, coords, sd)
    _, p32b, r32b, st32b, _ = forward("fp32", feats, coords, sd, sparse=False)
    print(f"fp32 sparse first layer vs fp32 dense first layer (summation order only): prob {rel(p32b, p32):.2e} reg {rel(r32b, r32):.2e}")
    for n in names:
        print(f"   fp32 order-only {n:16s} y {rel(st32b.layers[n].y.t, st32.layers[n].y.t):.2e}  a {rel(st32b.layers[n].a.t, st32.layers[n].a.t):.2e}")
    del st32b
    _, p16, r16, st16, _ = forward("bf16", feats, coords, sd)
    print(f"bf16 vs fp32 maps: prob {rel(p16, p32):.2e} reg {rel(r16, r32):.2e}")
    for n in names:
        y32, a32 = st32.layers[n].y.t, st32.layers[n].a.t
        y16, a16 = st16.layers[n].y.t.float(), st16.layers[n].a.t.float()
        C = y32.shape[-1]
        yf = y32.reshape(-1, C).double()
        mean, std = yf.mean(0), yf.std(0)
        ratio = (mean.abs() / (std + 1e-30)).cpu().numpy()
        bias = P[n]["bias"].double()
        rb = (bias.abs() / (std + 1e-30)).cpu().numpy()
        mask_flip = float(((a32.reshape(-1) > 0) != (a16.reshape(-1) > 0)).double().mean()) if a32.shape == a16.shape else float("nan")
        print(f"{n:16s} y rel-L2 {rel(y16, y32):.2e}  a rel-L2 {rel(a16, a32):.2e}  relu-mask flips {mask_flip:.2e}  |mean|/std max {ratio.max():.1f} "
              f"median {np.median(ratio):.2f}  |bias|/std max {rb.max():.1f} median {np.median(rb):.2f}  std min {float(std.min()):.2e}")


if __name__ == "__main__":
    main()
