"""Which bf16-stored tensor class moves the benchmarked mode's gradients away from the fp32 parity mode's?

One full-size car step (B=2, the step tests/test_gpu_bf16_parity.py::test_bf16_step_vs_fp32_step runs) in the fp32 mode,
then in the bf16 mode with one tensor class at a time promoted to fp32 storage (vnNetConfig.grad_storage bits, see
include/voxelnet_hip.h), printing per parameter group: relative L2 distance and cosine to the fp32 step's gradient, and a
step time for each variant (forward + loss + backward, HIP events over --steps repetitions).

    python tools/grad_attribution.py [--steps 20] [--variants 0,8,1,9,...]  > gpurun_out/attr.log
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "voxelnet-pytorch_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import torch_ref as tr  # noqa: E402  (closed-form weights only: the fixtures' fill)
from voxelnet_amd import model as M  # noqa: E402
from voxelnet_amd import synth  # noqa: E402
from voxelnet_amd.config import grid_config  # noqa: E402
from voxelnet_amd.voxelize import voxelize_device  # noqa: E402

DEV = "cuda:0"
GROUPS = [("heads", ("prob_conv", "reg_conv")), ("deconv3", ("deconv3",)), ("block3", ("block3",)), ("deconv2", ("deconv2",)),
          ("block2", ("block2",)), ("deconv1", ("deconv1",)), ("block1", ("block1",)), ("middle", ("middle_layer",)),
          ("vfe", ("feature_net",))]
BITS = {1: "dcat f32", 2: "dx f32", 4: "y f32", 8: "exact heads dgrad"}


def group_of(name):
    for g, keys in GROUPS:
        if any(k in name for k in keys):
            return g
    raise KeyError(name)


def is_dead_bias(k):
    return (k.endswith("conv.bias") and "prob_conv" not in k and "reg_conv" not in k) or k.endswith("deconv.bias")


def bf16_valued_weights(sd):
    """conv / deconv / head weights rounded to the nearest bf16 value (what the bf16 mode's weight packing reads)"""
    out = {}
    for k, v in sd.items():
        conv_w = k.endswith("conv.weight") or k.endswith("deconv.weight")
        out[k] = v.bfloat16().float() if conv_w else v
    return out


def run_step(mode, gs, feats, coords, targets, steps, round_w=False):
    M.set_precision(mode)
    m = M.RPN3D("Car")
    sd = tr.make_state_dict("Car")
    m.load_state_dict(bf16_valued_weights(sd) if round_w else sd)
    m = m.to(DEV).train()
    m.grad_storage = gs
    batch = (None, None, feats, None, coords, None, None)
    res = m(batch, DEV, targets=targets)
    res[2].backward()
    torch.cuda.synchronize()
    out = (res[0].detach().double().cpu(), res[1].detach().double().cpu(),
           {k: p.grad.detach().double().cpu().clone() for k, p in m.named_parameters()}, float(res[2]))
    ms = float("nan")
    if steps > 0:
        for _ in range(3):
            m.zero_grad(set_to_none=True)
            m(batch, DEV, targets=targets)[2].backward()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            m.zero_grad(set_to_none=True)
            m(batch, DEV, targets=targets)[2].backward()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / steps
    del m
    torch.cuda.empty_cache()
    return out, ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--variants", default="0,8,1,9,2,11,4,13,15,f0w,f16,f16w")
    ap.add_argument("--per-param", action="store_true")
    a = ap.parse_args()
    grid = grid_config("Car")
    frames = synth.workload_frames(2, batch=2)
    feats, coords = [], []
    for b, f in enumerate(frames):
        fb, cb, _ = voxelize_device(torch.from_numpy(f).to(DEV), grid, b, coord_cols=4)
        feats.append(fb)
        coords.append(cb)
    rng = np.random.default_rng(99)
    pos = (rng.random((2, 200, 176, 2)) < 0.002).astype(np.float32)
    neg = ((rng.random((2, 200, 176, 2)) < 0.98) & (pos == 0)).astype(np.float32)
    tgt = (rng.standard_normal((2, 200, 176, 14)) * 0.1).astype(np.float32)
    targets = tuple(torch.from_numpy(x).to(DEV) for x in (pos, neg, tgt))
    t0 = time.time()
    ref, ms_ref = run_step("fp32", 0, feats, coords, targets, min(a.steps, 5))
    print(f"fp32 step: loss {ref[3]:.6f}, {ms_ref:.2f} ms/step (fwd+loss+bwd)  [{time.time() - t0:.0f} s]", flush=True)
    for var in a.variants.split(","):
        # "<bits>" = bf16 mode with grad_storage bits; "f<bits>" = fp32 mode (16: activations rounded to bf16 values);
        # a trailing "w": the conv weights are rounded to bf16 values first (fp32 mode: the bf16 mode always reads them so)
        round_w = var.endswith("w")
        var = var.rstrip("w")
        mode = "fp32" if var.startswith("f") else "bf16"
        gs = int(var.lstrip("f"))
        out, ms = run_step(mode, gs, feats, coords, targets, a.steps if mode == "bf16" else min(a.steps, 5), round_w)
        what = " + ".join(BITS[b] for b in BITS if gs & b) or ("all bf16 (round 2)" if mode == "bf16" else "plain")
        if mode == "fp32":
            what = "fp32 kernels, " + ("activations" if gs & 16 else "nothing") + (" and conv weights" if round_w else "") + " rounded to bf16 values"
        line = [f"{mode} grad_storage {gs:2d} ({what}): loss {out[3]:.6f} (rel {abs(out[3] - ref[3]) / abs(ref[3]):.2e}), {ms:.3f} ms/step"]
        for i, nm in enumerate(("prob", "reg")):
            x, y = out[i], ref[i]
            line.append(f"   {nm} map: max err / max {float((x - y).abs().max() / y.abs().max()):.2e}, rel-L2 {float((x - y).norm() / y.norm()):.2e}")
        acc = {}
        for k, gb in out[2].items():
            if is_dead_bias(k):
                continue
            gf = ref[2][k]
            l2 = float((gb - gf).norm() / (gf.norm() + 1e-30))
            cos = float((gb * gf).sum() / (gb.norm() * gf.norm() + 1e-30))
            kind = "w" if k.endswith("weight") and ("conv." in k or "fcn." in k) else "bn/b"
            acc.setdefault((group_of(k), kind), []).append((l2, cos, k))
            if a.per_param:
                line.append(f"      {k:52s} rel-L2 {l2:.3f}  cos {cos:.3f}")
        for g, _ in GROUPS:
            parts = []
            for kind in ("w", "bn/b"):
                v = acc.get((g, kind))
                if v:
                    parts.append(f"{kind}: rel-L2 max {max(x[0] for x in v):.3f} mean {sum(x[0] for x in v) / len(v):.3f}, cos min {min(x[1] for x in v):.3f}")
            line.append(f"   {g:8s} " + " | ".join(parts))
        allv = [x for v in acc.values() for x in v]
        worst = max(allv)
        line.append(f"   ALL: worst rel-L2 {worst[0]:.3f} ({worst[2]}), min cos {min(x[1] for x in allv):.3f}")
        print("\n".join(line), flush=True)


if __name__ == "__main__":
    main()
