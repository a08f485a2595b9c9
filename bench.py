#!/usr/bin/env python3
"""Benchmark of the VoxelNet training hot path on MI355X (BASELINE.json metric:
point-clouds/sec fwd+bwd, KITTI car voxel grid, batch=2 per GPU).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = the reference's train step (train.py:148-155) on one batch of synthetic KITTI-shaped
frames whose raw (N,4) point clouds are already resident in HBM:
    voxelize (HIP) -> VFE x2 + max (HIP) -> [sparse->dense scatter folded into the rulebook first Conv3d] -> 3 Conv3d +
    RPN (MFMA implicit GEMM) -> loss (model.py:310-352) -> backward of all of it -> [N>1: bucketed RCCL all-reduce
    overlapped with backward] -> clip_grad_norm_(5) + SGD(lr=0.01) step (fused: vn_clip_sgd) -> zero_grad.
Rank 0 prints ONE JSON line.  `roofline` is measured live (HIP events around every launch of the
MFMA kernels in eager steps right after the timed region); `cpu_baseline` times the oracle (PyTorch-CPU restatement of the
reference's op sequence + the C voxelizer) on this box's host cores, on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "voxelnet-pytorch_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16_DENSE_TFLOPS = 2500.0   # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
FLOP_PER_PC_FWD_BWD = 1732.3e9    # dense-equivalent conv/deconv/head FLOPs, fwd + bwd (x3)


def synthetic_targets(B, h, w, seed, device):
    """seeded pos/neg/targets maps with the shapes utils.generate_targets returns (model.py:309)"""
    rng = np.random.default_rng(seed)
    pos = (rng.random((B, h, w, 2)) < 0.002).astype(np.float32)
    neg = ((rng.random((B, h, w, 2)) < 0.98) & (pos == 0)).astype(np.float32)
    tgt = (rng.standard_normal((B, h, w, 14)) * 0.1).astype(np.float32)
    return tuple(torch.from_numpy(a).to(device) for a in (pos, neg, tgt))


def cpu_baseline(frames_np, threads):
    """oracle (kind 'port'): C voxelizer + PyTorch-CPU fwd+bwd of the reference's op sequence, 1 frame"""
    from oracle import torch_ref as tr
    from oracle import voxelize as ov
    torch.set_num_threads(threads)
    cloud = frames_np[0]
    t0 = time.perf_counter()
    v = ov.voxelize(cloud, "Car")
    f, _, c = ov.prepare_voxel([v])
    feats, coords = [torch.from_numpy(f[0])], [torch.from_numpy(c[0])]
    rng = np.random.default_rng(1)
    dp = torch.from_numpy((rng.standard_normal((1, 2, 200, 176)) * 1e-2).astype(np.float32))
    dr = torch.from_numpy((rng.standard_normal((1, 14, 200, 176)) * 1e-2).astype(np.float32))
    tr.forward_backward(feats, coords, tr.make_state_dict("Car"), (10, 400, 352), "Car", dp, dr)
    dt = time.perf_counter() - t0
    return {"value": 1.0 / dt, "unit": "point-clouds/s", "cores": threads, "kind": "port",
            "sample": "1 step on 1 car frame (batch=1): C voxelizer + PyTorch-CPU fwd+bwd of the reference op "
                      "sequence, %.1f s" % dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "bf16x3"])
    ap.add_argument("--batch", type=int, default=2, help="frames per GPU (BASELINE configs[1]: 2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="experimental: replay the post-voxelizer part of the step from captured HIP graphs")
    ap.add_argument("--torch-optim", action="store_true",
                    help="torch's clip_grad_norm_ + SGD instead of the fused vn_clip_sgd tail (same arithmetic)")
    ap.add_argument("--timer-steps", type=int, default=3)
    ap.add_argument("--force-reducer", action="store_true",
                    help="diagnostic: run the DDP bucket path (flat buckets, segmented backward) on one GPU")
    ap.add_argument("--static-voxels", action="store_true",
                    help="diagnostic: voxelize once, outside the timed steps (NOT the benchmark configuration)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (there is no CPU path)"
    # rehearsal aid (NOT a benchmark configuration): VN_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and uses gloo, so
    # the N>1 code path (broadcast, bucketed all-reduce overlapped with the segmented backward, MAX over ranks) can be
    # exercised on a one-GPU box
    share = os.environ.get("VN_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from voxelnet_amd import engine as E
    from voxelnet_amd import model as M
    from voxelnet_amd import parallel, synth
    from voxelnet_amd.config import GRADIENT_CLIP, LR, grid_config
    from voxelnet_amd.voxelize import VoxelBuffers, voxelize_device_async

    M.set_precision(args.precision)
    torch.manual_seed(1234)                      # same initial weights on every rank
    model = M.RPN3D("Car").to(dev)
    model.train(True)                            # train.py:148
    named = list(model.named_parameters())
    params = [p for _, p in named]
    from voxelnet_amd.optim import ClipSGD
    # train.py:130 + 153-154: clip_grad_norm_(5) + SGD(lr=0.01) as two HIP launches (csrc/optim.hip);
    # --torch-optim runs torch's own clip_grad_norm_ + SGD instead (same arithmetic, ~12 launches)
    opt = torch.optim.SGD(params, lr=LR) if args.torch_optim else ClipSGD(params, LR, GRADIENT_CLIP)
    if world > 1:
        for p in params:
            dist.broadcast(p.data, 0)
    if world > 1 or args.force_reducer:
        model.grad_reducer = parallel.GradAllReducer(named)

    B = args.batch
    grid = grid_config("Car")
    frames_np = synth.workload_frames(2, batch=B, frame0=rank * B)   # weak scaling: own frames per rank
    frames = [torch.from_numpy(f).to(dev) for f in frames_np]       # resident in HBM before timing
    h, w = model.rpn_output_shape
    targets = synthetic_targets(B, h, w, 99 + rank, dev)

    # Voxelization is software-pipelined one step ahead on its own HIP stream (the input-pipeline stage of the
    # step): the K read-back that sizes its outputs (utils.py:69-71 returns (K,T,7)/(K,3)/(K,) arrays) then only
    # never stalls the training queue (capacity-sized outputs, K read by the gather kernel from device memory,
    # asynchronous K copy to pinned memory: voxelize_device_async).  Every timed step still runs one
    # voxelization of its B frames (for the next step) and one train step (on the buffers voxelized during the
    # previous one).
    vox_stream = torch.cuda.Stream()
    pending = {}

    slots = [[VoxelBuffers(pts.shape[0], grid, 4, dev) for pts in frames] for _ in range(3)]   # 3-deep ring
    ring = {"i": 0, "use": 0}
    slot_free = [None, None, None]     # event recorded on the training stream after the last consumer of the slot

    def launch_voxelize():
        si = ring["i"] % 3
        bufs = slots[si]
        ring["i"] += 1
        if slot_free[si] is not None:
            # write-after-read: the train step that consumed this slot (VFE forward AND backward read its feature
            # buffer) must have finished before the voxelizer overwrites it — the host runs several steps ahead
            vox_stream.wait_event(slot_free[si])
        with torch.cuda.stream(vox_stream):
            pending["next"] = [voxelize_device_async(pts, grid, b, coord_cols=4, buffers=bufs[b])
                               for b, pts in enumerate(frames)]

    def voxelize_batch():
        if args.static_voxels and "static" in pending:
            return pending["static"]
        if "next" not in pending:
            launch_voxelize()
        handles = pending.pop("next")
        feats, coords = [], []
        for hdl in handles:
            f, c, _ = hdl.result()                       # waits for the (long finished) K copy only
            feats.append(f)
            coords.append(c)
        torch.cuda.current_stream().wait_event(handles[-1].event)
        launch_voxelize()
        if args.static_voxels:
            torch.cuda.synchronize()
            pending["static"] = (feats, coords)
        return feats, coords

    def fwd_bwd(feats, coords):
        batch = (None, None, feats, None, coords, None, None)
        out = model(batch, dev, targets=targets)
        out[2].backward()                                                  # train.py:151
        return out[2]

    def reduce_grads():
        if model.grad_reducer is not None:
            model.grad_reducer.finish(named)

    def optim():
        if args.torch_optim:
            torch.nn.utils.clip_grad_norm_(params, GRADIENT_CLIP)          # train.py:153
        opt.step()                                                         # train.py:154 (ClipSGD: both lines)

    def step_eager():
        feats, coords = voxelize_batch()
        loss = fwd_bwd(feats, coords)
        if not args.static_voxels:         # this step's slot may be re-used once the backward (queued above) is done
            si = ring["use"] % 3
            ring["use"] += 1
            ev = slot_free[si] if slot_free[si] is not None else torch.cuda.Event()
            ev.record()
            slot_free[si] = ev
        reduce_grads()
        optim()
        opt.zero_grad(set_to_none=True)                                    # train.py:155
        return loss

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(max(1, args.warmup)):
        loss = step_eager()
    sync_all()

    # ---- HIP-graph mode: everything after the voxelizer is replayed from two captured graphs ----------
    # (voxelization reads K back to size its outputs, so it stays eager; the RCCL all-reduce runs between
    #  the fwd+bwd graph and the clip+SGD graph).  Shapes are static because the frames of this benchmark
    #  are; a change of K re-captures.
    mode = "eager"
    step = step_eager
    if args.graph:
        try:
            feats0, coords0 = voxelize_batch()
            st_feats = [f.clone() for f in feats0]
            st_coords = [c.clone() for c in coords0]
            opt.zero_grad(set_to_none=True)
            if model.grad_reducer is not None:
                model.grad_reducer.defer_allreduce = True          # bucket copies are captured, collectives are not
            torch.cuda.synchronize()
            g1 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1):
                st_loss = fwd_bwd(st_feats, st_coords)
            if model.grad_reducer is not None:
                model.grad_reducer.finish(named, launch_deferred=True)   # p.grad -> static bucket views
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2):
                optim()
            torch.cuda.synchronize()

            def step_graph():
                feats, coords = voxelize_batch()
                for s_, f_ in zip(st_feats, feats):
                    if s_.shape != f_.shape:
                        raise RuntimeError("voxel count changed: static-shape graph is stale")
                    s_.copy_(f_)
                for s_, c_ in zip(st_coords, coords):
                    s_.copy_(c_)
                g1.replay()
                if model.grad_reducer is not None:
                    model.grad_reducer.allreduce_all()
                g2.replay()
                return st_loss
            step = step_graph
            for _ in range(2):
                loss = step()
            sync_all()
            mode = "hipgraph"
        except Exception as e:   # noqa: BLE001 - report and fall back to the eager step
            if rank == 0:
                import traceback
                traceback.print_exc()
                print(f"[bench] graph capture failed ({type(e).__name__}); running eager", file=sys.stderr)
            torch.cuda.synchronize()
            if model.grad_reducer is not None:
                model.grad_reducer.defer_allreduce = False
                model.grad_reducer.reset()
            opt.zero_grad(set_to_none=True)
            step = step_eager
            mode = "eager"

    sync_all()
    t0 = time.perf_counter()
    step_events = []
    for _ in range(args.steps):
        loss = step()
        if os.environ.get("VN_BENCH_STEP_TIMES"):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            step_events.append(ev)
    t_enq = time.perf_counter() - t0       # host time to enqueue the K steps (the GPU may still be running)
    sync_all()
    dt = time.perf_counter() - t0
    if step_events and rank == 0:
        print("[bench] per-step ms:", " ".join(f"{a.elapsed_time(b):.2f}" for a, b in zip(step_events, step_events[1:])),
              file=sys.stderr)
    assert torch.isfinite(loss).item(), "non-finite loss"
    ranks_in_sync = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # self-check of the data-parallel path (outside the timed region): every rank started from rank 0's weights
        # and applied the same averaged gradients, so the parameters must still be bit-identical on all ranks
        with torch.no_grad():
            cs = torch.stack([torch.stack([p.double().sum(), p.double().abs().sum()]) for p in params]).sum(0)
        lo, hi = cs.clone(), cs.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        ranks_in_sync = bool(torch.equal(lo, hi))
        if not ranks_in_sync and rank == 0:
            print("[bench] WARNING: parameter checksums differ between ranks", file=sys.stderr)

    # ---- per-kernel durations: HIP events around every MFMA-kernel launch, eager steps on the same inputs
    timer = None
    if not args.no_kernel_timer and rank == 0 or (not args.no_kernel_timer and world > 1):
        if mode == "hipgraph":
            if model.grad_reducer is not None:
                model.grad_reducer.defer_allreduce = False
                model.grad_reducer.reset()
            opt.zero_grad(set_to_none=True)
        timer = E.KernelTimer()
        E.TIMER = timer
        native = model.native_executor
        model.native_executor = False      # per-launch events need the per-launch (Python) orchestration:
        for _ in range(args.timer_steps):  # same kernels, same shapes, same inputs
            step_eager()
        sync_all()
        model.native_executor = native
        E.TIMER = None

    if rank == 0:
        value = world * B * args.steps / dt
        res = {
            "metric": "point-clouds/sec fwd+bwd, KITTI car voxel grid, batch=2",
            "value": value, "unit": "point-clouds/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"bf16": "bf16", "fp32": "f32", "bf16x3": "bf16x3"}[args.precision], "data": "synthetic",
            "config": {"workload": "KITTI car config (voxel 0.2x0.2x0.4 m, grid 10x400x352, T=35), batch=%d per GPU, "
                                   "fwd+bwd train step (BASELINE configs[1])" % B,
                       "global_batch": world * B, "points_per_frame": int(frames_np[0].shape[0]),
                       "parallelism": "dp%d" % world,
                       "step": "voxelize+VFE+scatter+Conv3d+RPN fwd, loss, bwd, clip_grad_norm, SGD",
                       "launch_mode": mode + ("+native-executor" if model.native_executor else "")},
            "host_enqueue_ms_per_step": 1e3 * t_enq / args.steps,
            "model_flops_fraction_of_bf16_peak": value / world * FLOP_PER_PC_FWD_BWD / (PEAK_BF16_DENSE_TFLOPS * 1e12),
        }
        if ranks_in_sync is not None:
            res["ranks_in_sync"] = ranks_in_sync       # parameters bit-identical on all ranks after the timed steps
        if timer is not None:
            summ = timer.summary()
            kern = {}
            for k, (n, fl, ms) in summ.items():
                kern[k] = {"launches_per_step": n / args.timer_steps, "ms_per_step": ms / args.timer_steps,
                           "achieved_tflops": fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0}
            dom = max(summ, key=lambda k: summ[k][2])
            n, fl, ms = summ[dom]
            ach = fl / (ms * 1e-3) / 1e12
            peak = PEAK_BF16_DENSE_TFLOPS if args.precision != "fp32" else 157.3
            # HBM-side bytes per launch of the convolution family: NOT measurable here (PMC counters need rocprofv3);
            # taken from the committed counter passes on this same command (profiles/, tools/pmc_family.py), bf16/B=2 only
            traffic = None
            pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_traffic.json")
            if dom == "k_gather_gemm" and args.precision == "bf16" and args.batch == 2 and os.path.exists(pmc):
                with open(pmc) as fh:
                    traffic = json.load(fh).get("traffic_bytes_per_launch")
            res["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                               "frac": ach / peak, "traffic": traffic,
                               "avg_launch_us": 1e3 * ms / n, "launches": n,
                               "note": "algorithmic (dense-equivalent) FLOPs of all launches / summed HIP-event time, "
                                       "%d eager steps on the same inputs right after the timed region" % args.timer_steps}
            res["kernels"] = kern
        if world == 1 and not args.no_cpu_baseline:
            try:
                ncpu = len(os.sched_getaffinity(0))
            except AttributeError:
                ncpu = os.cpu_count() or 1
            res["cpu_baseline"] = cpu_baseline(frames_np, max(1, min(ncpu, 16)))   # a 1-GPU box's CPU share is 16
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
