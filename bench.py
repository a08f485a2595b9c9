#!/usr/bin/env python3
"""Benchmark of the VoxelNet training hot path on MI355X (BASELINE.json metric:
point-clouds/sec fwd+bwd, KITTI car voxel grid, batch=2 per GPU).

    python bench.py --gpus 1 --steps K --warmup W                      (default workload: BASELINE configs[1])
    python bench.py --config ped | dense                               (BASELINE configs[2] / configs[4])
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = the reference's train step (train.py:148-155) on one batch of synthetic KITTI-shaped frames whose raw (N,4)
point clouds are already resident in HBM:
    voxelize (HIP) -> VFE x2 + max (HIP) -> [sparse->dense scatter folded into the rulebook first Conv3d] -> 3 Conv3d +
    RPN (MFMA implicit GEMM) -> loss (model.py:310-352) -> backward of all of it -> [N>1: bucketed RCCL all-reduce
    overlapped with backward] -> clip_grad_norm_(5) + SGD(lr=0.01) step (fused: vn_clip_sgd) -> zero_grad.
(ped: the reference's loss cannot run for Pedestrian / Cyclist — its anchor grid does not match the network's output,
SURVEY.md 8a-a8 — so the backward starts from a fixed seeded upstream gradient)

Targets: by default every timed step generates the RPN targets from the frames' label lines ON THE DEVICE inside the step
(model.py:309 is part of RPN3D.forward in the reference too); --precomputed-targets takes seeded target maps instead (the
round-1/2 behaviour; the default run also reports that rate as `value_precomputed_targets`).
The timed region of `--steps` steps is the headline (`value`); it is repeated `--windows` - 1 more times in the same run and
`value_min / value_median / value_max` give the spread over all windows (box-to-box variance on this pool is larger).

Rank 0 prints ONE JSON line.
  roofline      the dominant kernel family — the implicit-GEMM convolutions (k_conv_patch + k_gather_gemm: forward and data
                gradient of every layer) — measured live: HIP events around every launch ON THE STREAM IT IS LAUNCHED ON,
                inside the same native executor path the timed region runs (vn_net_timing_begin/_read), in
                `--timer-steps` steps on the same inputs right after the timed region (events inside the timed region
                would cost host time there).  achieved = algorithmic FLOPs (2 x MACs of the layer
                definition; a data gradient counts the layer's forward FLOPs) / summed launch time.
  kernels       the same for every family of the step; HBM-bound families carry algorithmic bytes and GB/s.
  cpu_baseline  the oracle (PyTorch-CPU restatement of the reference's op sequence + the C voxelizer) timed on this
                box's host cores on a bounded sample.
"""
import argparse
import ctypes
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
PEAK_F32_MATRIX_TFLOPS = 157.3    # v_mfma_f32_16x16x4_f32 (the fp32 parity mode)
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s
# dense-equivalent conv/deconv/head FLOPs per point cloud, forward + backward (x3)
FLOP_PER_PC = {"car": 1732.3e9, "dense": 1732.3e9, "ped": 929.0e9}

CONFIGS = {
    # name: (BASELINE.json configs index, synth workload id, class, T, default batch, description)
    "car": (1, 2, "Car", 35, 2, "KITTI car config (voxel 0.2x0.2x0.4 m, grid 10x400x352, T=35)"),
    "ped": (2, 3, "Pedestrian", 45, 2, "KITTI pedestrian/cyclist config (grid 10x200x240, T=45)"),
    "dense": (4, 5, "Car", 64, 4, "dense synthetic scene (~300k pts/frame, 40k voxels, T=64, car grid)"),
}
KIND_NAMES = ["conv_fwd", "conv_dgrad", "wgrad", "bn_apply", "bn_bwd_reduce", "bn_bwd_apply", "bn_finalize", "unpack_wgrads",
              "pack_weights", "first_layer_sparse", "misc"]


def synthetic_targets(B, h, w, seed, device):
    """seeded pos/neg/targets maps with the shapes utils.generate_targets returns (model.py:309)"""
    rng = np.random.default_rng(seed)
    pos = (rng.random((B, h, w, 2)) < 0.002).astype(np.float32)
    neg = ((rng.random((B, h, w, 2)) < 0.98) & (pos == 0)).astype(np.float32)
    tgt = (rng.standard_normal((B, h, w, 14)) * 0.1).astype(np.float32)
    return tuple(torch.from_numpy(a).to(device) for a in (pos, neg, tgt))


def cpu_model_string():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def pin_rank_to_cores(local_rank, local_world):
    """contiguous slice `local_rank` of the cores this process may run on (contiguous ids share a socket / NUMA node on
    the usual enumeration) -> the list it was pinned to, or None when there is nothing to split / it is switched off"""
    if os.environ.get("VN_BENCH_NO_AFFINITY") == "1" or not hasattr(os, "sched_setaffinity") or local_world < 2:
        return None
    try:
        cores = sorted(os.sched_getaffinity(0))
        per = len(cores) // local_world
        if per < 1:
            return None
        mine = cores[local_rank * per:(local_rank + 1) * per]
        os.sched_setaffinity(0, mine)
        return mine
    except OSError:
        return None


def cpu_baseline(frames_np, cls, T, threads):
    """oracle (kind 'port') on the host cores: B=2 fwd+bwd after one warm step (the metric's shape) and B=1 forward only
    (BASELINE configs[0]); C voxelizer + PyTorch-CPU restatement of the reference's op sequence."""
    from oracle import torch_ref as tr
    from oracle import voxelize as ov
    from voxelnet_amd.config import grid_config
    torch.set_num_threads(threads)
    grid = grid_config(cls, T=T)
    hw = (grid.H // grid.block1_stride, grid.W // grid.block1_stride)
    nb = min(2, len(frames_np))
    rng = np.random.default_rng(1)
    dp = torch.from_numpy((rng.standard_normal((nb, 2) + hw) * 1e-2).astype(np.float32))
    dr = torch.from_numpy((rng.standard_normal((nb, 14) + hw) * 1e-2).astype(np.float32))

    def step(nfr, backward):
        vs = [ov.voxelize(frames_np[b], cls, T=T) for b in range(nfr)]
        f, _, c = ov.prepare_voxel(vs)
        feats, coords = [torch.from_numpy(x) for x in f], [torch.from_numpy(x) for x in c]
        sd = tr.make_state_dict(cls)
        if backward:
            tr.forward_backward(feats, coords, sd, grid.dims, cls, dp[:nfr], dr[:nfr])
        else:
            with torch.no_grad():
                tr.middle_rpn(tr.feature_net(feats, coords, sd, grid.dims, True), sd, cls, True)

    def timed(nfr, backward, n):
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            step(nfr, backward)
            ts.append(time.perf_counter() - t0)
        return sorted(ts)

    step(nb, True)                       # warm (allocator, thread pool, oneDNN primitive caches)
    fb = timed(nb, True, 3)              # three timed steps, the MEDIAN is reported
    t_fb = fb[1]
    step(1, False)
    f1 = timed(1, False, 3)
    t_f = f1[1]
    return {"value": nb / t_fb, "unit": "point-clouds/s", "cores": threads, "kind": "port", "cpu": cpu_model_string(),
            "value_min": nb / fb[-1], "value_max": nb / fb[0],
            "sample": "1 warm + 3 timed steps (median; %.1f / %.1f / %.1f s), batch=%d fwd+bwd: C voxelizer + PyTorch-CPU "
                      "restatement of the reference op sequence; also batch=1 forward only (BASELINE configs[0]), median of 3: "
                      "%.2f point-clouds/s (%.1f s)" % (fb[0], fb[1], fb[2], nb, 1.0 / t_f, t_f),
            "fwd_only_b1_value": 1.0 / t_f}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="car", choices=sorted(CONFIGS),
                    help="car = BASELINE configs[1] (the metric's workload, default), ped = configs[2], dense = configs[4]")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "fp32x3"])
    ap.add_argument("--batch", type=int, default=0, help="frames per GPU (default: the config's: 2, 2, 4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--no-parity-mode", action="store_true", help="skip the short fp32 (parity mode) throughput run")
    ap.add_argument("--torch-optim", action="store_true",
                    help="torch's clip_grad_norm_ + SGD instead of the fused vn_clip_sgd tail (same arithmetic)")
    ap.add_argument("--timer-steps", type=int, default=3)
    ap.add_argument("--separate-calls", action="store_true",
                    help="issue the step as forward / loss.backward() / optimizer.step() (rounds 1-5) instead of ONE library "
                         "call (RPN3D.train_step -> vn_net_step): same kernels, same results, more host work")
    ap.add_argument("--force-reducer", action="store_true",
                    help="diagnostic: run the DDP bucket path (flat buckets, bucket events) on one GPU")
    ap.add_argument("--static-voxels", action="store_true",
                    help="diagnostic: voxelize once, outside the timed steps (NOT the benchmark configuration)")
    ap.add_argument("--precomputed-targets", action="store_true",
                    help="seeded target maps instead of generating the RPN targets from label lines inside every step")
    ap.add_argument("--windows", type=int, default=5, help="timed windows of --steps steps each (the first is the headline)")
    ap.add_argument("--direct-rccl", action="store_true",
                    help="N > 1: all-reduce through the library's own RCCL wrapper (vn_allreduce_bucket) instead of torch.distributed")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # N > 1: each rank's host threads on a slice of the cores of their own, set in-process BEFORE the first GPU call (no
    # taskset / numactl wrapper: a process that has touched the GPU must not exec).  One step costs ~2 ms of one core's
    # time per rank against ~3.6 ms on the GPU; eight ranks migrating over the same cores is the first thing that would
    # show in the per-rank enqueue times below.  VN_BENCH_NO_AFFINITY=1 leaves the scheduler alone.
    affinity = pin_rank_to_cores(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))) if world > 1 else None
    assert torch.cuda.is_available(), "bench.py needs an MI355X (there is no CPU path)"
    # rehearsal aid (NOT a benchmark configuration): VN_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and uses gloo, so
    # the N>1 code path (broadcast, bucketed all-reduce overlapped with the backward, MAX over ranks) can be
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
    # rehearsal aid (NOT a benchmark configuration): VN_BENCH_FORCE_DIST=1 on a one-GPU box creates a ONE-rank RCCL process
    # group and makes the reducer issue its collectives anyway (VN_FORCE_COLLECTIVE): torch.distributed's nccl backend,
    # all_reduce(async_op) from the communication stream and the waits run on hardware before an N > 1 run exists
    force_dist = world == 1 and os.environ.get("VN_BENCH_FORCE_DIST") == "1"
    if force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29547")
        os.environ["VN_FORCE_COLLECTIVE"] = "1"
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    if args.direct_rccl and world > 1 and not share and os.environ.get("VN_DIRECT_RCCL_UNSAFE") != "1":
        # the library's own RCCL entry (vn_allreduce_bucket) has only ever run on a ONE-rank communicator (this pool hands
        # out one GPU per call): it must not be what an 8-GPU measurement silently runs through
        raise SystemExit("--direct-rccl with N > 1 has never run on two devices: set VN_DIRECT_RCCL_UNSAFE=1 to try it "
                         "(the default path is torch.distributed over the same RCCL)")

    from voxelnet_amd import _lib
    from voxelnet_amd import engine as E
    from voxelnet_amd import model as M
    from voxelnet_amd import parallel, synth
    from voxelnet_amd.config import GRADIENT_CLIP, LR, grid_config
    from voxelnet_amd.optim import ClipSGD
    from voxelnet_amd.voxelize import VoxelBatch, VoxelBuffers, pipeline_stream, voxelize_device_async

    cfg_index, workload_id, cls, T, default_batch, cfg_desc = CONFIGS[args.config]
    B = args.batch or default_batch
    grid = grid_config(cls, T=T)
    with_loss = cls == "Car"

    def build_model(precision):
        M.set_precision(precision)
        torch.manual_seed(1234)                      # same initial weights on every rank
        m = M.RPN3D(cls).to(dev)
        m.train(True)                                # train.py:148
        return m

    model = build_model(args.precision)
    named = list(model.named_parameters())
    params = [p for _, p in named]
    # train.py:130 + 153-154: clip_grad_norm_(5) + SGD(lr=0.01) as two HIP launches (csrc/optim.hip);
    # --torch-optim runs torch's own clip_grad_norm_ + SGD instead (same arithmetic, ~12 launches)
    opt = torch.optim.SGD(params, lr=LR) if args.torch_optim else ClipSGD(params, LR, GRADIENT_CLIP)
    if world > 1:
        for p in params:
            dist.broadcast(p.data, 0)
    if world > 1 or args.force_reducer or force_dist:
        model.grad_reducer = parallel.GradAllReducer(named, direct_rccl=True if args.direct_rccl else None)

    frames_np = synth.workload_frames(workload_id, batch=B, frame0=rank * B)   # weak scaling: own frames per rank
    frames = [torch.from_numpy(f).to(dev) for f in frames_np]                  # resident in HBM before timing
    hf, wf = grid.H // grid.block1_stride, grid.W // grid.block1_stride

    def make_targets():
        if with_loss:
            return synthetic_targets(B, hf, wf, 99 + rank, dev), None
        rng = np.random.default_rng(4100 + rank)     # seeded N(0,1) * 1e-3 upstream gradient
        return None, (torch.from_numpy((rng.standard_normal((B, 2, hf, wf)) * 1e-3).astype(np.float32)).to(dev),
                      torch.from_numpy((rng.standard_normal((B, 14, hf, wf)) * 1e-3).astype(np.float32)).to(dev))
    targets, upstream = make_targets()
    # label lines of this rank's frames (six boxes of the class + a DontCare line each): the default step turns them into
    # targets on the device (voxelnet_amd/targets.py -> vn_rpn_targets), as model.py:309 does inside RPN3D.forward
    labels = np.empty(B, dtype=object)
    for b in range(B):
        labels[b] = synth.synth_labels(cls, 6, seed=7000 + rank * B + b)
    gen_targets = with_loss and not args.precomputed_targets
    mode = {"gen": gen_targets}

    # Voxelization is software-pipelined one step ahead on its own HIP stream (the input-pipeline stage of the step): the
    # K read-back that sizes its outputs (utils.py:69-71 returns (K,T,7)/(K,3)/(K,) arrays) never stalls the training
    # queue (capacity-sized outputs, K read by the gather kernel from device memory, asynchronous K copy to pinned memory:
    # voxelize_device_async).  Every timed step still runs one voxelization of its B frames (for the next step) and one
    # train step (on the buffers voxelized during the previous one).
    vox_stream = pipeline_stream(dev)      # (the input pipeline's stream, shared with the target generator: 4 streams in all with a reducer)
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
        # the model's first act is to concatenate the per-frame tensors (RPN3D.detect): done here on the voxelizer's stream,
        # one step ahead like the voxelization itself (voxelize.VoxelBatch; DeviceCollate does the same)
        feats = VoxelBatch.ahead(feats, vox_stream, torch.float32)
        coords = VoxelBatch.ahead(coords, vox_stream, torch.int64)
        launch_voxelize()
        if args.static_voxels:
            torch.cuda.synchronize()
            pending["static"] = (feats, coords)
        return feats, coords

    state = {"model": model, "opt": opt, "params": params, "named": named}

    def fwd_bwd(feats, coords):
        m = state["model"]
        if with_loss:
            if mode["gen"]:
                out = m((None, labels, feats, None, coords, None, None), dev)          # targets from the labels, on the device
            else:
                out = m((None, None, feats, None, coords, None, None), dev, targets=targets)
            out[2].backward()                                              # train.py:151
            return out[2]
        prob, reg = m.detect(feats, coords)
        torch.autograd.backward([prob, reg], list(upstream))
        return prob.flatten()[0]

    gap_events = [] if os.environ.get("VN_BENCH_GAPS") else None   # diagnostic: idle time of the training stream between steps

    host_delay = float(os.environ.get("VN_BENCH_HOST_DELAY_US", "0")) * 1e-6   # diagnostic: a busy-wait per step on the host —
    #                                    does the step time follow (the host is on the critical path) or not (the GPU is)?

    def step_eager():
        if host_delay > 0:
            t_end = time.perf_counter() + host_delay
            while time.perf_counter() < t_end:
                pass
        if gap_events is not None:
            g0 = torch.cuda.Event(enable_timing=True)
            g0.record()
        feats, coords = voxelize_batch()
        one_call = (with_loss and not args.separate_calls and not args.torch_optim and state.get("exposed") is None)
        if one_call:
            # model.py:298-362 + train.py:151-154 as ONE library call (reducer exchange and ClipSGD included)
            m = state["model"]
            out = m.train_step((None, labels if mode["gen"] else None, feats, None, coords, None, None), dev, state["opt"],
                               targets=None if mode["gen"] else targets)
            loss = out[2]
        else:
            loss = fwd_bwd(feats, coords)
        if not args.static_voxels:         # this step's slot may be re-used once the backward (queued above) is done
            si = ring["use"] % 3
            ring["use"] += 1
            ev = slot_free[si] if slot_free[si] is not None else torch.cuda.Event()
            ev.record()
            slot_free[si] = ev
        m = state["model"]
        if one_call:
            state["opt"].zero_grad(set_to_none=True)                       # train.py:155
            if gap_events is not None:
                g1 = torch.cuda.Event(enable_timing=True)
                g1.record()
                gap_events.append((g0, g1))
            return loss
        if m.grad_reducer is not None:
            if state.get("exposed") is not None:       # (N > 1 diagnostic steps only: how long the main stream waits for the reducer)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                m.grad_reducer.finish(state["named"])
                e1.record()
                state["exposed"].append((e0, e1))
            else:
                m.grad_reducer.finish(state["named"])
        if args.torch_optim:
            torch.nn.utils.clip_grad_norm_(state["params"], GRADIENT_CLIP)  # train.py:153
        state["opt"].step()                                                # train.py:154 (ClipSGD: both lines)
        state["opt"].zero_grad(set_to_none=True)                           # train.py:155
        if gap_events is not None:
            g1 = torch.cuda.Event(enable_timing=True)
            g1.record()
            gap_events.append((g0, g1))
        return loss

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(max(1, args.warmup)):
        loss = step_eager()
    sync_all()
    t0 = time.perf_counter()
    step_events = []
    for _ in range(args.steps):
        loss = step_eager()
        if os.environ.get("VN_BENCH_STEP_TIMES"):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            step_events.append(ev)
    t_enq = time.perf_counter() - t0       # host time to enqueue the K steps (the GPU may still be running)
    sync_all()
    dt = time.perf_counter() - t0
    if gap_events is not None and rank == 0:
        ge = gap_events[-args.steps:]
        gaps = sorted(ge[i][1].elapsed_time(ge[i + 1][0]) * 1e3 for i in range(len(ge) - 1))
        inside = sorted(a.elapsed_time(b) * 1e3 for a, b in ge)
        print("VN_BENCH_GAPS: training-stream time between the last launch of a step and the first of the next: median %.1f us, "
              "p90 %.1f, max %.1f; first-to-last launch of a step: median %.1f us"
              % (gaps[len(gaps) // 2], gaps[int(len(gaps) * 0.9)], gaps[-1], inside[len(inside) // 2]), file=sys.stderr)
    # ---- the spread: the same timed region again, --windows - 1 times (every rank: a step holds the collectives)
    window_dts = [dt]
    for _ in range(max(0, args.windows - 1)):
        sync_all()
        tw = time.perf_counter()
        for _ in range(args.steps):
            step_eager()
        sync_all()
        window_dts.append(time.perf_counter() - tw)
    dt_pre = None
    if gen_targets:                      # the same window with precomputed target maps (what rounds 1-2 timed)
        mode["gen"] = False
        for _ in range(2):
            step_eager()
        sync_all()
        tw = time.perf_counter()
        for _ in range(args.steps):
            step_eager()
        sync_all()
        dt_pre = time.perf_counter() - tw
        mode["gen"] = True
    if step_events and rank == 0:
        print("[bench] per-step ms:", " ".join(f"{a.elapsed_time(b):.2f}" for a, b in zip(step_events, step_events[1:])),
              file=sys.stderr)
    assert torch.isfinite(loss).item(), "non-finite loss"
    ranks_in_sync, grad_checksums_equal = None, None
    per_rank = None
    if world > 1:
        # what each rank saw, before the MAX: the slowest rank sets `value`; the spread says whether one rank (host-bound,
        # a slow device) holds the others in the collectives
        mine = torch.tensor([1e3 * dt / args.steps, 1e3 * t_enq / args.steps], dtype=torch.float64, device=dev)
        lo_, hi_ = mine.clone(), mine.clone()
        dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
        per_rank = {"ms_per_step_min": float(lo_[0]), "ms_per_step_max": float(hi_[0]),
                    "host_enqueue_in_loop_ms_per_step_min": float(lo_[1]), "host_enqueue_in_loop_ms_per_step_max": float(hi_[1])}
        t = torch.tensor([dt, dt_pre or 0.0] + window_dts, dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0].item())
        dt_pre = float(t[1].item()) if dt_pre is not None else None
        window_dts = [float(v) for v in t[2:].tolist()]
        # self-check of the data-parallel path (outside the timed region): every rank started from rank 0's weights
        # and applied the same averaged gradients, so the parameters must still be bit-identical on all ranks ...
        with torch.no_grad():
            cs = torch.stack([torch.stack([p.double().sum(), p.double().abs().sum()]) for p in params]).sum(0)
        lo, hi = cs.clone(), cs.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        ranks_in_sync = bool(torch.equal(lo, hi))
        # ... and so must the averaged gradients of the last step (they are still in the reducer's buckets)
        gcs = torch.tensor([model.grad_reducer.checksum()], dtype=torch.float64, device=dev)
        glo, ghi = gcs.clone(), gcs.clone()
        dist.all_reduce(glo, op=dist.ReduceOp.MIN)
        dist.all_reduce(ghi, op=dist.ReduceOp.MAX)
        grad_checksums_equal = bool(torch.equal(glo, ghi))
        if not (ranks_in_sync and grad_checksums_equal) and rank == 0:
            print("[bench] WARNING: parameter / gradient checksums differ between ranks", file=sys.stderr)

    # ---- host cost of one step, un-throttled: the enqueue of ONE step into an EMPTY queue (the timed loop's enqueue time is
    #      paced by the GPU: the host runs into the queue's back-pressure), median of 7
    #      (every rank runs them: a step holds the gradient collectives)
    samples = []
    for _ in range(7):
        torch.cuda.synchronize()
        th = time.perf_counter()
        step_eager()
        samples.append(time.perf_counter() - th)
    torch.cuda.synchronize()
    host_free = 1e3 * sorted(samples)[len(samples) // 2]
    if world > 1:           # the line carries the SLOWEST rank's host cost (and the fastest beside it)
        hf_ = torch.tensor([host_free], dtype=torch.float64, device=dev)
        hlo, hhi = hf_.clone(), hf_.clone()
        dist.all_reduce(hlo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hhi, op=dist.ReduceOp.MAX)
        per_rank["host_enqueue_ms_per_step_min"], per_rank["host_enqueue_ms_per_step_max"] = float(hlo.item()), float(hhi.item())
        host_free = float(hhi.item())
    sync_all()

    # ---- N > 1: how long the main stream waits for the gradient exchange at the end of a step (the part of the bucketed
    #      all-reduce that the backward did not hide), HIP events around GradAllReducer.finish on the training stream
    exposed_ms = None
    if model.grad_reducer is not None:
        state["exposed"] = []
        for _ in range(10):
            step_eager()
        torch.cuda.synchronize()
        ex = sorted(a.elapsed_time(b) for a, b in state["exposed"])
        state["exposed"] = None
        t = torch.tensor([ex[len(ex) // 2]], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        exposed_ms = float(t.item())
        sync_all()

    # ---- per-kernel durations: the native executor's own HIP events around every launch (same path as the timed region),
    #      torch events around the launch groups the Python side issues (voxelizer, VFE, loss, optimizer)
    recs, sect = None, None
    native_timed = bool(model.native_executor)
    if not args.no_kernel_timer and (rank == 0 or world > 1):
        sect = E.KernelTimer()
        E.SECTIONS = sect
        recs = []
        buf = (_lib.VnTimingRecord * 4096)()
        n_rec = ctypes.c_int32(0)
        for _ in range(args.timer_steps):
            if native_timed:
                _lib.call("vn_net_timing_begin", model._net_handle(dev), 4096)
            step_eager()
            torch.cuda.synchronize()
            if native_timed:
                _lib.call("vn_net_timing_read", model._net_handle(dev), buf, 4096, ctypes.byref(n_rec))
                recs += [(r.kind, r.layer, r.ms, r.flops, r.bytes) for r in buf[:n_rec.value]]
        sync_all()
        E.SECTIONS = None

    # ---- the same step in the fp32 parity mode (the mode the <= 1e-3 parity tests run in) and in bf16x3, a few steps each,
    #      and how far the three modes' RPN maps are apart on this batch (same weights: the seed of build_model)
    parity = None
    if rank == 0 and world == 1 and args.precision == "bf16" and not args.no_parity_mode:
        feats0, coords0 = voxelize_batch()
        maps = {}

        def maps_of(m_):
            with torch.no_grad():
                pr, rg = m_.detect(feats0, coords0)
            torch.cuda.synchronize()
            return pr.double(), rg.double()
        maps["bf16"] = maps_of(build_model("bf16"))
        parity = {}
        for prec, nsteps, note in (("fp32", 5, "fp32 operands on v_mfma_f32_16x16x4_f32: the mode of the <= 1e-3 parity tests"),
                                   ("fp32x3", 5, "fp32-sized storage, every conv / weight-gradient product as three bf16 MFMAs on hi / lo "
                                                 "splits (VN_F32X3); round 5: activations and gradients are STORED split by the BatchNorm "
                                                 "passes (VN_F32X3S), no split work in the kernels; native executor: the fast mode INSIDE "
                                                 "the 1e-3 map tolerance")):
            pm = build_model(prec)
            maps[prec] = maps_of(pm)
            state.update(model=pm, params=list(pm.parameters()), named=list(pm.named_parameters()),
                         opt=ClipSGD(list(pm.parameters()), LR, GRADIENT_CLIP))
            for _ in range(2):
                step_eager()
            torch.cuda.synchronize()
            tp = time.perf_counter()
            for _ in range(nsteps):
                step_eager()
            torch.cuda.synchronize()
            tpe = time.perf_counter() - tp
            parity[prec] = {"dtype": {"fp32": "f32"}.get(prec, prec), "value": B * nsteps / tpe, "unit": "point-clouds/s",
                            "ms_per_step": 1e3 * tpe / nsteps, "note": note}
            state.update(model=model, params=params, named=named, opt=opt)
            del pm
        M.set_precision(args.precision)
        for prec in ("bf16", "fp32x3"):
            (pa, ra), (pb, rb) = maps[prec], maps["fp32"]
            err = {"prob_max_over_max": float((pa - pb).abs().max() / pb.abs().max()), "prob_rel_l2": float((pa - pb).norm() / pb.norm()),
                   "reg_max_over_max": float((ra - rb).abs().max() / rb.abs().max()), "reg_rel_l2": float((ra - rb).norm() / rb.norm())}
            (parity[prec] if prec in parity else parity.setdefault("bf16", {}))["map_error_vs_fp32"] = err
        parity["value"], parity["dtype"], parity["ms_per_step"] = parity["fp32"]["value"], "f32", parity["fp32"]["ms_per_step"]

    if rank == 0:
        value = world * B * args.steps / dt
        # fp32x3 evaluates every algorithmic (fp32) product as THREE bf16 MFMA products: the roof for algorithmic FLOPs
        # is a third of the dense bf16 MFMA peak
        peak = {"fp32": PEAK_F32_MATRIX_TFLOPS, "fp32x3": PEAK_BF16_DENSE_TFLOPS / 3.0}.get(
            args.precision, PEAK_BF16_DENSE_TFLOPS)
        metric = "point-clouds/sec fwd+bwd, KITTI car voxel grid, batch=2"
        if args.config != "car":
            metric = "point-clouds/sec fwd+bwd, %s, batch=%d" % (cfg_desc, B)
        vals = sorted(world * B * args.steps / w for w in window_dts)
        res = {
            "metric": metric,
            "value": value, "unit": "point-clouds/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            # the spread over --windows timed regions of --steps steps in this run (the first one is `value`)
            "value_min": vals[0], "value_median": vals[len(vals) // 2], "value_max": vals[-1], "windows": len(vals),
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"bf16": "bf16", "fp32": "f32", "fp32x3": "f32 storage / bf16x3 products"}[args.precision], "data": "synthetic",
            "config": {"workload": "%s, batch=%d per GPU, fwd+bwd train step (BASELINE configs[%d])" % (cfg_desc, B, cfg_index),
                       "global_batch": world * B, "points_per_frame": int(frames_np[0].shape[0]),
                       "parallelism": "dp%d" % world,
                       "step": "voxelize+VFE+scatter+Conv3d+RPN fwd, %s, bwd, clip_grad_norm, SGD"
                               % (("RPN targets from the label lines (device) + loss" if gen_targets else "loss on precomputed target maps")
                                  if with_loss else "seeded upstream gradient (the reference's loss is undefined for this class)"),
                       "with_target_generation": bool(gen_targets),
                       "launch_mode": "eager" + ("+native-executor" if model.native_executor else "")
                                      + ("+one-call-step" if (with_loss and not args.separate_calls and not args.torch_optim
                                                              and model._step_fused_ok(args.precision, opt)) else "")},
            # un-throttled: one step enqueued into an empty queue (what the host needs); in_loop: the timed loop's enqueue
            # time, which the GPU paces through queue back-pressure
            "host_enqueue_ms_per_step": host_free, "host_enqueue_in_loop_ms_per_step": 1e3 * t_enq / args.steps,
            # dense-equivalent model FLOPs (the first Conv3d's skipped zeros NOT subtracted) over the whole step
            "model_flops_fraction_of_peak": value / world * FLOP_PER_PC[args.config] / (peak * 1e12),
        }
        res["library"] = _lib.load().vn_build_info().decode()       # build + every VN_* tuning override in effect
        if per_rank is not None:
            per_rank["cpu_affinity_rank0"] = affinity if affinity is None else "%d cores: %d-%d" % (len(affinity), affinity[0], affinity[-1])
            res["per_rank"] = per_rank          # host_enqueue_ms_per_step above is the MAX over ranks
        if dt_pre is not None:
            res["value_precomputed_targets"] = world * B * args.steps / dt_pre
        if model.grad_reducer is not None:
            red = model.grad_reducer
            res["allreduce"] = {"exposed_ms_per_step": exposed_ms, "bucket_bytes": [int(b["flat"].numel() * 4) for b in red.buckets],
                                "path": "vn_allreduce_bucket (library RCCL wrapper)" if red.comm is not None else "torch.distributed (%s)" % (
                                    dist.get_backend() if world > 1 else ("nccl, ONE-rank group, collectives forced: rehearsal" if force_dist else "world 1: no collective")),
                                "collective_calls_per_step": red.comm_calls,   # VN_COMM_CALLS (parallel.MERGE_PLANS): buckets merged into fewer calls
                                "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default (4)"),
                                "comm_stream": red.comm_stream_kind,      # VN_COMM_STREAM=private | pipeline (default): parallel.py
                                "rccl_version_bound_by_library": int(_lib.load().vn_comm_rccl_version()),
                                "torch_nccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else None,
                                "note": "exposed = median time the training stream waits in GradAllReducer.finish (HIP events, 10 steps, "
                                        "MAX over ranks): the part of the bucketed all-reduce the backward did not hide"}
        if ranks_in_sync is not None:
            res["ranks_in_sync"] = ranks_in_sync       # parameters bit-identical on all ranks after the timed steps
            res["grad_checksums_equal"] = grad_checksums_equal
        if recs is not None:
            ns = float(args.timer_steps)
            fam = {}
            for kind, layer, ms, fl, by in recs:
                n, t, f, b = fam.get(kind, (0, 0.0, 0.0, 0.0))
                fam[kind] = (n + 1, t + ms, f + fl, b + by)
            kern = {}
            for kind, (n, t, f, b) in sorted(fam.items()):
                e = {"launches_per_step": n / ns, "ms_per_step": t / ns, "avg_launch_us": 1e3 * t / n}
                if f > 0:
                    e["gflop_per_step"] = f / ns / 1e9
                    e["achieved_tflops"] = f / (t * 1e-3) / 1e12
                    e["frac_of_mfma_peak"] = e["achieved_tflops"] / peak
                if b > 0:
                    e["algorithmic_mb_per_step"] = b / ns / 1e6
                    e["achieved_gbs"] = b / (t * 1e-3) / 1e9
                    e["frac_of_hbm_peak"] = e["achieved_gbs"] / PEAK_HBM_GBS
                kern[KIND_NAMES[kind]] = e
            for name, (n, by, ms) in sect.summary().items():
                extra = 0.0
                if name == "voxelize":               # outputs: 28*K*T + 40*K per frame (K known on the host by now)
                    extra = sum((28.0 * T + 40.0) * s.k_host.item() for s in slots[0]) * (n / len(frames))
                kern[name] = {"launches_per_step": n / ns, "ms_per_step": ms / ns, "avg_launch_us": 1e3 * ms / n,
                              "algorithmic_mb_per_step": (by + extra) / ns / 1e6,
                              "achieved_gbs": (by + extra) / (ms * 1e-3) / 1e9,
                              "frac_of_hbm_peak": (by + extra) / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS}
            res["kernels"] = kern
            if os.environ.get("VN_BENCH_LAYER_TIMES"):       # diagnostic: per (family, layer) mean duration
                per = {}
                for kind, layer, ms, fl, by in recs:
                    n, t = per.get((kind, layer), (0, 0.0))
                    per[(kind, layer)] = (n + 1, t + ms)
                for (kind, layer), (n, t) in sorted(per.items()):
                    print("[bench] %-18s layer %3d  x%.1f/step  %8.1f us each" % (KIND_NAMES[kind], layer, n / ns, 1e3 * t / n),
                          file=sys.stderr)
            if fam:
                # dominant family: the implicit-GEMM convolutions (forward + data gradient share the two kernels)
                n = fam.get(0, (0, 0, 0, 0))[0] + fam.get(1, (0, 0, 0, 0))[0]
                t = fam.get(0, (0, 0, 0, 0))[1] + fam.get(1, (0, 0, 0, 0))[1]
                f = fam.get(0, (0, 0, 0, 0))[2] + fam.get(1, (0, 0, 0, 0))[2]
                ach = f / (t * 1e-3) / 1e12
                # HBM-side bytes per launch of that family: PMC counters cannot be read in-process; taken from the committed
                # rocprofv3 --pmc passes of this same command (profiles/, tools/pmc_family.py), car / bf16 / batch 2 only
                traffic, traffic_wgrad, traffic_source = None, None, None
                pmcs = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_traffic.json")) \
                    if os.path.isdir(os.path.join(ROOT, "profiles")) else []
                if args.config == "car" and args.precision == "bf16" and B == 2 and pmcs:
                    with open(os.path.join(ROOT, "profiles", pmcs[-1])) as fh:         # the latest round's passes
                        pj = json.load(fh)
                    # the counter passes name the library build they profiled (tools/pmc_family.py: vn_build_id); a figure
                    # from another build is not this build's traffic: null, with the reason
                    running = _lib.load().vn_build_id().decode()
                    if pj.get("library_build_id") == running:
                        traffic = pj.get("traffic_bytes_per_launch")
                        traffic_wgrad = (pj.get("wgrad") or {}).get("traffic_bytes_per_launch")
                        traffic_source = ("committed file profiles/%s (rocprofv3 --pmc FETCH_SIZE x2 / WRITE_SIZE passes of this command "
                                          "on library build %s = the running one, tools/final_run.sh + tools/pmc_family.py): counters "
                                          "cannot be collected inside this process, so the figure is NOT measured in this run"
                                          % (pmcs[-1], running))
                    else:
                        traffic_source = ("null: profiles/%s was collected on library build %s, this run is build %s — re-run "
                                          "tools/final_run.sh's PMC passes for this build" % (pmcs[-1], pj.get("library_build_id"), running))
                res["roofline"] = {
                    "kernel": "k_conv_patch + k_gather_gemm (implicit-GEMM convolutions: forward + data gradient)",
                    "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                    "traffic": traffic, "traffic_source": traffic_source, "avg_launch_us": 1e3 * t / n, "launches_per_step": n / ns,
                    "gflop_per_step": f / ns / 1e9, "ms_per_step": t / ns,
                    "peak_note": {"fp32x3": "2500 / 3 TFLOP/s: three bf16 MFMA products per algorithmic fp32 product",
                                  "fp32": "v_mfma_f32_16x16x4_f32 dense peak"}.get(args.precision, "dense bf16 MFMA peak"),
                    "note": "algorithmic FLOPs (SURVEY.md 8d; the launches that skip constant data — the rulebook first layer, "
                            "the row-list data gradients at its active sites — with the FLOPs they execute) / summed "
                            "HIP-event time of every launch of the family, events on the launch's own stream inside the "
                            "native executor, %d steps on the same inputs right after the timed region" % args.timer_steps}
                # the other MFMA family: the weight gradients (side stream)
                if 2 in fam:
                    nw, tw_, fw_, _ = fam[2]
                    res["roofline_wgrad"] = {"kernel": "k_wgrad / k_wgrad_patch (weight gradients)", "bound": "mfma",
                                             "achieved": fw_ / (tw_ * 1e-3) / 1e12, "peak": peak, "unit": "TFLOP/s",
                                             "frac": fw_ / (tw_ * 1e-3) / 1e12 / peak, "launches_per_step": nw / ns, "ms_per_step": tw_ / ns,
                                             "traffic": traffic_wgrad, "traffic_source": traffic_source}
                # FLOPs the MFMA pipes actually execute in a step (first layer: rulebook, not dense-equivalent)
                fx = sum(fam.get(k, (0, 0, 0, 0))[2] for k in (0, 1, 2)) / ns
                res["executed_mfma_flops_fraction_of_peak"] = fx / (1e-3 * res["ms_per_step"]) / (peak * 1e12)
                res["roofline_hbm"] = [dict(family=k, **{kk: v[kk] for kk in ("ms_per_step", "algorithmic_mb_per_step",
                                                                              "achieved_gbs", "frac_of_hbm_peak")})
                                       for k, v in kern.items() if "achieved_gbs" in v and "achieved_tflops" not in v]
        if parity is not None:
            res["parity_mode"] = parity
        if world == 1 and not args.no_cpu_baseline:
            try:
                ncpu = len(os.sched_getaffinity(0))
            except AttributeError:
                ncpu = os.cpu_count() or 1
            res["cpu_baseline"] = cpu_baseline(frames_np, cls, T, max(1, min(ncpu, 16)))   # a 1-GPU box's CPU share is 16
        print(json.dumps(res))
    if model.grad_reducer is not None:
        model.grad_reducer.close()          # (the direct path's RCCL communicator)
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
