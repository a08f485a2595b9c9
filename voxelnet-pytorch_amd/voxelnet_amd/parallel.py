"""Data parallelism for the train step: one process per GPU, torch.distributed (backend
"nccl" == RCCL on ROCm, over xGMI), gradients averaged with bucketed all-reduces that are
launched on a side stream WHILE the backward is still running.

The reference has no distributed code at all; the partition is the natural
one — each rank trains on its own point clouds, BatchNorm statistics stay per replica (as in
stock DDP), and the only exchange is the all-reduce of the 6,809,392 gradient elements
(27.2 MB fp32) once per step.

Bucketing follows the backward's execution order (heads/deconv3/block3 first, the Conv3d
stack and the VFE last): `layer grads ready` -> copied into the flat bucket -> when the bucket
is complete an event is recorded on the compute stream, the comm stream waits for it and
starts the all-reduce.  xGMI is point-to-point (7 links/GPU), so a handful of multi-MB buckets
is the right granularity: each all-reduce is large enough to be bandwidth-bound and the last,
tiny bucket (the VFE parameters, whose gradients finish last) is what is left exposed.
"""
import ctypes
import os

import torch
import torch.distributed as dist

# backward order of the parameter groups (net.middle_backward + VFE at the very end)
BUCKET_PLAN = [
    ["heads", "deconv3", "block3"],                 # ~17.2 MB fp32
    ["deconv2", "block2", "deconv1"],               # ~5.3 MB
    ["block1"],                                     # ~3.0 MB
    ["middle_layer"],                               # ~1.8 MB: launched when middle_layer.0's gradients are final — beside the VFE backward
    ["vfe"],                                        # 9.6 KB: the only bucket that is exposed (round 3: it used to ride with middle_layer,
    #                                                 which kept that 1.8 MB all-reduce waiting for the VFE backward, ~135 us)
]


# Which buckets travel in ONE collective call (round 5).  A torch.distributed all_reduce costs ~75 us of host time per call
# (ProcessGroupNCCL bookkeeping, two stream hops, the scaling launch) — measured with a one-rank RCCL group on one MI355X:
# five calls per step take the step from 549.8 to 496.6 point-clouds/s (`profiles/r05_ab_one_rank_rccl.md`).  The buckets are
# slices of ONE flat buffer, so consecutive ones can be reduced by a single call when the LAST of them is ready:
#   "2" (default): [heads .. block1] = 25.4 MB when block1's gradients are final (the three Conv3d layers' backward, ~1 ms,
#                  still runs beside it) + [middle_layer, vfe] = 1.8 MB at the end of the VFE backward (the exposed tail);
#   "5": every bucket by itself (round 2-4's schedule).
# torch.distributed collectives: "0" (default on the nccl backend) = synchronous op issued from the communication stream,
# "1" = async_op on ProcessGroupNCCL's internal stream (rounds 2-4).  One-rank RCCL rehearsal on one MI355X
# (profiles/r05_ab_collective_path.md): async -8.75 %, sync -7.73 % against the same step without a collective.
_COMM_ASYNC = os.environ.get("VN_COMM_ASYNC")
MERGE_PLANS = {"5": [(0,), (1,), (2,), (3,), (4,)], "2": [(0, 1, 2), (3, 4)], "1": [(0, 1, 2, 3, 4)]}


_HEADS_W = ["middle_rpn.prob_conv.conv.weight", "middle_rpn.reg_conv.conv.weight"]
_HEADS_B = ["middle_rpn.prob_conv.conv.bias", "middle_rpn.reg_conv.conv.bias"]


def group_of(param_name):
    """state_dict key -> bucket group name"""
    if param_name.startswith("feature_net."):
        return "vfe"
    n = param_name.split(".")[1]
    if n in ("prob_conv", "reg_conv"):
        return "heads"
    return n


class GradAllReducer:
    """Flat-bucket gradient averaging.  Works with any backend (gloo on CPU in the tests)."""

    def __init__(self, named_params, process_group=None, plan=BUCKET_PLAN, use_side_stream=True, direct_rccl=None):
        """direct_rccl: all-reduce through the library's own RCCL wrapper (vn_allreduce_bucket on a communicator made by
        vn_comm_create; csrc/comm.hip) instead of torch.distributed's all_reduce.  Default: the environment variable
        VN_DIRECT_RCCL=1; torch.distributed (backend "nccl" == RCCL) otherwise.  Same collective either way."""
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        named_params = list(named_params)
        self.buckets = []          # dicts: flat, views{name: tensor}, pending(set), handle
        self.where = {}            # param name -> bucket index
        # every bucket is a slice of ONE flat buffer (bucket starts aligned to 64 elements), so that consecutive buckets can
        # travel in one collective call (MERGE_PLANS)
        sizes = []
        for groups in plan:
            members = [(n, p) for n, p in named_params if group_of(n) in groups]
            sizes.append(sum(p.numel() for _, p in members))
        starts, tot = [], 0
        for sz in sizes:
            starts.append(tot)
            tot += (sz + 63) // 64 * 64
        dev0, dt0 = named_params[0][1].device, named_params[0][1].dtype
        self.flat_all = torch.zeros(max(tot, 1), dtype=dt0, device=dev0)
        self._span = []            # per bucket: (start, end) in flat_all
        self.fused_views = {}      # model.HEADS_W / HEADS_B -> the (16,768) / (16,) views over both heads' gradients
        for bi, groups in enumerate(plan):
            members = [(n, p) for n, p in named_params if group_of(n) in groups]
            if not members:
                continue
            total = sum(p.numel() for _, p in members)
            flat = self.flat_all[starts[bi]:starts[bi] + total]
            self._span.append((starts[bi], starts[bi] + total))
            # the two heads' weights (and biases) back to back at the bucket's end: the executor's fused (16,768) heads
            # gradient then IS the two parameters' gradients (model._grad_views does the same in its own flat buffer)
            names_here = [n for n, _ in members]
            if all(h in names_here for h in _HEADS_W + _HEADS_B):
                members = [(n, p) for n, p in members if n not in _HEADS_W + _HEADS_B] + \
                          [(n, dict(members)[n]) for n in _HEADS_W + _HEADS_B]
            views, off, start = {}, 0, {}
            for n, p in members:
                views[n] = flat[off:off + p.numel()].view_as(p)
                start[n] = off
                off += p.numel()
                self.where[n] = len(self.buckets)
            if all(h in start for h in _HEADS_W + _HEADS_B) and views[_HEADS_W[0]].numel() + views[_HEADS_W[1]].numel() == 16 * 768:
                self.fused_views["__heads_weight_fused__"] = flat[start[_HEADS_W[0]]:start[_HEADS_W[0]] + 16 * 768].view(16, 768, 1, 1)
                self.fused_views["__heads_bias_fused__"] = flat[start[_HEADS_B[0]]:start[_HEADS_B[0]] + 16]
            self.buckets.append({"flat": flat, "views": views, "pending": set(), "handle": None, "names": list(views),
                                 "index": len(self.buckets)})
        missing = [n for n, _ in named_params if n not in self.where]
        assert not missing, f"parameters without a bucket: {missing[:3]}"
        self.cuda = self.buckets[0]["flat"].is_cuda
        merge = os.environ.get("VN_COMM_CALLS", "2")
        if merge not in MERGE_PLANS or plan is not BUCKET_PLAN or len(self.buckets) != len(BUCKET_PLAN):
            merge = "5" if len(self.buckets) == len(BUCKET_PLAN) else None
        self.merge_plan = MERGE_PLANS[merge] if merge else [(i,) for i in range(len(self.buckets))]
        self.comm_calls = len(self.merge_plan)
        self._merge_of = {bi: g for g in self.merge_plan for bi in g}
        self._ready = set()        # buckets of this step whose gradients are final (their merge group waits for all members)
        # The collectives are issued from the input pipeline's stream (voxelnet_amd.voxelize.pipeline_stream), not from a
        # stream of their own: torch.distributed runs an NCCL collective on ITS internal stream, ordered behind the stream it
        # is called from — with a private communication stream the process had five busy streams (training, executor side,
        # pipeline, this one, torch's) on the HIP runtime's four hardware queues, i.e. two of them serialised against each
        # other.  The pipeline stream is idle while the backward runs (the next batch was voxelized at
        # the start of the step), so waiting for the bucket events there delays nothing.
        # UNTESTED with real RCCL kernels on that queue (no N > 1 hardware run exists yet): with the
        # collectives of step i on the pipeline stream, batch i+1's copies / crop / voxelizer / target generation queue behind
        # them.  VN_COMM_STREAM=private gives the reducer a stream of its own again (the five-stream arrangement) so that the
        # first multi-GPU run can compare the two; `comm_stream_kind` says which one this reducer uses (bench.py prints it).
        self.comm_stream = None
        self.comm_stream_kind = None
        if self.cuda and use_side_stream:
            if os.environ.get("VN_COMM_STREAM", "pipeline") == "private":
                self.comm_stream = torch.cuda.Stream(device=self.buckets[0]["flat"].device)
                self.comm_stream_kind = "private"
            else:
                from .voxelize import pipeline_stream
                self.comm_stream = pipeline_stream(self.buckets[0]["flat"].device)
                self.comm_stream_kind = "pipeline (shared with the input pipeline and the target generator)"
        self.defer_allreduce = False   # True: grad_ready only fills the buckets (HIP-graph capture); allreduce_all() later
        # rehearsal aid (bench.py VN_BENCH_FORCE_DIST=1 on a ONE-GPU box): issue the collectives even at world size 1, so that
        # process-group creation over RCCL, all_reduce(async_op) from the communication stream and handle.wait() run on real
        # hardware before the first multi-GPU run does
        self.force_collective = os.environ.get("VN_FORCE_COLLECTIVE") == "1" and dist.is_initialized()
        self.comm = None               # ncclComm_t of the direct path
        if direct_rccl is None:
            direct_rccl = os.environ.get("VN_DIRECT_RCCL") == "1"
        if direct_rccl:
            if self.world > 1 and os.environ.get("VN_DIRECT_RCCL_UNSAFE") != "1":
                # vn_allreduce_bucket has only ever run on a one-rank communicator (the build pool hands out one GPU per
                # call): refuse to be the path of a multi-GPU run until someone asks for it by name
                from . import _lib
                raise _lib.VoxelnetHipError("GradAllReducer(direct_rccl=True) with world size %d: the library's own RCCL entry has "
                                            "never run on two devices — set VN_DIRECT_RCCL_UNSAFE=1 to try it; the default "
                                            "(torch.distributed, backend nccl = the same RCCL) needs nothing" % self.world)
            if self.cuda:
                self._init_direct()
        self.reset()

    def _init_direct(self):
        """one RCCL communicator over the ranks of the process group: rank 0's unique id travels through
        torch.distributed's object broadcast (any backend), then every rank calls vn_comm_create on its device"""
        from . import _lib
        ident = (ctypes.c_ubyte * 128)()
        if self.rank == 0:
            _lib.call("vn_comm_unique_id", ident)
        if self.world > 1:
            box = [bytes(ident)]
            dist.broadcast_object_list(box, src=0, group=self.pg)
            ident = (ctypes.c_ubyte * 128).from_buffer_copy(box[0])
        h = ctypes.c_void_p()
        with _lib.on_device(self.buckets[0]["flat"].device):
            _lib.call("vn_comm_create", ctypes.byref(h), ident, self.world, self.rank)
        self.comm = h

    def close(self):
        """destroy the direct path's communicator (idempotent; also run by __del__ and by bench.py at exit)"""
        if self.comm is not None:
            from . import _lib
            torch.cuda.synchronize()
            _lib.load().vn_comm_destroy(self.comm)
            self.comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001 - interpreter shutdown
            pass

    def _bucket_ready(self, bi):
        """bucket bi's gradients are final: the collective of its merge group runs when the group is complete -> the flat
        slice to reduce now, or None"""
        self._ready.add(bi)
        g = self._merge_of[bi]
        if not all(m in self._ready for m in g):
            return None
        return self.flat_all[self._span[g[0]][0]:self._span[g[-1]][1]]

    def _reduce(self, flat):
        """mean over the ranks of a flat gradient slice, in place, on the CURRENT stream; returns a work handle or None"""
        if self.comm is not None:
            from . import _lib
            _lib.call("vn_allreduce_bucket", self.comm, flat.data_ptr(), flat.numel(), 1.0 / self.world,
                      _lib.raw_stream())
            return None
        # async_op=False (VN_COMM_ASYNC=0): torch >= 2.7 runs a synchronous NCCL collective ON THE CALLER'S STREAM — here the
        # communication stream — instead of on ProcessGroupNCCL's internal stream behind two event hops: one busy HIP
        # stream less in the process (the runtime maps streams onto four hardware queues)
        async_op = (_COMM_ASYNC != "0") if _COMM_ASYNC is not None else dist.get_backend(self.pg) != "nccl"
        if dist.get_backend(self.pg) == "nccl":      # RCCL averages itself (ncclAvg): no scaling launch in front of the collective
            return dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.pg, async_op=async_op)
        flat.div_(self.world)
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.pg, async_op=async_op)

    def reset(self):
        for b in self.buckets:
            b["pending"] = set(b["names"])
            b["handle"] = None
        self._ready = set()

    def grad_ready(self, name, grad):
        """called as soon as a parameter's gradient exists (in backward order)"""
        b = self.buckets[self.where[name]]
        if grad.data_ptr() != b["views"][name].data_ptr():   # the kernels may already have written into the bucket
            b["views"][name].copy_(grad)
        b["pending"].discard(name)
        if not b["pending"]:
            self._launch(b)

    def _launch(self, b):
        if (self.world == 1 and self.comm is None and not self.force_collective) or self.defer_allreduce:
            return
        flat = self._bucket_ready(b["index"])
        if self.comm_stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.comm_stream.wait_event(ev)          # (every member of a merge group puts its wait on the comm stream)
            if flat is not None:
                with torch.cuda.stream(self.comm_stream):
                    b["handle"] = self._reduce(flat)
        elif flat is not None:
            b["handle"] = self._reduce(flat)

    def launch_bucket(self, bi, wait_fn=None, after_event=None, prelude=None):
        """All-reduce bucket bi on the comm stream once (a) wait_fn(comm_stream) has made the comm stream wait for whatever
        produces the bucket's gradients (e.g. vn_net_wait_bucket: the native executor's per-group events), (b) the
        optional torch event has passed and (c) prelude() — small copies into the bucket, issued on the comm stream —
        has run.  The calling (compute) stream is never blocked."""
        b = self.buckets[bi]
        b["pending"] = set()
        st = self.comm_stream
        live = (self.world > 1 or self.comm is not None or self.force_collective) and not self.defer_allreduce
        if st is None:                       # CPU tensors / no side stream: plain, in order
            if prelude is not None:
                prelude()
            flat = self._bucket_ready(bi) if live else None
            if flat is not None:
                b["handle"] = self._reduce(flat)
            return
        with torch.cuda.stream(st):
            if after_event is not None:
                st.wait_event(after_event)
            if wait_fn is not None:
                wait_fn(st)
            if prelude is not None:
                prelude()
            flat = self._bucket_ready(bi) if live else None
            if flat is not None:
                b["handle"] = self._reduce(flat)

    def allreduce_all(self):
        """deferred mode: all-reduce every (already filled) bucket now, largest first, and wait"""
        if self.world == 1 and self.comm is None:
            return
        hs = [self._reduce(self.flat_all[self._span[g[0]][0]:self._span[g[-1]][1]]) for g in self.merge_plan]
        for h in hs:
            if h is not None:
                h.wait()

    def finish(self, named_params, launch_deferred=False):
        """wait for every bucket and point .grad of each parameter at its averaged view"""
        for b in self.buckets:
            assert not b["pending"], f"bucket never completed: {sorted(b['pending'])[:3]}"
            if b["handle"] is not None:
                b["handle"].wait()
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        for n, p in named_params:
            p.grad = self.buckets[self.where[n]]["views"][n]
        self.reset()

    def checksum(self):
        """sum of all averaged gradients (identical on every rank after finish())"""
        return float(sum(b["flat"].double().sum().item() for b in self.buckets))
