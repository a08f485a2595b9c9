"""CPU, world_size 2 over gloo: the bucketed gradient averaging used for data parallelism
(voxelnet_amd/parallel.py) — bucket plan covers all 104 parameters in backward order, buckets
launch as soon as their last gradient arrives, the result equals the mean of the per-rank
gradients on every rank, and the self-check checksum agrees across ranks."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _named_params():
    from oracle import torch_ref as tr
    sd = tr.make_state_dict("Car")
    return [(k, sd[k].clone()) for k in tr.param_keys(sd)]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path[:0] = [root, os.path.join(root, "voxelnet-pytorch_amd")]
        from voxelnet_amd import parallel
        named = [(k, torch.nn.Parameter(v)) for k, v in _named_params()]
        red = parallel.GradAllReducer(named)
        assert sum(b["flat"].numel() for b in red.buckets) == 6809392
        assert len(red.buckets) == 5
        # gradients arrive in backward order (reverse of the state_dict / execution order)
        launched = []
        orig = red._launch
        red._launch = lambda b: (launched.append(len(launched)), orig(b))[1]
        g = {}
        for i, (k, p) in enumerate(reversed(named)):
            gen = torch.Generator().manual_seed(1000 * rank + i)
            g[k] = torch.randn(p.shape, generator=gen)
        order = [k for k, _ in named if parallel.group_of(k) in ("heads",)]
        order += [k for k, _ in reversed(named) if k not in order]
        for k in order:
            red.grad_ready(k, g[k])
        assert len(launched) == 5
        red.finish(named)
        # expected mean over ranks, recomputed locally from the seeds
        for i, (k, p) in enumerate(reversed(named)):
            exp = sum(torch.randn(p.shape, generator=torch.Generator().manual_seed(1000 * r + i)) for r in range(world)) / world
            assert torch.allclose(p.grad, exp, atol=1e-6), k
        cs = torch.tensor([red.checksum()], dtype=torch.float64)
        lst = [torch.zeros_like(cs) for _ in range(world)]
        dist.all_gather(lst, cs)
        assert all(abs(float(x) - float(cs)) < 1e-9 for x in lst)
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_bucket_plan_covers_every_parameter_in_backward_order():
    from voxelnet_amd import parallel
    named = [(k, torch.nn.Parameter(v)) for k, v in _named_params()]
    red = parallel.GradAllReducer(named)
    sizes = [b["flat"].numel() * 4 / 1e6 for b in red.buckets]
    assert abs(sum(sizes) - 27.24) < 0.01
    assert sizes[0] > sizes[1] > sizes[2] > sizes[3]          # heads+deconv3+block3 first and largest
    assert all(parallel.group_of(k) for k, _ in named)


def test_bench_pins_each_rank_to_its_own_slice_of_the_cores():
    """bench.py (N > 1): every rank restricts itself to a contiguous slice of the cores before its first GPU call
    (eight ranks on one host is the first multi-GPU run this code will see)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    before = sorted(os.sched_getaffinity(0))
    if len(before) < 2:
        pytest.skip("one core")
    try:
        a = bench.pin_rank_to_cores(0, 2)
        assert a == before[:len(before) // 2] and sorted(os.sched_getaffinity(0)) == a
        os.sched_setaffinity(0, before)
        b = bench.pin_rank_to_cores(1, 2)
        assert b == before[len(before) // 2:2 * (len(before) // 2)] and not set(a) & set(b)
        os.sched_setaffinity(0, before)
        assert bench.pin_rank_to_cores(0, 1) is None and sorted(os.sched_getaffinity(0)) == before     # one rank: untouched
        assert bench.pin_rank_to_cores(0, 10 * len(before)) is None                                      # fewer cores than ranks
    finally:
        os.sched_setaffinity(0, before)


def test_direct_rccl_is_refused_for_more_than_one_rank(monkeypatch):
    """the library's own RCCL wrapper has only ever run on a one-rank communicator: GradAllReducer(direct_rccl=True) must not
    become the path of a multi-GPU run silently (bench.py --direct-rccl raises the same way before any GPU work)"""
    import torch
    import torch.distributed as dist
    from voxelnet_amd import _lib, parallel
    from voxelnet_amd import model as M
    named = [(n, p) for n, p in M.RPN3D("Car").named_parameters()]
    # a fake two-rank world (no process group needed: the constructor only asks for the sizes)
    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "get_world_size", lambda group=None: 2)
    monkeypatch.setattr(dist, "get_rank", lambda group=None: 0)
    monkeypatch.delenv("VN_DIRECT_RCCL_UNSAFE", raising=False)
    monkeypatch.delenv("VN_DIRECT_RCCL", raising=False)
    with pytest.raises(_lib.VoxelnetHipError, match="never run on two devices"):
        parallel.GradAllReducer(named, direct_rccl=True)
    monkeypatch.setenv("VN_DIRECT_RCCL", "1")                  # the environment default must be refused the same way
    with pytest.raises(_lib.VoxelnetHipError, match="never run on two devices"):
        parallel.GradAllReducer(named)
    monkeypatch.delenv("VN_DIRECT_RCCL")
    red = parallel.GradAllReducer(named)                      # the default path (torch.distributed) constructs
    assert red.world == 2 and red.comm is None and len(red.buckets) == 5
    monkeypatch.setenv("VN_DIRECT_RCCL_UNSAFE", "1")           # asked for by name: allowed (CPU parameters: no communicator made)
    red = parallel.GradAllReducer(named, direct_rccl=True)
    assert red.comm is None
    # one world: nothing to refuse
    monkeypatch.setattr(dist, "get_world_size", lambda group=None: 1)
    monkeypatch.delenv("VN_DIRECT_RCCL_UNSAFE")
    parallel.GradAllReducer(named, direct_rccl=True)
