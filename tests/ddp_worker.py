"""One rank of the data-parallel rehearsal on a ONE-GPU box (tests/test_gpu_multiproc.py): two fresh processes share
cuda:0, talk over gloo, and run two train steps through the path the 8-GPU run takes — native executor, single-call
backward with per-group events, GradAllReducer.launch_bucket on the comm stream (voxelnet_amd/model.py,
voxelnet_amd/parallel.py) — with the collective swapped from RCCL to gloo (RCCL refuses two ranks on one device).

Checks, written to <outdir>/rank<r>.txt ("ok" or the traceback):
  (a) the averaged gradients of step 1 equal the mean of the two ranks' local gradients, recomputed here in ONE process
      without any reducer (each rank knows both ranks' inputs);
  (b) after two steps with the fused clip + SGD tail the parameters of both ranks are bit-identical (SHA-256)."""
import hashlib
import os
import sys
import traceback
from dataclasses import replace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "voxelnet-pytorch_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def rank_inputs(rank, dev):
    """two frames of the tiny-grid fixture per rank: rank 1 sees a different subset of the voxels than rank 0"""
    import numpy as np
    import torch
    g = np.load(os.path.join(ROOT, "tests", "golden", "middle_tiny_car.npz"))
    feats = torch.from_numpy(g["features"])
    coords = torch.from_numpy(g["coords"])
    lens = [int(x) for x in g["feat_lens"]]
    fs, cs = list(torch.split(feats, lens)), list(torch.split(coords, lens))
    if rank == 1:
        fs = [f[::2].contiguous() for f in fs]
        cs = [c[::2].contiguous() for c in cs]
    t = np.load(os.path.join(ROOT, "tests", "golden", "rpn3d_tiny.npz"))
    targets = tuple(torch.from_numpy(np.roll(t[k], rank, axis=1)).to(dev) for k in ("pos", "neg", "targets"))
    return [f.to(dev) for f in fs], [c.to(dev) for c in cs], targets


def main():
    rank, world, port, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    out = os.path.join(outdir, f"rank{rank}.txt")
    try:
        import torch
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        from oracle import torch_ref as tr
        from voxelnet_amd import model as M
        from voxelnet_amd import parallel
        from voxelnet_amd.optim import ClipSGD
        M.set_precision("bf16")

        def build():
            m = M.RPN3D("Car")
            m.load_state_dict(tr.make_state_dict("Car"))
            m.feature_net._grid = replace(m.feature_net._grid, H=16, W=24)
            return m.to(dev).train()

        def local_grads(r):
            m = build()
            f, c, tg = rank_inputs(r, dev)
            res = m((None, None, f, None, c, None, None), dev, targets=tg)
            res[2].backward()
            torch.cuda.synchronize()
            return {k: p.grad.detach().clone() for k, p in m.named_parameters()}

        expect = None
        for r in range(world):
            g = local_grads(r)
            expect = g if expect is None else {k: expect[k] + g[k] for k in g}
        expect = {k: v / world for k, v in expect.items()}

        m = build()
        named = list(m.named_parameters())
        params = [p for _, p in named]
        m.grad_reducer = parallel.GradAllReducer(named)
        assert m.grad_reducer.comm_stream is not None and m.overlap_wgrad and m.native_executor
        calls = []
        orig = m.grad_reducer.launch_bucket
        m.grad_reducer.launch_bucket = lambda bi, **kw: (calls.append(bi), orig(bi, **kw))[1]
        opt = ClipSGD(params, 0.01, 5.0)
        f, c, tg = rank_inputs(rank, dev)
        for step in range(2):
            res = m((None, None, f, None, c, None, None), dev, targets=tg)
            res[2].backward()
            m.grad_reducer.finish(named)
            torch.cuda.synchronize()
            if step == 0:
                assert calls == [0, 1, 2, 3, 4], calls          # the native-executor bucket path, not grad_ready
                for k, p in named:
                    e = expect[k]
                    tol = 1e-5 * float(e.abs().max()) + 1e-12
                    assert torch.allclose(p.grad, e, rtol=1e-4, atol=tol), (k, float((p.grad - e).abs().max()), tol)
            opt.step()
            opt.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        h = hashlib.sha256()
        for _, p in named:
            h.update(p.detach().cpu().numpy().tobytes())
        digests = [None] * world
        dist.all_gather_object(digests, h.hexdigest())
        assert len(set(digests)) == 1, digests
        cs = m.grad_reducer.checksum() if hasattr(m.grad_reducer, "checksum") else 0.0
        dist.barrier()
        dist.destroy_process_group()
        with open(out, "w") as fh:
            fh.write(f"ok {digests[0]} {cs}\n")
    except Exception:  # noqa: BLE001
        with open(out, "w") as fh:
            fh.write(traceback.format_exc())
        raise


if __name__ == "__main__":
    main()
