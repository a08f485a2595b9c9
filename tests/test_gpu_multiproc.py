"""GPU: rehearsal of the N > 1 data-parallel path on a one-GPU box: two fresh processes share cuda:0
over gloo and run two train steps through the native executor's bucket-event backward and GradAllReducer.launch_bucket —
the path bench.py takes at --gpus 2/4/8, with RCCL swapped for gloo.  The ranks are started by conftest.py BEFORE this
process touches the GPU (tests/ddp_worker.py holds the checks: averaged gradients == mean of the per-rank local gradients
recomputed in one process; parameters bit-identical across ranks after two clip + SGD steps)."""
import os

import pytest

from conftest import DDP_REHEARSAL

pytestmark = pytest.mark.gpu


def test_two_ranks_share_one_gpu_native_bucket_path():
    if "error" in DDP_REHEARSAL:
        pytest.fail("could not start the rank processes: " + DDP_REHEARSAL["error"])
    if "procs" not in DDP_REHEARSAL:
        pytest.skip("rank processes were not started (not a `-m gpu` session)")
    outs = []
    for r, p in enumerate(DDP_REHEARSAL["procs"]):
        try:
            rc = p.wait(timeout=900)
        except Exception:  # noqa: BLE001
            p.kill()
            rc = -9
        res = os.path.join(DDP_REHEARSAL["dir"], f"rank{r}.txt")
        text = open(res).read() if os.path.exists(res) else ""
        log = open(os.path.join(DDP_REHEARSAL["dir"], f"rank{r}.log")).read()[-3000:]
        assert rc == 0 and text.startswith("ok"), f"rank {r}: exit {rc}\n{text}\n{log}"
        outs.append(text.split())
    assert outs[0][1] == outs[1][1]          # the same parameter digest on both ranks
