"""GPU: bench.py's output contract.  The driver parses ONE JSON line of `python bench.py` (train.py:145-155 is the loop it
times; BASELINE.json names the metric and the workload); a refactor of bench.py must not silently drop a key the driver or
the judge reads.  conftest.py starts `bench.py --steps 2 --warmup 1 --force-reducer` (N = 1 with the data-parallel bucket
path attached) as its own process before this process touches the GPU; here its line is parsed."""
import json
import os

import pytest

from conftest import BENCH_RUN

pytestmark = pytest.mark.gpu


def test_bench_line_carries_the_keys_the_driver_reads():
    if "proc" not in BENCH_RUN:
        pytest.skip("bench.py was not started (not a `-m gpu` session, or VN_NO_DDP_REHEARSAL=1)")
    p = BENCH_RUN["proc"]
    try:
        rc = p.wait(timeout=900)
    except Exception:  # noqa: BLE001
        p.kill()
        rc = -9
    err = open(os.path.join(BENCH_RUN["dir"], "bench.err")).read()[-3000:]
    out = open(os.path.join(BENCH_RUN["dir"], "bench.json")).read().strip().splitlines()
    assert rc == 0 and out, f"bench.py exit {rc}\n{err}"
    d = json.loads(out[-1])
    # the driver's contract (one line, these keys)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["metric"].startswith("point-clouds/sec fwd+bwd, KITTI car voxel grid, batch=2")
    assert "BASELINE configs[1]" in d["config"]["workload"] and "model" not in d["config"]
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["dtype"] == "bf16" and d["data"] == "synthetic" and d["vs_baseline"] is None and d["value"] > 0
    # the tier's two objects
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and 0.0 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic"] is None or r["traffic"] > 0
    assert "traffic_source" in r                                   # says where `traffic` comes from, or why it is null
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0
    # the data-parallel path at world size 1 (--force-reducer): the fields the first SCALE run will be read by
    a = d["allreduce"]
    assert a["exposed_ms_per_step"] is not None and a["exposed_ms_per_step"] >= 0.0
    assert len(a["bucket_bytes"]) == 5 and sum(a["bucket_bytes"]) == 4 * 6809392
    assert a["comm_stream"] in ("private",) or a["comm_stream"].startswith("pipeline")
    assert "library" in d and "build" in d["library"] and "host_enqueue_ms_per_step" in d
