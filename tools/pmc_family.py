"""HBM-side traffic of the implicit-GEMM convolution family (k_conv_patch + k_gather_gemm, the launches bench.py's
KernelTimer labels "k_gather_gemm") from two rocprofv3 --pmc passes -> profiles/<name>.json, read by bench.py for the
roofline object's `traffic`.  FETCH_SIZE is doubled (gfx950: 128-B requests tallied at 64 B; MI355X_MICROARCH.md)."""
import csv, glob, json, sys, collections
def load(d, counter):
    f = (glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"))[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return rows
def family(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if n.startswith("k_conv_patch") or (n.startswith("k_gather_gemm") ):
        return "conv"
    if n.startswith("k_wgrad") and not n.startswith("k_wgrad_const") and not n.startswith("k_wgrad_reduce"):
        return "wgrad"
    if n.startswith("k_unpack_wgrads"):
        return "unpack"
    return None
fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
steps = int(sys.argv[3])
def per_step(rows, fam="conv"):
    # last `steps` steps: delimit by k_vfe_rows launches
    idx = [i for i, r in enumerate(rows) if "k_vfe_rows" in r["Kernel_Name"]]
    sel = rows[idx[-steps - 1]:idx[-1]] if len(idx) > steps else rows
    tot, n = 0.0, 0
    for r in sel:
        if family(r["Kernel_Name"]) == fam:
            tot += float(r["Counter_Value"]); n += 1
    return tot / steps, n / steps
f_kb, nf = per_step(fe)
w_kb, nw = per_step(wr)
others = {}
for fam in ("wgrad", "unpack"):      # the weight-gradient family and its unpack (no traffic figure for them)
    fk, n1 = per_step(fe, fam)
    wk, _ = per_step(wr, fam)
    if n1 > 0:
        others[fam] = {"launches_per_step": n1, "fetch_bytes_per_step": 2 * fk * 1024, "write_bytes_per_step": wk * 1024,
                       "traffic_bytes_per_launch": (2 * fk + wk) * 1024 / n1}
out = {"family": "k_conv_patch + k_gather_gemm (+ the two row-list launches)", "launches_per_step": nf,
       "fetch_bytes_per_step": 2 * f_kb * 1024, "write_bytes_per_step": w_kb * 1024,
       "traffic_bytes_per_launch": (2 * f_kb + w_kb) * 1024 / nf,
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python3 bench.py --steps 3 --warmup 2 "
                 "--no-cpu-baseline --no-kernel-timer`; FETCH_SIZE x2 (gfx950 correction)"}
out.update(others)
# which library build these passes profiled (bench.py only reports `traffic` when it is the build that is running)
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "voxelnet-pytorch_amd")]
from voxelnet_amd import _lib
out["library_build_id"] = _lib.load().vn_build_id().decode()
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out, indent=1))
