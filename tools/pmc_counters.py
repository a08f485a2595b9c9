"""Per-kernel table of arbitrary SQ counters from one or more rocprofv3 --pmc passes (each pass in its own directory),
for the kernels whose name contains a given substring — e.g. what bounds the VFE kernels:

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d gpurun_out/pmc_vfe_a ... bench.py
    rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY -d gpurun_out/pmc_vfe_b ... bench.py
    python tools/pmc_counters.py k_vfe 3 gpurun_out/pmc_vfe_a gpurun_out/pmc_vfe_b > profiles/r03_pmc_vfe.txt

Counters are summed over the dispatches of the last <steps> steps (delimited by k_vfe_rows, one per step) and printed per
kernel and per step, with the ratios that say what a wave spends its cycles on:
  valu_busy  = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES   (x4: SQ_ACTIVE_INST_* count quad-cycles on gfx9)
  lds_wait   = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES
  any_wait   = SQ_WAIT_ANY / SQ_WAVE_CYCLES
(The SQ_* cycle counters are per-wave sums over the chip; ratios between them do not depend on the sampling.)"""
import collections
import csv
import glob
import sys

pat, steps, dirs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
counts = collections.defaultdict(int)
names = []
for d in dirs:
    f = (glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"))[0]
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        e = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "t0": int(r["Start_Timestamp"])})
        e[r["Counter_Name"]] = float(r["Counter_Value"])
        if r["Counter_Name"] not in names:
            names.append(r["Counter_Name"])
    order = sorted(disp.values(), key=lambda e: e["t0"])
    marks = [i for i, e in enumerate(order) if "k_vfe_rows" in e["name"]]
    sel = order[marks[-steps - 1]:marks[-1]] if len(marks) > steps else order
    first = d == dirs[0]
    for e in sel:
        if pat not in e["name"]:
            continue
        k = e["name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if first:
            counts[k] += 1
        for n in names:
            if n in e:
                agg[k][n] += e[n]
print(f"kernels matching '{pat}', last {steps} steps; counter sums per step")
hdr = f"{'kernel':28s} {'n/step':>6s} " + " ".join(f"{n[3:][:16]:>16s}" for n in names) + "   valu_busy lds_wait any_wait"
print(hdr)
for k, a in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0.0)):
    wc = a.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    line = f"{k[:28]:28s} {counts[k] / steps:6.1f} " + " ".join(f"{a.get(n, 0.0) / steps:16.4g}" for n in names)
    line += f"   {4.0 * a.get('SQ_ACTIVE_INST_VALU', 0.0) / wc:9.3f} {a.get('SQ_WAIT_INST_LDS', 0.0) / wc:8.3f} {a.get('SQ_WAIT_ANY', 0.0) / wc:8.3f}"
    print(line)
