"""Per-kernel MFMA-pipe utilisation from one rocprofv3 --pmc pass
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE ...
(north_star: "rocprof ... MFMA utilisation against chip peak"; SURVEY.md 8d: report MFMA-busy next to the
dense-equivalent rate, because the first Conv3d's zero work is skipped).

  SQ_VALU_MFMA_BUSY_CYCLES   cycles a SIMD's matrix pipe is busy, summed over the chip's 1024 SIMDs
  GRBM_GUI_ACTIVE            GPU-active cycles, summed over the 8 XCDs (MI355X_MICROARCH.md: divide by 8)
  SQ_INSTS_VALU_MFMA_MOPS_*  executed MFMA math operations / 512
=> busy fraction of a kernel = MFMA_BUSY / (1024 * GUI_ACTIVE / 8); executed FLOP = MOPS * 512.
usage: python tools/pmc_mfma.py <pmc dir> <steps> [out.txt]"""
import collections, csv, glob, sys

d, steps = sys.argv[1], int(sys.argv[2])
f = (glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"))[0]
rows = list(csv.DictReader(open(f)))
# delimit the last `steps` steps by the first VFE kernel of each step (k_vfe_rows_p1 in a train step, k_vfe_rows otherwise)
disp = collections.OrderedDict()
for r in rows:
    key = int(r["Dispatch_Id"])
    e = disp.setdefault(key, {"name": r["Kernel_Name"], "t0": int(r["Start_Timestamp"]), "t1": int(r["End_Timestamp"])})
    e[r["Counter_Name"]] = float(r["Counter_Value"])
order = sorted(disp.values(), key=lambda e: e["t0"])
marks = [i for i, e in enumerate(order) if "k_vfe_rows" in e["name"]]
sel = order[marks[-steps - 1]:marks[-1]] if len(marks) > steps else order


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    return "torch:" + n.split("at::native::")[1][:40] if "at::native" in n else n


agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0, 0.0])
for e in sel:
    a = agg[short(e["name"])]
    a[0] += 1
    a[1] += e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    a[2] += e.get("GRBM_GUI_ACTIVE", 0.0)
    a[3] += (e.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) + e.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0)) * 512.0
    a[4] += (e["t1"] - e["t0"]) / 1e3
lines = []
tot_busy = sum(a[1] for a in agg.values())
tot_act = sum(a[2] for a in agg.values())
tot_flop = sum(a[3] for a in agg.values())
lines.append(f"last {steps} steps of the profiled run; per step: executed MFMA FLOP {tot_flop / steps / 1e9:.1f} GFLOP, "
             f"MFMA-busy SIMD-cycles {tot_busy / steps:.3e}, GPU-active cycles (per XCD) {tot_act / 8 / steps:.3e}")
lines.append(f"whole-step MFMA busy fraction while the GPU is active: {tot_busy / (1024 * tot_act / 8):.3f}")
lines.append(f"{'kernel':46s} {'n/step':>6s} {'us/step':>8s} {'MFMA busy':>9s} {'GFLOP/step':>10s} {'TFLOP/s':>8s}")
for name, a in sorted(agg.items(), key=lambda kv: -kv[1][4]):
    if a[2] <= 0:
        continue
    busy = a[1] / (1024 * a[2] / 8)
    tf = a[3] / (a[4] * 1e-6) / 1e12 if a[4] > 0 else 0.0
    lines.append(f"{name[:46]:46s} {a[0] / steps:6.1f} {a[4] / steps:8.1f} {busy:9.3f} {a[3] / steps / 1e9:10.1f} {tf:8.0f}")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write(out + "\n")
