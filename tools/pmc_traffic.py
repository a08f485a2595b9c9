"""Per-kernel HBM-side traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KB units).
gfx950: FETCH_SIZE counts 128-B requests as 64 B -> doubled (MI355X_MICROARCH.md, HBM section).

usage: python tools/pmc_traffic.py <fetch pass dir> <write pass dir> [steps]
The table covers the LAST `steps` steps of each profiled run, delimited by the step's first VFE kernel (k_vfe_rows, one per
step) — the profiled bench.py command also runs warm-up, host-enqueue and window steps, so dividing every launch of the
trace by the `--steps` argument (what this script did in round 3) over-counts the per-step columns."""
import collections
import csv
import glob
import sys


def load(d, counter, steps):
    f = (glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"))[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "k_vfe_rows" in r["Kernel_Name"]]
    if len(marks) > steps:
        rows, used = rows[marks[-steps - 1]:marks[-1]], steps
    else:                       # fewer step markers than asked for: everything, normalised by the steps that ARE there
        used = max(1, len(marks))
    out = collections.defaultdict(list)
    for r in rows:
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if "at::native" in n:
            n = "torch:" + n.split("at::native::")[1][:36]
        out[(n, r["Grid_Size"])].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return out, used


steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
(fe, nf), (wr, nw) = load(sys.argv[1], "FETCH_SIZE", steps), load(sys.argv[2], "WRITE_SIZE", steps)
rows = []
for k, v in fe.items():
    w = wr.get(k, [(0, 0)])
    f_mb = 2 * sum(x[0] for x in v) / len(v) / 1024
    w_mb = sum(x[0] for x in w) / len(w) / 1024
    us = sum(x[1] for x in v) / len(v) / 1e3
    rows.append((len(v) / nf * us, k, len(v) / nf, f_mb, w_mb, us))
rows.sort(reverse=True)
tot_f = sum(r[3] * r[2] for r in rows)
tot_w = sum(r[4] * r[2] for r in rows)
print(f"last {nf} steps of the profiled run; per step: fetch {tot_f:.0f} MB (x2-corrected), write {tot_w:.0f} MB")
print(f"{'kernel':44s} {'grid':>9s} {'n/step':>6s} {'fetch MB':>9s} {'write MB':>9s} {'us':>7s} {'TB/s':>6s}   (MB and us per launch)")
for t, k, n, f, w, us in rows[:60]:
    print(f"{k[0][:44]:44s} {k[1]:>9s} {n:6.1f} {f:9.1f} {w:9.1f} {us:7.1f} {(f+w)/us:6.2f}")
