"""Per-kernel HBM-side traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KB units).
gfx950: FETCH_SIZE counts 128-B requests as 64 B -> doubled (MI355X_MICROARCH.md, HBM section)."""
import csv, sys, collections, glob
def load(d, counter):
    f = (glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"))[0]
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter: continue
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if "at::native" in n: n = "torch:" + n.split("at::native::")[1][:36]
        out[(n, r["Grid_Size"])].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return out
fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
rows = []
for k, v in fe.items():
    w = wr.get(k, [(0, 0)])
    f_mb = 2 * sum(x[0] for x in v) / len(v) / 1024
    w_mb = sum(x[0] for x in w) / len(w) / 1024
    us = sum(x[1] for x in v) / len(v) / 1e3
    rows.append((len(v) / steps * us, k, len(v) / steps, f_mb, w_mb, us))
rows.sort(reverse=True)
tot_f = sum(r[3] * r[2] for r in rows); tot_w = sum(r[4] * r[2] for r in rows)
print(f"per step: fetch {tot_f:.0f} MB, write {tot_w:.0f} MB")
print(f"{'kernel':44s} {'grid':>9s} {'n/step':>6s} {'fetch MB':>9s} {'write MB':>9s} {'us':>7s} {'TB/s':>6s}")
for t, k, n, f, w, us in rows[:60]:
    print(f"{k[0][:44]:44s} {k[1]:>9s} {n:6.1f} {f:9.1f} {w:9.1f} {us:7.1f} {(f+w)/us:6.2f}")
