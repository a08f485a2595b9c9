"""Summarise a rocprofv3 kernel_trace.csv: per-kernel ms/step over the last N steps (steps found by k_vfe_rows)."""
import csv, sys, collections
path, nsteps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_vfe_rows' in r['Kernel_Name']]
start = idx[-nsteps - 1]; end = idx[-1]
sel = rows[start:end]
span = (int(sel[-1]['End_Timestamp']) - int(sel[0]['Start_Timestamp'])) / 1e6 / nsteps
agg = collections.defaultdict(lambda: [0, 0.0])
for r in sel:
    n = r['Kernel_Name']
    n = n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    if 'at::native' in n: n = 'torch:' + n.split('at::native::')[1][:40]
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    agg[n][0] += 1; agg[n][1] += d
busy = sum(v[1] for v in agg.values()) / nsteps
print(f"steps {nsteps}: span {span:.2f} ms/step, busy {busy:.2f} ms/step, launches/step {len(sel)/nsteps:.0f}")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:32]:
    print(f"  {t/nsteps:7.3f} ms  {c/nsteps:6.1f}x  {n[:90]}")
