"""Print one step of a rocprofv3 kernel_trace.csv in launch order: start offset, duration, gap to the previous kernel
on the same queue, queue id, kernel name.  usage: trace_order.py kernel_trace.csv [step-from-end]"""
import csv, sys
path = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_vfe_rows' in r['Kernel_Name']]
sel = rows[idx[-back - 1]:idx[-back]]
t0 = int(sel[0]['Start_Timestamp'])
last = {}
for r in sel:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    if 'at::native' in n: n = 'torch:' + n.split('at::native::')[1][:50]
    q = r.get('Queue_Id', '?')
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - last[q]) / 1e3 if q in last else 0.0
    last[q] = e
    print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:7.1f} us  gap {gap:7.1f}  q{q}  {n[:80]}")
