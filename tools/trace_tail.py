"""End of one train step from a rocprofv3 kernel trace (tools/profile_mode.sh): start, duration and hardware queue of
every kernel of the last `window_us` microseconds of the second-to-last complete step — which stream the step waits for.
usage: python tools/trace_tail.py <kernel_trace.csv> [window_us]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) if len(sys.argv) > 2 else 2000.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_opt_update" in r["Kernel_Name"]]
b, e = idx[-3], idx[-2]
t0 = int(rows[b]["End_Timestamp"])
span = (int(rows[e]["End_Timestamp"]) - t0) / 1e3
print(f"step span {span:.1f} us (under the profiler)")
for r in rows[b + 1:e + 1]:
    s = (int(r["Start_Timestamp"]) - t0) / 1e3
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if s > span - win:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        print(f"{s:9.1f} {d:8.1f} q{r['Queue_Id']} {name[:70]}")
