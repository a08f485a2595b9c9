# per-kernel durations of the VFE alone (tools/bench_vfe.py) under rocprofv3: tools/trace_vfe.sh <tag> [dense|random]
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
tag=$1; shift
rm -rf $O/$tag
rocprofv3 --kernel-trace --stats -d $O/$tag -o s --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_vfe.py "$@" > $O/$tag.log 2>&1
cat $O/$tag.log | grep -v "^W\|amdgpu.ids"
python3 - $O/$tag <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "vfe" in r["Name"]:
        print(f'{r["Name"].split("(")[0][-28:]:28s} calls {r["Calls"]:>4s}  avg {float(r["AverageNs"]) / 1e3:8.1f} us  min {float(r["MinNs"]) / 1e3:8.1f}')
PY
