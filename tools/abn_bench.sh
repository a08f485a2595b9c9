#!/bin/bash
# Interleaved N-way comparison of bench.py inside ONE gpurun call (box-to-box variance on this pool is 6-12 %):
#   tools/abn_bench.sh <tag> <rounds> "<env 0>" "<env 1>" ... [-- extra bench.py args]
# an env of "-" means no override; arm 0 is the baseline.  Prints value / ms_per_step of every run (JSON lines ->
# gpurun_out/<tag>_<k>_<i>.json) and, at the end, tools/ab_stats.py's table: mean +- sd per arm, the paired difference of
# every arm against arm 0 with its standard error, and a verdict ONLY at >= 2 standard errors over >= 5 rounds
# (fewer rounds or a smaller difference print "no verdict": VERDICT round 4, item 5).
tag=$1; rounds=$2; shift 2
envs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done
[ "$1" = "--" ] && shift
mkdir -p gpurun_out
rm -f gpurun_out/${tag}_*_*.json gpurun_out/${tag}_arms.txt
for k in "${!envs[@]}"; do echo "${envs[$k]}" >> gpurun_out/${tag}_arms.txt; done
[ "$rounds" -lt 5 ] && echo "note: $rounds rounds < 5 — ab_stats.py will print no verdict"
for i in $(seq 1 "$rounds"); do
  for k in "${!envs[@]}"; do
    e=${envs[$k]}; [ "$e" = "-" ] && e=""
    env $e timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-parity-mode --no-kernel-timer "$@" \
        > gpurun_out/${tag}_${k}_$i.json 2> gpurun_out/${tag}_${k}_$i.err || { echo "run $k/$i failed"; tail -3 gpurun_out/${tag}_${k}_$i.err; exit 1; }
    python - "$k/$i" "${envs[$k]}" gpurun_out/${tag}_${k}_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
print(f"{sys.argv[1]} [{sys.argv[2]}] {d['value']:.1f} pc/s  {d['ms_per_step']:.3f} ms/step", flush=True)
PY
  done
done
python tools/ab_stats.py "$tag" gpurun_out | tee gpurun_out/${tag}_stats.md
