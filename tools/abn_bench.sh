#!/bin/bash
# Interleaved N-way comparison of bench.py inside ONE gpurun call (box-to-box variance on this pool is 6-12 %):
#   tools/abn_bench.sh <tag> <rounds> "<env 0>" "<env 1>" ... [-- extra bench.py args]
# an env of "-" means no override.  Prints value / ms_per_step of every run; JSON lines -> gpurun_out/<tag>_<k>_<i>.json
tag=$1; rounds=$2; shift 2
envs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done
[ "$1" = "--" ] && shift
mkdir -p gpurun_out
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
