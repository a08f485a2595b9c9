#!/bin/bash
# Interleaved A/B of bench.py inside ONE gpurun call (box-to-box variance on this pool is 6-12 %):
#   tools/ab_bench.sh <tag> <rounds> "<env of A>" "<env of B>" [extra bench.py args]
# prints value / ms_per_step of every run; the JSON lines go to gpurun_out/<tag>_{A,B}<i>.json
tag=$1; rounds=$2; envA=$3; envB=$4; shift 4
mkdir -p gpurun_out
for i in $(seq 1 "$rounds"); do
  for v in A B; do
    if [ $v = A ]; then e=$envA; else e=$envB; fi
    env $e timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-parity-mode --no-kernel-timer "$@" \
        > gpurun_out/${tag}_$v$i.json 2> gpurun_out/${tag}_$v$i.err || { echo "run $v$i failed"; tail -3 gpurun_out/${tag}_$v$i.err; exit 1; }
    python - "$v$i" "$e" gpurun_out/${tag}_$v$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
print(f"{sys.argv[1]} [{sys.argv[2]}] {d['value']:.1f} pc/s  {d['ms_per_step']:.3f} ms/step", flush=True)
PY
  done
done
