#!/bin/bash
# the working tree's library against the previous commit's (tools/ubench/bin/libprev.so, built from `git archive HEAD`):
# a test subset under the new library, then interleaved step A/Bs.  usage: exp_ab_prev.sh "<pytest args>" <tag> <rounds> [configs...]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PREV=$GRAFT_REPO_ROOT/tools/ubench/bin/libprev.so
tests=$1; tag=$2; rounds=$3; shift 3
timeout -k 10 600 python -m pytest $tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1 || { tail -30 gpurun_out/${tag}_tests.log; exit 1; }
tail -2 gpurun_out/${tag}_tests.log
for cfg in "$@"; do
  if [ "$cfg" = car ]; then extra=""; else extra="--config $cfg"; fi
  bash tools/abn_bench.sh ${tag}_$cfg $rounds "VN_LIB_PATH=$PREV" "-" -- $extra || exit 1
done
# VFE alone (both builds), when asked for: VFE_ALONE=1
if [ -n "$VFE_ALONE" ]; then
  for i in 1 2 3; do for w in car dense; do
    echo "prev $w: $(VN_LIB_PATH=$PREV timeout -k 10 120 python tools/bench_vfe.py $w 2>&1 | grep ' ms' | tr '\n' ' ')"
    echo "new  $w: $(timeout -k 10 120 python tools/bench_vfe.py $w 2>&1 | grep ' ms\|rows:' | tr '\n' ' ')"
  done; done | tee gpurun_out/${tag}_vfe_alone.txt
fi
