# rocprofv3 kernel trace of bench.py in one precision mode (default fp32x3) -> gpurun_out/x3_per_step.txt (per-step kernel table)
set -e
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
rocprofv3 --kernel-trace --stats -d $O/x3stats -o s --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --precision ${1:-fp32x3} --steps 20 --warmup 5 --no-cpu-baseline --no-parity-mode --windows 1 > $O/x3stats.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_summary.py $(ls gpurun_out/x3stats/*/*kernel_trace.csv gpurun_out/x3stats/*kernel_trace.csv 2>/dev/null | head -1) 10 > gpurun_out/x3_per_step.txt 2>&1
tail -1 gpurun_out/x3stats.log | cut -c1-300
