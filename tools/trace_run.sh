cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $O/tr1
rocprofv3 --kernel-trace -d $O/tr1 -o s --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-parity-mode --no-kernel-timer > $O/tr1.log 2>&1
ls $O/tr1
