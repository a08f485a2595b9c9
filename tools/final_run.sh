# round-end measurement set (run on the GPU box through gpurun): tests, smoke, bench, rocprofv3 stats + PMC passes
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/ -m gpu -x -q > gpurun_out/final_tests.log 2>&1
tail -3 gpurun_out/final_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1
tail -1 gpurun_out/final_smoke.log
python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
cat gpurun_out/final_bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/fstats -o s --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/fstats.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/fpmc_fetch -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timer > $GRAFT_REPO_ROOT/gpurun_out/fpmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/fpmc_write -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timer > $GRAFT_REPO_ROOT/gpurun_out/fpmc_write.log 2>&1
cd $GRAFT_REPO_ROOT
ls gpurun_out/fstats gpurun_out/fpmc_fetch gpurun_out/fpmc_write
