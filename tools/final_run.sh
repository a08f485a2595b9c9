# round-end measurement set (run on the GPU box through gpurun): tests, smoke, bench (three configs), rocprofv3 kernel
# stats of the SAME command, the two HBM-traffic PMC passes and the MFMA-busy PMC pass (each --pmc pass on its own, never
# combined with tracing).  Summaries are copied to profiles/ by hand afterwards (gpurun_out/ is scratch).
R=${R:-r05}
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/ -m gpu -x -q > gpurun_out/final_tests.log 2>&1
tail -3 gpurun_out/final_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1
tail -1 gpurun_out/final_smoke.log
# HBM traffic of THIS build first (two PMC passes of their own, never combined with tracing): bench.py only reports
# `roofline.traffic` from a profiles/*_pmc_traffic.json that names the running library build (vn_build_id), so the passes
# run before the bench and their summary goes to profiles/ on the box as well as to gpurun_out/ (which travels back)
( cd /tmp && export TMPDIR=/tmp && O=$GRAFT_REPO_ROOT/gpurun_out && \
  rocprofv3 --pmc FETCH_SIZE -d $O/fpmc_fetch -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-parity-mode --no-kernel-timer --windows 1 > $O/fpmc_fetch.log 2>&1 && \
  rocprofv3 --pmc WRITE_SIZE -d $O/fpmc_write -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-parity-mode --no-kernel-timer --windows 1 > $O/fpmc_write.log 2>&1 )
python tools/pmc_traffic.py gpurun_out/fpmc_fetch gpurun_out/fpmc_write 3 > gpurun_out/${R}_pmc_traffic_per_kernel.txt
python tools/pmc_family.py gpurun_out/fpmc_fetch gpurun_out/fpmc_write 3 gpurun_out/${R}_pmc_traffic.json > /dev/null
cp gpurun_out/${R}_pmc_traffic.json profiles/${R}_pmc_traffic.json
python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
cat gpurun_out/final_bench.json
python bench.py --config ped --no-cpu-baseline > gpurun_out/final_bench_ped.json 2> gpurun_out/final_bench_ped.err
python bench.py --config dense --no-cpu-baseline > gpurun_out/final_bench_dense.json 2> gpurun_out/final_bench_dense.err
# the fast mode inside the 1e-3 map tolerance, with its own roofline (peak = 2500 / 3 TFLOP/s: three bf16 products per fp32 product)
python bench.py --precision fp32x3 --no-cpu-baseline > gpurun_out/final_bench_fp32x3.json 2> gpurun_out/final_bench_fp32x3.err
# the N > 1 code path, rehearsed with two ranks on the one GPU of this box (gloo carries the collective)
VN_BENCH_SHARE_GPU=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/final_bench_n2_rehearsal.json 2> gpurun_out/final_bench_n2_rehearsal.err
tail -c 600 gpurun_out/final_bench_n2_rehearsal.json
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
rocprofv3 --kernel-trace --stats -d $O/fstats -o s --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-mode --windows 1 > $O/fstats.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -d $O/fpmc_mfma -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-parity-mode --no-kernel-timer --windows 1 > $O/fpmc_mfma.log 2>&1
# what bounds the VFE kernels: two counter passes of their own
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $O/fpmc_vfe_a -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-parity-mode --no-kernel-timer --windows 1 > $O/fpmc_vfe_a.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY -d $O/fpmc_vfe_b -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-parity-mode --no-kernel-timer --windows 1 > $O/fpmc_vfe_b.log 2>&1
# the same per-kernel view for the fp32x3 mode (its roofline object in profiles/r05_bench_fp32x3.json comes from HIP events; this is the rocprofv3 side)
rocprofv3 --kernel-trace --stats -d $O/fstats_x3 -o s --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --precision fp32x3 --steps 20 --warmup 5 --no-cpu-baseline --no-parity-mode --windows 1 > $O/fstats_x3.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_summary.py $(ls gpurun_out/fstats_x3/*/*kernel_trace.csv gpurun_out/fstats_x3/*kernel_trace.csv 2>/dev/null | head -1) 10 > gpurun_out/${R}_fp32x3_per_step.txt 2>&1 || true
cp $(ls gpurun_out/fstats_x3/*/*kernel_stats.csv gpurun_out/fstats_x3/*kernel_stats.csv 2>/dev/null | head -1) gpurun_out/${R}_fp32x3_kernel_stats.csv || true
python tools/pmc_counters.py k_vfe 3 gpurun_out/fpmc_vfe_a gpurun_out/fpmc_vfe_b > gpurun_out/${R}_pmc_vfe.txt 2>&1 || true
# (every post-processing script takes "the last 3 steps of the profiled run" and finds the step boundaries itself: the
#  profiled command also runs warm-up / window / host-enqueue steps, 17 in all — round 3's per-step header divided all of
#  them by 3)
python tools/pmc_mfma.py gpurun_out/fpmc_mfma 3 gpurun_out/${R}_pmc_mfma_per_kernel.txt | head -40
python tools/trace_summary.py $(ls gpurun_out/fstats/*/*kernel_trace.csv gpurun_out/fstats/*kernel_trace.csv 2>/dev/null | head -1) 10 > gpurun_out/${R}_bench_per_step.txt 2>&1 || true
cp $(ls gpurun_out/fstats/*/*kernel_stats.csv gpurun_out/fstats/*kernel_stats.csv 2>/dev/null | head -1) gpurun_out/${R}_bench_kernel_stats.csv || true
# two-stream timelines of one executor step (the executor's own event brackets): car and dense
python tools/step_timeline.py > gpurun_out/${R}_car_timeline.txt 2>&1 || true
python tools/step_timeline.py --dense > gpurun_out/${R}_dense_timeline.txt 2>&1 || true
ls gpurun_out/fstats gpurun_out/fpmc_fetch gpurun_out/fpmc_write gpurun_out/fpmc_mfma
