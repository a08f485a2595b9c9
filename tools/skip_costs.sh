# What does each launch family cost INSIDE the step?  bench.py with every launch of one VN_T_* kind dropped (results wrong,
# timing right), on a -DVN_DIAG_SKIP build:
#   cd voxelnet-pytorch_amd/csrc && make OUT=../../tools/ubench/bin/libskip.so BUILD=/tmp/skip_build EXTRA=-DVN_DIAG_SKIP
#   tools/skip_costs.sh            (on the GPU box)
cd $GRAFT_REPO_ROOT
export VN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ubench/bin/libskip.so
names=(conv_fwd conv_dgrad wgrad bn_apply bn_bwd_reduce bn_bwd_apply bn_finalize unpack_wgrads pack_weights first_layer_sparse misc)
run() { VN_SKIP=$1 timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-parity-mode --no-kernel-timer --windows 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-22s %7.1f pc/s  %.3f ms/step' % ('$2', d['value'], d['ms_per_step']))"; }
run 0 nothing_skipped
for k in 0 1 2 3 4 5 6 7 8 10; do run $((1 << k)) "skip_${names[$k]}"; done
run $(( (1<<2) | (1<<7) | (1<<8) )) skip_wgrad+unpack+pack
run 0 nothing_skipped
