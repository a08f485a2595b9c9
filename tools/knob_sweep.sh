cd $GRAFT_REPO_ROOT
run() { env $1 timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-parity-mode --no-kernel-timer --windows 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1))"; }
for i in 1 2; do
for e in "VN_X=1" "VN_M0_MAIN=0" "VN_EARLY_UNPACK=0" "VN_WG_BLOCKS=192" "VN_WG_BLOCKS=320" "VN_WGP_BLOCKS=192" "VN_BOX_SIDE=0"; do run "$e"; done
done
