# bench.py under each given environment in turn, twice: tools/knob_sweep.sh "VN_X=1" "VN_WG_LIST_BLOCKS=512" ...
cd $GRAFT_REPO_ROOT
run() { env $1 timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-parity-mode --no-kernel-timer --windows 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1))"; }
for i in 1 2; do
for e in "$@"; do run "$e"; done
done
