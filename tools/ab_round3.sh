cd $GRAFT_REPO_ROOT
run() { env "$@" python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', d['value'], d['ms_per_step'], 'igemm', round(d['roofline']['avg_launch_us']*d['roofline']['launches_per_step']/1e3,2), 'wgrad', round(d['roofline_wgrad']['avg_launch_us']*d['roofline_wgrad']['launches_per_step']/1e3,2), 'bn', round(d['roofline_hbm_bn']['avg_call_us'],1))
"; }
for i in 1 2; do
run A=1
run EVK_WGRAD_TR=0
run EVK_LAZY_RES=0 EVK_RELU_BITS=0
run EVK_WGRAD_TR=0 EVK_LAZY_RES=0 EVK_RELU_BITS=0
done
