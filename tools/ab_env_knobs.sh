# dev tool (GPU box): runtime environment switches against the default step, one box.  gpurun_out/x6/knobs.txt
mkdir -p gpurun_out/x6
last() { grep '^{"metric"' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['host_unblocked_ms_per_step'])"; }
run() { env "$@" timeout 300 python bench.py --no-cpu-baseline --no-graph-line 2>/dev/null | last "$*" >> gpurun_out/x6/knobs.txt; }
run A=0
run HSA_ENABLE_INTERRUPT=0
run GPU_MAX_HW_QUEUES=2
run GPU_MAX_HW_QUEUES=8
run HSA_ENABLE_SDMA=0
run HIP_FORCE_DEV_KERNARG=0
run AMD_SERIALIZE_KERNEL=0 HSA_DISABLE_CACHE=0 ROC_ACTIVE_WAIT_TIMEOUT=1000
run HSA_XNACK=0
run PYTORCH_NO_HIP_MEMORY_CACHING=0 PYTORCH_HIP_ALLOC_CONF=expandable_segments:True
run EVK_WGRAD_STREAM=0
run A=1
cat gpurun_out/x6/knobs.txt
