# dev tool (GPU box): HIP / ROCclr runtime switches that change how kernels, signals and kernel arguments are fenced, against the
# default step, interleaved twice on one box.  -> gpurun_out/knobs2.txt
cd $GRAFT_REPO_ROOT
last() { grep '^{"metric"' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['host_unblocked_ms_per_step'])"; }
run() { env "$@" timeout 300 python bench.py --no-cpu-baseline --no-graph-line --no-kernel-timer 2>/dev/null | last "$*" >> gpurun_out/knobs2.txt; }
: > gpurun_out/knobs2.txt
for r in 1 2; do
run A=0
run ROC_SYSTEM_SCOPE_SIGNAL=0
run ROC_SKIP_KERNEL_ARG_COPY=1
run AMD_OPT_FLUSH=0
run GPU_FLUSH_ON_EXECUTION=1
run DEBUG_CLR_MAX_BATCH_SIZE=4096
run ROC_SIGNAL_POOL_SIZE=256
run HSA_ENABLE_INTERRUPT=0
run DEBUG_HIP_KERNARG_COPY_OPT=0
done
cat gpurun_out/knobs2.txt
