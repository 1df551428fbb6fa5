# dev tool (GPU box): how the two streams share the chip with and without the half-chip split of side-stream weight gradients
# (rocprofv3 --kernel-trace of a short bench run each way; tools/stream_timeline.py)  -> gpurun_out/timeline_shared_{0,1}.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 0 1; do
  O=$R/gpurun_out/tl_$v; rm -rf $O; mkdir -p $O
  EVK_WGRAD_SHARED=$v rocprofv3 --kernel-trace -d $O/stats -o farseg -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-graph-line --no-kernel-timer > $O/log.txt 2>&1
  (cd $R && python tools/stream_timeline.py $(ls $O/stats/*.db $O/stats/*/*.db 2>/dev/null | head -1) > gpurun_out/timeline_shared_$v.txt 2>&1)
  grep '^{"metric"' $O/log.txt | tail -1 | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('EVK_WGRAD_SHARED=$v under the profiler:', b['value'], 'tiles/s', b['ms_per_step'], 'ms')" >> $R/gpurun_out/timeline_shared_$v.txt
  rm -rf $O
done
