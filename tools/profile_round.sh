# Round profile on the GPU box: kernel-trace stats + the two PMC traffic passes of bench.py; outputs under gpurun_out/prof_round.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_round; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/stats -o farseg -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph-line > $O/bench_under_rocprof.log 2>&1
EVK_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -d $O/stats1 -o farseg -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph-line > $O/bench_under_rocprof_single_stream.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o r -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph-line --no-kernel-timer > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o r -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph-line --no-kernel-timer > $O/write.log 2>&1
cd $R
python bench.py > $O/bench.log 2>&1
tail -1 $O/bench.log > $O/bench.json
grep "^{\"metric\"" $O/bench_under_rocprof.log | tail -1 > $O/bench_under_rocprof.json
python tools/rocpd_summary.py $(ls $O/stats/*.db | head -1) $O/kernel_stats
python tools/rocpd_summary.py $(ls $O/stats1/*.db | head -1) $O/kernel_stats_single_stream 28 single
python tools/stream_timeline.py $(ls $O/stats/*.db | head -1) > $O/stream_timeline.txt 2>&1
grep "^{\"metric\"" $O/bench_under_rocprof_single_stream.log | tail -1 > $O/bench_under_rocprof_single_stream.json
rm -rf $O/stats1
python tools/traffic_from_pmc.py $(ls $O/fetch/*.db | head -1) $(ls $O/write/*.db | head -1) $O/traffic.json
rm -rf $O/fetch $O/write   # keep the merged-back payload small; the stats db stays
