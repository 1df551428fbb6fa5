# dev tool (GPU box): per-kernel times of the bench step, single-stream (every kernel alone), under rocprofv3 --kernel-trace
# usage: bash tools/kstats.sh <tag> [env assignments...]   -> gpurun_out/kstats_<tag>.md
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1; shift
O=$R/gpurun_out/kstats_$tag; rm -rf $O; mkdir -p $O
env EVK_WGRAD_STREAM=0 "$@" rocprofv3 --kernel-trace --stats -d $O/stats -o farseg -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-graph-line > $O/bench.log 2>&1
cd $R
python tools/rocpd_summary.py $(ls $O/stats/*.db | head -1) $O/ks 16 single > /dev/null
mv $O/ks.md $R/gpurun_out/kstats_$tag.md; rm -rf $O
