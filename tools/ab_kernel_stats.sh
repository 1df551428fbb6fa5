# dev tool (GPU box): rocprofv3 kernel stats of bench.py for two settings of one environment switch, side by side.
# usage: bash tools/ab_kernel_stats.sh EVK_PACKED 1 0     -> gpurun_out/ab/<var>_<value>_kernel_stats.md
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab; mkdir -p $O
for v in $2 $3; do
  rm -rf $O/s_$v
  env $1=$v rocprofv3 --kernel-trace --stats -d $O/s_$v -o r -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/$1_$v.log 2>&1
  (cd $R && python tools/rocpd_summary.py $(ls $O/s_$v/*.db | head -1) $O/$1_${v}_kernel_stats)
  rm -rf $O/s_$v
done
