# dev tool (GPU box): forward / backward phases of the step with the head's pyramid levels on one stream and on two
# (EVK_HEAD_BRANCH=0 / 1) under rocprofv3 --kernel-trace -> gpurun_out/head_branch_timeline.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 0 1; do
  O=$R/gpurun_out/hb_$v; rm -rf $O; mkdir -p $O
  EVK_HEAD_BRANCH=$v rocprofv3 --kernel-trace -d $O/stats -o farseg -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-graph-line --no-kernel-timer > $O/log.txt 2>&1
  DB=$(ls $O/stats/*.db $O/stats/*/*.db 2>/dev/null | head -1)
  echo "EVK_HEAD_BRANCH=$v" >> $R/gpurun_out/head_branch_timeline.txt
  (cd $R && python tools/phase_timeline.py $DB >> gpurun_out/head_branch_timeline.txt 2>&1; python tools/stream_timeline.py $DB 2>&1 | sed -n 2,9p >> gpurun_out/head_branch_timeline.txt)
  rm -rf $O
done
