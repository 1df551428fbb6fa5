# dev tool (GPU box): the captured step (bench.py --graph) against the number of side-stream launches per fork
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  r=$(python bench.py --no-cpu-baseline --no-graph-line --no-kernel-timer 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"); echo "eager round $round: $r"
  for b in 32 8 4 2 1; do
    r=$(EVK_WGRAD_BATCH=$b python bench.py --graph --no-cpu-baseline --no-graph-line 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "graph EVK_WGRAD_BATCH=$b round $round: $r"
  done
done
