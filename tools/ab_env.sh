# dev tool (GPU box): bench.py A/B over values of ONE environment variable, interleaved, two rounds: tiles/s per run
# usage: bash tools/ab_env.sh VAR v1 v2 [v3 ...]
cd $GRAFT_REPO_ROOT
var=$1; shift
for round in 1 2; do
  for v in "$@"; do
    r=$(env $var=$v python bench.py --no-cpu-baseline --no-graph-line --no-kernel-timer 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "$var=$v round $round: $r"
  done
done
