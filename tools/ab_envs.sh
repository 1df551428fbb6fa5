# dev tool (GPU box): bench.py A/B over VALUES of one environment variable, interleaved, three rounds
# usage: bash tools/ab_envs.sh NAME v1 v2 ...
cd $GRAFT_REPO_ROOT
name=$1; shift
for round in $(seq 1 ${ROUNDS:-3}); do
  for v in "$@"; do
    r=$(env $name=$v python bench.py --no-cpu-baseline --no-graph-line --no-kernel-timer 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "$name=$v round $round: $r"
  done
done
