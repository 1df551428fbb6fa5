# dev tool (GPU box): bench.py over a LIST of environment settings ("A=1 B=2" strings), interleaved, ROUNDS rounds
cd $GRAFT_REPO_ROOT
for round in $(seq 1 ${ROUNDS:-3}); do
  for e in "$@"; do
    r=$(env $e python bench.py --no-cpu-baseline --no-graph-line --no-kernel-timer 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "[$e] round $round: $r"
  done
done
