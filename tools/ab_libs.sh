# dev tool (GPU box): bench.py A/B between the in-tree library and SEVERAL variant builds (tools/build_variant.sh), interleaved, three rounds
# usage: bash tools/ab_libs.sh <variant name> [<variant name> ...]
cd $GRAFT_REPO_ROOT
for round in $(seq 1 ${ROUNDS:-3}); do
  for name in default "$@"; do
    if [ $name = default ]; then e="EVK_X=0"; else e="EVK_LIB=$GRAFT_REPO_ROOT/ever_amd/lib/variants/libever_hip_$name.so"; fi
    r=$(env $e python bench.py --no-cpu-baseline --no-graph-line --no-kernel-timer 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "$name round $round: $r"
  done
done
