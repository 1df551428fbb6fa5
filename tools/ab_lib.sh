# dev tool (GPU box): bench.py A/B between the in-tree library and ONE variant build (tools/build_variant.sh), interleaved, three rounds
# usage: bash tools/ab_lib.sh <variant name>
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/ever_amd/lib/variants/libever_hip_$1.so
for round in 1 2 3; do
  for lib in default $V; do
    if [ $lib = default ]; then e="EVK_X=0"; else e="EVK_LIB=$lib"; fi
    r=$(env $e python bench.py --no-cpu-baseline --no-graph-line --no-kernel-timer 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "$(basename $lib) round $round: $r"
  done
done
