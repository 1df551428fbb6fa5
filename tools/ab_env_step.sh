# same-box A/B of one environment switch on the bench step: bash tools/ab_env_step.sh VAR [passes]
V=$1; N=${2:-2}
for i in $(seq $N); do for c in 0 1; do
env $V=$c python bench.py --no-cpu-baseline --no-graph-line 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V=$c', d['value'], d['ms_per_step'])"
done; done
