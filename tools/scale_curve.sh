# The multi-GPU scaling curve in one command, for the first box that has more than one MI355X:
#   bash tools/scale_curve.sh [max_gpus=8] [steps=30] [warmup=10]
# For N = 1, 2, 4, ... max_gpus: `python bench.py --gpus N` (which starts its N ranks itself, one per GPU, RCCL over xGMI)
#   -> gpurun_out/scale/bench_N.json: the whole-job line with per_rank {host_enqueue, host_unblocked, exposed_allreduce} ms
# then at N = max_gpus a rocprofv3 --kernel-trace of the same job, rank 0's database summarised by tools/stream_timeline.py
#   -> gpurun_out/scale/timeline_rank0.txt: which HIP queue RCCL's kernels land on relative to the backward / side / comm
#   streams and how much of the all-reduce runs beside the backward (the open question of the gradient exchange)
# and a table: N, tiles/s, ms/step, speed-up over N = 1, max exposed all-reduce ms, max host_unblocked ms.
# Nothing here computes an "efficiency" for the driver: it records what a reader needs to see where a curve bends.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
MAXN=${1:-8}; STEPS=${2:-30}; WARM=${3:-10}
O=$R/gpurun_out/scale; mkdir -p $O
have=$(python -c "import torch; print(torch.cuda.device_count())")
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
n=1
while [ $n -le $MAXN ]; do
  if [ $n -gt $have ]; then echo "N=$n: only $have GPU(s) on this box, stopping"; break; fi
  python bench.py --gpus $n --steps $STEPS --warmup $WARM --no-cpu-baseline --no-graph-line 2> $O/err_$n.txt | grep '^{"metric"' | tail -1 > $O/bench_$n.json
  [ -s $O/bench_$n.json ] || { echo "N=$n: no bench line (see $O/err_$n.txt)"; tail -5 $O/err_$n.txt; break; }
  last=$n; n=$((n * 2))
done
if [ "${last:-0}" -gt 1 ]; then
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d $O/trace -o farseg -- python $R/bench.py --gpus $last --steps 8 --warmup 3 --no-cpu-baseline --no-graph-line --no-kernel-timer > $O/trace.log 2>&1 )
  db=$(ls -S $(find $O/trace -name '*.db') 2>/dev/null | head -1)
  [ -n "$db" ] && python tools/stream_timeline.py $db > $O/timeline_rank0.txt 2>&1
  rm -rf $O/trace
fi
python - "$O" <<'PY'
import glob, json, os, sys
O = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(O, 'bench_*.json')):
    try:
        d = json.load(open(f))
    except ValueError:
        continue
    pr = d.get('per_rank', {})
    rows.append((d['n_gpus'], d['value'], d['ms_per_step'], max(pr.get('exposed_allreduce_ms_per_step', [0.0])),
                 max(pr.get('host_unblocked_ms_per_step', [d.get('host_unblocked_ms_per_step', 0.0)]))))
rows.sort()
if rows:
    base = rows[0][1] / rows[0][0]
    print('n_gpus  tiles/s   ms/step  x(N=1)  max exposed all-reduce ms  max host_unblocked ms')
    for n, v, ms, ex, hu in rows:
        print(f'{n:6d} {v:8.1f} {ms:9.2f} {v / base:7.2f} {ex:26.3f} {hu:21.2f}')
PY
