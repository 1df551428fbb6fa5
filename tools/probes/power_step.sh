# dev probe (GPU box): board power (rocm-smi, twice a second) while the bench's training step runs 400 times
cd $GRAFT_REPO_ROOT
python bench.py --steps 400 --warmup 10 --no-cpu-baseline --no-graph-line --no-kernel-timer > /tmp/b.log 2>&1 &
pid=$!
sleep 6
for i in $(seq 1 12); do rocm-smi --showpower --showclocks --json 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)['card0']; print(d.get('Current Socket Graphics Package Power (W)'), d.get('sclk clock speed:'))"; sleep 0.5; done
wait $pid
tail -1 /tmp/b.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], 'tiles/s', d['ms_per_step'], 'ms/step')"
