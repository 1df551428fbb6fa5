# dev tool (GPU box): the eager step's host cost against the number of CPUs the enqueuing threads are kept on
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for hc in 0 1 2 4 8; do
  python bench.py --host-cores $hc --steps 30 --warmup 5 --no-cpu-baseline --no-graph-line 2>/dev/null | python -c "
import sys, json
b = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{\"metric\"')][-1])
print('host-cores $hc:', b['value'], 'tiles/s', b['ms_per_step'], 'ms; host unblocked', b['host_unblocked_ms_per_step'], 'enqueue', b['host_enqueue_ms_per_step'], 'affinity', b['host_affinity'])"
done
done
echo "taskset -c 0 (launch mask of one CPU, restored after the runtime's initialisation):"
taskset -c 0 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-graph-line 2>/dev/null | python -c "
import sys, json
b = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{\"metric\"')][-1])
print(b['value'], b['ms_per_step'], 'host unblocked', b['host_unblocked_ms_per_step'], 'affinity', b['host_affinity'])"
