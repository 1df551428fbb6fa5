python bench.py --no-graph-line --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/b.json
python bench.py --no-graph-line --no-cpu-baseline --no-kernel-timer 2>/dev/null | tail -1 > gpurun_out/b2.json
( time python bench.py 2>/dev/null | tail -1 > gpurun_out/b3.json ) 2>&1 | grep real
