export EVK_BENCH_MARKS=1
echo ddp1; EVK_BENCH_FORCE_DDP=1 python bench.py --ddp flat --no-cpu-baseline --no-graph-line --no-kernel-timer 2>&1 | grep -E "^weight|metric|Warn|warn" | cut -c1-230
echo ddp0; EVK_WGRAD_STREAM=0 EVK_BENCH_FORCE_DDP=1 python bench.py --ddp flat --no-cpu-baseline --no-graph-line --no-kernel-timer 2>&1 | grep -E "^weight|metric" | cut -c1-230
echo plain; python bench.py --no-cpu-baseline --no-graph-line --no-kernel-timer 2>&1 | grep -E "^weight|metric|warn" | cut -c1-230
echo torchddp; EVK_BENCH_FORCE_DDP=1 python bench.py --ddp torch --no-cpu-baseline --no-graph-line --no-kernel-timer 2>&1 | grep -E "^weight|metric|warn" | cut -c1-230
python -m pytest tests/test_wgrad_stream_gpu.py tests/test_ddp_gpu.py tests/test_graph_gpu.py -q 2>&1 | grep -E "passed|failed|Error" | tail -5
