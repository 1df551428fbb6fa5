# The round's secondary bench lines and layer tables, one gpurun call: gpurun_out/final/*.json|txt (copy to profiles/ by hand)
O=gpurun_out/final; mkdir -p $O
last() { grep '^{"metric"' | tail -1; }
for c in c3 c4 c5; do
  python bench.py --config $c --no-cpu-baseline --no-graph-line 2>$O/err_$c.txt | last > $O/bench_$c.json
  python tools/layer_table.py $c > $O/layer_table_$c.txt 2>&1
done
python tools/layer_table.py c2 > $O/layer_table_c2.txt 2>&1
python bench.py --conv-math bf16x3 --no-cpu-baseline --no-graph-line 2>/dev/null | last > $O/bench_conv_math_bf16x3.json
python bench.py --conv-math bf16 --no-cpu-baseline --no-graph-line 2>/dev/null | last > $O/bench_conv_math_bf16.json
EVK_BENCH_FORCE_DDP=1 python bench.py --ddp flat --no-cpu-baseline --no-graph-line 2>/dev/null | last > $O/bench_ddp_flat_world1.json
python bench.py --host-cores 0 --no-cpu-baseline --no-graph-line 2>/dev/null | last > $O/bench_host_full.json   # unpinned: every CPU of the launch mask
EVK_WGRAD_STREAM=0 python bench.py --no-cpu-baseline --no-graph-line 2>/dev/null | last > $O/bench_single_stream.json
python bench.py --host-cores 2 --no-cpu-baseline --no-graph-line 2>/dev/null | last > $O/bench_host_2cores.json
python bench.py --host-cores 1 --no-cpu-baseline --no-graph-line 2>/dev/null | last > $O/bench_host_1core.json   # (--host-cores, not taskset: the HIP runtime may reset the mask at initialisation)
python bench.py --no-cpu-baseline --no-graph-line 2>/dev/null | last > $O/g_eager.json
python bench.py --graph --no-cpu-baseline --no-graph-line 2>/dev/null | last > $O/g_graph.json
python bench.py --no-cpu-baseline --no-graph-line 2>/dev/null | last > $O/g_eager2.json
python bench.py --graph --no-cpu-baseline --no-graph-line 2>/dev/null | last > $O/g_graph2.json
python bench.py --host-cores 1 --graph --no-cpu-baseline --no-graph-line 2>/dev/null | last > $O/g_graph_1core.json
for f in $O/*.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f', d['value'], d['unit'], d['ms_per_step'])"; done
