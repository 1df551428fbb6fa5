python -m pytest tests -x -q -m gpu 2>&1 | tail -4
for i in 1 2; do
EVK_C1_DMA=0 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dma0', d['value'], d['ms_per_step'])"
EVK_C1_DMA=1 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dma1', d['value'], d['ms_per_step'])"
done
