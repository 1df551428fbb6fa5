# dev tool (GPU box): the captured step (bench.py --graph) with coarse forks of the weight-gradient branch
# (EVK_WGRAD_BATCH) and more hardware queues (GPU_MAX_HW_QUEUES), against the eager two-stream step.  gpurun_out/x3/graph.txt
mkdir -p gpurun_out/x3
last() { grep '^{"metric"' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
python bench.py --no-cpu-baseline --no-graph-line 2>/dev/null | last "eager" >> gpurun_out/x3/graph.txt
for q in 4 8; do for b in 1 8 16 32 64; do
  GPU_MAX_HW_QUEUES=$q EVK_WGRAD_BATCH=$b timeout 300 python bench.py --graph --no-cpu-baseline --no-graph-line 2>/dev/null | last "graph queues=$q batch=$b" >> gpurun_out/x3/graph.txt
done; done
EVK_WGRAD_STREAM=0 python bench.py --graph --no-cpu-baseline --no-graph-line 2>/dev/null | last "graph single-stream" >> gpurun_out/x3/graph.txt
python bench.py --no-cpu-baseline --no-graph-line 2>/dev/null | last "eager" >> gpurun_out/x3/graph.txt
python - <<'P' >> gpurun_out/x3/graph.txt 2>&1
import torch, ever_amd as er
from ever_amd.hip import functional as F
m = er.module.FarSeg(dict(encoder=dict(resnet_type='resnet18', in_channels=4), head=dict(fpn=dict(in_channels_list=(64,128,256,512), out_channels=256), fs_relation=dict(scene_embedding_channels=512)))).cuda().train()
x = torch.randn(2,4,64,64,device='cuda'); y = torch.randint(0,2,(2,64,64),device='cuda')
for _ in range(2):
    l = m.loss(m.head(m.en(x)), y); sum(l.values()).backward()
torch.cuda.synchronize(); print('wgrad_stream_stats', F.wgrad_stream_stats)
P
cat gpurun_out/x3/graph.txt
