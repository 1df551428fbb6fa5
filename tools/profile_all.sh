# The round's evidence in ONE gpurun call (outputs under gpurun_out/rNN/, to be copied to profiles/rNN_* by tools/collect_profiles.py):
#   kernel stats both stream modes + traffic (tools/profile_round.sh), pipe utilisation (tools/pmc_bench.sh), the secondary
#   bench lines and layer tables (tools/final_lines.sh), the floor-vs-passes table, the probes behind DESIGN 2.10.
# usage: bash tools/profile_all.sh r05
tag=${1:-r05}
R=$GRAFT_REPO_ROOT
cd $R
bash tools/profile_round.sh
bash tools/pmc_bench.sh $tag
cd $R
bash tools/final_lines.sh > gpurun_out/final_lines.log 2>&1
python tools/floor_table.py c2 > gpurun_out/final/floor_table_c2.txt 2>&1
python tools/probes/power_probe.py 2>&1 | grep -v "INFO\|amdgpu" > gpurun_out/final/power_probe.txt
( cd tools/probes && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/dma_patterns dma_patterns.hip && /tmp/dma_patterns > $R/gpurun_out/final/dma_patterns.txt 2>&1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/dma_issue dma_issue.hip && /tmp/dma_issue > $R/gpurun_out/final/dma_issue.txt 2>&1 )
python tools/ab_c1sp.py e128,d128,q128,s128,t128,s64 > gpurun_out/final/ab_c1sp.txt 2>&1
mkdir -p gpurun_out/final; python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > gpurun_out/final/gputest_tail.txt
