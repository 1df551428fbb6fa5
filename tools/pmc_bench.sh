# Pipe-utilisation counters of the dominant kernels of one bench step (rocprofv3 --pmc, one pass per counter set, no
# trace domains beside --kernel-trace).  usage: bash tools/pmc_bench.sh [tag] [env assignments...] -> gpurun_out/pmc_bench_<tag>.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=${1:-r03}; shift
for e in "$@"; do export "$e"; done
O=$R/gpurun_out/pmcb_$tag; mkdir -p $O
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE"; do
i=$((i+1))
timeout 400 rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o r -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph-line --no-kernel-timer > $O/p$i.log 2>&1 < /dev/null
done
cd $R
python tools/pmc_bench_summary.py $O/p1/*.db $O/p2/*.db $O/p3/*.db $O/p4/*.db gpurun_out/pmc_bench_$tag.json > gpurun_out/pmc_bench_$tag.txt 2>&1
rm -rf $O/p1 $O/p2 $O/p3 $O/p4
