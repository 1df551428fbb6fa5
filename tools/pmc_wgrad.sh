# dev tool: PMC counters of the wave-specialised weight gradient (tools/ablate_wgrad.py) under an EVK_WG_DBG setting.
# usage: bash tools/pmc_wgrad.sh <tag> [env assignments...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
for e in "$@"; do export "$e"; done
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"; do
i=$((i+1))
timeout 200 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmcwg/${tag}_$i -o r -- python $R/tools/ablate_wgrad.py > $R/gpurun_out/pmcwg_${tag}_$i.log 2>&1 < /dev/null
done
cd $R; for i in 1 2 3 4 5; do python tools/pmc_summary.py $(ls gpurun_out/pmcwg/${tag}_$i/*.db | head -1) "conv_wgrad_x3ws_kernel<128, 256, 2>"; done
