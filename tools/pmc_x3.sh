# dev tool: PMC counters of one conv kernel.  usage: bash tools/pmc_x3.sh <which> <kernel-name-pattern> [env assignments...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
which=$1; pat=$2; shift 2
for e in "$@"; do export "$e"; done
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE"; do
i=$((i+1))
rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmcx3/${which}_$i -o r -- python $R/tools/one_conv.py $which 128 128 256 256 3 1 1 3 > $R/gpurun_out/pmcx3_${which}_$i.log 2>&1
done
cd $R; for i in 1 2 3 4; do python tools/pmc_summary.py $(ls gpurun_out/pmcx3/${which}_$i/*.db | head -1) $pat; done
