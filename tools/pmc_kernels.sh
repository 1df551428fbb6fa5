# dev tool: PMC counters of the convolution kernels as tools/check_f16x2.py launches them (one pass per counter set).
# usage: bash tools/pmc_kernels.sh <kernel-name-pattern> [env assignments...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
pat=$1; shift
for e in "$@"; do export "$e"; done
export CHECK=0
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE"; do
i=$((i+1))
timeout 300 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmck/$i -o r -- python $R/tools/check_f16x2.py > $R/gpurun_out/pmck_$i.log 2>&1 < /dev/null
done
cd $R; for i in 1 2 3 4; do python tools/pmc_summary.py $(ls gpurun_out/pmck/$i/*.db | head -1) "$pat"; done
rm -rf gpurun_out/pmck
