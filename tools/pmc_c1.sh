# dev tool: PMC counters of one one-tap convolution form (tools/one_c1.py), one pass per counter set.
# usage: bash tools/pmc_c1.sh <kernel-name-pattern> "<one_c1 args>" [env assignments...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
pat=$1; shift
args=$1; shift
for e in "$@"; do export "$e"; done
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA"; do
i=$((i+1))
timeout 300 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmcc/$i -o r -- python $R/tools/one_c1.py $args > $R/gpurun_out/pmcc_$i.log 2>&1 < /dev/null
done
cd $R; for i in 1 2 3 4 5; do python tools/pmc_summary.py $(ls gpurun_out/pmcc/$i/*.db | head -1) "$pat"; done
rm -rf gpurun_out/pmcc
