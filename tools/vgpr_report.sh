# dev tool (no GPU needed): VGPRs / LDS per kernel of the convolution and BatchNorm sources, one line per instantiation
# (hipcc -Rpass-analysis=kernel-resource-usage).  usage: bash tools/vgpr_report.sh > /tmp/now.txt; diff profiles/r04_vgprs.txt /tmp/now.txt
# A shared epilogue that grows by a branch raises the count of every kernel that instantiates it: the one-tap DMA kernel must
# stay <= 128 (two workgroups of 512 threads per CU), the halo kernel's 12-wave forms <= 168.
cd "$(dirname "$0")/../ever_amd/csrc"
for f in conv1x1_dma conv1x1_ps2 conv1x1_sp conv3x3_halo_x3 conv3x3_wino_x3 conv_igemm_x3 conv_igemm_x3ws conv_wgrad_x3ws conv_wgrad_tr bn; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $f.hip -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
    grep -E "Function Name|VGPRs:|LDS Size" | paste - - - | sed -E 's/\[-Rpass[^]]*\]//g; s/remark: //g; s/[a-z_0-9]+\.hip:[0-9]+:[0-9]+: //g' |
    awk -v f=$f '{print f, $3, "vgprs", $5, "lds", $NF}' | c++filt | sort
done
