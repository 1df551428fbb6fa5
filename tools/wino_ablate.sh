# dev tool (GPU box): where the time of the Winograd F(2,3) 3x3 kernel goes — the direct halo kernel against it on every 3x3 /
# stride-1 layer shape of the step (EVK_WINO=0 / 1), then 3x3x256 @128^2 through builds with parts of the kernel compiled out
# (tools/build_variant.sh wino_ablN conv3x3_wino_x3.hip -DEVK_WINO_ABL=N, built beforehand in the build container; the bits:
# 1 no weight DMA, 2 no halo loads / transform, 4 no MFMA, 16 no output stores).  -> gpurun_out/wino_ablate.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/wino_ablate.txt; : > $O
for w in 0 1; do EVK_WINO=$w python tools/wino_probe.py 2>/dev/null | grep -v amdgpu >> $O; done
for a in 1 2 3 4 16 20; do
  L=ever_amd/lib/variants/libever_hip_wino_abl$a.so
  [ -f $L ] || continue
  echo "EVK_WINO_ABL=$a (timing only, wrong results)" >> $O
  EVK_LIB=$PWD/$L ONLY=fpn.256-256@128 python tools/wino_probe.py 2>/dev/null | grep "fpn.256" >> $O
done
cat $O
