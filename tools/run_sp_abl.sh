# dev tool (GPU box): compile-time ablations of conv1x1_sp.hip on the small-map one-tap shapes (build: tools/build_variant.sh sp_<name> conv1x1_sp.hip -DEVK_SP_ABL=<bits>)
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/ever_amd/lib/variants
for shape in "32 1024 256" "32 256 1024" "64 512 128" "16 2048 512"; do
  for f in s128 s64; do
    EVK_TUNE=1 EVK_X3_FORCE=$f python tools/time_c1.py $shape 1 0 2>&1 | grep -v "INFO\|amdgpu"
    for n in e s d dc k kb kr krb krsb; do
      EVK_LIB=$V/libever_hip_sp_$n.so EVK_TUNE=1 EVK_X3_FORCE=$f python tools/time_c1.py $shape 1 0 2>&1 | grep -v "INFO\|amdgpu"
    done
  done
done
