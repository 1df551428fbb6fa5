cd $GRAFT_REPO_ROOT
for shape in "128 256 256" "128 64 256" "128 256 128" "64 128 512"; do
  for rep in 1 2; do
    EVK_TUNE=1 EVK_X3_FORCE=q128 python tools/time_c1.py $shape 1 1 2>&1 | grep -v "INFO\|amdgpu"
    EVK_LIB=$GRAFT_REPO_ROOT/ever_amd/lib/variants/libever_hip_p2nt.so EVK_TUNE=1 EVK_X3_FORCE=q128 python tools/time_c1.py $shape 1 1 2>&1 | grep -v "INFO\|amdgpu"
  done
done
