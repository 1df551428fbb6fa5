cd $GRAFT_REPO_ROOT
for shape in "128 256 256" "128 64 256"; do
  for f in q128; do
    EVK_TUNE=1 EVK_X3_FORCE=$f python tools/time_c1.py $shape 1 0 2>&1 | grep -v "INFO\|amdgpu"
    for a in 73 201 329 457; do
      EVK_LIB=$GRAFT_REPO_ROOT/ever_amd/lib/variants/libever_hip_p2abl$a.so EVK_TUNE=1 EVK_X3_FORCE=$f python tools/time_c1.py $shape 1 0 2>&1 | grep -v "INFO\|amdgpu"
    done
  done
done
