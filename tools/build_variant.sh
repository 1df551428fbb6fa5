#!/bin/bash
# dev tool: a second build of libever_hip.so with one source compiled under extra flags, for same-box A/B through EVK_LIB.
# usage: tools/build_variant.sh NAME source.hip "-DEVK_PS_ABL=11 ..."   ->  ever_amd/lib/variants/libever_hip_NAME.so
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; src=$2; flags=$3
mkdir -p $R/build/var_$name $R/ever_amd/lib/variants
make -C $R/ever_amd/csrc -j8 >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $flags -c $R/ever_amd/csrc/$src -o $R/build/var_$name/${src%.hip}.o
objs=$(ls $R/build/obj/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $R/build/var_$name/${src%.hip}.o -o $R/ever_amd/lib/variants/libever_hip_$name.so
echo built ever_amd/lib/variants/libever_hip_$name.so
