#!/bin/bash
# experiment builds of the C-ABI library: tools/exp/build_tu_variant.sh <tu> <tag> [-DFLAG ...]  ->  build/exp/lib_<tag>.so
# (csrc/<tu>.hip recompiled with the extra flags, the other objects taken from build/obj)
set -e
tu=$1; tag=$2; shift; shift
mkdir -p build/exp
extra=""
[ "$tu" = "env_kernels" ] && extra="-mllvm -amdgpu-sched-strategy=max-ilp"
[ "$tu" = "policy_split_kernels" ] && extra="-mllvm -amdgpu-mfma-vgpr-form"
[ "$tu" = "policy_csplit_kernels" ] && extra="-mllvm -amdgpu-mfma-vgpr-form"
[ "$tu" = "policy_splith_kernels" ] && extra="-mllvm -amdgpu-mfma-vgpr-form"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -fno-slp-vectorize $extra "$@" -c -o build/exp/${tu}_$tag.o rllab_amd/csrc/$tu.hip
objs=$(ls build/obj/*.o | grep -v "/$tu.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/exp/lib_$tag.so build/exp/${tu}_$tag.o $objs
echo built build/exp/lib_$tag.so
