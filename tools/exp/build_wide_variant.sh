#!/bin/bash
# experiment builds of the C-ABI library: tools/exp/build_wide_variant.sh <tag> [-DFLAG ...]  ->  build/exp/lib_<tag>.so
# (policy_wide_kernels.hip recompiled with the extra flags, the other objects taken from build/obj)
set -e
tag=$1; shift
mkdir -p build/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -fno-slp-vectorize "$@" -c -o build/exp/wide_$tag.o rllab_amd/csrc/policy_wide_kernels.hip
objs=$(ls build/obj/*.o | grep -v policy_wide_kernels)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/exp/lib_$tag.so build/exp/wide_$tag.o $objs
echo built build/exp/lib_$tag.so
