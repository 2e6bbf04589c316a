#!/bin/bash
# experiment builds of the C-ABI library: tools/exp/build_variant.sh <tag> [-DFLAG ...]  ->  build/exp/lib_<tag>.so
# (env_kernels.hip recompiled with the extra flags, the other objects taken from build/obj)
set -e
tag=$1; shift
mkdir -p build/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp "$@" -c -o build/exp/env_$tag.o rllab_amd/csrc/env_kernels.hip
objs=$(ls build/obj/*.o | grep -v env_kernels)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/exp/lib_$tag.so build/exp/env_$tag.o $objs
echo built build/exp/lib_$tag.so
