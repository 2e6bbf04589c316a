#!/bin/bash
# round 5: what bounds fvp_split64_kernel?  Timing ablations (wrong results), one library each -> build/exp/lib_a64_*.so.
# run HERE, then on the GPU box: python tools/exp/fvp_split_ab.py 64
set -e
cd "$(dirname "$0")/../.."
rm -f build/exp/lib_*.so
b() { bash tools/exp/build_tu_variant.sh policy_split_kernels "$@" > /dev/null; }
b a64_0base &
b a64_mfma -DRL_ABL_MFMA=1 &
b a64_split -DRL_ABL_SPLIT=1 &
b a64_ops -DRL_ABL_OPS=1 &
wait
b a64_thin -DRL_ABL_THIN=1 &
b a64_mfma_split -DRL_ABL_MFMA=1 -DRL_ABL_SPLIT=1 &
b a64_all -DRL_ABL_MFMA=1 -DRL_ABL_SPLIT=1 -DRL_ABL_OPS=1 -DRL_ABL_THIN=1 &
wait
ls build/exp/*.so
