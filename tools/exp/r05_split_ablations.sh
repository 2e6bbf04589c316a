#!/bin/bash
# round 5: what bounds fvp_split_kernel?  Timing ablations (wrong results, same instruction stream otherwise), one library
# each: build/exp/lib_abl_<name>.so.  Run HERE, then on the GPU box: python tools/exp/fvp_split_ab.py
set -e
cd "$(dirname "$0")/../.."
rm -f build/exp/lib_*.so
b() { bash tools/exp/build_tu_variant.sh policy_split_kernels "$@" > /dev/null; }
b abl_0base &
b abl_mfma -DRL_ABL_MFMA=1 &
b abl_split -DRL_ABL_SPLIT=1 &
b abl_fetch -DRL_ABL_FETCH=1 &
wait
b abl_ops -DRL_ABL_OPS=1 &
b abl_mfma_split -DRL_ABL_MFMA=1 -DRL_ABL_SPLIT=1 &
b abl_fetch_ops -DRL_ABL_FETCH=1 -DRL_ABL_OPS=1 &
b abl_all -DRL_ABL_MFMA=1 -DRL_ABL_SPLIT=1 -DRL_ABL_FETCH=1 -DRL_ABL_OPS=1 &
wait
b abl_nodot_nopk -DRL_SPLIT_DOT2=0 -DRL_SPLIT_PK=0 &
wait
ls build/exp/
