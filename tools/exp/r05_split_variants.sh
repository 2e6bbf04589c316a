#!/bin/bash
# round 5: the split Fisher-vector product's instruction diet, one library per switch combination
#   build/exp/lib_split_<DOT2><PK>.so   (RL_SPLIT_DOT2, RL_SPLIT_PK of csrc/policy_split_kernels.hip)
# run HERE (cross-compiles), then on the GPU box:  python tools/exp/with_libs.py tools/exp/fvp_split_time.py 1
set -e
cd "$(dirname "$0")/../.."
rm -f build/exp/lib_*.so
for v in "0 0" "1 0" "0 1" "1 1"; do
  set -- $v
  bash tools/exp/build_tu_variant.sh policy_split_kernels split_$1$2 -DRL_SPLIT_DOT2=$1 -DRL_SPLIT_PK=$2 "${@:3}" &
done
wait
ls -la build/exp/lib_split_*.so
