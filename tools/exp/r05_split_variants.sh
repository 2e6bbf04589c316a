#!/bin/bash
# round 5: the split Fisher-vector product's build switches, one library per combination -> build/exp/lib_v_*.so
# (RL_SPLIT_ASM_DMA / RL_SPLIT_DOT2 / RL_SPLIT_PK / RL_SPLIT_SCALAR_SUB of csrc/policy_split_kernels.hip).
# run HERE (cross-compiles), then on the GPU box:  python tools/exp/fvp_split_ab.py
set -e
cd "$(dirname "$0")/../.."
rm -f build/exp/lib_*.so
b() { bash tools/exp/build_tu_variant.sh policy_split_kernels "$@" > /dev/null; }
b v_dma_dot_pk -DRL_SPLIT_ASM_DMA=1 -DRL_SPLIT_DOT2=1 -DRL_SPLIT_PK=1 &
b v_dma_plain -DRL_SPLIT_ASM_DMA=1 -DRL_SPLIT_DOT2=0 -DRL_SPLIT_PK=0 &
b v_dma_scalar -DRL_SPLIT_ASM_DMA=1 -DRL_SPLIT_DOT2=0 -DRL_SPLIT_PK=0 -DRL_SPLIT_SCALAR_SUB=1 &
b v_dma_scalar_ilp -DRL_SPLIT_ASM_DMA=1 -DRL_SPLIT_DOT2=0 -DRL_SPLIT_PK=0 -DRL_SPLIT_SCALAR_SUB=1 -mllvm -amdgpu-sched-strategy=max-ilp &
wait
b v_nodma_dot_pk -DRL_SPLIT_ASM_DMA=0 -DRL_SPLIT_DOT2=1 -DRL_SPLIT_PK=1 &
b v_dma_dot -DRL_SPLIT_ASM_DMA=1 -DRL_SPLIT_DOT2=1 -DRL_SPLIT_PK=0 &
b v_dma_dot_pk_ilp -DRL_SPLIT_ASM_DMA=1 -DRL_SPLIT_DOT2=1 -DRL_SPLIT_PK=1 -mllvm -amdgpu-sched-strategy=max-ilp &
b v_dma_abl_mfma -DRL_SPLIT_ASM_DMA=1 -DRL_ABL_MFMA=1 &
wait
b v_dma_abl_split -DRL_SPLIT_ASM_DMA=1 -DRL_ABL_SPLIT=1 &
b v_dma_abl_fetch -DRL_SPLIT_ASM_DMA=1 -DRL_ABL_FETCH=1 &
wait
ls build/exp/*.so
