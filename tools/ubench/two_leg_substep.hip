// tools/ubench/two_leg_substep.hip -- the sub-step of csrc/dyn_two_legs.h in its one-body-per-scalar-lane instantiation,
// alone in a loop: what rollout_two_leg_wave_kernel runs four (HalfCheetah) / two (Walker2D) times per env-step, in a
// translation unit that compiles in a second instead of env_kernels.hip's four minutes.  Not meant to be run: its ISA
// is the evidence for the instruction count of the sub-step (profiles/r05_notes.md section 5).
//
//   tools/ubench/two_leg_substep.sh          -> "loop: 316 instrs, 312 vector, 0 nop, 88 with a lane move ..."
//
// The exchange context is the one env_kernels.hip uses (DppBodyLanes): quad_perm lane moves for parent / child / cyclic
// partner / broadcast, row_ror:8 for the other leg.
#include <hip/hip_runtime.h>
#include "../../rllab_amd/csrc/dyn_cheetah.h"
using namespace rl;
struct TwoLegQuadMoves {
    enum : int { UP = 0 | 0 << 2 | 1 << 4 | 2 << 6, DOWN = 1 | 2 << 2 | 3 << 4 | 3 << 6, NXT = 0 | 2 << 2 | 3 << 4 | 1 << 6,
                 PRV = 0 | 3 << 2 | 1 << 4 | 2 << 6, ROOT = 0, FIRST = 1 | 1 << 2 | 1 << 4 | 1 << 6, OTHER_ROW_ROR8 = 0x128 };
    template <int CTRL> __device__ __forceinline__ static float mv(float v) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
    }
};
struct DppBodyLanes : TwoLegQuadMoves {
    bool is_root, is_leaf;
    __device__ __forceinline__ float up(float v) const { return mv<UP>(v); }
    __device__ __forceinline__ float down(float v) const { return mv<DOWN>(v); }
    __device__ __forceinline__ float nxt(float v) const { return mv<NXT>(v); }
    __device__ __forceinline__ float prv(float v) const { return mv<PRV>(v); }
    __device__ __forceinline__ float root(float v) const { return mv<ROOT>(v); }
    __device__ __forceinline__ float first(float v) const { return mv<FIRST>(v); }
    __device__ __forceinline__ float other(float v) const { return mv<OTHER_ROW_ROR8>(v); }
    __device__ __forceinline__ float sel_root(float a, float b) const { return is_root ? a : b; }
    __device__ __forceinline__ float sel_leaf(float a, float b) const { return is_leaf ? a : b; }
};
__global__ void __launch_bounds__(64) substep_kernel(float* st, const float* act, int n) {
    using Legs = HalfCheetah::Legs;
    const int lane = threadIdx.x & 63, role = lane & 3, leg = (lane >> 3) & 1;
    const Legs::LaneK<float> kc = Legs::lane_constants<float>(leg, role);
    const DppBodyLanes dpp{{}, role == 0, role == 3};
    Legs::State<float> s;
    float* p = st + lane * 9;
    s.q = p[0]; s.w = p[1]; s.om = p[2]; s.sn = p[3]; s.cs = p[4]; s.p1 = p[5]; s.p2 = p[6]; s.v1 = p[7]; s.v2 = p[8];
    const float a = act[lane];
#pragma unroll 1
    for (int it = 0; it < n; ++it) Legs::substep<float, float, DppBodyLanes>(dpp, kc, s, a, 0.0025f);
    p[0] = s.q; p[1] = s.w; p[2] = s.om; p[3] = s.sn; p[4] = s.cs; p[5] = s.p1; p[6] = s.p2; p[7] = s.v1; p[8] = s.v2;
}
