// device_rng.h -- in-kernel random draws for production runs (device only).
//
// The integer stream is Philox4x32-10 (rl_math.h), pinned bit-for-bit against a
// numpy restatement in tests/.  The float transforms below use gfx950 hardware
// transcendentals (v_log_f32 / v_sin_f32 / v_cos_f32) and are therefore NOT part
// of any bit-exact contract: parity runs inject pre-generated draws instead
// (the `eps` / `reset_draws` arguments of the C ABI), exactly as SURVEY.md
// section 7 "RNG" prescribes, because the reference's single global np.random
// stream (gaussian_mlp_policy.py:128, box2d_env.py:194-217) cannot be
// reproduced by a lock-step sampler anyway.
#pragma once
#include "rl_math.h"

namespace rl {

__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& z0, float& z1) {
    float u1 = u32_to_unit_open(a);
    float u2 = u32_to_unit(b);
    float r = __builtin_sqrtf(-2.0f * __logf(u1));
    float ang = 6.28318530717958647692f * u2;
    z0 = r * __cosf(ang);
    z1 = r * __sinf(ang);
}

// Fill d[0..COUNT) with U[0,1) (NORMAL == false) or N(0,1) (NORMAL == true)
// draws for (env, step) under `purpose`.
template <int COUNT, bool NORMAL>
__device__ __forceinline__ void philox_draws(float* d, uint64_t seed, uint32_t env, uint64_t step,
                                             uint32_t purpose) {
    constexpr int BLOCKS = (COUNT + 3) / 4;
#pragma unroll
    for (int b = 0; b < BLOCKS; ++b) {
        Philox4 p = philox4x32_10(env, (uint32_t)step, (uint32_t)(step >> 32), purpose + (uint32_t)b,
                                  (uint32_t)seed, (uint32_t)(seed >> 32));
        float v[4];
        if (NORMAL) {
            box_muller(p.v[0], p.v[1], v[0], v[1]);
            box_muller(p.v[2], p.v[3], v[2], v[3]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = u32_to_unit(p.v[k]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (4 * b + k < COUNT) d[4 * b + k] = v[k];
    }
}

}  // namespace rl
