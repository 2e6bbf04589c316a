/* rllab_amd.h -- C ABI of librllab_amd.so, the MI355X (gfx950) engine underneath
 * the rllab Env / Policy / Sampler / optimizer seams.
 *
 * The reference (rll/rllab) is pure Python: its plug-in boundary is duck typing
 * (SURVEY.md section 8b).  This header is what a maintainer binds with ctypes
 *  from exactly those seams.  Each entry point names the
 * reference function it replaces.
 *
 * Conventions
 *  - every pointer argument is a DEVICE pointer owned by the caller unless the
 *    comment says "host";
 *  - per-env arrays are struct-of-arrays "planes": a quantity with D components
 *    for N envs is float[D][N]; trajectory quantities are float[D][T][N];
 *  - `stream` is a hipStream_t (may be NULL = default stream); calls only
 *    enqueue work, they never synchronise;
 *  - return value 0 = ok, negative = error; rl_last_error() (host string,
 *    thread-local) describes the last failure.  No exceptions cross the ABI.
 */
#ifndef RLLAB_AMD_H
#define RLLAB_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum rl_env_kind {
    RL_ENV_CARTPOLE = 0,         /* rllab/envs/box2d/cartpole_env.py:10-56 */
    RL_ENV_DOUBLE_PENDULUM = 1,  /* rllab/envs/box2d/double_pendulum_env.py:11-61 */
    RL_ENV_SWIMMER = 2,          /* rllab/envs/mujoco/swimmer_env.py:10-62 (swimmer-style planar chain) */
    RL_ENV_HALF_CHEETAH = 3,     /* rllab/envs/mujoco/half_cheetah_env.py:14-56 (cheetah-style planar tree) */
    RL_ENV_CARTPOLE_SWINGUP = 4, /* rllab/envs/box2d/cartpole_swingup_env.py:14-61 */
    RL_ENV_WALKER2D = 5,         /* rllab/envs/mujoco/walker2d_env.py:15-59 (walker-style planar biped) */
    RL_ENV_HOPPER = 6,           /* rllab/envs/mujoco/hopper_env.py:19-62 (hopper-style planar monoped) */
    RL_ENV_INVERTED_DOUBLE_PENDULUM = 7  /* rllab/envs/mujoco/inverted_double_pendulum_env.py:10-58 */
};

enum rl_status {
    RL_OK = 0,
    RL_ERR_ARG = -1,       /* bad argument (unknown env kind, null pointer, bad size) */
    RL_ERR_UNSUPPORTED = -2, /* valid request the engine has no kernel for (e.g. hidden sizes) */
    RL_ERR_HIP = -3        /* HIP runtime error; rl_last_error() has hipGetErrorString */
};

/* Human-readable description of the last error on this thread (host string). */
const char* rl_last_error(void);

/* Library / ABI version, bumped when a signature changes.  11: rl_rollout_lds_bytes, rl_mlp_forward_ws (every network
 * shape of the two kernel families as a function on planes), RL_CFG_LIMIT_MUJOCO, rl_policy_fvp_variant's value 2.
 * 12: rl_policy_batch.gate + rl_line_search_decide (the line search decided on the device); rl_launch_opts (rl_rollout_args.opts,
 * rl_policy_batch.opts, a trailing `variant` / `spin_limit` argument of rl_lfb_normal_eq / rl_peer_allreduce_sum) in place
 * of the library's getenv reads; rl_rollout_plan_query; per-layer hidden activations (layer_activations, RL_ACT_IDENTITY);
 * rl_running_norm (NormalizedEnv's running observation / reward normalisation inside the fused rollout).
 * 13: rl_env_terminates; rl_rollout_args.reset_at_start == 0 continues from last_obs (the sampler's further launches
 * until batch_size whole-path samples are in, running normalisation included); RL_CFG_CONTACT_MUJOCO, and RL_CFG_LIMIT_MUJOCO
 * for the legged envs; rl_rollout_args.std_layer_activations, identity layers on the cooperative (wide) kernels.
 * 14: rl_policy_batch.obs_absmax (the two-way f16 split Fisher-vector product), rl_launch_opts.fvp_split = 4 / 5,
 * rl_policy_fvp_variant's value 4. */
int rl_abi_version(void);

/* Static facts about an env kind: observation / action / persisted-state sizes,
 * number of random draws one reset consumes and whether they are N(0,1)
 * (1) or U[0,1) (0).  Replaces Env.observation_space / action_space shape
 * queries (rllab/envs/base.py:31-50).  All outputs are host ints. */
int rl_env_query(int kind, int* obs_dim, int* act_dim, int* state_dim,
                 int* reset_draws, int* reset_is_normal);

/* 1 when a path of this env kind can end before max_path_length (the env's own done rule: CartpoleEnv.is_current_done,
 * rllab/envs/box2d/cartpole_env.py:53-56; hopper / walker2d / inverted double pendulum), 0 when the reference env's done
 * is always False (swimmer_env.py:44, half_cheetah_env.py:45, double_pendulum_env.py:60-61), negative on an unknown kind.
 * A sampler that has to return at least batch_size samples in whole paths (rllab/algos/batch_polopt.py:23-34,
 * sandbox/rocky/tf/samplers/vectorized_sampler.py:55) knows from this whether n_envs x max_path_length lock steps are
 * that many samples by construction or have to be counted. */
int rl_env_terminates(int kind);

/* Action bounds of the un-normalised env (host arrays of act_dim floats).
 * Replaces Box2DEnv.action_space (box2d_env.py:99-103) and
 * MujocoEnv.action_space (mujoco_env.py:85-90). */
int rl_env_action_bounds(int kind, float* lb_host, float* ub_host);

/* Constructor options of the reference's env classes that the kernels honour at run time.  A HOST struct;
 * NULL wherever an `rl_env_cfg*` is taken means "the env's defaults" (what rl_env_default_cfg writes).
 *   ctrl_cost_coeff  SwimmerEnv / Walker2DEnv / HopperEnv(ctrl_cost_coeff=..)
 *                    (rllab/envs/mujoco/swimmer_env.py:15-21, walker2d_env.py:21-27, hopper_env.py:27-35)
 *   alive_coeff      HopperEnv(alive_coeff=..)                          (hopper_env.py:27-35)
 *   action_noise     Box2DEnv / MujocoEnv(action_noise=..): applied = action + 0.5 (ub - lb) * action_noise * N(0,1),
 *                    after the reward captured the action (box2d_env.py:163-175,219-226, mujoco_env.py:175-187)
 *   obs_noise        Box2DEnv(obs_noise=..): observation + obs_noise * N(0,1) entry-wise (box2d_env.py:194-218)
 *   frame_skip       Box2DEnv(frame_skip=..): world steps per env step; 0 = the env's default
 *   flags            RL_CFG_* below
 *   link_len         DoublePendulumEnv: length of both links; the reference draws it once per env object when
 *                    template_args = {noise: True} (double_pendulum_env.py:17-21), else 1.  0 = the model's.
 *   reserved         0
 *   action_noise_z   DEVICE pointer or NULL: injected N(0,1) draws of the action noise (parity runs);
 *                    rl_vecenv_step: float[act_dim][n], rl_rollout_gaussian_mlp: float[T][act_dim][n].
 *                    NULL = Philox4x32-10 keyed (seed; env, step, ACT_NOISE)
 *   obs_noise_z      the same for the observation noise: float[obs_dim][n] (step / reset),
 *                    float[T+1][obs_dim][n] (rollout; slice 0 = first observation, t + 1 = after step t) */
typedef struct rl_env_cfg {
    float ctrl_cost_coeff;
    float alive_coeff;
    float action_noise;
    float obs_noise;
    int32_t frame_skip;
    int32_t flags;
    float link_len;
    float reserved;
    const float* action_noise_z;
    const float* obs_noise_z;
} rl_env_cfg;

enum rl_env_cfg_flags {
    RL_CFG_POLE_FOLLOWS_CART = 1, /* CartpoleEnv / CartpoleSwingupEnv: reset moves the pole body with the cart, so
                                   * the hinge starts closed (engine option; the reference's reset leaves the pole
                                   * origin at the XML pose, cartpole_env.py:28-43 -- DESIGN.md section 5) */
    RL_CFG_FIXED_START = 2,       /* InvertedDoublePendulumEnv(random_start=False)
                                   * (inverted_double_pendulum_env.py:20,47-58) */
    RL_CFG_LIMIT_MUJOCO = 4,      /* SwimmerEnv(limit_model="mujoco") (engine option): the two hinge limits act through
                                   * MuJoCo's documented soft-constraint model with the MJCF's own solreflimit = "0.02 1"
                                   * and solimplimit = "0 .8 .03" (vendor/mujoco_models/swimmer.xml:31,34): reference
                                   * acceleration + impedance, constraint forces f >= 0 minimising
                                   * 1/2 f'(A + R)f + f'(a0 - a_ref), solved exactly for the (at most two) active rows --
                                   * instead of the default penalty spring-damper; such launches run the scalar sub-step
                                   * program (the four-lanes-per-env rollout is built for the default).
                                   * HalfCheetahEnv / Walker2DEnv / HopperEnv(limit_model="mujoco"): the hinges' limits as rows
                                   * of the projected Gauss-Seidel constraint solve of csrc/dyn_mjc.h, parameters from
                                   * vendor/mujoco_models/half_cheetah.xml:38 (hopper / walker2d: MuJoCo's defaults). */
    RL_CFG_CONTACT_MUJOCO = 8     /* HalfCheetahEnv / Walker2DEnv / HopperEnv(contact_model="mujoco") (engine option): the
                                   * capsule end spheres' floor contacts through the same soft-constraint model -- condim 3,
                                   * pyramidal cone (two edge rows J_n +- mu J_t per contact in the plane), the MJCFs' own
                                   * solref / solimp / margin (half_cheetah.xml:39, hopper.xml:5, walker2d.xml:6), up to 8
                                   * contacts at once, 100 Gauss-Seidel sweeps -- instead of the default spring-damper
                                   * penalty.  Either flag takes these envs to the env-per-lane rollout kernels. */
};

/* The options env `kind` runs with by default (host struct out). */
int rl_env_default_cfg(int kind, rl_env_cfg* cfg_host);

/* MujocoEnv.get_body_com("torso") / get_body_comvel("torso") (rllab/envs/mujoco/mujoco_env.py:232-238,
 * rllab/mujoco_py/mjcore.py:58-81: subtree linear momentum / subtree mass) from the persisted state planes:
 *   com4  float[4][n] = (forward, up) position and (forward, up) velocity of the torso subtree's centre of mass
 * RL_ERR_UNSUPPORTED for the Box2D-style env kinds. */
int rl_vecenv_com(int kind, int n, const float* state, float* com4, void* stream);

/* Env.reset for the envs selected by `mask` (NULL = all).
 *   state   float[state_dim][n]   in/out (persisted solver state survives reset
 *                                 where the reference's does)
 *   ts      int32[n]              steps since reset, zeroed for reset envs
 *   draws   float[reset_draws][n] injected random draws (parity mode) or NULL:
 *                                 then Philox4x32-10 keyed by `seed`, counter
 *                                 (env index + env_offset, step_counter, RESET)
 *   obs     float[obs_dim][n]     written for reset envs only
 * Replaces CartpoleEnv.reset (cartpole_env.py:28-43), DoublePendulumEnv.reset
 * (double_pendulum_env.py:32-41), MujocoEnv.reset (mujoco_env.py:109-123) and
 * VecEnvExecutor.reset (sandbox/rocky/tf/envs/vec_env_executor.py:30-33). */
int rl_vecenv_reset(int kind, int n, float* state, int32_t* ts, const uint8_t* mask,
                    const float* draws, uint64_t seed, uint64_t step_counter,
                    int env_offset, const rl_env_cfg* cfg, float* obs, void* stream);

/* rl_vecenv_step for launches replayed from a hipGraph (the per-step loop of an arbitrary policy is launch-bound:
 * policy kernels + one step kernel per transition, T times; captured once and replayed, it pays one graph launch per
 * transition).  Kernel arguments are frozen at capture, so the RNG counter of the transition comes from a device
 * word, advanced between replays by rl_counter_add (itself a node of the graph).  No injected reset draws. */
int rl_vecenv_step_graph(int kind, int n_envs, int normalize, float scale_reward, int max_path_length,
                         int auto_reset, float* state, int32_t* ts, const float* actions, uint64_t seed,
                         const uint64_t* step_counter_dev, int env_offset, const rl_env_cfg* cfg, float* obs,
                         float* reward, uint8_t* done, void* stream);
int rl_counter_add(uint64_t* counter_dev, uint64_t increment, void* stream);

/* Observation of the state planes as they are, without a transition: obs[obs_dim][n] = observe(state).
 * Env.get_current_obs of the reference env bases (rllab/envs/box2d/box2d_env.py:210-218,
 * rllab/envs/mujoco/mujoco_env.py:118-131); used after a state was written from outside. */
int rl_vecenv_observe(int kind, int n_envs, const float* state, float* obs, void* stream);

/* One lock-step Env.step over n envs, with the VecEnvExecutor contract: ts += 1,
 * done |= ts >= max_path_length, done envs are reset inside the call and the
 * returned obs is the post-reset observation
 * (sandbox/rocky/tf/envs/vec_env_executor.py:16-28).
 *   normalize != 0 applies NormalizedEnv's action affine map + clip
 *                  (rllab/envs/normalized_env.py:78-92); scale_reward as there.
 *   actions float[act_dim][n]; reward float[n]; done uint8[n].
 *   auto_reset == 0 gives plain Env.step semantics (rllab/envs/base.py:7-22): the
 *   terminal observation is returned and the caller resets.
 * Replaces NormalizedEnv.step -> Box2DEnv.step / MujocoEnv.forward_dynamics. */
int rl_vecenv_step(int kind, int n, int normalize, float scale_reward, int max_path_length,
                   int auto_reset, float* state, int32_t* ts, const float* actions,
                   const float* reset_draws, uint64_t seed, uint64_t step_counter,
                   int env_offset, const rl_env_cfg* cfg, float* obs, float* reward, uint8_t* done,
                   void* stream);

/* NormalizedEnv(normalize_obs=True / normalize_reward=True) inside the fused rollout (rllab/envs/normalized_env.py:33-49,
 * 78-92): per env copy the running estimates
 *     mean <- (1 - alpha) mean + alpha x;   var <- (1 - alpha) var + alpha (x - mean)^2        (float64, the updated mean)
 * and the whitened values  (x - mean) / (sqrt(var) + 1e-8)  /  reward / (sqrt(var_r) + 1e-8)  that the policy sees and the
 * trajectory records.  Update order of the reference's executor over n NormalizedEnv copies: every step's observation --
 * the TERMINAL one included -- feeds its copy's estimate, a finished copy is then reset and the reset observation feeds
 * it once more and is returned whitened; the reward estimate is fed once per step, the scale_reward factor applies
 * after the normalisation.  The arrays are state, read and written in place (they persist over rollouts and are
 * shared with the per-transition path's NormalizingVecEnv).  Takes the generic rollout kernels. */
typedef struct rl_running_norm {
    double* obs_mean;          /* double[obs_dim][n]   (ignored unless normalize_obs) */
    double* obs_var;           /* double[obs_dim][n] */
    double* reward_mean;       /* double[n]            (ignored unless normalize_reward) */
    double* reward_var;        /* double[n] */
    double obs_alpha, reward_alpha;
    int32_t normalize_obs, normalize_reward;
} rl_running_norm;

/* Launch-shape requests.  Every field: 0 = the library's own rule (what every production call passes).  The non-zero
 * values exist for A/B timing and so that the parity tests can run every launch shape against the host build; the
 * Python binding fills the struct from the RLLAB_* environment switches of INTEGRATION.md section 4 (rllab_amd/_lib.py::
 * launch_opts) -- the library itself reads no environment variable.  rl_rollout_plan reports what a request resolves to. */
typedef struct rl_launch_opts {
    int32_t rollout_epw;          /* generic rollout kernels: 16 / 64 envs per wavefront (also keeps the Swimmer / two-leg
                                   * lane-group kernels from being chosen) */
    int32_t rollout_wpb;          /* lane-group shapes: 1 / 2 / 4 wavefronts per workgroup */
    int32_t swimmer_lane_kernel;  /* 1: the generic env-per-lane kernel for the Swimmer */
    int32_t swimmer_coop;         /* wide / deep policy on the Swimmer: 1 force, 2 forbid the four-wavefront shape */
    int32_t two_leg_lane_kernel;  /* 2: forbid the one-leg-per-lane kernels (HalfCheetah / Walker2D) */
    int32_t two_leg_wave_kernel;  /* 1 force, 2 forbid one env per wavefront */
    int32_t fvp_split;            /* rl_policy_fvp: 1 = f32 matrix instructions only, 2 = the cooperative split kernel for
                                   * every shape it is built for, 3 = the 16-sample-tile split kernel (four wavefronts per
                                   * SIMD) for the (32, 32) shapes it is built for, 4 = the two-way f16 split kernels (also
                                   * the library's choice, 0, whenever rl_policy_batch.obs_absmax is set and the shape is
                                   * theirs), 5 = the three-way bf16 split kernels although obs_absmax is set */
    int32_t fvp_split_wps;        /* 1: fvp_split_kernel with one wavefront per SIMD (register-resident operands); 4: the
                                   * 16-sample-tile kernel (fvp_split = 3) with four wavefronts per SIMD instead of three */
    int32_t lfb_valu;             /* rl_lfb_normal_eq: 1 = the register-blocked vector kernel */
    int32_t reserved[7];          /* reserved[0]: timing ablations of fvp_split16_kernel (WRONG results; tools/exp/fvp_split16_time.py) */
} rl_launch_opts;

/* Arguments of the fused rollout: T lock-step iterations of
 *   policy.get_actions -> env.step -> record -> auto-reset
 * for n envs in ONE launch (every env is independent, so no grid-wide
 * synchronisation exists).  Replaces the per-step Python loop of rollout()
 * (rllab/sampler/utils.py:18-31) as driven by BatchSampler.obtain_samples
 * (rllab/algos/batch_polopt.py:22-34), with GaussianMLPPolicy.get_actions
 * (rllab/policies/gaussian_mlp_policy.py:132-137) evaluated in-kernel. */
typedef struct rl_rollout_args {
    int32_t kind;             /* rl_env_kind */
    int32_t n_envs;
    int32_t horizon;          /* T: steps recorded per env in this call */
    int32_t max_path_length;  /* forced done when ts reaches it */
    int32_t normalize;        /* NormalizedEnv action map on/off */
    int32_t reset_at_start;   /* 1: reset every env before step 0.  0: carry on from state / ts; with last_obs != NULL the
                                 first observation is last_obs as the previous launch, rl_vecenv_step or rl_vecenv_reset left
                                 it (observation noise and whitening included, no estimate is fed twice) on the generic /
                                 wide / dual kernels -- the lane-group kernels re-derive it from the state */
    int32_t hidden0, hidden1; /* tanh MLP hidden sizes: each 32, 64 or 128 (narrower layers: zero padding, exact) */
    int32_t hidden2;          /* third hidden layer, 0 = two layers (network.py:36-101 takes any hidden_sizes tuple) */
    int32_t env_offset;       /* global index of env 0 (multi-GPU sharding) */
    float scale_reward;
    float log_min_std;        /* log_std floor, log(min_std) (gaussian_mlp_policy.py:100-101) */
    uint64_t seed;
    uint64_t step_counter;    /* global step index of t = 0 (RNG counter base) */
    float* state;             /* float[state_dim][n]  in/out */
    int32_t* ts;              /* int32[n]             in/out */
    const float* theta;       /* flat policy params, reference layout W0,b0,W1,b1,[W2,b2,]Wout,bout,log_std,
                                 W stored [in][out] row-major (parameterized.py:54-58) */
    const float* eps;         /* NULL or float[act_dim][T][n] injected N(0,1) policy noise */
    const float* reset_draws; /* NULL or float[T+1][reset_draws][n] injected reset draws;
                                 slice 0 = initial reset, slice t+1 = reset after step t */
    float* obs;               /* float[obs_dim][T][n]  observation the action was computed from */
    float* actions;           /* float[act_dim][T][n] */
    float* means;             /* float[act_dim][T][n]  agent_info "mean" */
    float* rewards;           /* float[T][n] */
    uint8_t* dones;           /* uint8[T][n]  env done OR ts == max_path_length */
    float* last_obs;          /* NULL or float[obs_dim][n]: observation after the last step (post-reset), in/out */
    const rl_env_cfg* cfg;    /* host; NULL = the env's defaults */
    const float* theta_std;   /* NULL (log_std is the last row of theta), or the parameters of a log-std NETWORK on the
                                 observation -- GaussianMLPPolicy(adaptive_std=True) / std_network=...
                                 (gaussian_mlp_policy.py:60-98) -- in the same layout as theta (W0,b0,..,Wout,bout and
                                 act_dim unused floats); theta then holds the mean network the same way */
    float* log_stds;          /* with theta_std: float[act_dim][T][n] agent_info "log_std" (floored at log_min_std) */
    int32_t std_hidden0, std_hidden1, std_hidden2;   /* hidden sizes of the log-std network (as hidden0..2) */
    int32_t layer_activations;/* hidden activations of the (mean) network per layer, as rl_policy_batch.layer_activations:
                                 0 = tanh layers; rectify / identity layers on the equal-width (32,32) / (64,64) kernels */
    const rl_launch_opts* opts;   /* host; NULL = every launch rule the library's own */
    const rl_running_norm* norm;  /* host; NULL = no running normalisation (NormalizedEnv's defaults) */
    int32_t std_layer_activations;/* with theta_std: hidden activations of the log-std network (as layer_activations).  Next to a
                                     log-std network both words may hold tanh / identity layers only: a ONE-hidden-layer
                                     network (gaussian_mlp_policy.py:60-90 takes any hidden sizes) runs as (H, H) with W1 = I */
    int32_t reserved_tail;
} rl_rollout_args;

int rl_rollout_gaussian_mlp(const rl_rollout_args* args, void* stream);

/* Which kernel, in which shape, rl_rollout_gaussian_mlp(args) launches -- the launch rules as data (host query, launches
 * nothing; only kind, n_envs, horizon, the hidden sizes, theta_std != NULL, cfg->flags and opts are read).  bench.py names
 * the kernel of its roofline from here, HipVecEnv.takes_rollout_of asks `supported` before it commits a policy to the
 * fused path, the parity tests assert the shape they meant to run. */
enum rl_rollout_kernel {
    RL_ROLLOUT_UNSUPPORTED = 0,      /* no fused kernel: sample through rl_vecenv_step */
    RL_ROLLOUT_GENERIC = 1,          /* rollout_kernel<Env, H, H, EPW>: (32,32) / (64,64), weights in registers */
    RL_ROLLOUT_WIDE = 2,             /* rollout_wide_kernel<Env, EPW>: two or three layers of 32 / 64 / 128, fragments in LDS */
    RL_ROLLOUT_DUAL = 3,             /* rollout_dual_kernel<Env, EPW>: mean net + log-std net */
    RL_ROLLOUT_SWIMMER_QUAD = 4,     /* rollout_swimmer_quad_kernel<H>: four lanes per env in the physics */
    RL_ROLLOUT_SWIMMER_QUAD_WIDE = 5,
    RL_ROLLOUT_SWIMMER_QUAD_COOP = 6,/* ... the network split over four wavefronts per group of 16 envs */
    RL_ROLLOUT_TWO_LEG_WAVE = 7,     /* rollout_two_leg_wave_kernel<Env, H>: one env per wavefront */
    RL_ROLLOUT_TWO_LEG_QUAD = 8,     /* rollout_two_leg_quad_kernel<Env, H>: one leg per lane, 16 envs per wavefront */
    RL_ROLLOUT_TWO_LEG_QUAD_WIDE = 9
};
typedef struct rl_rollout_plan {
    int32_t kernel;                    /* rl_rollout_kernel */
    int32_t envs_per_wavefront;        /* 64, 16 or 1 */
    int32_t wavefronts;                /* of the whole launch */
    int32_t wavefronts_per_workgroup;
    int32_t workgroups;
    int32_t lds_bytes;                 /* dynamic LDS of one workgroup (0: static only) */
    int32_t lds_limit;                 /* of a CU: 160 KB */
    int32_t reserved;
    char name[96];                     /* the kernel's name with its template arguments, e.g. "rollout_swimmer_quad_kernel<32>" */
} rl_rollout_plan;
int rl_rollout_plan_query(const rl_rollout_args* args, rl_rollout_plan* plan);

/* LDS bytes the fused rollout of a WIDE / DEEP policy (any hidden sizes other than (32,32) / (64,64): two or three tanh
 * layers of 32 / 64 / 128 units, weight fragments in LDS) or of a policy with a log-std NETWORK (std_hidden0 != 0:
 * both networks' fragments in LDS) needs at its smallest launch shape (one wavefront per workgroup), and the limit of
 * a CU (160 KB).  *bytes = 0: the sizes are not a kernel shape at all.  rl_rollout_gaussian_mlp lowers its
 * wavefronts-per-workgroup until the workgroup fits and returns RL_ERR_UNSUPPORTED when *bytes > *limit; a caller asks
 * here first and samples such a policy through rl_vecenv_step instead (rllab/policies/gaussian_mlp_policy.py:21-98 is
 * free-form in its hidden sizes; e.g. mean (128,128) + log-std (128,128) on a 20-observation env needs 172 KB). */
int rl_rollout_lds_bytes(int env_kind, int hidden0, int hidden1, int hidden2, int std_hidden0, int std_hidden1,
                         int std_hidden2, size_t* bytes, size_t* limit);

/* Segmented reverse linear-recurrence scans over [T][n] planes, fused:
 *   delta[t] = r[t] + gamma * V[t+1] * (1 - end[t]) - V[t]
 *   adv[t]   = delta[t] + gamma*lambda * (1 - end[t]) * adv[t+1]
 *   ret[t]   = r[t]     + gamma        * (1 - end[t]) * ret[t+1]
 * where end[t] = done[t] | (t == T-1) and V after a path's last step is 0.
 * gamma / lambda are doubles (the reference's Python floats); accumulates in f64,
 * stores f32.  Replaces the per-path loop of
 * BaseSampler.process_samples (rllab/sampler/base.py:57-66) and
 * special.discount_cumsum (rllab/misc/special.py:107-111).
 *   values: f64 plane [T][n] of baseline predictions (the reference predicts in
 *   float64), may be NULL (ZeroBaseline / LinearFeatureBaseline before first fit).
 *   undiscounted: optional [T][n] plane of the undiscounted return-to-go (its value at a path
 *   start is the path's total return, the "AverageReturn" statistic of base.py:93-103). */
int rl_gae(int T, int n, const float* rewards, const double* values, const uint8_t* dones,
           double gamma, double lambda, float* adv, float* ret, float* undiscounted, void* stream);

/* y[t] = x[t] + discount * (1 - end[t]) * y[t+1] on a [T][n] plane
 * (special.discount_cumsum with path boundaries; dones may be NULL = one path
 * per column). */
int rl_discount_cumsum(int T, int n, const float* x, const uint8_t* dones, double discount,
                       float* y, void* stream);

/* ---- BaseSampler.process_samples / LinearFeatureBaseline on dense [T][n] planes ----------------
 * (rllab/sampler/base.py:48-104, rllab/baselines/linear_feature_baseline.py:6-43) */

/* Path indexing + baseline prediction in one pass over the columns.
 *   tin    int32[T][n]  step index inside its path (the arange(l) of linear_feature_baseline.py:18)
 *   valid  uint8[T][n]  1 unless whole_paths != 0 and no done flag follows in the column (the
 *                       trailing incomplete path the reference's samplers drop,
 *                       batch_polopt.py:30-34, vectorized_sampler.py:72-97)
 *   values double[T][n] or NULL: phi(obs, tin) . coeffs, coeffs = double[2*obs_dim+4] in the
 *                       reference's feature order, or zeros when coeffs == NULL (before the first fit)
 *   obs    float[obs_dim][T][n]; obs_dim <= 21. */
int rl_path_scan(int T, int n, int obs_dim, const uint8_t* dones, const float* obs, const double* coeffs,
                 int whole_paths, int32_t* tin, uint8_t* valid, double* values, void* stream);

/* Scratch (device bytes) for rl_sample_stats / rl_lfb_normal_eq. */
size_t rl_process_workspace_bytes(int obs_dim);

/* Number of doubles rl_sample_stats writes. */
int rl_sample_stats_cols(void);

/* One-pass float64 statistics over the valid samples (deterministic two-level reduction):
 *   out[0] count, [1..2] sum / sum^2 of (return - ret_shift), [3..4] of (baseline - ret_shift),
 *   [5..6] of (return - baseline), [7..8] of the advantage, [9] valid paths (tin == 0),
 *   [10..11] sum / sum^2 of (undiscounted path return - und_shift), [12] sum of discounted
 *   returns at path starts, [13..14] sum / sum^2 of the per-path progress, [15] min advantage,
 *   [16] max / [17] min path return, [18] max / [19] min progress.
 * Feeds explained_variance_1d (misc/special.py:51-59), center / shift_advantages
 * (algos/util.py:7-12), the return statistics of base.py:93-103 and the envs' forward-progress
 * diagnostics (swimmer_env.py:48-62).  The shifts only condition the one-pass variances; pass any
 * value near the respective means.
 *   progress: NULL, or a float[T][n_cols] plane x (e.g. one observation component): a path's progress is
 *   x[its last step] - x[its first step]; n_cols = n (envs per row of the [T][n] planes). */
int rl_sample_stats(size_t n_samples, const float* returns, const double* baselines,
                    const float* advantages, const float* undiscounted, const int32_t* tin,
                    const uint8_t* valid, double ret_shift, double und_shift, const float* progress,
                    int n_cols, void* workspace, size_t workspace_bytes, double* out, void* stream);

/* adv_out = valid ? (adv_in - mean) / denom + shift : 0   (algos/util.py:7-12). */
int rl_adv_finish(size_t n_samples, const float* adv_in, const uint8_t* valid, double mean, double denom,
                  double shift, float* adv_out, void* stream);

/* Normal equations of LinearFeatureBaseline.fit (linear_feature_baseline.py:25-36) over the valid
 * samples, float64, without materialising the feature matrix:
 *   out = [ Phi^T Phi (F*F, row-major) | Phi^T returns (F) ],  F = 2*obs_dim + 4. */
int rl_lfb_normal_eq(size_t n_samples, int obs_dim, const float* obs, const int32_t* tin,
                     const float* returns, const uint8_t* valid, void* workspace, size_t workspace_bytes,
                     double* out, int variant, void* stream);   /* variant: 0 = matrix-core kernel (the library's choice),
                                                                 * 1 = register-blocked vector kernel (rl_launch_opts.lfb_valu) */

/* One dense batch for the fused GaussianMLPPolicy update kernels.  Per-sample arrays
 * are planes with the sample axis last (B = n_samples). */
typedef struct rl_policy_batch {
    int32_t n_samples;         /* B */
    int32_t obs_dim, act_dim;  /* Do, Da */
    int32_t hidden0, hidden1;  /* tanh MLP hidden sizes: each 32, 64 or 128 */
    int32_t hidden2;           /* third hidden layer, 0 = two layers */
    float inv_count;           /* 1 / (global number of valid samples) */
    float log_min_std;         /* log_std floor */
    const float* theta;        /* [P] flat params, reference layout (see rl_rollout_args.theta) */
    const float* obs;          /* [Do][B] */
    const float* actions;      /* [Da][B] */
    const float* advantages;   /* [B] */
    const float* old_means;    /* [Da][B]  agent_infos["mean"] */
    const float* old_log_std;  /* [Da]     agent_infos["log_std"] (one constant row) */
    const float* weights;      /* [B] 0/1 validity */
    float* activations;        /* NULL, or rl_policy_activation_bytes() bytes
                                * of device scratch: rl_policy_grad (vpg == 0)
                                * leaves the hidden activations of every sample there and rl_policy_fvp reads them
                                * instead of re-evaluating the forward pass.  The caller guarantees that an FVP call
                                * which passes the buffer uses the same obs / theta as the gradient call that filled
                                * it -- in ConjugateGradientOptimizer.optimize (conjugate_gradient_optimizer.py:
                                * 229-296) the 11 f_Hx_plain evaluations follow f_grad at the same parameters. */
    float kl_penalty;          /* rl_policy_grad / rl_policy_grad_loss: the gradient becomes that of
                                *   (-sum_b w {lr | logp} adv + kl_penalty * sum_b w KL(old_b || new_b)) * inv_count,
                                * the penalised objective PenaltyLbfgsOptimizer hands to L-BFGS
                                * (rllab/optimizers/penalty_lbfgs_optimizer.py:66-79): PPO on NPO's surrogate
                                * (vpg == 0, rllab/algos/ppo.py:8-22), GaussianMLPRegressor on the log-likelihood under
                                * its mean-KL trust region (vpg != 0, gaussian_mlp_regressor.py:126-143).  0 = none. */
    int32_t activation;        /* RL_ACT_TANH (policies) or RL_ACT_RECTIFY (GaussianMLPRegressor's default hidden
                                * nonlinearity, gaussian_mlp_regressor.py:31; loss and vpg gradient only, act_dim 1,
                                * hidden 32x32) */
    const rl_launch_opts* opts;/* host; NULL = the library's own kernel choice (rl_policy_fvp_variant reports it) */
    int32_t layer_activations; /* 0: every hidden layer uses `activation`.  Else two bits per hidden layer l (bits 2l, 2l+1)
                                * holding rl_activation + 1 (0 in a field = `activation`): a GaussianMLPPolicy with a rectify
                                * hidden_nonlinearity, or with ONE hidden layer -- its kernel copy is the two-layer net whose
                                * second layer is the identity (W1 = I, b1 = 0, RL_ACT_IDENTITY) -- runs on the equal-width
                                * two-layer kernels (rllab/policies/gaussian_mlp_policy.py:21-69, rllab/core/network.py:36-101
                                * take any hidden_sizes / nonlinearity).  Every pass of those kernels takes the codes; the
                                * split-operand products, the three-layer / wide kernels and rl_mlp_* take tanh layers only. */
    int32_t reserved_pad;
    const int32_t* gate;       /* NULL, or a device word: rl_policy_loss_kl returns without evaluating anything when
                                * *gate != 0 at the time the launch RUNS (its out4 is then unspecified).  The word is
                                * rl_line_search_decide's "a candidate has been accepted" flag: the loss passes of
                                * the candidates enqueued behind an accepted one cost a launch, not a pass over the
                                * batch.  Ignored by every other entry point. */
    float* obs_absmax;         /* NULL, or one device float.  With `activations` set, rl_policy_grad / rl_policy_grad_loss
                                * (vpg == 0) leave max_{d,b} |obs[d][b]| of the batch there (no extra pass: the gradient pass holds the
                                * observations), and rl_policy_fvp -- under the same guarantee as for `activations` -- takes
                                * it as the bound from which the two-way f16 split product (policy_splith_kernels.hip)
                                * scales its operands; a larger value than the true maximum is valid (coarser scales), a
                                * smaller one is not; the scales also assume weights in [0, 1] (they are 0 / 1 validity).  NULL: the three-way
                                * bf16 split product, which needs no bound. */
} rl_policy_batch;

enum rl_activation { RL_ACT_TANH = 0, RL_ACT_RECTIFY = 1, RL_ACT_IDENTITY = 2 };

/* Scratch the three calls below need (device memory, caller-owned, reusable). */
size_t rl_policy_workspace_bytes(int obs_dim, int act_dim, int hidden0, int hidden1, int hidden2);

/* Size of rl_policy_batch.activations for a batch of n_samples (one float per sample and hidden unit, padded to
 * 32-sample tiles); 0 for hidden sizes the kernels do not take (each layer 32 / 64 / 128 units, hidden2 = 0: two layers). */
size_t rl_policy_activation_bytes(int n_samples, int hidden0, int hidden1, int hidden2);

/* out4 (device, 4 doubles) = [ sum_b w lr adv, sum_b w KL, sum_b w logp adv, max_b KL ] at theta:
 * surrogate loss = -out4[0]*inv_count, mean KL = out4[1]*inv_count.  Replaces the compiled
 * f_loss / f_constraint / f_loss_constraint of ConjugateGradientOptimizer
 * (rllab/optimizers/conjugate_gradient_optimizer.py:194-215) on NPO's surr_loss / mean_kl
 * (rllab/algos/npo.py:72-82) and VPG's f_kl (rllab/algos/vpg.py:100-107). */
int rl_policy_loss_kl(const rl_policy_batch* batch, void* workspace, size_t workspace_bytes,
                      double* out4, void* stream);

/* grad_out (device, P doubles) = d/dtheta of -sum_b w lr adv * inv_count (vpg == 0, f_grad of
 * conjugate_gradient_optimizer.py:199-203) or of -sum_b w logp adv * inv_count (vpg != 0,
 * rllab/algos/vpg.py:91 through FirstOrderOptimizer, first_order_optimizer.py:63-65). */
int rl_policy_grad(const rl_policy_batch* batch, int vpg, void* workspace, size_t workspace_bytes,
                   double* grad_out, void* stream);

/* rl_policy_grad and rl_policy_loss_kl in ONE pass over the batch: ConjugateGradientOptimizer.optimize evaluates
 * f_loss and f_grad back to back at the same parameters (conjugate_gradient_optimizer.py:247-251), and so does
 * FirstOrderOptimizer around its step (first_order_optimizer.py:96,109, vpg != 0); the forward pass, likelihood
 * ratio and KL of the gradient pass are the ones the loss needs.
 * workspace: rl_policy_workspace_bytes (holds both kinds of partial rows). */
int rl_policy_grad_loss(const rl_policy_batch* batch, int vpg, void* workspace, size_t workspace_bytes,
                        double* grad_out, double* out4, void* stream);

/* fvp_out (device, P doubles) = Fisher-vector product of the mean KL with `vec` [P] at
 * theta_new == theta_old, WITHOUT the reg_coeff*vec term.  Replaces f_Hx_plain of PerlmutterHvp
 * (conjugate_gradient_optimizer.py:27-46).  Uses obs, weights, inv_count, theta only. */
int rl_policy_fvp(const rl_policy_batch* batch, const float* vec, void* workspace,
                  size_t workspace_bytes, double* fvp_out, void* stream);

/* Which arithmetic rl_policy_fvp runs this batch's products in (host query, launches nothing):
 *   0  f32 matrix instructions (v_mfma_f32_32x32x2_f32 / 16x16x4_f32): bit-identical with or without `activations`;
 *   1  bf16 matrix instructions on three-way split operands with f32 accumulation (six cross terms per product, the
 *      dropped ones at most 2^-23 of |a b|, 2^-28 in the mean): cached products of two 32-unit tanh layers on a batch of whole 32-sample
 *      tiles.  Same result to f32 rounding, not bit for bit.  RLLAB_FVP_SPLIT=0 in the environment selects 0.
 *   2  the same arithmetic in the cooperative tiling (csrc/policy_csplit_kernels.hip: parts images in LDS, transposing
 *      reads for the sample-axis products): cached products of two or three tanh layers of 32 / 64 / 128 units with a
 *      128-unit layer (RLLAB_FVP_SPLIT=2: every such net that is not all-32) on whole 32-sample tiles.
 *   4  f16 matrix instructions on TWO-way split operands, lo part scaled by 2^11, three cross terms per product with f32
 *      accumulation (csrc/policy_splith_kernels.hip; closer to float64 than an f32 fma chain, tools/ubench/f16_split.hip),
 *      every operand class under a per-launch power-of-two scale with worst-case bounds: the cached products of 1 for the
 *      (32, 32) and (64, 64) shapes, when rl_policy_batch.obs_absmax is set (RLLAB_FVP_SPLIT=5:
 *      variant 1 all the same). */
int rl_policy_fvp_variant(const rl_policy_batch* batch);

/* ---- policies whose log-std is a NETWORK (GaussianMLPPolicy(adaptive_std=True) / std_network=...,
 * rllab/policies/gaussian_mlp_policy.py:60-98; the reference's regression test tests/regression_tests/test_issue_3.py).
 * Mean and log-std networks run as plain functions on planes, the Gaussian head sits between them:
 *   loss / KL        rl_mlp_forward (mean net), rl_mlp_forward (std net), rl_gaussian_head(g = NULL)
 *   gradient         ... rl_gaussian_head(g_mean, g_log_std), rl_mlp_backward x 2; flat gradient = [mean net | std net]
 *   Fisher x vector  rl_mlp_forward with `vec` x 2 (tangents), rl_gaussian_fisher, rl_mlp_backward x 2
 * rl_policy_batch carries the network: n_samples, obs_dim, act_dim, hidden0..2 (two or three tanh layers of 32 / 64 /
 * 128 units, hidden2 = 0 for two), theta = [W0,b0,W1,b1,(W2,b2,)Wout,bout | act_dim unused floats] (the policy layout
 * with its log_std row ignored), obs, weights; the other fields are not read.  Two equal layers of 32 / 64 units on a
 * HIP-native (obs_dim, act_dim) pair run one wavefront per tile; every other shape runs the cooperative kernels, whose
 * operand images live in the caller's workspace -- rl_mlp_forward_ws (rl_mlp_backward takes one anyway;
 * rl_policy_workspace_bytes sizes it). */

/* out[act_dim][B] = network(obs); with vec (same layout as theta) also dout = d/d eps network_{theta + eps vec}(obs). */
int rl_mlp_forward(const rl_policy_batch* batch, const float* vec, float* out, float* dout, void* stream);
/* The same with a workspace: any network shape of the two kernel families (rl_mlp_forward fails with RL_ERR_ARG for the
 * shapes that need one). */
int rl_mlp_forward_ws(const rl_policy_batch* batch, const float* vec, void* workspace, size_t workspace_bytes, float* out,
                      float* dout, void* stream);

/* grad_out (device, P doubles; the trailing act_dim entries are zero) = d/dtheta sum_b sum_k cotangent[k][b] out_k(b).
 * workspace: rl_policy_workspace_bytes. */
int rl_mlp_backward(const rl_policy_batch* batch, const float* cotangent, void* workspace, size_t workspace_bytes,
                    double* grad_out, void* stream);

/* The diagonal-Gaussian head on planes [act_dim][B] (rllab/distributions/diagonal_gaussian.py:14-69,
 * rllab/algos/npo.py:72-82, rllab/algos/vpg.py:91): out4 as rl_policy_loss_kl; when g_mean / g_log_std are given,
 * the cotangents of (-sum_b w {lr | logp} adv + kl_penalty sum_b w KL) * inv_count on the mean / log-std planes
 * (vpg != 0: logp instead of lr).  log_std is the raw network output, floored at log_min_std here (the floor's
 * zero derivative included).  workspace: rl_gaussian_head_workspace_bytes(). */
size_t rl_gaussian_head_workspace_bytes(void);
int rl_gaussian_head(size_t n_samples, int act_dim, const float* mean, const float* log_std, const float* actions,
                     const float* advantages, const float* old_means, const float* old_log_stds,
                     const float* weights, float inv_count, float log_min_std, int vpg, float kl_penalty,
                     float* g_mean, float* g_log_std, void* workspace, size_t workspace_bytes, double* out4,
                     void* stream);

/* The Fisher metric of the mean KL at old == new, applied to output tangents (it is diagonal in (mean, log_std)):
 *   g_mean = w inv_count dmean 2 / (2 v + 1e-8),   g_log_std = w inv_count dlog_std 4 v (2 v - e) / (2 v + e)^2,
 * v = exp(2 max(log_std, log_min_std)), e = 1e-8; g_log_std = 0 where the floor is active. */
int rl_gaussian_fisher(size_t n_samples, int act_dim, const float* dmean, const float* dlog_std, const float* log_std,
                       const float* weights, float inv_count, float log_min_std, float* g_mean, float* g_log_std,
                       void* stream);

/* Vector algebra of krylov.cg (rllab/misc/krylov.py:7-39) for the TRPO descent direction
 * (conjugate_gradient_optimizer.py:253-256), one launch per iteration, float64 like the reference.
 *   rl_cg_init : x = 0, r = p = b, p32 = (float)p, scal = {r.r, active = 1, 0, 0}
 *   rl_cg_step : given fvp = F p (rl_policy_fvp output, already summed over ranks):
 *                z = fvp + reg_coeff p; v = rdotr / p.z; x += v p; r -= v z; mu = r.r / rdotr;
 *                p = r + mu p; p32 = (float)p; the `rdotr < residual_tol: break` of the reference
 *                freezes x, r, p from then on (scal[1] = 0) instead of returning to the host.
 * n <= 65536; b, x, r, p, fvp: double[n]; p32: float[n]; scal: double[4] = {rdotr, active,
 * last p.Ap, steps taken}.  All device pointers. */
int rl_cg_init(int n, const double* b, double* x, double* r, double* p, float* p32, double* scal,
               void* stream);
int rl_cg_step(int n, const double* fvp, double reg_coeff, double residual_tol, double* x, double* r,
               double* p, float* p32, double* scal, void* stream);

/* rl_policy_fvp (vec = p32) and rl_cg_step in two launches instead of three, for a single rank (no all-reduce sits
 * between the product and the CG algebra): the workgroup that finishes the row reduction of F p last runs the CG
 * iteration.  Same arithmetic in the same order as rl_policy_fvp followed by rl_cg_step.
 *   fvp_scratch  double[P]: receives F p
 *   ticket       one uint32, zero before the first call; every call leaves it zero */
int rl_policy_fvp_cg_step(const rl_policy_batch* batch, void* workspace, size_t workspace_bytes, double reg_coeff,
                          double residual_tol, double* x, double* r, double* p, float* p32, double* scal,
                          double* fvp_scratch, unsigned int* ticket, void* stream);

/* The step ConjugateGradientOptimizer.optimize forms after CG
 * (rllab/optimizers/conjugate_gradient_optimizer.py:257-262), float64, one launch:
 *   xHx = x . (a - b + reg_coeff x);  beta = sqrt(2 max_constraint * (1 / (xHx + 1e-8)))  (NaN -> 1);
 *   step = beta x;  out = {xHx, beta}.
 * Two ways to supply H x = F x + reg_coeff x:
 *   a = F x from rl_policy_fvp (summed over ranks), b = NULL         -- evaluated afresh, as the reference does;
 *   a = CG's right-hand side g, b = CG's residual r, reg_coeff = 0   -- CG's invariant r = g - H x (rl_cg_step
 *                                                                       iterates on H = F + reg I), no extra pass. */
int rl_trpo_step(int n, const double* x, const double* a, const double* b, double reg_coeff,
                 double max_constraint, double* step, double* out, void* stream);

/* One candidate of its backtracking line search (:266-274): theta = (float)(prev - ratio * step),
 * prev: float[n] (the parameters before the update), step: double[n], theta: float[n]. */
int rl_line_search_point(int n, const float* prev, const double* step, double ratio, float* theta,
                         void* stream);

/* The accept test of that line search ON THE DEVICE, so that several candidates (and whatever follows the update) can
 * be enqueued without a host round trip per candidate (conjugate_gradient_optimizer.py:262-274:
 *     for n_iter, ratio in enumerate(backtrack_ratio ** arange(max_backtracks)):
 *         cur_param = prev_param - ratio * flat_descent_step;  set_param_values(cur_param)
 *         loss, constraint_val = f_loss_constraint(...)
 *         if loss < loss_before and constraint_val <= max_constraint_val: break ).
 * One launch, one workgroup, after candidate `candidate`'s rl_policy_loss_kl (+ the gather of its sums over the ranks):
 *   sums    double[rows][4]  the candidate's {sum w lr adv, sum w KL, sum w logp adv, max KL} per rank, rank order
 *   before  double[rows][4]  the same sums at the parameters the search started from
 *   loss = -(sum over rows of column 0) * inv_count, constraint = (sum of column 1) * inv_count -- the arithmetic of
 *   the host path (float64, rows added in rank order), NaN compares false exactly as in the reference
 *   state   double[2 + 4 * K]: state[0] = 1 once a candidate was accepted, state[1] = its index,
 *           state[2 + 4 k .. 5 + 4 k] = candidate k's folded sums (three sums, one max) -- written for every candidate
 *           that was actually evaluated (i.e. while state[0] was still 0 when its decide launch ran)
 *   gate    int32[1]: mirrors state[0] (rl_policy_batch.gate of the later candidates' loss passes)
 * If, after this decision, no candidate has been accepted and next_ratio > 0, the same launch writes the NEXT
 * candidate  theta = (float)(prev - next_ratio * step)  (rl_line_search_point's arithmetic); pass next_ratio = 0 after
 * the last speculative candidate.  The host reads `state` once, after everything is enqueued. */
int rl_line_search_decide(int rows, const double* sums, const double* before, double inv_count, double max_constraint,
                          int candidate, double* state, int32_t* gate, int n, const float* prev, const double* step,
                          double next_ratio, float* theta, void* stream);

/* One Adam step of FirstOrderOptimizer (rllab/optimizers/first_order_optimizer.py:21-22,62-76: lasagne.updates.adam
 * on the flat parameters), float64 arithmetic, in place:
 *   m = beta1 m + (1 - beta1) g;  v = beta2 v + (1 - beta2) g^2;  theta = (float)(theta - a_t m / (sqrt(v) + epsilon))
 * with a_t = lr sqrt(1 - beta2^t) / (1 - beta1^t) formed by the caller.  theta: float[n]; grad, m, v: double[n]. */
int rl_adam_step(int n, float* theta, const double* grad, double* m, double* v, double a_t, double beta1,
                 double beta2, double epsilon, void* stream);

/* ---- one-shot peer all-reduce (one node, <= 8 GPUs, one process per GPU) ------------------------------------
 * The sharded update's only exchange is a sum over ranks of small float64 vectors -- the flat gradient and each
 * Fisher-vector product of CG (the reference is single-process, its counterpart is the sum
 * inside f_grad / f_Hx_plain, rllab/optimizers/conjugate_gradient_optimizer.py:194-215,27-46).  Instead of a
 * host-issued collective per vector, every rank writes its row into every peer's MAILBOX over xGMI, raises a flag,
 * waits for the world's flags in its own mailbox and sums the rows in rank order: one launch on the update's
 * stream, bit-identical on all ranks, no host call between the product and the CG algebra.
 *   rl_peer_mailbox_bytes  size of one rank's mailbox for vectors of up to max_n doubles (0: bad arguments)
 *   rl_peer_alloc / free   device memory that can be exported (fine-grained, zero-filled; the ONE place the library
 *                          allocates: IPC needs an allocation of its own, not a slice of a caching allocator's)
 *   rl_peer_export         64-byte IPC handle of a mailbox (host buffer out) -- ship it to the peers with any host
 *                          channel (torch.distributed.all_gather_object at start-up)
 *   rl_peer_open / close   map a peer's mailbox from its handle
 *   rl_peer_allreduce_sum  data[0..n) <- sum over ranks, in rank order.  mailboxes: HOST array of `world` device
 *                          pointers (entry `rank` = own mailbox, the others = rl_peer_open results); seq = 1, 2, 3 ...
 *                          the same on every rank for the same reduction; err_dev: one device int, zero before the
 *                          first call, set non-zero (1 + missing rank) if a peer's row did not arrive in time.
 * Needs HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC) in the environment of every rank. */
size_t rl_peer_mailbox_bytes(int world, int max_n);
int rl_peer_alloc(size_t bytes, void** dev_ptr_out);
int rl_peer_free(void* dev_ptr);
int rl_peer_export(void* dev_ptr, void* handle_out64_host);
int rl_peer_open(const void* handle64_host, void** dev_ptr_out);
int rl_peer_close(void* dev_ptr);
int rl_peer_allreduce_sum(int n, double* data, int rank, int world, void* const* mailboxes_host, int max_n,
                          uint64_t seq, int* err_dev, int64_t spin_limit, void* stream);   /* spin_limit: 0 = give up on a
                                                                 * silent peer after 10 s of wall clock only; > 0 also after
                                                                 * that many polls (tests of the give-up path) */

/* Debug / test hook: fill out[4*count] with Philox4x32-10 blocks for counters
 * (c0 + i, c1, c2, c3), key (k0, k1), i = 0..count-1.  Device buffer. */
int rl_debug_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                    int count, uint32_t* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RLLAB_AMD_H */
