// envs.h -- the set of env dynamics compiled into the engine.
#pragma once
#include "dyn_cartpole.h"
#include "dyn_cheetah.h"
#include "dyn_double_pendulum.h"
#include "dyn_hopper.h"
#include "dyn_idp.h"
#include "dyn_swimmer.h"
#include "dyn_walker.h"

// extra `case` labels for RL_DISPATCH_ENV (env_kernels.hip)
#define RL_EXTRA_ENV_CASES(CALL)                                                \
    case RL_ENV_DOUBLE_PENDULUM: { using E = rl::DoublePendulum; return CALL; } \
    case RL_ENV_SWIMMER: { using E = rl::Swimmer; return CALL; }                \
    case RL_ENV_HALF_CHEETAH: { using E = rl::HalfCheetah; return CALL; }       \
    case RL_ENV_CARTPOLE_SWINGUP: { using E = rl::CartpoleSwingup; return CALL; } \
    case RL_ENV_WALKER2D: { using E = rl::Walker2D; return CALL; }              \
    case RL_ENV_HOPPER: { using E = rl::Hopper; return CALL; }                  \
    case RL_ENV_INVERTED_DOUBLE_PENDULUM: { using E = rl::InvertedDoublePendulum; return CALL; }

// extra cases for the host oracle dispatch (oracle/env_host.cpp)
#define ORACLE_EXTRA_ENV_CASES(FN, ...)                 \
    case 1: return FN<rl::DoublePendulum>(__VA_ARGS__); \
    case 2: return FN<rl::Swimmer>(__VA_ARGS__);        \
    case 3: return FN<rl::HalfCheetah>(__VA_ARGS__);    \
    case 4: return FN<rl::CartpoleSwingup>(__VA_ARGS__); \
    case 5: return FN<rl::Walker2D>(__VA_ARGS__);       \
    case 6: return FN<rl::Hopper>(__VA_ARGS__);         \
    case 7: return FN<rl::InvertedDoublePendulum>(__VA_ARGS__);
#define ORACLE_EXTRA_ENV_CASES_R(FN, R, ...)               \
    case 1: return FN<rl::DoublePendulum, R>(__VA_ARGS__); \
    case 2: return FN<rl::Swimmer, R>(__VA_ARGS__);        \
    case 3: return FN<rl::HalfCheetah, R>(__VA_ARGS__);    \
    case 4: return FN<rl::CartpoleSwingup, R>(__VA_ARGS__); \
    case 5: return FN<rl::Walker2D, R>(__VA_ARGS__);       \
    case 6: return FN<rl::Hopper, R>(__VA_ARGS__);         \
    case 7: return FN<rl::InvertedDoublePendulum, R>(__VA_ARGS__);
