// envs.h -- the set of env dynamics compiled into the engine.
#pragma once
#include "dyn_cartpole.h"

// extra `case` labels for RL_DISPATCH_ENV as envs are added
#define RL_EXTRA_ENV_CASES(CALL)
