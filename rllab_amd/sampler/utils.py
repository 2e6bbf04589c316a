"""``rollout`` -- one episode of a single Python env with a single policy
(mirrors rllab/sampler/utils.py:6-43).  Kept for arbitrary user envs / policies
and for ``sim_policy``-style evaluation; HIP-native envs are sampled by the
lock-step ``VectorizedSampler`` instead."""
import numpy as np

from rllab_amd.misc import tensor_utils


def rollout(env, agent, max_path_length=np.inf, animated=False, speedup=1, always_return_paths=False):
    observations, actions, rewards, agent_infos, env_infos = [], [], [], [], []
    o = env.reset()
    agent.reset()
    path_length = 0
    if animated:
        env.render()
    while path_length < max_path_length:
        a, agent_info = agent.get_action(o)
        next_o, r, d, env_info = env.step(a)
        observations.append(env.observation_space.flatten(o))
        rewards.append(r)
        actions.append(env.action_space.flatten(a))
        agent_infos.append(agent_info)
        env_infos.append(env_info)
        path_length += 1
        if d:
            break
        o = next_o
        if animated:
            env.render()
    if animated and not always_return_paths:
        return
    return dict(
        observations=tensor_utils.stack_tensor_list(observations),
        actions=tensor_utils.stack_tensor_list(actions),
        rewards=tensor_utils.stack_tensor_list(rewards),
        agent_infos=tensor_utils.stack_tensor_dict_list(agent_infos),
        env_infos=tensor_utils.stack_tensor_dict_list(env_infos),
    )


def truncate_paths(paths, max_samples):
    """Keep whole paths while the total stays >= max_samples, then shorten the last
    one so the total is exactly max_samples (mirrors
    rllab/sampler/parallel_sampler.py:129-155; pinned by tests/test_sampler.py)."""
    paths = list(paths)
    total = sum(len(p["rewards"]) for p in paths)
    while len(paths) > 0 and total - len(paths[-1]["rewards"]) >= max_samples:
        total -= len(paths.pop(-1)["rewards"])
    if len(paths) > 0:
        last = paths.pop(-1)
        keep = len(last["rewards"]) - (total - max_samples)
        out = dict()
        for k, v in last.items():
            if k in ("observations", "actions", "rewards"):
                out[k] = tensor_utils.truncate_tensor_list(v, keep)
            elif k in ("env_infos", "agent_infos"):
                out[k] = tensor_utils.truncate_tensor_dict(v, keep)
            else:
                raise NotImplementedError
        paths.append(out)
    return paths
