"""Single-env sampling helpers for arbitrary Python envs / policies and for ``sim_policy``-style
evaluation: ``rollout`` (API of rllab/sampler/utils.py:6-43) and ``truncate_paths``
(rllab/sampler/parallel_sampler.py:129-155).  HIP-native envs are sampled by the lock-step
``VectorizedSampler`` instead."""
import numpy as np

from rllab_amd.misc import tensor_utils


def rollout(env, agent, max_path_length=np.inf, animated=False, speedup=1, always_return_paths=False):
    """One episode: reset, then act / step until ``done`` or ``max_path_length`` steps.  Returns the path
    dict (observations, actions, rewards, agent_infos, env_infos stacked over time), or None for an
    animated run unless ``always_return_paths``."""
    obs_space, act_space = env.observation_space, env.action_space
    transitions = []                      # (flat obs, flat action, reward, agent_info, env_info)
    obs = env.reset()
    agent.reset()
    show = env.render if animated else (lambda: None)
    show()
    while len(transitions) < max_path_length:
        action, agent_info = agent.get_action(obs)
        step = env.step(action)
        transitions.append((obs_space.flatten(obs), act_space.flatten(action), step[1], agent_info, step[3]))
        if step[2]:
            break
        obs = step[0]
        show()
    if animated and not always_return_paths:
        return None
    o, a, r, ai, ei = (list(col) for col in zip(*transitions)) if transitions else ([], [], [], [], [])
    return dict(observations=tensor_utils.stack_tensor_list(o), actions=tensor_utils.stack_tensor_list(a),
                rewards=tensor_utils.stack_tensor_list(r), agent_infos=tensor_utils.stack_tensor_dict_list(ai),
                env_infos=tensor_utils.stack_tensor_dict_list(ei))


_ARRAY_KEYS = ("observations", "actions", "rewards")
_DICT_KEYS = ("env_infos", "agent_infos")


def truncate_paths(paths, max_samples):
    """Trim a list of paths to exactly ``max_samples`` samples: drop whole trailing paths while the rest still
    covers the budget, then cut the new last path (pinned by the reference's tests/test_sampler.py:
    130 -> 100 + 30)."""
    paths = list(paths)
    lengths = [len(p["rewards"]) for p in paths]
    total = sum(lengths)
    while paths and total - lengths[-1] >= max_samples:
        total -= lengths.pop()
        paths.pop()
    if not paths:
        return paths
    keep = lengths[-1] - (total - max_samples)
    last, cut = paths.pop(), dict()
    for key, val in last.items():
        if key in _ARRAY_KEYS:
            cut[key] = tensor_utils.truncate_tensor_list(val, keep)
        elif key in _DICT_KEYS:
            cut[key] = tensor_utils.truncate_tensor_dict(val, keep)
        else:
            raise NotImplementedError("truncate_paths: unexpected path key %r" % key)
    return paths + [cut]
