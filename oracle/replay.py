"""Replay checker (test infrastructure): feed the actions a fused GPU rollout
recorded through the host float32 build of the same env dynamics and demand
bit-identical observations, rewards and done flags.

At each path start the host env is reset with the injected draws when the caller has
them (parity mode); otherwise the state is rebuilt from the recorded observation (possible
for Cartpole and Swimmer, whose observation determines the reset state), so the check also
works for production rollouts that drew their resets from the in-kernel Philox stream.
Persisted solver state (Cartpole joint impulses) is carried across resets exactly
as the kernel carries it.
"""
import numpy as np

from oracle import host_env as H


def replay_check(vec_env, traj, max_envs=64, verbose=False, reset_draws=None, action_noise_z=None, obs_noise_z=None):
    """Raises AssertionError on the first mismatching bit.  Returns the number of
    env-steps compared.  Only paths that start at t == 0 or after a recorded done
    are replayed; the first path of each env needs the env's state at the start of
    the rollout, which for reset_at_start rollouts is a reset state.
    The host env runs under the executor's options (``vec_env.cfg_overrides``); with action / observation noise on,
    the N(0,1) draws the rollout was given must be passed here too ([T, Da, N] / [T+1, Do, N])."""
    kind = vec_env.kind
    T, N = traj.T, traj.N
    n_chk = min(N, max_envs)
    obs = traj.obs[:, :, :n_chk].cpu().numpy()
    act = traj.actions[:, :, :n_chk].cpu().numpy()
    rew = traj.rewards[:, :n_chk].cpu().numpy()
    done = traj.dones[:, :n_chk].cpu().numpy()
    compared = 0
    if reset_draws is not None:
        reset_draws = np.asarray(reset_draws, np.float32)   # [T+1][R][N]: slice t = reset before step t
    cfg = {k: v for k, v in getattr(vec_env, "cfg_overrides", {}).items()}
    if action_noise_z is not None:
        action_noise_z = np.asarray(action_noise_z, np.float32)
    if obs_noise_z is not None:
        obs_noise_z = np.asarray(obs_noise_z, np.float32)
    noisy_obs = H.make_cfg(kind, cfg).obs_noise != 0.0
    for n in range(n_chk):
        env = H.HostEnv(kind, np.float32, normalize=vec_env.normalize, cfg=cfg)
        fresh = True
        ts = 0
        o_step = None        # the observation the host's previous step() returned
        for t in range(T):
            if fresh:
                if reset_draws is not None:
                    env.reset(reset_draws[t, :, n], zobs=obs_noise_z[t, :, n] if noisy_obs else None)
                else:
                    assert not noisy_obs, "a noisy observation does not determine the reset state"
                    set_state_from_obs(env, obs[:, t, n])
                fresh = False
                ts = 0
            # the observation recorded at step t carries noise slice t (slice 0 = the first, t = after step t - 1): inside a
            # path it is what the previous step() returned (for the Hopper under the soft-constraint models that carries the
            # step's own constraint forces, which observe() cannot re-derive from the state), at a path start observe()'s
            o_host = o_step if o_step is not None else env.observe(obs_noise_z[t, :, n] if noisy_obs else None)
            assert np.array_equal(o_host.view(np.uint32), obs[:, t, n].view(np.uint32)), \
                "obs mismatch env %d t %d: host %r gpu %r" % (n, t, o_host, obs[:, t, n])
            o_step, r, d = env.step(act[:, t, n], zact=None if action_noise_z is None else action_noise_z[t, :, n],
                                    zobs=obs_noise_z[t + 1, :, n] if noisy_obs else None)
            ts += 1
            if vec_env.max_path_length > 0 and ts >= vec_env.max_path_length:
                d = True
            r = np.float32(r) * np.float32(vec_env.scale_reward)
            assert np.float32(r).view(np.uint32) == rew[t, n].view(np.uint32), \
                "reward mismatch env %d t %d: host %r gpu %r" % (n, t, r, rew[t, n])
            assert bool(d) == bool(done[t, n]), "done mismatch env %d t %d" % (n, t)
            compared += 1
            if d:
                fresh = True
                o_step = None
    if verbose:
        print("replay_check: %d env-steps bit-identical" % compared)
    return compared


def set_state_from_obs(env, o):
    """Write the reset state encoded by a path-start observation into ``env.state``
    (keeping persisted solver state), using the env's own reset() so that derived
    quantities are computed by the dynamics source itself."""
    kind = env.kind
    o = np.asarray(o, np.float32)
    if kind in (0, 4):   # Cartpole / CartpoleSwingup: same bodies, same observation
        # Cartpole.reset maps u -> lo + u*(hi-lo); instead of inverting that affine map
        # in float (not exactly invertible) we reset with any draws and then overwrite
        # the four reset values and the pole centre the same way reset() derives it.
        env.reset(np.zeros(4, np.float32))
        s = env.state
        s[0], s[3], s[8], s[11] = o[0], o[1], o[2], o[3]
        sn, cs = H.sincos_f32(np.array([o[2]], np.float32))
        s[6] = -sn[0] * np.float32(0.5)
        s[7] = np.float32(0.86602540378443864676) + cs[0] * np.float32(0.5)
        if env.cfg.flags & H.CFG_POLE_FOLLOWS_CART:
            s[6] = np.float32(s[6] + np.float32(o[0]))
        return
    if kind == 2:  # Swimmer: obs[:10] = (qpos, qvel) is the whole state
        env.state[:] = o[:10]
        return
    raise NotImplementedError("set_state_from_obs: env kind %d" % kind)
