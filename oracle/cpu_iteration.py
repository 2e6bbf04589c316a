"""CPU port of the rest of the reference iteration -- process_samples and the TRPO update -- timed on paths
of the CPU sampler (``cpu_baseline.iteration`` of bench.py; TEST INFRASTRUCTURE, never imported by the product).

  process : oracle/np_reference.process_samples       = BaseSampler.process_samples, rllab/sampler/base.py:48-104
            (per-path Python loop, scipy-style discount_cumsum, LinearFeatureBaseline lstsq)
  update  : oracle/np_reference.cg_optimize            = ConjugateGradientOptimizer.optimize,
            rllab/optimizers/conjugate_gradient_optimizer.py:229-296, driven by float64 torch-CPU closures of the
            reference formulas (npo.py:72-82, diagonal_gaussian.py:14-69, PerlmutterHvp :27-55) where the
            reference calls its compiled Theano functions (Theano is not installable here).
"""
import time

import numpy as np
import torch

from oracle import np_reference as R


def _closures(sizes, obs, act, adv, old_mean, old_log_std):
    def unflatten(flat):
        layers, n = [], 0
        for a, b in zip(sizes[:-1], sizes[1:]):
            W = flat[n:n + a * b].reshape(a, b)
            n += a * b
            layers.append((W, flat[n:n + b]))
            n += b
        return layers, flat[n:n + sizes[-1]]

    def dist(flat):
        layers, log_std = unflatten(flat)
        h = obs
        for i, (W, b) in enumerate(layers):
            h = h @ W + b
            if i < len(layers) - 1:
                h = torch.tanh(h)
        return h, torch.clamp(log_std, min=float(np.log(1e-6))).expand_as(h)

    def logli(x, mean, log_std):
        z = (x - mean) / torch.exp(log_std)
        return -log_std.sum(-1) - 0.5 * (z ** 2).sum(-1) - 0.5 * mean.shape[-1] * np.log(2 * np.pi)

    def surr(flat):
        mean, log_std = dist(flat)
        lr = torch.exp(logli(act, mean, log_std) - logli(act, old_mean, old_log_std))
        return -(lr * adv).mean()

    def kl(flat):
        mean, log_std = dist(flat)
        os_, ns = torch.exp(old_log_std), torch.exp(log_std)
        num = (old_mean - mean) ** 2 + os_ ** 2 - ns ** 2
        return (num / (2 * ns ** 2 + 1e-8) + log_std - old_log_std).sum(-1).mean()
    return surr, kl


def timed_process_and_update(paths, theta, hidden, discount=0.99, gae_lambda=1.0, step_size=0.01, max_samples=100000):
    """Seconds of process_samples and of one TRPO update on (a prefix of) ``paths``; returns a dict."""
    keep, n = [], 0
    for p in paths:
        keep.append(p)
        n += len(p["rewards"])
        if n >= max_samples:
            break
    t0 = time.time()
    samples, _ = R.process_samples(keep, R.LinearFeatureBaseline(), discount, gae_lambda)
    t_process = time.time() - t0
    f64 = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64))
    obs, act, adv = f64(samples["observations"]), f64(samples["actions"]), f64(samples["advantages"])
    om, ols = f64(samples["agent_infos"]["mean"]), f64(samples["agent_infos"]["log_std"])
    sizes = (obs.shape[1],) + tuple(hidden) + (act.shape[1],)
    surr, kl = _closures(sizes, obs, act, adv, om, ols)
    as_t = lambda th: torch.as_tensor(th, dtype=torch.float64)
    f_loss = lambda th: float(surr(as_t(th)))
    f_kl = lambda th: float(kl(as_t(th)))

    def f_grad(th):
        t = as_t(th).requires_grad_(True)
        return torch.autograd.grad(surr(t), t)[0].numpy()

    def f_hx(th, x):
        t = as_t(th).requires_grad_(True)
        g = torch.autograd.grad(kl(t), t, create_graph=True)[0]
        return torch.autograd.grad((g * as_t(x)).sum(), t)[0].numpy()
    # [100k x 32] float64 matmuls: more threads than this only add synchronisation (measured on the 256-core bench host)
    threads_before = torch.get_num_threads()
    torch.set_num_threads(min(16, threads_before))
    t0 = time.time()
    _, info = R.cg_optimize(np.asarray(theta, dtype=np.float64).copy(), f_loss, f_grad, f_kl, f_hx, step_size)
    t_update = time.time() - t0
    used = torch.get_num_threads()
    torch.set_num_threads(threads_before)
    return dict(samples=int(obs.shape[0]), process_s=t_process, update_s=t_update,
                backtrack_iters=int(info["backtrack_iters"]), torch_threads=used)
