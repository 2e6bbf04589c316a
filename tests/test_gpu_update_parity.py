"""GPU parity of the fused update kernels (C ABI: rl_policy_loss_kl / rl_policy_grad /
rl_policy_fvp) against float64 torch autograd of the reference formulas
(npo.py:72-82, diagonal_gaussian.py:14-69, PerlmutterHvp), and of the device CG /
line-search control flow against the numpy oracle (oracle/np_reference.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(4, 1, 32), (11, 1, 32), (13, 2, 32), (13, 2, 64), (20, 3, 32), (20, 6, 32), (20, 6, 64), (21, 6, 64)]


def _policy(do, da, h, seed=0):
    from rllab_amd.envs.env_spec import EnvSpec
    from rllab_amd.policies.gaussian_mlp_policy import GaussianMLPPolicy
    from rllab_amd.spaces import Box
    np.random.seed(seed)
    spec = EnvSpec(Box(-np.ones(do), np.ones(do)), Box(-np.ones(da), np.ones(da)))
    pol = GaussianMLPPolicy(spec, hidden_sizes=(h, h))
    # non-trivial biases / log_std so every gradient block is exercised
    theta = pol.get_param_values()
    theta += 0.1 * np.random.randn(theta.size)
    pol.set_param_values(theta)
    return pol


def _inputs(pol, B, seed=1, ragged=True, old_equals_new=False):
    rng = np.random.RandomState(seed)
    dev = pol.flat_params.device
    do, da = pol.obs_dim, pol.action_dim
    obs = torch.as_tensor(rng.randn(do, B).astype(np.float32), device=dev)
    with torch.no_grad():
        mean_now = pol.mean_planes(obs.double(), pol.flat_params.double())
    ls = pol.effective_log_std().detach()
    if old_equals_new:
        old_mean = mean_now.float()
        old_ls = ls.clone()
    else:
        old_mean = (mean_now + 0.05 * torch.as_tensor(rng.randn(da, B), device=dev)).float()
        old_ls = ls + torch.as_tensor(0.03 * rng.randn(da).astype(np.float32), device=dev)
    act = (old_mean + torch.exp(old_ls)[:, None] * torch.as_tensor(rng.randn(da, B).astype(np.float32), device=dev))
    adv = torch.as_tensor(rng.randn(B).astype(np.float32), device=dev)
    w = torch.ones(B, dtype=torch.float32, device=dev)
    if ragged:
        w[torch.as_tensor(rng.rand(B) < 0.1, device=dev)] = 0.0
        w[0] = 1.0          # never an all-masked batch (1 / count would be inf)
    inv = 1.0 / w.double().sum()
    return (obs, act, adv, old_mean, old_ls.reshape(-1, 1), w, inv)


def _closures(pol):
    from rllab_amd.algos.trpo import TRPO

    class _A(object):
        pass
    dist = pol.distribution

    def surr(flat, obs, act, adv, om, ols, w, inv):
        new = pol.dist_info_planes(obs.double(), flat.double())
        lr = dist.likelihood_ratio_sym(act.double(), dict(mean=om.double(), log_std=ols.double()), new, axis=0)
        return -(lr * adv.double() * w.double()).sum() * inv

    def kl(flat, obs, act, adv, om, ols, w, inv):
        new = pol.dist_info_planes(obs.double(), flat.double())
        k = dist.kl_sym(dict(mean=om.double(), log_std=ols.double()), new, axis=0)
        return (k * w.double()).sum() * inv

    def vpg(flat, obs, act, adv, om, ols, w, inv):
        new = pol.dist_info_planes(obs.double(), flat.double())
        ll = dist.log_likelihood_sym(act.double(), new, axis=0)
        return -(ll * adv.double() * w.double()).sum() * inv
    return surr, kl, vpg


@pytest.mark.parametrize("do,da,h", SHAPES)
@pytest.mark.parametrize("B", [1, 63, 1000, 70001])
def test_loss_kl_grad_vs_float64_autograd(do, da, h, B):
    pol = _policy(do, da, h)
    ops = pol.fused_ops()
    assert ops is not None
    inp = _inputs(pol, B)
    surr, kl, vpg = _closures(pol)
    flat64 = pol.flat_params.detach().double().requires_grad_(True)
    l64, k64, v64 = surr(flat64, *inp), kl(flat64, *inp), vpg(flat64, *inp)
    s = ops.loss_stats(inp)
    assert abs(float(-s[0]) - float(l64)) <= 2e-5 * max(1.0, abs(float(l64)))
    assert abs(float(s[1]) - float(k64)) <= 2e-5 * max(1e-2, abs(float(k64)))
    assert abs(float(-s[2]) - float(v64)) <= 2e-5 * max(1.0, abs(float(v64)))
    g64 = torch.autograd.grad(l64, flat64)[0]
    g = ops.loss_grad(inp)
    assert float((g - g64).abs().max()) <= 2e-5 * max(1e-3, float(g64.abs().max()))
    gv64 = torch.autograd.grad(v64, flat64)[0]
    gv = ops.loss_grad(inp, vpg=True)
    assert float((gv - gv64).abs().max()) <= 2e-5 * max(1e-3, float(gv64.abs().max()))


@pytest.mark.parametrize("do,da,h", SHAPES)
def test_fvp_equals_kl_hessian_at_theta_old(do, da, h):
    """F v from the kernel == grad(grad(mean_kl) . v) (PerlmutterHvp) in float64 when the old
    distribution is the current one."""
    pol = _policy(do, da, h)
    ops = pol.fused_ops()
    B = 5000
    inp = _inputs(pol, B, old_equals_new=True)
    _, kl, _ = _closures(pol)
    rng = np.random.RandomState(3)
    # float64 reference: old dist recomputed in float64 so that old == new exactly
    flat64 = pol.flat_params.detach().double().requires_grad_(True)
    with torch.no_grad():
        om64 = pol.mean_planes(inp[0].double(), flat64.detach())
    inp64 = (inp[0], inp[1], inp[2], om64, pol.effective_log_std().detach().double().reshape(-1, 1), inp[5], inp[6])
    g = torch.autograd.grad(kl(flat64, *inp64), flat64, create_graph=True)[0]
    for trial in range(3):
        v = torch.as_tensor(rng.randn(flat64.numel()), device=flat64.device)
        hv64 = torch.autograd.grad((g * v).sum(), flat64, retain_graph=True)[0]
        hv = ops.fvp(inp, v)
        assert float((hv - hv64).abs().max()) <= 5e-5 * float(hv64.abs().max())


@pytest.mark.parametrize("do,da,h", SHAPES)
@pytest.mark.parametrize("B", [1, 63, 1000, 70001])
def test_fvp_on_cached_activations_is_the_same_product(do, da, h, B):
    """rl_policy_grad with batch.activations set leaves the hidden activations in device memory and
    rl_policy_fvp then skips the forward pass: the same gradient and the same product, bit for bit (the
    fragments are stored exactly as the matrix pipe produced them); the cache is dropped as soon as the
    parameters or the batch change."""
    pol = _policy(do, da, h)
    ops = pol.fused_ops()
    inp = _inputs(pol, B, old_equals_new=True)
    rng = np.random.RandomState(5)
    v = torch.as_tensor(rng.randn(pol.flat_params.numel()), device="cuda")
    want_g, want_hv = ops.loss_grad(inp), ops.fvp(inp, v)
    assert ops._acts_tag is None
    g = ops.loss_grad(inp, keep_activations=True)
    assert ops._acts_tag is not None and ops._acts.numel() == ((B + 31) // 32) * 2 * h * 32 * 4
    hv = ops.fvp(inp, v)
    assert torch.equal(g, want_g)
    assert torch.equal(hv, want_hv)
    x, xhx = ops.cg(inp, want_g, 4, 1e-5)
    ops._acts_tag = None
    x2, xhx2 = ops.cg(inp, want_g, 4, 1e-5)
    assert torch.equal(x, x2) and torch.equal(xhx, xhx2)
    # a parameter update invalidates the cache: the next product recomputes the forward pass
    ops.loss_grad(inp, keep_activations=True)
    with torch.no_grad():
        pol.flat_params.add_(0.01)
    hv_new = ops.fvp(inp, v)
    ops._acts_tag = None
    assert torch.equal(hv_new, ops.fvp(inp, v)) and not torch.equal(hv_new, want_hv)
    # VPG gradients never keep activations
    ops.loss_grad(inp, vpg=True, keep_activations=True)
    assert ops._acts_tag is None


def test_fvp_is_symmetric_psd():
    pol = _policy(13, 2, 32)
    ops = pol.fused_ops()
    inp = _inputs(pol, 4096, old_equals_new=True)
    rng = np.random.RandomState(0)
    n = pol.flat_params.numel()
    u = torch.as_tensor(rng.randn(n), device="cuda")
    v = torch.as_tensor(rng.randn(n), device="cuda")
    fu, fv = ops.fvp(inp, u), ops.fvp(inp, v)
    assert abs(float(u.dot(fv)) - float(v.dot(fu))) <= 1e-4 * abs(float(u.dot(fv)))
    assert float(u.dot(fu)) > 0 and float(v.dot(fv)) > 0


def test_log_std_floor_blocks_its_gradient():
    pol = _policy(4, 1, 32)
    theta = pol.get_param_values()
    theta[-1] = np.log(1e-8)   # below min_std = 1e-6 (tests/algos/test_trpo.py of the reference)
    pol.set_param_values(theta)
    ops = pol.fused_ops()
    inp = _inputs(pol, 512, old_equals_new=True)
    g = ops.loss_grad(inp)
    assert torch.isfinite(g).all() and float(g[-1]) == 0.0
    hv = ops.fvp(inp, torch.ones_like(g))
    assert torch.isfinite(hv).all() and float(hv[-1]) == 0.0


@pytest.mark.parametrize("algo", ["trpo", "tnpg"])
def test_trpo_step_fused_matches_oracle_control_flow(quiet_logger, algo):
    """One ConjugateGradientOptimizer.optimize with the fused kernels vs the numpy restatement
    of the reference control flow (oracle/np_reference.cg_optimize) driven by float64 torch
    closures: same accepted step (parameters within 1e-4 relative of the step norm).  "tnpg": the optimizer as
    rllab/algos/tnpg.py:16-21 configures it (max_backtracks = 1: the full natural-gradient step, taken once)."""
    from oracle import np_reference as R
    from rllab_amd.optimizers.conjugate_gradient_optimizer import ConjugateGradientOptimizer
    from rllab_amd.algos.tnpg import TNPG
    from rllab_amd.algos.npo import pick_optimizer
    pol = _policy(13, 2, 32)
    inp = _inputs(pol, 20000, old_equals_new=True, ragged=True)
    surr, kl, _ = _closures(pol)
    theta0 = pol.get_param_values()
    dev = pol.flat_params.device

    def as_t(th):
        return torch.as_tensor(th, dtype=torch.float64, device=dev)
    f_loss = lambda th: float(surr(as_t(th), *inp))
    f_kl = lambda th: float(kl(as_t(th), *inp))

    def f_grad(th):
        t = as_t(th).requires_grad_(True)
        return torch.autograd.grad(surr(t, *inp), t)[0].cpu().numpy()

    def f_hx(th, x):
        t = as_t(th).requires_grad_(True)
        g = torch.autograd.grad(kl(t, *inp), t, create_graph=True)[0]
        return torch.autograd.grad((g * as_t(x)).sum(), t)[0].cpu().numpy()
    if algo == "tnpg":
        want, info = R.cg_optimize(theta0.copy(), f_loss, f_grad, f_kl, f_hx, 0.01, max_backtracks=1)
        opt = pick_optimizer(None, None, ConjugateGradientOptimizer, max_backtracks=1)    # what TNPG.__init__ builds
        assert opt._max_backtracks == 1
    else:
        want, info = R.cg_optimize(theta0.copy(), f_loss, f_grad, f_kl, f_hx, 0.01)
        opt = ConjugateGradientOptimizer()
    opt.update_opt(loss=surr, target=pol, leq_constraint=(kl, 0.01), fused=pol.fused_ops())
    opt.optimize(inp)
    got = pol.get_param_values()
    step = np.abs(want - theta0).max()
    if algo == "trpo":
        assert step > 0 and not info["rejected"]
    if info["rejected"]:                       # the single full step broke the constraint: both sides restore theta
        assert np.array_equal(got, theta0)
    else:
        assert np.abs(got - want).max() <= 2e-3 * step
    assert opt.last_backtrack_iters == info["backtrack_iters"]


def test_device_cg_matches_krylov_and_oracle():
    """rl_cg_init / rl_cg_step + rl_policy_fvp (FusedGaussianMLPOps.cg) == krylov.cg on the same
    operator == the numpy restatement of the reference krylov.cg (oracle) on the float64
    autograd Hessian-vector product."""
    from oracle import np_reference as R
    from rllab_amd.misc import krylov
    pol = _policy(13, 2, 32)
    ops = pol.fused_ops()
    inp = _inputs(pol, 30000, old_equals_new=True)
    rng = np.random.RandomState(5)
    n = pol.flat_params.numel()
    g = torch.as_tensor(rng.randn(n) * 1e-2, device="cuda")
    reg = 1e-5
    x_dev, xHx = ops.cg(inp, g, 10, reg)
    hx = ops.hvp_approach()
    hx.update_opt(None, pol, None, reg)
    f = hx.build_eval(inp)
    x_t = krylov.cg(f, g, cg_iters=10)
    assert float((x_dev - x_t).abs().max()) <= 1e-9 * float(x_t.abs().max())
    assert abs(float(xHx) - float(x_t.dot(f(x_t)))) <= 1e-6 * abs(float(xHx))
    # oracle: reference CG loop in numpy on the float64 autograd HVP
    _, kl, _ = _closures(pol)
    flat64 = pol.flat_params.detach().double().requires_grad_(True)
    with torch.no_grad():
        om64 = pol.mean_planes(inp[0].double(), flat64.detach())
    inp64 = (inp[0], inp[1], inp[2], om64, pol.effective_log_std().detach().double().reshape(-1, 1), inp[5], inp[6])
    gk = torch.autograd.grad(kl(flat64, *inp64), flat64, create_graph=True)[0]

    def f_Ax(v):
        vt = torch.as_tensor(v, device="cuda")
        return (torch.autograd.grad((gk * vt).sum(), flat64, retain_graph=True)[0] + reg * vt).cpu().numpy()
    x_np = R.cg(f_Ax, g.cpu().numpy(), cg_iters=10)
    assert np.abs(x_dev.cpu().numpy() - x_np).max() <= 2e-3 * np.abs(x_np).max()
    # early exit: a tiny right-hand side trips `rdotr < residual_tol` after the first iteration; the
    # device loop freezes x there, exactly where the reference breaks out
    g_small = g * 1e-6
    x1, _ = ops.cg(inp, g_small, 10, reg)
    x1_np = R.cg(f_Ax, g_small.cpu().numpy(), cg_iters=10)
    x1_one = R.cg(f_Ax, g_small.cpu().numpy(), cg_iters=1)
    assert np.array_equal(x1_np, x1_one)                      # the oracle did stop after one iteration
    assert np.abs(x1.cpu().numpy() - x1_np).max() <= 2e-3 * np.abs(x1_np).max()


def test_trpo_step_and_line_search_point_kernels():
    """rl_trpo_step / rl_line_search_point == the float64 formulas of
    conjugate_gradient_optimizer.py:257-274 (numpy restatement), including the NaN -> 1 rule."""
    from rllab_amd import _lib
    rng = np.random.RandomState(3)
    for n in (1, 7, 1572, 5900):
        x = rng.randn(n)
        fx = 0.3 * x + 0.01 * rng.randn(n)
        reg, delta = 1e-5, 0.01
        xd, fd = (torch.as_tensor(a, device="cuda") for a in (x, fx))
        step = torch.empty(n, dtype=torch.float64, device="cuda")
        out = torch.empty(2, dtype=torch.float64, device="cuda")
        _lib.check(_lib.lib.rl_trpo_step(n, _lib.ptr(xd), _lib.ptr(fd), None, reg, delta, _lib.ptr(step), _lib.ptr(out),
                                         _lib.stream_ptr()))
        xHx = float(x.dot(fx + reg * x))
        beta = np.sqrt(2.0 * delta * (1.0 / (xHx + 1e-8)))
        got = out.cpu().numpy()
        assert abs(got[0] - xHx) <= 1e-12 * max(1.0, abs(xHx))
        assert abs(got[1] - beta) <= 1e-12 * beta
        assert np.allclose(step.cpu().numpy(), beta * x, rtol=1e-12, atol=0)
        prev = rng.randn(n).astype(np.float32)
        pd = torch.as_tensor(prev, device="cuda")
        theta = torch.empty(n, dtype=torch.float32, device="cuda")
        for ratio in (1.0, 0.8, 0.8 ** 7):
            _lib.check(_lib.lib.rl_line_search_point(n, _lib.ptr(pd), _lib.ptr(step), ratio, _lib.ptr(theta),
                                                     _lib.stream_ptr()))
            want = (prev.astype(np.float64) - ratio * step.cpu().numpy()).astype(np.float32)
            assert np.array_equal(theta.cpu().numpy(), want)
    # negative curvature -> sqrt of a negative number -> NaN -> step size 1 (reference :260-261)
    x = np.ones(4)
    xd, fd = torch.as_tensor(x, device="cuda"), torch.as_tensor(-x, device="cuda")
    step = torch.empty(4, dtype=torch.float64, device="cuda")
    out = torch.empty(2, dtype=torch.float64, device="cuda")
    _lib.check(_lib.lib.rl_trpo_step(4, _lib.ptr(xd), _lib.ptr(fd), None, 0.0, 0.01, _lib.ptr(step), _lib.ptr(out),
                                     _lib.stream_ptr()))
    assert float(out[1]) == 1.0 and np.array_equal(step.cpu().numpy(), x)


@pytest.mark.parametrize("case", ["plain", "many_backtracks", "all_rejected", "padded_net", "hook"])
def test_line_search_decided_on_the_device_is_the_host_search(quiet_logger, case):
    """rl_line_search_decide (K = 3 candidates enqueued without a host read, later passes gated off) against the host
    loop of conjugate_gradient_optimizer.py:262-296: the same accepted candidate, bit-identical parameters, the same
    ``backtrack_iters``, the same LossAfter / MeanKL, also when more than K candidates are needed (the host loop takes
    over at candidate K), when every candidate is rejected (parameters restored), and for a zero-padded layout (the
    kernels' parameter copy has to follow each candidate).  ``hook``: the callback that queues the next rollout runs
    exactly once, after the search is enqueued, and sees the FINAL parameters."""
    from rllab_amd.optimizers.conjugate_gradient_optimizer import ConjugateGradientOptimizer
    from rllab_amd.envs.env_spec import EnvSpec
    from rllab_amd.policies.gaussian_mlp_policy import GaussianMLPPolicy
    from rllab_amd.spaces import Box

    def run(n_device):
        np.random.seed(4)
        if case == "padded_net":
            spec = EnvSpec(Box(-np.ones(13), np.ones(13)), Box(-np.ones(2), np.ones(2)))
            pol = GaussianMLPPolicy(spec, hidden_sizes=(24, 20))
            pol.set_param_values(pol.get_param_values() + 0.1 * np.random.randn(pol.get_param_values().size))
        else:
            pol = _policy(13, 2, 32)
        inp = _inputs(pol, 20000, old_equals_new=True, ragged=True)
        surr, kl, _ = _closures(pol)
        kw = {}
        delta = 0.01
        if case == "all_rejected":
            kw = dict(backtrack_ratio=0.999, max_backtracks=5)
        opt = ConjugateGradientOptimizer(**kw)
        opt._device_line_search = n_device
        ops = pol.fused_ops()
        assert ops is not None
        opt.update_opt(loss=surr, target=pol, leq_constraint=(kl, delta), fused=ops)
        if case in ("many_backtracks", "all_rejected"):
            # inflate the initial step threefold: KL ~ 9 delta ratio^2, so 0.8^k has to fall below 1/3 (k = 5) -- or never does
            orig = ops.cg_step_vector

            def big(*a, **k):
                step, stats = orig(*a, **k)
                return step * 3.0, stats
            ops.cg_step_vector = big
        seen = []
        if case == "hook":
            opt._after_enqueue = lambda: seen.append(pol.flat_params.detach().clone())
        theta0 = pol.flat_params.detach().clone()
        opt.optimize(inp)
        after = pol.flat_params.detach().clone()
        out = dict(after=after, n=opt.last_backtrack_iters, before=opt.last_before, loss=opt.loss(inp),
                   kl=opt.constraint_val(inp), moved=not torch.equal(after, theta0), seen=seen)
        return out
    dev, host = run(3), run(0)
    assert torch.equal(dev["after"], host["after"])
    assert dev["n"] == host["n"] and dev["before"] == host["before"]
    assert dev["loss"] == host["loss"] and dev["kl"] == host["kl"]
    if case == "plain":
        assert dev["moved"] and dev["n"] <= 2
    if case == "many_backtracks":
        assert dev["moved"] and dev["n"] >= 3, dev["n"]           # the host loop had to take over behind the device's three
    if case == "all_rejected":
        assert not dev["moved"] and dev["n"] == 4
    if case == "hook":
        assert len(dev["seen"]) == 1 and torch.equal(dev["seen"][0], dev["after"]) and host["seen"] == []


def test_line_search_decide_kernel_by_hand():
    """rl_line_search_decide on hand-made sums: rows folded in rank order, NaN never accepted, `<` on the loss and `<=`
    on the constraint (conjugate_gradient_optimizer.py:272), later candidates leave an accepted state alone, the next
    candidate's parameters are written only while nothing is accepted."""
    from rllab_amd import _lib
    dev = "cuda"
    n = 5
    prev = torch.arange(n, dtype=torch.float32, device=dev)
    step = torch.ones(n, dtype=torch.float64, device=dev)
    theta = torch.full((n,), -7.0, dtype=torch.float32, device=dev)
    before = torch.tensor([[2.0, 0.0, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0]], dtype=torch.float64, device=dev)   # loss_before = -3 inv
    inv, delta = 0.5, 0.01

    def decide(state, gate, sums, k, nxt):
        t = torch.tensor(sums, dtype=torch.float64, device=dev)
        _lib.check(_lib.lib.rl_line_search_decide(t.shape[0], _lib.ptr(t), _lib.ptr(before), inv, delta, k, _lib.ptr(state),
                                                  _lib.ptr(gate), n, _lib.ptr(prev), _lib.ptr(step), nxt, _lib.ptr(theta),
                                                  _lib.stream_ptr()))
    state = torch.zeros(2 + 4 * 4, dtype=torch.float64, device=dev)
    gate = torch.zeros(1, dtype=torch.int32, device=dev)
    decide(state, gate, [[2.0, 0.0, 1.0, 0.1], [1.0, 0.0, 1.0, 0.3]], 0, 0.5)       # equal loss: not `<` -> rejected
    assert state[0] == 0 and gate[0] == 0 and torch.equal(theta, (prev.double() - 0.5).float())
    assert state[2:6].tolist() == [3.0, 0.0, 2.0, 0.3]
    decide(state, gate, [[float("nan"), 0.0, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0]], 1, 0.25)   # NaN loss: rejected
    assert state[0] == 0 and torch.equal(theta, (prev.double() - 0.25).float())
    decide(state, gate, [[3.0, 0.03, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0]], 2, 0.125)   # better loss, kl = 0.015 > delta: rejected
    assert state[0] == 0 and torch.equal(theta, (prev.double() - 0.125).float())
    decide(state, gate, [[3.0, 0.02, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0]], 3, 0.0625)  # kl = 0.01 == delta: `<=` accepts
    assert state[0] == 1 and state[1] == 3 and gate[0] == 1
    assert torch.equal(theta, (prev.double() - 0.125).float())                      # the accepted point stays
    keep = state.clone()
    decide(state, gate, [[9.0, 0.0, 0.0, 0.0], [9.0, 0.0, 0.0, 0.0]], 2, 0.5)       # behind an accepted candidate: untouched
    assert torch.equal(state.view(torch.int64), keep.view(torch.int64))            # (bitwise: one record holds the NaN)
    assert torch.equal(theta, (prev.double() - 0.125).float())


def test_deferred_reads_give_the_same_numbers(quiet_logger):
    """The optimizer's asynchronous reads change when the host waits, not what it reads:
    last_before == loss / constraint evaluated up front, loss()/constraint_val() after the step come
    from the accepted candidate, and the cached values survive until the parameters change."""
    from rllab_amd.optimizers.conjugate_gradient_optimizer import ConjugateGradientOptimizer
    pol = _policy(13, 2, 32)
    inp = _inputs(pol, 20000, old_equals_new=True, ragged=True)
    surr, kl, _ = _closures(pol)
    opt = ConjugateGradientOptimizer()
    ops = pol.fused_ops()
    opt.update_opt(loss=surr, target=pol, leq_constraint=(kl, 0.01), fused=ops)
    l0, k0 = opt.loss(inp), opt.constraint_val(inp)
    assert abs(l0 - float(surr(pol.flat_params.double(), *inp))) <= 2e-5 * max(1.0, abs(l0))
    ops.release()
    opt.optimize(inp)
    assert opt.last_before == (l0, k0)
    l1, k1 = opt.loss(inp), opt.constraint_val(inp)
    assert l1 < l0 and 0.0 < k1 <= 0.01
    assert abs(l1 - float(surr(pol.flat_params.double(), *inp))) <= 2e-5 * max(1.0, abs(l1))
    assert abs(k1 - float(kl(pol.flat_params.double(), *inp))) <= 2e-5 * max(1e-3, abs(k1))


def test_prefetched_gradient_pass_changes_nothing_but_the_time(quiet_logger):
    """optimizer.prefetch(inputs) launches the gradient pass of the next optimize(inputs) early (process_samples does,
    through NPO.prefetch_update): the update that follows is bit-identical to one without it; a record made for other
    inputs or other parameters is not used."""
    from rllab_amd.optimizers.conjugate_gradient_optimizer import ConjugateGradientOptimizer

    def run(prefetch, disturb=None):
        np.random.seed(4)
        pol = _policy(13, 2, 32)
        inp = _inputs(pol, 20000, old_equals_new=True, ragged=True)
        surr, kl, _ = _closures(pol)
        opt = ConjugateGradientOptimizer()
        ops = pol.fused_ops()
        opt.update_opt(loss=surr, target=pol, leq_constraint=(kl, 0.01), fused=ops)
        if prefetch:
            opt.prefetch(inp)
            assert opt._pre is not None
        if disturb == "params":
            with torch.no_grad():
                pol.flat_params.mul_(1.0)              # same values, a new parameter version: the record is stale
        if disturb == "inputs":
            inp = tuple(t.clone() if torch.is_tensor(t) else t for t in inp)
        opt.optimize(inp)
        assert opt._pre is None
        return pol.get_param_values().copy(), opt.last_before, opt.last_backtrack_iters

    base = run(False)
    for kw in (dict(prefetch=True), dict(prefetch=True, disturb="params"), dict(prefetch=True, disturb="inputs")):
        got = run(**kw)
        assert np.array_equal(got[0], base[0]) and got[1] == base[1] and got[2] == base[2], kw


def test_process_samples_starts_the_update(quiet_logger, monkeypatch):
    """TRPO through BatchPolopt's pieces with and without RLLAB_UPDATE_PREFETCH: identical parameters after three
    iterations, and with it on the optimizer really finds its gradient ready."""
    from rllab.algos.trpo import TRPO
    from rllab.baselines.linear_feature_baseline import LinearFeatureBaseline
    from rllab.envs.mujoco.swimmer_env import SwimmerEnv
    from rllab.envs.normalized_env import normalize
    from rllab.misc import ext
    from rllab.policies.gaussian_mlp_policy import GaussianMLPPolicy
    from rllab_amd.optimizers import conjugate_gradient_optimizer as cgo

    def run(flag, in_training_loop=True):
        monkeypatch.setenv("RLLAB_UPDATE_PREFETCH", flag)
        ext.set_seed(3)
        env = normalize(SwimmerEnv())
        policy = GaussianMLPPolicy(env_spec=env.spec, hidden_sizes=(32, 32))
        algo = TRPO(env=env, policy=policy, baseline=LinearFeatureBaseline(env_spec=env.spec), batch_size=128 * 50,
                    max_path_length=50, n_itr=3, discount=0.99, step_size=0.01, sampler_args=dict(n_envs=128))
        algo.start_worker()
        algo.init_opt()
        used = []
        for itr in range(3):
            paths = algo.sampler.obtain_samples(itr)
            # what BatchPolopt.train_iteration does around its process_samples call: the update follows on this batch
            algo._update_follows = in_training_loop
            sd = algo.sampler.process_samples(itr, paths)
            algo._update_follows = False
            used.append(getattr(algo.optimizer, "_pre", None) is not None)
            algo.log_diagnostics(paths)
            algo.optimize_policy(itr, sd)
        return policy.get_param_values().copy(), used

    p1, used1 = run("1")
    p0, used0 = run("0")
    assert used1 == [True, True, True] and used0 == [False, False, False]
    assert np.array_equal(p1, p0)
    # a caller that processes samples on its own (no optimize_policy promised) pays no extra pass
    p2, used2 = run("1", in_training_loop=False)
    assert used2 == [False, False, False] and np.array_equal(p2, p0)


@pytest.mark.parametrize("hidden", [(32, 32), (100, 50, 25)])
def test_frozen_log_std_updates_on_the_kernels(quiet_logger, hidden):
    """GaussianMLPPolicy(learn_std=False): the log_std row is a parameter but not trainable
    (gaussian_mlp_policy.py:21-58).  The fused passes zero its gradient, the Fisher matrix does not couple it to the
    network at theta_old, so the device CG / line search on the full vector never move it -- and the network moves as
    under the reference's computation in the trainable subspace (the same optimizer on float64 autograd closures)."""
    from rllab_amd.envs.env_spec import EnvSpec
    from rllab_amd.optimizers.conjugate_gradient_optimizer import ConjugateGradientOptimizer
    from rllab_amd.policies.gaussian_mlp_policy import GaussianMLPPolicy
    from rllab_amd.spaces import Box

    def make():
        np.random.seed(7)
        spec = EnvSpec(Box(-np.ones(13), np.ones(13)), Box(-np.ones(2), np.ones(2)))
        pol = GaussianMLPPolicy(spec, hidden_sizes=hidden, learn_std=False, init_std=0.7)
        theta = pol.get_param_values()                       # (trainable and frozen entries alike)
        pol.set_param_values(theta + 0.1 * np.random.randn(theta.size))
        return pol

    results = []
    for fused in (True, False):
        pol = make()
        assert pol._flat_index(trainable=True) is not None
        inp = _inputs(pol, 30000, old_equals_new=True, ragged=True)
        surr, kl, _ = _closures(pol)
        opt = ConjugateGradientOptimizer()
        ops = pol.fused_ops() if fused else None
        if fused:
            assert ops is not None and ops.masks_frozen
        opt.update_opt(loss=surr, target=pol, leq_constraint=(kl, 0.01), fused=ops)
        before = pol.flat_params.detach().clone()
        opt.optimize(inp)
        after = pol.flat_params.detach().clone()
        frozen = torch.ones_like(before, dtype=torch.bool)
        frozen[pol._flat_index(trainable=True)] = False
        assert torch.equal(after[frozen], before[frozen])                 # the log_std row did not move, bit for bit
        assert float((after - before).abs().max()) > 0
        assert 0.0 < opt.constraint_val(inp) <= 0.01
        results.append((after.double().cpu().numpy(), opt.last_backtrack_iters))
    (a, ba), (b, bb) = results
    assert ba == bb
    assert np.abs(a - b).max() <= 2e-4 * max(1.0, np.abs(b).max())


def test_frozen_log_std_with_a_caller_given_hvp(quiet_logger):
    """learn_std=False + an explicit ``hvp_approach`` (conjugate_gradient_optimizer.py:118-140 of the reference lets the
    caller pick FiniteDifferenceHvp): the fused passes still serve loss / gradient, but CG and the line search run in
    the trainable subspace (no device CG), so every vector must have the trainable length -- the configuration the
    round-4 advisor found broken.  The step must land where the default (device CG) path lands."""
    from rllab_amd.envs.env_spec import EnvSpec
    from rllab_amd.optimizers.conjugate_gradient_optimizer import ConjugateGradientOptimizer, FiniteDifferenceHvp
    from rllab_amd.policies.gaussian_mlp_policy import GaussianMLPPolicy
    from rllab_amd.spaces import Box

    def make():
        np.random.seed(7)
        spec = EnvSpec(Box(-np.ones(13), np.ones(13)), Box(-np.ones(2), np.ones(2)))
        pol = GaussianMLPPolicy(spec, hidden_sizes=(32, 32), learn_std=False, init_std=0.7)
        pol.set_param_values(pol.get_param_values() + 0.1 * np.random.randn(pol.get_param_values().size))
        return pol
    out = []
    for hvp in (None, FiniteDifferenceHvp(base_eps=1e-2)):
        pol = make()
        inp = _inputs(pol, 30000, old_equals_new=True, ragged=True)
        surr, kl, _ = _closures(pol)
        opt = ConjugateGradientOptimizer(hvp_approach=hvp)
        ops = pol.fused_ops()
        assert ops.masks_frozen
        opt.update_opt(loss=surr, target=pol, leq_constraint=(kl, 0.01), fused=ops)
        before = pol.flat_params.detach().clone()
        opt.optimize(inp)
        after = pol.flat_params.detach().clone()
        frozen = torch.ones_like(before, dtype=torch.bool)
        frozen[pol._flat_index(trainable=True)] = False
        assert torch.equal(after[frozen], before[frozen]) and float((after - before).abs().max()) > 0
        assert 0.0 < opt.constraint_val(inp) <= 0.01
        out.append((after - before).double().cpu().numpy())
    # finite differences of gradients at float32 parameters are noisy (the reference's are too): same direction, same
    # trust-region length, not the same digits
    a, b = out
    assert a.dot(b) / (np.linalg.norm(a) * np.linalg.norm(b)) > 0.9
    assert 0.5 < np.linalg.norm(b) / np.linalg.norm(a) < 2.0


def test_cg_residual_gives_the_same_step_as_a_fresh_product():
    """d^T H d from CG's invariant (H d = g - r) vs from one more Fisher-vector product (what the reference
    evaluates, conjugate_gradient_optimizer.py:258-260): same step vector to ~1e-6 relative."""
    pol = _policy(13, 2, 32)
    ops = pol.fused_ops()
    inp = _inputs(pol, 50000, old_equals_new=True)
    g = ops.loss_grad(inp, keep_activations=True)
    s_res, st_res = ops.cg_step_vector(inp, g, 10, 1e-5, 0.01, reuse_cg_residual=True)
    s_new, st_new = ops.cg_step_vector(inp, g, 10, 1e-5, 0.01, reuse_cg_residual=False)
    a, b = st_res.cpu().numpy(), st_new.cpu().numpy()
    assert abs(a[0] - b[0]) <= 1e-5 * abs(b[0]) and abs(a[1] - b[1]) <= 1e-5 * b[1]
    assert float((s_res - s_new).abs().max()) <= 1e-5 * float(s_new.abs().max())
    # and the rl_trpo_step (a - b) form itself
    from rllab_amd import _lib
    rng = np.random.RandomState(0)
    n = 100
    x, aa, bb = (torch.as_tensor(rng.randn(n), device="cuda") for _ in range(3))
    step = torch.empty(n, dtype=torch.float64, device="cuda")
    out = torch.empty(2, dtype=torch.float64, device="cuda")
    _lib.check(_lib.lib.rl_trpo_step(n, _lib.ptr(x), _lib.ptr(aa), _lib.ptr(bb), 0.25, 0.01, _lib.ptr(step),
                                     _lib.ptr(out), _lib.stream_ptr()))
    want = float(x.dot(aa - bb + 0.25 * x))
    assert abs(float(out[0]) - want) <= 1e-12 * max(1.0, abs(want))


@pytest.mark.parametrize("do,da,h", [(13, 2, 32), (20, 6, 64), (4, 1, 32)])
def test_grad_pass_also_returns_the_loss_sums(do, da, h):
    """rl_policy_grad_loss == rl_policy_grad + rl_policy_loss_kl at the same point (same per-sample
    expressions; the sums differ at most by their association)."""
    pol = _policy(do, da, h)
    inp = _inputs(pol, 70001)
    a, b = pol.fused_ops(), pol.fused_ops()
    g1 = a.loss_grad(inp, keep_activations=(h == 32), with_loss=True)
    s1 = np.array(a.loss_stats_host(inp))            # served from the gradient pass: no loss pass launched
    s2 = np.array(b.loss_stats_host(inp))
    g2 = b.loss_grad(inp)
    assert torch.equal(g1, g2)
    assert np.all(np.abs(s1[:3] - s2[:3]) <= 1e-12 * np.maximum(1.0, np.abs(s2[:3]))) and s1[3] == s2[3]
    # the VPG gradient pass (d log p) hands back the same sums
    c = pol.fused_ops()
    g3 = c.loss_grad(inp, vpg=True, with_loss=True)
    s3 = np.array(c.loss_stats_host(inp))
    assert torch.equal(g3, b.loss_grad(inp, vpg=True))
    assert np.all(np.abs(s3[:3] - s2[:3]) <= 1e-12 * np.maximum(1.0, np.abs(s2[:3]))) and s3[3] == s2[3]


def test_adam_step_kernel_matches_the_oracle_adam():
    """rl_adam_step (in place on the float32 parameters, float64 moments) against oracle/np_reference.py::Adam -- the
    restatement of lasagne.updates.adam (first_order_optimizer.py:21-22,62-76), itself pinned by hand-derived steps in
    tests/test_oracle_golden.py -- over eight steps with changing gradients: moments to float64 rounding (the kernel
    contracts b1 m + (1 - b1) g into a fused multiply-add), parameters to one float32 unit in the last place.  The
    oracle carries its own float64 parameters forward; the kernel's float32 copy is compared with them every step and
    may only drift by the accumulated roundings of its own stores."""
    from oracle import np_reference as R
    from rllab_amd.optimizers.first_order_optimizer import _Adam
    rng = np.random.RandomState(2)
    n = 1250
    theta0 = rng.randn(n).astype(np.float32)
    tb = torch.as_tensor(theta0, device="cuda")
    b = _Adam(learning_rate=1e-2)
    st = R.Adam([theta0.astype(np.float64)], learning_rate=1e-2)
    want = theta0.astype(np.float64)
    ulp = 0.0
    for k in range(8):
        g = rng.randn(n) * 10.0 ** rng.randint(-6, 2, size=n)       # gradients over eight decades, some below epsilon
        g[rng.rand(n) < 0.02] = 0.0
        want_prev = tb.double().cpu().numpy()                      # step the oracle from the kernel's own float32 point
        want = R.adam_step(st, want_prev, g)
        b.step_in_place(tb, torch.as_tensor(g, device="cuda"))
        assert b.t == st.t == k + 1
        assert np.allclose(b.m.cpu().numpy(), st.m[0], rtol=1e-13, atol=1e-300), k
        assert np.allclose(b.v.cpu().numpy(), st.v[0], rtol=1e-13, atol=1e-300), k
        got = tb.double().cpu().numpy()
        # one float32 rounding of the stored result
        assert np.all(np.abs(got - want) <= 6e-8 * np.maximum(np.abs(want), 1e-30) + 1e-45), (k, np.abs(got - want).max())
    assert float(np.abs(tb.cpu().numpy() - theta0).max()) > 1e-3    # it moved


def test_vpg_iterations_on_c2_shapes_against_the_oracle(quiet_logger):
    """BASELINE config C2 (CartpoleEnv, 4096 parallel envs, VPG, GaussianMLPPolicy(32, 32), horizon 100): three
    iterations of the product's VPG.  Each iteration's surrogate loss and gradient (rl_policy_grad_loss, vpg = 1) are
    held to the oracle's float64 numpy back-propagation of vpg.py:66-76 on the same batch (1e-5 / 2e-5), and the
    parameter step the product took to the oracle's Adam (persisting moments, one step per iteration, vpg.py:26-34)
    driven by the oracle's own gradient."""
    from oracle import np_reference as R
    from rllab_amd.algos.npo import npo_inputs
    from rllab_amd.algos.vpg import VPG
    from rllab_amd.baselines.linear_feature_baseline import LinearFeatureBaseline
    from rllab_amd.envs.box2d.cartpole_env import CartpoleEnv
    from rllab_amd.envs.normalized_env import normalize
    from rllab_amd.misc import ext, logger
    from rllab_amd.policies.gaussian_mlp_policy import GaussianMLPPolicy
    ext.set_seed(11)
    env = normalize(CartpoleEnv())
    pol = GaussianMLPPolicy(env_spec=env.spec, hidden_sizes=(32, 32))
    lr = 1e-2
    algo = VPG(env=env, policy=pol, baseline=LinearFeatureBaseline(env_spec=env.spec), batch_size=4096 * 100,
               max_path_length=100, n_itr=3, discount=0.99, optimizer_args=dict(learning_rate=lr),
               sampler_args=dict(n_envs=4096, seed=11))
    algo.start_worker()
    algo.init_opt()
    npol = R.NumpyGaussianMLP(4, 1, (32, 32), min_std=pol.min_std)
    theta = pol.get_param_values().astype(np.float64)
    st = R.Adam([theta], learning_rate=lr)
    for itr in range(3):
        paths = algo.sampler.obtain_samples(itr)
        sd = algo.sampler.process_samples(itr, paths)
        obs, act, adv, _, _, w, _ = [x.double().cpu().numpy() if torch.is_tensor(x) else x for x in npo_inputs(pol, sd)]
        # (Cartpole terminates: the batch runs on past 100 lock steps until 409600 samples are in finished paths)
        assert obs.shape == (4, paths.traj.B) and int(paths.traj.valid.sum()) >= 409600
        theta = pol.get_param_values().astype(np.float64)           # the float32 point the product steps from
        loss64, g64 = R.vpg_surrogate_and_grad(npol, theta, obs.T, act.T, adv, w)
        algo.optimize_policy(itr, sd)
        tab = logger.get_tabular()
        assert abs(float(tab["LossBefore"]) - loss64) <= 1e-5 * max(1.0, abs(loss64)), (itr, tab["LossBefore"], loss64)
        # the product's gradient at the same point (a second, deterministic evaluation of the pass optimize() ran)
        pol_probe = GaussianMLPPolicy(env_spec=env.spec, hidden_sizes=(32, 32))
        pol_probe.set_param_values(theta)
        g = pol_probe.fused_ops().loss_grad(npo_inputs(pol, sd), vpg=True).cpu().numpy()
        assert np.abs(g - g64).max() <= 2e-5 * np.abs(g64).max(), (itr, np.abs(g - g64).max(), np.abs(g64).max())
        # the step: oracle Adam on the ORACLE's gradient.  Adam divides by sqrt(v): where |g| is within the gradient
        # tolerance of zero the step direction is not determined, so those entries get the full step as slack
        want = R.adam_step(st, theta, g64)
        got = pol.get_param_values().astype(np.float64)
        slack = np.where(np.abs(g64) > 1e-3 * np.abs(g64).max(), 0.02 * lr, 2.1 * lr)
        assert np.all(np.abs(got - want) <= slack + 1e-6 * np.abs(want)), (itr, np.abs(got - want).max())
        assert np.abs(got - theta).max() > 0.5 * lr                  # a real step was taken
        logger.dump_tabular()


@pytest.mark.parametrize("do,da,h", [(13, 2, 32), (20, 6, 64), (4, 1, 32)])
def test_fused_fvp_cg_step_is_the_three_launch_sequence(do, da, h):
    """rl_policy_fvp_cg_step (row reduction's last workgroup runs the CG iteration) == rl_policy_fvp + rl_cg_step,
    bit for bit: same partial rows, same fixed reduction order, same float64 algebra -- over a whole CG run
    including the early exit, with and without the activation cache."""
    pol = _policy(do, da, h)
    inp = _inputs(pol, 20011, old_equals_new=True)
    g = torch.as_tensor(np.random.RandomState(5).randn(pol.flat_params.numel()), device="cuda")
    for cached in (False, True):
        for iters in (3, 10, 40):                 # 40: the residual test fires and freezes the iterates
            res = []
            for fuse in (True, False):
                ops = pol.fused_ops()
                ops.fuse_cg = fuse
                if cached:
                    ops.loss_grad(inp, keep_activations=True)
                x, xHx = ops.cg(inp, g, iters, 1e-5)
                step, stats = ops.cg_step_vector(inp, g, iters, 1e-5, 0.01)
                res.append((x.clone(), xHx.clone(), step.clone(), stats.clone()))
                ops.release()
            for a, b in zip(*res):
                assert torch.equal(a, b), (cached, iters)
