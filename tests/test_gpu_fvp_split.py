"""The split-operand Fisher-vector products -- csrc/policy_split_kernels.hip (bf16 matrix instructions on three-way split
f32 operands, six cross terms; variant 1) and csrc/policy_splith_kernels.hip (f16 matrix instructions on two-way split
operands under per-launch power-of-two scales, three cross terms; variant 4, the library's choice where it is built) --
against float64 autograd of the reference's mean KL (PerlmutterHvp,
rllab/optimizers/conjugate_gradient_optimizer.py:27-55) and against the f32-matrix-instruction product of the same
batch: each must be an f32-accurate product, not a reduced-precision one."""
import numpy as np
import pytest
import torch

from tests import test_gpu_update_parity as U

pytestmark = pytest.mark.gpu

SPLIT_SHAPES = [(4, 1), (6, 1), (11, 1), (13, 2), (13, 1), (20, 3), (20, 6), (21, 6)]   # every HIP-native env's default (32, 32) policy
SPLIT64_SHAPES = [(4, 1), (6, 1), (11, 1), (13, 2), (20, 3), (20, 6), (21, 6)]             # ... and (64, 64): fvp_split64_kernel
ALL_SPLIT = [(d, a, 32) for d, a in SPLIT_SHAPES] + [(d, a, 64) for d, a in SPLIT64_SHAPES]


SPLITH_SHAPES = ALL_SPLIT          # (round 6, later: the wide-head (32, 32) shapes too, one wavefront per SIMD)
ARITH = ["f16x2", "bf16x3"]


def _variant(ops, inp):
    return ops.fvp_variant(inp)


def _arith(monkeypatch, arith, shape):
    """Select the split arithmetic for the launches that follow; returns the variant rl_policy_fvp_variant must report:
    4 = two-way f16 (the library's choice for the shapes it is built for), 1 = three-way bf16 (RLLAB_FVP_SPLIT=5, or
    the library's choice elsewhere)."""
    if arith == "bf16x3":
        monkeypatch.setenv("RLLAB_FVP_SPLIT", "5")
        return 1
    monkeypatch.delenv("RLLAB_FVP_SPLIT", raising=False)
    return 4 if tuple(shape) in SPLITH_SHAPES else 1


def _f64_products(pol, inp, vs):
    _, kl, _ = U._closures(pol)
    flat64 = pol.flat_params.detach().double().requires_grad_(True)
    with torch.no_grad():
        om64 = pol.mean_planes(inp[0].double(), flat64.detach())
    inp64 = (inp[0], inp[1], inp[2], om64, pol.effective_log_std().detach().double().reshape(-1, 1), inp[5], inp[6])
    g = torch.autograd.grad(kl(flat64, *inp64), flat64, create_graph=True)[0]
    return [torch.autograd.grad((g * v).sum(), flat64, retain_graph=True)[0] for v in vs]


def _blocks(pol, h=32):
    do, da = pol.obs_dim, pol.action_dim
    names, sizes = ("W0", "b0", "W1", "b1", "W2", "b2", "log_std"), (do * h, h, h * h, h, h * da, da, da)
    out, o = [], 0
    for n, s in zip(names, sizes):
        out.append((n, o, o + s))
        o += s
    return out


@pytest.mark.parametrize("arith", ARITH)
@pytest.mark.parametrize("do,da,h", ALL_SPLIT)
@pytest.mark.parametrize("B", [32, 4096, 64000])
def test_split_product_is_an_f32_accurate_product(do, da, h, B, arith, monkeypatch):
    if arith == "f16x2" and (do, da, h) not in SPLITH_SHAPES:
        pytest.skip("a shape of the bf16 kernels only")
    pol = U._policy(do, da, h)
    ops = pol.fused_ops()
    inp = U._inputs(pol, B, old_equals_new=True)
    rng = np.random.RandomState(7)
    vs = [torch.as_tensor(rng.randn(pol.flat_params.numel()), device="cuda") for _ in range(2)]
    want = _f64_products(pol, inp, vs)
    ops.loss_grad(inp, keep_activations=True)
    assert ops._acts_tag is not None
    monkeypatch.setenv("RLLAB_FVP_SPLIT", "0")
    assert _variant(ops, inp) == 0
    plain = [ops.fvp(inp, v) for v in vs]
    want_variant = _arith(monkeypatch, arith, (do, da, h))
    assert _variant(ops, inp) == want_variant            # the launch below IS that split kernel
    split = [ops.fvp(inp, v) for v in vs]
    assert ops._acts_tag is not None
    for hv_s, hv_p, hv64 in zip(split, plain, want):
        scale = float(hv64.abs().max())
        err_s, err_p = float((hv_s - hv64).abs().max()) / scale, float((hv_p - hv64).abs().max()) / scale
        worst = [(n, float((hv_s - hv64)[a:b].abs().max()) / scale) for n, a, b in _blocks(pol, h)]
        assert err_s <= 5e-5, worst                                    # the reference tolerance of the product
        assert err_s <= 2.0 * err_p + 2e-6, (err_s, err_p, worst)      # and no worse than the f32 matrix instructions
        assert not torch.equal(hv_s, hv_p)                             # (two different kernels did run)


def test_split_product_takes_only_its_batches(monkeypatch):
    """Whole 32-sample tiles, cached activations, two 32-unit tanh layers; everything else stays on the f32 matrix
    instructions and keeps the cached product bit-identical to the recomputed one."""
    pol = U._policy(13, 2, 32)
    ops = pol.fused_ops()
    for B, want in ((4096, 4), (4100, 0), (63, 0)):
        inp = U._inputs(pol, B, old_equals_new=True)
        ops.release()
        assert _variant(ops, inp) == 0                   # nothing cached yet
        ops.loss_grad(inp, keep_activations=True)
        assert _variant(ops, inp) == want
        if want:
            monkeypatch.setenv("RLLAB_FVP_SPLIT", "5")   # the three-way bf16 split on request
            assert _variant(ops, inp) == 1
            monkeypatch.delenv("RLLAB_FVP_SPLIT")
    pol64 = U._policy(13, 2, 64)
    ops64 = pol64.fused_ops()
    inp = U._inputs(pol64, 4096, old_equals_new=True)
    ops64.loss_grad(inp, keep_activations=True)
    assert _variant(ops64, inp) == 4                     # (64, 64): the one-wavefront-per-tile split kernels since round 5
    inp_r = U._inputs(pol64, 4100, old_equals_new=True)
    ops64.loss_grad(inp_r, keep_activations=True)
    assert _variant(ops64, inp_r) == 0                   # ... whole tiles only, like the 32-unit kernel
    ops64.loss_grad(inp, keep_activations=True)
    monkeypatch.setenv("RLLAB_FVP_SPLIT", "2")           # the cooperative split kernel takes them on request
    assert _variant(ops64, inp) == 2                     # (tests/test_gpu_csplit.py; slower there, profiles/r04_notes.md)
    monkeypatch.setenv("RLLAB_FVP_SPLIT", "0")
    assert _variant(ops64, inp) == 0


@pytest.mark.parametrize("arith", ARITH)
@pytest.mark.parametrize("do,da,h", [(13, 2, 32), (20, 6, 64)])
def test_cg_on_the_split_product_solves_the_same_system(do, da, h, arith, monkeypatch):
    """Ten CG iterations (krylov.cg, rllab/misc/krylov.py:7-39) on either product: the same solution to f32 accuracy."""
    pol = U._policy(do, da, h)
    ops = pol.fused_ops()
    inp = U._inputs(pol, 64000, old_equals_new=True)
    g = ops.loss_grad(inp, keep_activations=True)
    want_variant = _arith(monkeypatch, arith, (do, da, h))
    assert _variant(ops, inp) == want_variant
    x_s, xhx_s = ops.cg(inp, g, 10, 1e-5)
    monkeypatch.setenv("RLLAB_FVP_SPLIT", "0")
    x_p, xhx_p = ops.cg(inp, g, 10, 1e-5)
    assert float((x_s - x_p).abs().max()) <= 1e-4 * float(x_p.abs().max())
    assert abs(float(xhx_s) - float(xhx_p)) <= 1e-5 * abs(float(xhx_p))


@pytest.mark.parametrize("arith", ARITH)
@pytest.mark.parametrize("do,da,h,B", [(13, 2, 32, 4096 * 500), (20, 6, 64, 1024 * 500)])
def test_full_size_products_are_linear_symmetric_and_positive(do, da, h, B, arith, monkeypatch):
    """At BASELINE config C3's batch (4096 envs x 500 steps = 2 048 000 samples, ragged weights): the split product is
    linear in the vector, symmetric (v . F w == w . F v) and positive (v . F v > 0) -- properties of the Fisher matrix of
    rllab/optimizers/conjugate_gradient_optimizer.py:27-55 that do not need a float64 pass over two million samples."""
    pol = U._policy(do, da, h)
    ops = pol.fused_ops()
    inp = U._inputs(pol, B, old_equals_new=True)
    ops.loss_grad(inp, keep_activations=True)
    want_variant = _arith(monkeypatch, arith, (do, da, h))
    assert _variant(ops, inp) == want_variant
    rng = np.random.RandomState(11)
    v, w = (torch.as_tensor(rng.randn(pol.flat_params.numel()), device="cuda") for _ in range(2))
    Fv, Fw = ops.fvp(inp, v), ops.fvp(inp, w)
    scale = float(Fv.abs().max())
    # the kernel takes the vector in f32: compare against the product of the f32-rounded combination
    comb = (0.7 * v - 1.3 * w).float().double()
    lin = ops.fvp(inp, comb) - (0.7 * ops.fvp(inp, v.float().double()) - 1.3 * ops.fvp(inp, w.float().double()))
    assert float(lin.abs().max()) <= 2e-5 * scale
    vFw, wFv = float(v.dot(Fw)), float(w.dot(Fv))
    assert abs(vFw - wFv) <= 2e-5 * max(abs(vFw), float(v.dot(Fv)))
    assert float(v.dot(Fv)) > 0 and float(w.dot(Fw)) > 0


def _f64_products_chunked(pol, inp, vs, chunk=256 * 1024):
    """Float64 double-backward of the reference's mean KL over the WHOLE batch, in sample chunks: the mean KL is a
    weighted sum over samples with one global 1 / count, so the Hessian-vector product is the sum of the chunks'."""
    _, kl, _ = U._closures(pol)
    B = inp[0].shape[-1]
    flat64 = pol.flat_params.detach().double().requires_grad_(True)
    ls64 = pol.effective_log_std().detach().double().reshape(-1, 1)
    out = [torch.zeros_like(flat64) for _ in vs]
    for a in range(0, B, chunk):
        b = min(B, a + chunk)
        with torch.no_grad():
            om64 = pol.mean_planes(inp[0][:, a:b].double(), flat64.detach())
        c64 = (inp[0][:, a:b], inp[1][:, a:b], inp[2][a:b], om64, ls64, inp[5][a:b], inp[6])
        g = torch.autograd.grad(kl(flat64, *c64), flat64, create_graph=True)[0]
        for o, v in zip(out, vs):
            o += torch.autograd.grad((g * v).sum(), flat64, retain_graph=True)[0]
        del g
    return out


@pytest.mark.parametrize("arith", ARITH)
@pytest.mark.parametrize("do,da,h,n_envs", [(13, 2, 32, 4096), (20, 6, 64, 1024)])
def test_full_size_product_against_float64_double_backward(do, da, h, n_envs, arith, monkeypatch):
    """The product at the FULL batch of BASELINE configs C3 (4096 Swimmer envs x 500 steps = 2 048 000 samples, (32, 32))
    and C5's per-GPU shard (1024 HalfCheetah envs x 500 = 512 000 samples, (64, 64)) against a float64 double-backward of
    the reference's mean KL over every sample (PerlmutterHvp, conjugate_gradient_optimizer.py:27-55): the reference
    tolerance of the product (5e-5 of its largest entry), and, where the split-operand arithmetic runs, no worse than
    the f32-matrix-instruction product of the same batch."""
    pol = U._policy(do, da, h)
    ops = pol.fused_ops()
    inp = U._inputs(pol, n_envs * 500, old_equals_new=True)
    rng = np.random.RandomState(13)
    vs = [torch.as_tensor(rng.randn(pol.flat_params.numel()), device="cuda") for _ in range(2)]
    want = _f64_products_chunked(pol, inp, vs)
    ops.loss_grad(inp, keep_activations=True)
    want_variant = _arith(monkeypatch, arith, (do, da, h))
    variant = _variant(ops, inp)
    assert variant == want_variant
    got = [ops.fvp(inp, v) for v in vs]
    plain = None
    if variant in (1, 2, 4):
        monkeypatch.setenv("RLLAB_FVP_SPLIT", "0")
        assert _variant(ops, inp) == 0
        plain = [ops.fvp(inp, v) for v in vs]
        monkeypatch.delenv("RLLAB_FVP_SPLIT")
        _arith(monkeypatch, arith, (do, da, h))
    for i, (hv, hv64) in enumerate(zip(got, want)):
        scale = float(hv64.abs().max())
        err = float((hv - hv64).abs().max()) / scale
        print("full-size FVP (%d, %d) -> (%d, %d) on %d samples, variant %d: max error %.3g of the largest entry"
              % (do, da, h, h, n_envs * 500, variant, err))
        assert err <= 5e-5
        if plain is not None:
            err_p = float((plain[i] - hv64).abs().max()) / scale
            assert err <= 2.0 * err_p + 2e-6, (err, err_p)


SPLIT16_SHAPES = [(4, 1), (6, 1), (11, 1), (13, 2), (13, 1)]     # csrc/policy_split16_kernels.hip (at most two actions)


@pytest.mark.parametrize("do,da", SPLIT16_SHAPES)
@pytest.mark.parametrize("B", [32, 4096, 64000, 70016])
@pytest.mark.parametrize("wps", ["4", "3"])
def test_sixteen_sample_tile_product_is_the_same_f32_accurate_product(do, da, B, wps, monkeypatch):
    """rl_launch_opts.fvp_split = 3: the split product on 16-sample tiles / v_mfma_f32_16x16x32_bf16, four wavefronts per
    SIMD (round 6).  Same arithmetic as fvp_split_kernel in another order: within the reference tolerance of the float64
    product, no worse than the f32 matrix instructions, and a different kernel did run."""
    pol = U._policy(do, da, 32)
    ops = pol.fused_ops()
    inp = U._inputs(pol, B, old_equals_new=True)
    rng = np.random.RandomState(17)
    vs = [torch.as_tensor(rng.randn(pol.flat_params.numel()), device="cuda") for _ in range(2)]
    want = _f64_products(pol, inp, vs)
    ops.loss_grad(inp, keep_activations=True)
    monkeypatch.setenv("RLLAB_FVP_SPLIT", "0")
    plain = [ops.fvp(inp, v) for v in vs]
    monkeypatch.setenv("RLLAB_FVP_SPLIT", "5")           # the 32-sample-tile kernel of the same (bf16) arithmetic
    assert _variant(ops, inp) == 1
    split = [ops.fvp(inp, v) for v in vs]
    monkeypatch.setenv("RLLAB_FVP_SPLIT", "3")
    monkeypatch.setenv("RLLAB_FVP_SPLIT_WPS", wps)       # wavefronts per SIMD: four (128 registers) or three
    assert _variant(ops, inp) == 3                       # the launch below IS fvp_split16_kernel
    tile16 = [ops.fvp(inp, v) for v in vs]
    for hv_t, hv_s, hv_p, hv64 in zip(tile16, split, plain, want):
        scale = float(hv64.abs().max())
        err_t, err_p = float((hv_t - hv64).abs().max()) / scale, float((hv_p - hv64).abs().max()) / scale
        worst = [(n, float((hv_t - hv64)[a:b].abs().max()) / scale) for n, a, b in _blocks(pol, 32)]
        assert err_t <= 5e-5, worst
        assert err_t <= 2.0 * err_p + 2e-6, (err_t, err_p, worst)
        assert float((hv_t - hv_s).abs().max()) <= 2e-5 * scale          # the two split kernels agree to rounding
        assert not torch.equal(hv_t, hv_s)


def test_sixteen_sample_tile_product_leaves_other_shapes_to_the_shipped_kernels(monkeypatch):
    monkeypatch.setenv("RLLAB_FVP_SPLIT", "3")
    for do, da, h, want in ((20, 6, 32, 1), (20, 6, 64, 1), (13, 2, 64, 1)):
        pol = U._policy(do, da, h)
        ops = pol.fused_ops()
        inp = U._inputs(pol, 4096, old_equals_new=True)
        ops.loss_grad(inp, keep_activations=True)
        assert _variant(ops, inp) == want


def test_full_size_sixteen_sample_tile_product_matches_the_shipped_kernel(monkeypatch):
    """BASELINE config C3's batch (2 048 000 samples, ragged weights): both split kernels against each other."""
    pol = U._policy(13, 2, 32)
    ops = pol.fused_ops()
    inp = U._inputs(pol, 4096 * 500, old_equals_new=True)
    ops.loss_grad(inp, keep_activations=True)
    v = torch.as_tensor(np.random.RandomState(3).randn(pol.flat_params.numel()), device="cuda")
    monkeypatch.setenv("RLLAB_FVP_SPLIT", "5")
    ref = ops.fvp(inp, v)
    monkeypatch.setenv("RLLAB_FVP_SPLIT", "3")
    assert _variant(ops, inp) == 3
    got = ops.fvp(inp, v)
    assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    assert float(v.dot(got)) > 0


# ---- the two-way f16 split (variant 4): range.  f16 has five exponent bits, so everything rests on the per-launch scales ----

def _scaled_inputs(pol, B, obs_scale):
    inp = list(U._inputs(pol, B, old_equals_new=True))
    inp[0] = inp[0] * obs_scale
    with torch.no_grad():
        inp[3] = pol.mean_planes(inp[0].double(), pol.flat_params.double()).float()         # old mean == new mean
    return tuple(inp)


@pytest.mark.parametrize("do,da,h", [(13, 2, 32), (20, 6, 64)])
@pytest.mark.parametrize("obs_scale,vec_scale,log_std", [(1.0, 1e-8, 0.0), (1.0, 1e8, 0.0), (1e4, 1.0, 0.0), (1e-4, 1.0, 0.0),
                                                         (30.0, 1e3, -4.0), (1.0, 1.0, 3.0), (1e3, 1e-6, -5.0)])
def test_f16_split_product_keeps_f32_accuracy_over_magnitudes(do, da, h, obs_scale, vec_scale, log_std, monkeypatch):
    """Observations from 1e-4 to 1e4, directions from 1e-8 to 1e8, sigma from e^-5 to e^3 (the Fisher's 1 / sigma^2 from
    2e4 to 2e-3): no operand of the f16 matrix instructions overflows (the result is finite) and the product stays as
    close to float64 as the f32 matrix instructions' -- the scales are worst-case bounds computed per launch
    (policy_splith_kernels.hip::make_scales), max |obs| coming from the gradient pass (rl_policy_batch.obs_absmax)."""
    pol = U._policy(do, da, h)
    with torch.no_grad():
        pol.flat_params[-da:].fill_(log_std)          # the log_std row closes the flat vector
    ops = pol.fused_ops()
    inp = _scaled_inputs(pol, 4096, obs_scale)
    rng = np.random.RandomState(23)
    vs = [torch.as_tensor(rng.randn(pol.flat_params.numel()) * vec_scale, device="cuda") for _ in range(2)]
    # a direction whose blocks differ by ten orders of magnitude
    mixed = torch.as_tensor(rng.randn(pol.flat_params.numel()), device="cuda") * vec_scale
    for (n, a, b), f in zip(_blocks(pol, h), (1e5, 1.0, 1e-5, 1.0, 1e3, 1e-3, 1.0)):
        mixed[a:b] *= f
    vs.append(mixed)
    want = _f64_products(pol, inp, vs)
    ops.loss_grad(inp, keep_activations=True)
    assert float(ops._absmax) == float(inp[0].abs().max())
    assert _variant(ops, inp) == 4
    got = [ops.fvp(inp, v) for v in vs]
    monkeypatch.setenv("RLLAB_FVP_SPLIT", "0")
    plain = [ops.fvp(inp, v) for v in vs]
    for hv, hv_p, hv64 in zip(got, plain, want):
        assert bool(torch.isfinite(hv).all())
        scale = float(hv64.abs().max())
        err, err_p = float((hv - hv64).abs().max()) / scale, float((hv_p - hv64).abs().max()) / scale
        assert err <= 2.0 * err_p + 2e-6, (err, err_p)


def test_obs_absmax_is_what_the_gradient_pass_saw():
    """rl_policy_grad with the activation cache leaves max |obs| in rl_policy_batch.obs_absmax; a second batch overwrites
    it (the launcher zeroes the word first), smaller or larger."""
    pol = U._policy(13, 2, 32)
    ops = pol.fused_ops()
    for scale in (50.0, 0.01, 7.0):
        inp = _scaled_inputs(pol, 64000, scale)
        ops.loss_grad(inp, keep_activations=True)
        assert float(ops._absmax) == float(inp[0].abs().max())
