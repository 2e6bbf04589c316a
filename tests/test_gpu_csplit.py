"""The cooperative split-operand Fisher-vector product (csrc/policy_csplit_kernels.hip: 64-unit, wide and deep nets on
v_mfma_f32_32x32x16_bf16 with three-way split f32 operands, parts images in LDS, transposing reads for the products over
the sample axis) against float64 autograd of the reference's mean KL (PerlmutterHvp,
rllab/optimizers/conjugate_gradient_optimizer.py:27-55; GaussianMLPPolicy(hidden_sizes=...) is free-form,
rllab/policies/gaussian_mlp_policy.py:21-58) and against the f32-matrix-instruction product of the same batch."""
import numpy as np
import pytest
import torch

from tests import test_gpu_update_parity as U
from tests.test_gpu_fvp_split import _f64_products, _variant
from tests.test_gpu_wide_nets import _policy as _wide_policy

pytestmark = pytest.mark.gpu

# (obs, act, hidden): both cache layouts (the one-wavefront-per-tile family's fragments for the HIP-native pairs at
# (64, 64); the cooperative family's unit rows for everything else), two and four wavefronts per workgroup, one or two
# input k-blocks, narrow layers inside wide nets, heads of 1 .. 8 actions
SHAPES = [(13, 2, (64, 64)), (20, 6, (64, 64)), (4, 1, (64, 64)), (17, 8, (64, 64)), (13, 2, (128, 128)),
          (20, 6, (128, 128)), (13, 2, (100, 50, 25)), (21, 6, (128, 64)), (11, 1, (64, 32)), (13, 2, (32, 64)),
          (20, 3, (64, 64, 64)), (13, 2, (128, 128, 64)),
          # round 6, k-slices: a row tile shared by FOUR wavefronts (128 -> 32), slices of ONE k-block, a sample-axis product
          # of a single output tile split per sample block (32 -> 32 inside a four-wavefront workgroup), run-time shapes
          (13, 2, (128, 32)), (17, 6, (128, 32, 32)), (11, 3, (32, 32, 128)), (13, 2, (64, 128, 32))]


def _policy(do, da, hidden):
    if len(hidden) == 2 and hidden[0] == hidden[1] and hidden[0] in (32, 64):
        return U._policy(do, da, hidden[0])
    return _wide_policy(do, da, hidden)


@pytest.mark.parametrize("do,da,hidden", SHAPES)
@pytest.mark.parametrize("B", [32, 4096, 64000])
def test_cooperative_split_product_is_an_f32_accurate_product(do, da, hidden, B, monkeypatch):
    pol = _policy(do, da, hidden)
    ops = pol.fused_ops()
    assert ops is not None
    inp = U._inputs(pol, B, old_equals_new=True)
    rng = np.random.RandomState(7)
    vs = [torch.as_tensor(rng.randn(pol.flat_params.numel()), device="cuda") for _ in range(2)]
    want = _f64_products(pol, inp, vs)
    ops.loss_grad(inp, keep_activations=True)
    assert ops._acts_tag is not None
    monkeypatch.setenv("RLLAB_FVP_SPLIT", "0")
    assert _variant(ops, inp) == 0
    plain = [ops.fvp(inp, v) for v in vs]
    monkeypatch.delenv("RLLAB_FVP_SPLIT")
    # by default the kernel takes the nets with a 128-unit layer (where it is the faster one); RLLAB_FVP_SPLIT=2: every shape
    # (the HIP-native (obs, action) pairs at (64, 64) have the one-wavefront-per-tile split kernels since round 5: variant 4,
    # the two-way f16 split, since round 6)
    from tests.test_gpu_fvp_split import SPLIT64_SHAPES
    narrow64 = tuple(pol.kernel_layout().hidden3) == (64, 64, 0) and (do, da) in SPLIT64_SHAPES    # (padded widths count)
    assert _variant(ops, inp) == (2 if max(pol.kernel_layout().hidden3) == 128 else 4 if narrow64 else 0)
    monkeypatch.setenv("RLLAB_FVP_SPLIT", "2")
    assert _variant(ops, inp) == 2                       # the launch below IS the cooperative split kernel
    split = [ops.fvp(inp, v) for v in vs]
    for hv_s, hv_p, hv64 in zip(split, plain, want):
        scale = float(hv64.abs().max())
        err_s, err_p = float((hv_s - hv64).abs().max()) / scale, float((hv_p - hv64).abs().max()) / scale
        assert err_s <= 5e-5, (err_s, err_p)                           # the reference tolerance of the product
        assert err_s <= 2.0 * err_p + 2e-6, (err_s, err_p)             # and no worse than the f32 matrix instructions
        assert not torch.equal(hv_s, hv_p)                             # (two different kernels did run)


def test_cooperative_split_product_takes_only_its_batches(monkeypatch):
    """Whole 32-sample tiles and cached activations; everything else stays on the f32 matrix instructions."""
    pol = _policy(13, 2, (128, 64))
    ops = pol.fused_ops()
    assert max(pol.kernel_layout().hidden3) == 128
    for B, want in ((4096, 2), (4100, 0), (63, 0)):
        inp = U._inputs(pol, B, old_equals_new=True)
        ops.release()
        assert _variant(ops, inp) == 0                   # nothing cached yet
        ops.loss_grad(inp, keep_activations=True)
        assert _variant(ops, inp) == want


@pytest.mark.parametrize("do,da,hidden", [(20, 6, (64, 64)), (13, 2, (100, 50, 25))])
def test_cg_on_the_cooperative_split_product_solves_the_same_system(do, da, hidden, monkeypatch):
    """Ten CG iterations (krylov.cg, rllab/misc/krylov.py:7-39) on either product: the same solution to f32 accuracy."""
    pol = _policy(do, da, hidden)
    ops = pol.fused_ops()
    inp = U._inputs(pol, 64000, old_equals_new=True)
    g = ops.loss_grad(inp, keep_activations=True)
    monkeypatch.setenv("RLLAB_FVP_SPLIT", "2")
    assert _variant(ops, inp) == 2
    x_s, xhx_s = ops.cg(inp, g, 10, 1e-5)
    monkeypatch.setenv("RLLAB_FVP_SPLIT", "0")
    x_p, xhx_p = ops.cg(inp, g, 10, 1e-5)
    assert float((x_s - x_p).abs().max()) <= 2e-4 * float(x_p.abs().max())
    assert abs(float(xhx_s) - float(xhx_p)) <= 2e-5 * abs(float(xhx_p))


def test_products_are_linear_symmetric_and_positive_at_c5_size(monkeypatch):
    """C5's per-GPU batch (1024 envs x 500 steps, (20 -> 64 -> 64 -> 6)), ragged weights."""
    pol = _policy(20, 6, (64, 64))
    ops = pol.fused_ops()
    inp = U._inputs(pol, 1024 * 500, old_equals_new=True)
    ops.loss_grad(inp, keep_activations=True)
    monkeypatch.setenv("RLLAB_FVP_SPLIT", "2")
    assert _variant(ops, inp) == 2
    rng = np.random.RandomState(11)
    v, w = (torch.as_tensor(rng.randn(pol.flat_params.numel()), device="cuda") for _ in range(2))
    Fv, Fw = ops.fvp(inp, v), ops.fvp(inp, w)
    scale = float(Fv.abs().max())
    comb = (0.7 * v - 1.3 * w).float().double()
    lin = ops.fvp(inp, comb) - (0.7 * ops.fvp(inp, v.float().double()) - 1.3 * ops.fvp(inp, w.float().double()))
    assert float(lin.abs().max()) <= 2e-5 * scale
    vFw, wFv = float(v.dot(Fw)), float(w.dot(Fv))
    assert abs(vFw - wFv) <= 2e-5 * max(abs(vFw), float(v.dot(Fv)))
    assert float(v.dot(Fv)) > 0 and float(w.dot(Fw)) > 0
