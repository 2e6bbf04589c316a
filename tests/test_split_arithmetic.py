"""The arithmetic of csrc/policy_split_kernels.hip, restated in numpy and checked without a GPU: an f32 number is the
exact sum of three bf16 parts (successive round-to-nearest-even residuals), and the six cross terms hi hi + hi mid +
mid hi + hi lo + lo hi + mid mid of two split operands reproduce the float64 product to the size of an f32 rounding
(worst case 2^-23 of |a b|, 2^-27 in the mean, either sign).
(On the device: tools/ubench/bf16_split_layout.hip for the instruction-level facts, tests/test_gpu_fvp_split.py for the
Fisher-vector product itself.)"""
import numpy as np


def bf16_rne(x):
    """float32 -> the nearest bfloat16 (ties to even), returned as float32 (what v_cvt_pk_bf16_f32 computes)."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = u + 0x7FFF + ((u >> 16) & 1)
    return (u & 0xFFFF0000).astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, dtype=np.float32)
    hi = bf16_rne(x)
    r1 = (x - hi).astype(np.float32)
    mid = bf16_rne(r1)
    r2 = (r1 - mid).astype(np.float32)
    lo = bf16_rne(r2)
    return hi, mid, lo, r1, r2


def _samples(rng, n):
    mags = 10.0 ** rng.uniform(-20, 20, n)
    x = (rng.standard_normal(n) * mags).astype(np.float32)
    edge = np.array([0.0, -0.0, 1.0, -1.0, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 3.0e38, -3.0e38, 1.0e-30, 0.1, 1 / 3.0],
                    dtype=np.float32)
    return np.concatenate([x, edge])


def test_three_bf16_parts_are_the_number_exactly():
    x = _samples(np.random.RandomState(0), 200000)
    hi, mid, lo, r1, r2 = split3(x)
    # every residual subtraction is exact (checked in float64), the last residual is itself a bf16
    assert np.array_equal(r1.astype(np.float64), x.astype(np.float64) - hi.astype(np.float64))
    assert np.array_equal(r2.astype(np.float64), r1.astype(np.float64) - mid.astype(np.float64))
    assert np.array_equal(lo, r2)
    assert np.array_equal(hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64), x.astype(np.float64))
    # each part carries at most 8 significant bits: its low 16 bits as an f32 are zero
    for part in (hi, mid, lo):
        assert not np.any(part.view(np.uint32) & 0xFFFF)
    # the parts shrink by 2^-8 or faster (round to nearest: half an ulp of the part above)
    nz = x != 0
    assert np.all(np.abs(mid[nz]) <= np.abs(x[nz]) * 2.0 ** -8)
    assert np.all(np.abs(lo[nz]) <= np.abs(x[nz]) * 2.0 ** -16)


def test_six_cross_terms_are_an_f32_accurate_product():
    rng = np.random.RandomState(1)
    a = (rng.standard_normal(100000) * 10.0 ** rng.uniform(-6, 6, 100000)).astype(np.float32)
    b = (rng.standard_normal(100000) * 10.0 ** rng.uniform(-6, 6, 100000)).astype(np.float32)
    ah, am, al, _, _ = split3(a)
    bh, bm, bl, _, _ = split3(b)
    f64 = lambda v: v.astype(np.float64)
    six = f64(ah) * f64(bh) + f64(ah) * f64(bm) + f64(am) * f64(bh) + f64(ah) * f64(bl) + f64(al) * f64(bh) + f64(am) * f64(bm)
    exact = f64(a) * f64(b)
    # dropped: mid lo + lo mid + lo lo.  Rounding to 8 significant bits leaves a residual of at most 2^-8 of the
    # number (a value just above a power of two), so |mid| <= 2^-8 |x|, |lo| <= 2^-16 |x| and the dropped terms are at
    # most (2 * 2^-24 + 2^-32) |a b| -- the size of two f32 roundings -- in the worst case ...
    rel = np.abs(six - exact) / np.abs(exact)
    assert np.all(rel <= 2.0 ** -23 * 1.001)
    # ... and far below one f32 rounding (2^-24 worst case, 2^-25.5 in the mean) for all but the odd pair
    assert np.mean(rel) <= 2.0 ** -27 and np.mean(rel <= 2.0 ** -25) >= 0.95


def test_dot_products_of_split_operands_beat_an_f32_fma_chain():
    """K = 32 dot products (one layer of the 32-unit nets) with f32 accumulation of the six-term sums against an f32
    fused-multiply-add chain, both measured against float64."""
    rng = np.random.RandomState(2)
    A = rng.standard_normal((2000, 32)).astype(np.float32)
    Bm = (rng.standard_normal((2000, 32)) * 1e-3).astype(np.float32)
    exact = (A.astype(np.float64) * Bm.astype(np.float64)).sum(1)
    scale = np.abs(A.astype(np.float64) * Bm.astype(np.float64)).sum(1)
    ah, am, al, _, _ = split3(A)
    bh, bm, bl, _, _ = split3(Bm)
    acc = np.zeros(2000, dtype=np.float32)
    # the kernel's order: smallest terms first, each term's 32 products summed into the f32 accumulator
    for x, y in ((am, bm), (ah, bl), (al, bh), (ah, bm), (am, bh), (ah, bh)):
        acc = (acc.astype(np.float64) + (x.astype(np.float64) * y.astype(np.float64)).sum(1)).astype(np.float32)
    chain = np.zeros(2000, dtype=np.float32)
    for k in range(32):
        chain = (chain.astype(np.float64) + A[:, k].astype(np.float64) * Bm[:, k].astype(np.float64)).astype(np.float32)
    err_split = np.abs(acc - exact) / scale
    err_chain = np.abs(chain - exact) / scale
    assert err_split.max() <= 2.0 ** -23
    assert np.mean(err_split) <= np.mean(err_chain)
