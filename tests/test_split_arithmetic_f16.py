"""The arithmetic and the scale plan of csrc/policy_splith_kernels.hip (the Fisher-vector product on a two-way f16 split),
restated in numpy and checked without a GPU:
  * b = hi + 2^-11 lo' with hi = f16(b), lo' = f16(2^11 (b - hi)) reproduces b to 2^-22 |b| over f16's normal range and to an
    absolute 2^-36 below it; the residual subtraction is exact;
  * three cross terms (image form: hi hi + lo hi + hs lo', and the two-accumulator form) reproduce the float64 product to a
    few f32 roundings;
  * `make_scales` (typed in again here from the kernel source): for ANY maxima -- observations, directions, weights, 1 / sigma^2
    over forty orders of magnitude -- every quantity that becomes an f16 operand is bounded below 2^15 (f16's largest finite
    number is 65504), the images sit where their unscaled low parts are normal numbers, and the unscale factors undo the
    scales exactly;
  * a whole product J^T M J v of a small tanh net evaluated operand by operand in that arithmetic (numpy float16 parts,
    float32 accumulation) agrees with float64 like an f32 evaluation does.
(On the device: tools/ubench/f16_split.hip for the instruction-level facts, tests/test_gpu_fvp_split.py for the kernels.)"""
import numpy as np

f64 = lambda v: np.asarray(v, dtype=np.float64)


def split2(b):
    """(hi, lo') as float16 arrays: what v_cvt_pk_f16_f32 / v_fma_mix_f32 / v_fma_mixlo_f16 compute."""
    b = np.asarray(b, dtype=np.float32)
    hi = b.astype(np.float16)
    r = (b - hi.astype(np.float32)).astype(np.float32)
    assert np.array_equal(f64(r), f64(b) - f64(hi))                  # the residual is exact
    lo = (r * np.float32(2048.0)).astype(np.float16)
    return hi, lo


def image(a):
    """(hi, lo, hs) of a loop-invariant operand (already multiplied by its scale)."""
    a = np.asarray(a, dtype=np.float32)
    hi = a.astype(np.float16)
    lo = (a - hi.astype(np.float32)).astype(np.float16)
    hs = (hi.astype(np.float32) * np.float32(2.0 ** -11)).astype(np.float16)
    return hi, lo, hs


def test_two_f16_parts_carry_22_bits_over_the_normal_range_and_an_absolute_floor_below():
    rng = np.random.RandomState(0)
    x = (rng.standard_normal(400000) * 2.0 ** rng.uniform(-13, 14.5, 400000)).astype(np.float32)
    x = x[(np.abs(x) >= 2.0 ** -13) & (np.abs(x) < 2.0 ** 15)]
    hi, lo = split2(x)
    assert np.all(np.isfinite(hi.astype(np.float32))) and np.all(np.isfinite(lo.astype(np.float32)))
    err = np.abs(f64(hi) + f64(lo) / 2048.0 - f64(x))
    assert np.all(err <= np.abs(f64(x)) * 2.0 ** -22)
    assert np.mean(err / np.abs(f64(x))) <= 2.0 ** -24.5
    assert np.all(np.abs(f64(lo)) <= np.abs(f64(x)) * 1.0001)         # lo' never exceeds |b|: no overflow of the scaled part
    # below the normal range: an absolute floor (hi is a subnormal f16, lo' picks up what it left)
    t = (rng.standard_normal(100000) * 2.0 ** rng.uniform(-40, -14, 100000)).astype(np.float32)
    t = t[np.abs(t) < 2.0 ** -14]
    hi, lo = split2(t)
    assert np.all(np.abs(f64(hi) + f64(lo) / 2048.0 - f64(t)) <= 2.0 ** -36)


def test_three_cross_terms_are_an_f32_accurate_product():
    rng = np.random.RandomState(1)
    n = 200000
    a = (rng.standard_normal(n) * 2.0 ** rng.uniform(-3, 13.5, n)).astype(np.float32)      # an image: scaled to sit high
    b = (rng.standard_normal(n) * 2.0 ** rng.uniform(-12, 14.5, n)).astype(np.float32)     # a per-tile operand
    keep = (np.abs(a) >= 2.0 ** -3) & (np.abs(a) < 2.0 ** 14) & (np.abs(b) >= 2.0 ** -13) & (np.abs(b) < 2.0 ** 15)
    a, b = a[keep], b[keep]
    ah, al, ahs = image(a)
    bh, bl = split2(b)
    exact = f64(a) * f64(b)
    img = f64(ah) * f64(bh) + f64(al) * f64(bh) + f64(ahs) * f64(bl)
    rel = np.abs(img - exact) / np.abs(exact)
    assert rel.max() <= 2.0 ** -20.5 and rel.mean() <= 2.0 ** -23.5
    # both operands per tile: c0 + 2^-11 c1
    a2h, a2l = split2(a)
    two = f64(a2h) * f64(bh) + (f64(a2h) * f64(bl) + f64(a2l) * f64(bh)) / 2048.0
    rel2 = np.abs(two - exact) / np.abs(exact)
    assert rel2.max() <= 2.0 ** -20.5 and rel2.mean() <= 2.0 ** -23.5
    # an f32 product for scale: one rounding, 2^-24 worst case, 2^-25.4 in the mean
    f32 = f64((a * b).astype(np.float32))
    assert rel.mean() <= 4.0 * np.mean(np.abs(f32 - exact) / np.abs(exact))


# ---- make_scales, typed in again from policy_splith_kernels.hip -----------------------------------------------------------
def exp_of(x):
    """x < 2^E for finite x >= 0 (frexp's exponent), at least -30."""
    x = np.float32(x)
    e = int((x.view(np.uint32) >> 23) & 255) - 126
    return max(e, -30)


def make_scales(HL, DAL, bx, bv0, bv1, bv2, bw1, bw2, fkmax):
    Ex, Ev0, Ev1, Ev2, EW1, EW2, Efk = (exp_of(max(bx, 1.0)), exp_of(bv0), exp_of(bv1), exp_of(bv2), exp_of(bw1), exp_of(bw2),
                                        exp_of(fkmax))
    e_x = 6 - exp_of(bx)
    e_b = min(e_x, 5)
    e_w1 = 7 - EW1
    xt = 11 if e_x <= 5 else 1 + max(11, e_x)
    e_v0 = min(15 - xt - Ev0, 14 - Ev1 - e_w1 - e_x)
    s0 = e_v0 + e_x
    s1 = s0 + e_w1
    EZ0 = 5 + Ev0 + Ex
    EZ1 = 1 + max(HL + 1 + Ev1, HL + EW1 + EZ0)
    EM = 1 + max(HL + 1 + Ev2, HL + EW2 + EZ1)
    EG1 = DAL + EW2 + EM + Efk
    sg = min(100, max(-100, 15 - EG1))
    kg = -(HL + 7)
    return dict(e_x=e_x, e_b=e_b, e_w1=e_w1, e_v0=e_v0, s0=s0, s1=s1, sg=sg, kg=kg, e_v0b=e_v0 + e_x - e_b,
                u1=-sg, u0=-(sg + e_w1 + kg + e_x + 8), u0b=-(sg + e_w1 + kg + e_b + 8),
                EZ0=EZ0, EZ1=EZ1, EM=EM, EG1=EG1)


def test_no_f16_operand_can_overflow_whatever_the_maxima():
    """Worst-case magnitudes through the product, in exact rational arithmetic on the maxima: dz0~, the images, gz1~, gz0~
    and the transposed observations all stay below 2^15."""
    rng = np.random.RandomState(2)
    for H, HL in ((32, 5), (64, 6)):
        for DO, DA, DAL in ((13, 2, 1), (20, 6, 3), (31, 8, 3), (4, 1, 0)):
            for _ in range(1500):
                bx, bv0, bv1, bv2, bw1, bw2 = (float(np.float32(10.0 ** rng.uniform(lo, hi)))
                                               for lo, hi in ((-8, 7), (-8, 8), (-8, 8), (-8, 8), (-3, 3), (-3, 3)))
                fkmax = float(np.float32(10.0 ** rng.uniform(-12, 2)))       # max_k 2 / (2 sigma_k^2 + 1e-8) / N
                S = make_scales(HL, DAL, bx, bv0, bv1, bv2, bw1, bw2, fkmax)
                two = lambda e: 2.0 ** e
                lim = 2.0 ** 15
                # images
                assert bv0 * two(S["e_v0"]) < 2.0 ** 4 * 1.0000001
                assert bv0 * two(S["e_v0b"]) < lim                              # the db0 row
                assert bw1 * two(S["e_w1"]) < 2.0 ** 7 and bw1 * two(S["e_w1"]) >= 2.0 ** 6
                assert bv1 * two(S["s1"]) < lim
                # observations and the bias slot, plain and transposed with 2^8 x identity
                assert bx * two(S["e_x"]) < 2.0 ** 6 and two(S["e_b"]) <= 2.0 ** 5
                assert bx * two(S["e_x"]) * 256 < lim and two(S["e_b"]) * 256 < lim
                # dz0~ (hence dh0~): sum over the inputs and the bias slot
                dz0 = DO * (bv0 * two(S["e_v0"])) * (bx * two(S["e_x"])) + (bv0 * two(S["e_v0b"])) * two(S["e_b"])
                assert dz0 < lim
                # natural worst cases, then gz1~ = 2^sg gz1 and gz0~ = 2^kg (W1~ gz1~)
                Z0 = (DO + 1) * bv0 * max(bx, 1.0)
                Z1 = (H + 1) * bv1 + H * bw1 * Z0
                M = (H + 1) * bv2 + H * bw2 * Z1
                G1 = DA * bw2 * M * fkmax
                if -100 < 15 - S["EG1"] < 100:
                    assert G1 * two(S["sg"]) < lim, (G1, S)
                    acc = H * (bw1 * two(S["e_w1"])) * lim
                    assert acc * two(S["kg"]) <= lim
                # the unscale factors undo the scales: gW1 carries 2^sg; gW0 carries sg + e_w1 + kg (gz0~) + e_x (x~) + 8
                assert S["u1"] == -S["sg"] and S["u0"] == -(S["sg"] + S["e_w1"] + S["kg"] + S["e_x"] + 8)
                assert S["u0b"] - S["u0"] == S["e_x"] - S["e_b"]


# ---- a whole product in the arithmetic ------------------------------------------------------------------------------------
def _mm_img(A, b):
    """A (f32, already scaled) x b (f32) in the image form: float16 parts, float32 accumulation."""
    ah, al, ahs = image(A)
    bh, bl = split2(b)
    f = lambda v: v.astype(np.float32)
    return f(ahs) @ f(bl) + f(al) @ f(bh) + f(ah) @ f(bh)


def _mm_tr(a, b):
    """a^T-side operand as an image derived from its per-tile parts (x 2^-11), b per tile; contraction over samples."""
    ah, al = split2(a)
    bh, bl = split2(b)
    f = lambda v: v.astype(np.float32)
    k = np.float16(2.0 ** -11)
    a_lo, a_hs = (al * k).astype(np.float16), (ah * k).astype(np.float16)
    return f(a_hs) @ f(bl).T + f(a_lo) @ f(bh).T + f(ah) @ f(bh).T


def _fvp(W0, b0, W1, b1, W2, ls, x, w, v, mode):
    """J^T M J v of mean = W2^T tanh(W1^T tanh(W0^T x + b0) + b1) (+ b2), M = diag(2 / (2 sigma^2 + 1e-8)) w / N.
    mode 'f64' / 'f32': plain evaluation in that precision; 'f16x2': the kernel's arithmetic and scale plan."""
    dW0, db0, dW1, db1, dW2, db2 = v
    DO, H = W0.shape
    DA = W2.shape[1]
    N = float(w.sum())
    T = np.float64 if mode == "f64" else np.float32
    c = lambda a: np.asarray(a, dtype=T)
    h0 = np.tanh(c(W0).T @ c(x) + c(b0)[:, None])
    h1 = np.tanh(c(W1).T @ h0 + c(b1)[:, None])
    d0, d1 = 1 - h0 * h0, 1 - h1 * h1
    fk = c(2.0 / (2.0 * np.exp(2.0 * f64(ls)) + 1e-8))
    if mode != "f16x2":
        dh0 = d0 * (c(dW0).T @ c(x) + c(db0)[:, None])
        dh1 = d1 * (c(dW1).T @ h0 + c(W1).T @ dh0 + c(db1)[:, None])
        dmu = c(dW2).T @ h1 + c(W2).T @ dh1 + c(db2)[:, None]
        gmu = dmu * fk[:, None] * c(w)[None, :] / T(N)
        gz1 = (c(W2) @ gmu) * d1
        gz0 = (c(W1) @ gz1) * d0
        return [c(x) @ gz0.T, gz0.sum(1), h0 @ gz1.T, gz1.sum(1), h1 @ gmu.T, gmu.sum(1)]
    HL, DAL = int(np.log2(H)), int(np.ceil(np.log2(max(DA, 1))))
    mx = lambda *a: float(max(np.abs(t).max() for t in a))
    S = make_scales(HL, DAL, mx(x), mx(dW0, db0), mx(dW1, db1), mx(dW2, db2), mx(W1), mx(W2), float(fk.max() / N))
    p2 = lambda e: np.float32(2.0 ** e)
    xs = np.vstack([c(x) * p2(S["e_x"]), np.full((1, x.shape[1]), p2(S["e_b"]), dtype=np.float32)])
    A0 = np.hstack([c(dW0).T * p2(S["e_v0"]), (c(db0) * p2(S["e_v0b"]))[:, None]])
    dh0 = d0 * _mm_img(A0, xs)
    assert np.abs(dh0).max() < 2.0 ** 15
    dz1 = _mm_img(c(dW1).T * p2(S["s1"]), h0) + _mm_img(c(W1).T * p2(S["e_w1"]), dh0) + (c(db1) * p2(S["s1"]))[:, None]
    dh1 = d1 * dz1
    dmu = (c(dW2) * p2(S["s1"])).T @ h1 + c(W2).T @ dh1 + (c(db2) * p2(S["s1"]))[:, None]
    fkS = (fk / np.float32(N)) * p2(S["sg"] - S["s1"])
    gmu = c(w)[None, :] * (dmu * fkS[:, None])
    gz1 = (c(W2) @ gmu) * d1
    assert np.abs(gz1).max() < 2.0 ** 15
    gz0 = (_mm_img(c(W1) * p2(S["e_w1"]), gz1) * p2(S["kg"])) * d0
    assert np.abs(gz0).max() < 2.0 ** 15
    gW1 = _mm_tr(h0, gz1) * p2(S["u1"])
    gW0e = _mm_tr(xs * np.float32(256.0), gz0)
    gW0, gb0 = gW0e[:DO] * p2(S["u0"]), gW0e[DO] * p2(S["u0b"])
    return [gW0, gb0, gW1, gz1.sum(1) * p2(S["u1"]), (h1 @ gmu.T) * p2(S["u1"]), gmu.sum(1) * p2(S["u1"])]


def test_a_whole_product_in_the_f16_arithmetic_is_as_close_to_float64_as_an_f32_evaluation():
    rng = np.random.RandomState(3)
    DO, H, DA, B = 13, 32, 2, 512
    for obs_scale, vec_scale, log_std in ((1.0, 1.0, 0.0), (1e-4, 1.0, 0.0), (1e4, 1e-6, 0.0), (30.0, 1e6, -4.0), (1.0, 1e-8, 3.0)):
        W0 = rng.uniform(-1, 1, (DO, H)) * np.sqrt(6.0 / (DO + H))
        W1 = rng.uniform(-1, 1, (H, H)) * np.sqrt(6.0 / (2 * H))
        W2 = rng.uniform(-1, 1, (H, DA)) * np.sqrt(6.0 / (H + DA))
        b0, b1 = 0.1 * rng.standard_normal(H), 0.1 * rng.standard_normal(H)
        ls = np.full(DA, log_std)
        x = rng.standard_normal((DO, B)) * obs_scale
        w = (rng.rand(B) > 0.1).astype(np.float64)
        v = [rng.standard_normal(s) * vec_scale for s in ((DO, H), (H,), (H, H), (H,), (H, DA), (DA,))]
        f32in = lambda a: np.asarray(a, dtype=np.float32)
        args = [f32in(a) for a in (W0, b0, W1, b1, W2, ls, x, w)]
        v32 = [f32in(a) for a in v]
        ref = _fvp(*args, v32, "f64")
        f32 = _fvp(*args, v32, "f32")
        h16 = _fvp(*args, v32, "f16x2")
        cat = lambda parts: np.concatenate([f64(p).ravel() for p in parts])
        r, a32, a16 = cat(ref), cat(f32), cat(h16)
        assert np.all(np.isfinite(a16))
        scale = np.abs(r).max()
        e32, e16 = np.abs(a32 - r).max() / scale, np.abs(a16 - r).max() / scale
        assert e16 <= 2.0 * e32 + 2e-6, (obs_scale, vec_scale, log_std, e16, e32)
        # (observations of 1e4 saturate the tanh layers: 1 - h^2 cancels in f32 for BOTH evaluations, 5e-4 of float64)
        assert e16 <= 5e-5 or obs_scale >= 1e4
