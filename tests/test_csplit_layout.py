"""The index arithmetic of csrc/policy_csplit_kernels.hip restated in numpy (no GPU): parts images in LDS, the
transposing read ds_read_b64_tr_b16 (lane map as tools/ubench/bf16_split_layout.hip pins it on the device), the operand
maps of v_mfma_f32_32x32x16_bf16 and the operand images of cs_stage_kernel -- a forward layer product and a product over
the sample axis computed THROUGH these layouts must equal the plain matrix products, and every gradient entry must land
on the parameter the kernel's final row write names.  This is a model of the layouts only (values are small integers,
exact in bf16); accuracy of the split arithmetic is tests/test_split_arithmetic.py, the kernel itself is
tests/test_gpu_csplit.py."""
import numpy as np

CH, HSTR, TILE_IMG = 1088, 576, 6 * 1088


def frag_unit(r, half):
    return (r & 3) + 8 * (r >> 2) + 4 * half


def tr_unit(l32):
    return frag_unit(8 * ((l32 >> 4) & 1) + 4 * ((l32 >> 2) & 1) + (l32 & 3), (l32 >> 3) & 1)


def lane_off(lane):
    return (lane & 31) * 16 + (lane >> 5) * HSTR


def tr_off(lane):
    return (((lane >> 4) & 1) * 3) * CH + (8 * (lane >> 5) + ((lane & 15) >> 2)) * 16 + ((lane & 3) >> 1) * HSTR + (lane & 1) * 8


def publish(lds, base, t, frag):
    """frag[lane][16] (lane = sample + 32 half, register r = unit frag_unit(r, half)) -> chunks (2 t + kb, part 0)."""
    for lane in range(64):
        for kb in range(2):
            off = base + t * TILE_IMG + (kb * 3 + 0) * CH + lane_off(lane)
            lds[off // 2: off // 2 + 8] = frag[lane, 8 * kb: 8 * kb + 8]


def ds_read_tr16_b64(lds, addr):
    """addr[64] byte addresses -> out[64][4]: per 16-lane group, lane 4 r + q supplies row r, columns 4 q .. 4 q + 3 and
    lane i receives column i, rows 0..3."""
    out = np.zeros((64, 4), dtype=lds.dtype)
    for g in range(4):
        m = np.zeros((4, 16), dtype=lds.dtype)
        for i in range(16):
            r, q = i >> 2, i & 3
            a = addr[16 * g + i] // 2
            m[r, 4 * q: 4 * q + 4] = lds[a: a + 4]
        for i in range(16):
            out[16 * g + i] = m[:, i]
    return out


def cs_tr(lds, src_base, kbs):
    """operand[lane][8] of part 0"""
    op = np.zeros((64, 8), dtype=lds.dtype)
    for e in range(2):
        addr = [src_base + tr_off(lane) + kbs * 256 + 64 * e for lane in range(64)]
        op[:, 4 * e: 4 * e + 4] = ds_read_tr16_b64(lds, addr)
    return op


_LANES = np.arange(64)
_ROW = np.array([[frag_unit(r, lane >> 5) for r in range(16)] for lane in range(64)])      # [lane][r] -> row of D
_COL = np.repeat((_LANES & 31)[:, None], 16, axis=1)


def mfma_32x32x16(a, b):
    """a[lane][8], b[lane][8] -> D[lane][16]: A lane (i, h) element j and B lane (n, h) element j meet at k = 8 h + j;
    D register r of lane (n, h) is row frag_unit(r, h), column n."""
    A = np.concatenate([a[:32], a[32:]], axis=1)                 # [i][k = 8 h + j]
    Bm = np.concatenate([b[:32], b[32:]], axis=1).T              # [k][n]
    return (A @ Bm)[_ROW, _COL]


def fragment_of(mat_units_by_samples, t):
    """[units][32 samples] -> frag[lane][16] of row tile t"""
    return mat_units_by_samples[32 * t + _ROW, _COL]


def test_forward_product_through_the_parts_image_and_the_operand_image():
    rng = np.random.RandomState(0)
    Hin, Hout = 64, 64
    W = rng.randint(-3, 4, size=(Hin, Hout)).astype(np.float64)            # W[k][i], the reference's [in][out]
    h = rng.randint(-3, 4, size=(Hin, 32)).astype(np.float64)              # [unit][sample]
    lds = np.zeros(64 * 1024, dtype=np.float64)                            # one element per 2 bytes
    for t in range(Hin // 32):
        publish(lds, 0, t, fragment_of(h, t))
    for t_out in range(Hout // 32):
        acc = np.zeros((64, 16))
        for kbg in range(Hin // 16):
            a = np.zeros((64, 8))
            for lane in range(64):                                         # cs_stage_kernel, kind 1
                i, half = lane & 31, lane >> 5
                for j in range(8):
                    ku = 32 * (kbg >> 1) + frag_unit(8 * (kbg & 1) + j, half)
                    a[lane, j] = W[ku, 32 * t_out + i]
            b = np.zeros((64, 8))
            for lane in range(64):                                         # cs_gemm's B read: chunk kbg, this lane's slot
                off = (kbg * 3) * CH + lane_off(lane)
                b[lane] = lds[off // 2: off // 2 + 8]
            acc += mfma_32x32x16(a, b)
        want = fragment_of(W.T @ h, t_out)
        assert np.array_equal(acc, want)


def test_sample_axis_product_through_transposing_reads():
    rng = np.random.RandomState(1)
    Ha, Hb = 64, 64
    ha = rng.randint(-3, 4, size=(Ha, 32)).astype(np.float64)
    gz = rng.randint(-3, 4, size=(Hb, 32)).astype(np.float64)
    lds = np.zeros(64 * 1024, dtype=np.float64)
    base_a, base_b = 0, 4 * TILE_IMG
    for t in range(2):
        publish(lds, base_a, t, fragment_of(ha, t))
        publish(lds, base_b, t, fragment_of(gz, t))
    want = ha @ gz.T                                                        # gW[k_prev][k_cur]
    for ti in range(2):
        for tj in range(2):
            acc = np.zeros((64, 16))
            for kbs in range(2):
                acc += mfma_32x32x16(cs_tr(lds, base_a + ti * TILE_IMG, kbs), cs_tr(lds, base_b + tj * TILE_IMG, kbs))
            for lane in range(64):                                          # the kernel's final row write
                lj, lh = lane & 31, lane >> 5
                for r in range(16):
                    row = 32 * ti + tr_unit(frag_unit(r, lh))
                    col = 32 * tj + tr_unit(lj)
                    assert acc[lane, r] == want[row, col]


def test_input_image_rows_are_input_slots_and_mean_image_columns_are_action_slots():
    rng = np.random.RandomState(2)
    x = rng.randint(-3, 4, size=(32, 32)).astype(np.float64)                # [slot][sample]; slots >= 21 zero
    x[21:] = 0
    gz = rng.randint(-3, 4, size=(32, 32)).astype(np.float64)
    gmu = rng.randint(-3, 4, size=(8, 32)).astype(np.float64)               # [action slot][sample]
    lds = np.zeros(64 * 1024, dtype=np.float64)
    bx, bg, bm = 0, TILE_IMG, 2 * TILE_IMG
    for lane in range(64):                                                  # the kernel's x item: chunk kb, slots 16 kb + 8 half + j
        n, half = lane & 31, lane >> 5
        for kb in range(2):
            off = bx + (kb * 3) * CH + lane_off(lane)
            lds[off // 2: off // 2 + 8] = [x[16 * kb + 8 * half + j, n] for j in range(8)]
        if half == 0:                                                       # the mean's cotangent: lane (sample, half 0) = 8 slots
            off = bm + n * 16
            lds[off // 2: off // 2 + 8] = gmu[:, n]
    publish(lds, bg, 0, fragment_of(gz, 0))
    acc0 = np.zeros((64, 16)); acco = np.zeros((64, 16))
    for kbs in range(2):
        acc0 += mfma_32x32x16(cs_tr(lds, bx, kbs), cs_tr(lds, bg, kbs))     # gW0 += x^T gz
        acco += mfma_32x32x16(cs_tr(lds, bg, kbs), cs_tr(lds, bm, kbs))     # gWo += h^T gmu  (h := gz here)
    w0 = x @ gz.T
    wo = gz @ gmu.T                                                         # [unit][slot]
    for lane in range(64):
        lj, lh = lane & 31, lane >> 5
        for r in range(16):
            assert acc0[lane, r] == w0[frag_unit(r, lh), tr_unit(lj)]        # row = input slot, column = unit
            if lj < 8:
                assert acco[lane, r] == wo[tr_unit(frag_unit(r, lh)), lj]    # column lj = action slot lj
            else:
                assert acco[lane, r] == 0


def test_output_layer_images():
    """kinds 3/4 (rows = action slots, k = this wavefront's units) and 5 (rows = units, k = action slots): dmu's partial
    over the wavefront's 32 units lands in registers r < 4 as slot r + 4 half; gz = Wo gmu from the mean image."""
    rng = np.random.RandomState(3)
    H, DA = 64, 6
    Wo = rng.randint(-3, 4, size=(H, DA)).astype(np.float64)
    h = rng.randint(-3, 4, size=(H, 32)).astype(np.float64)
    gmu = rng.randint(-3, 4, size=(DA, 32)).astype(np.float64)
    lds = np.zeros(64 * 1024, dtype=np.float64)
    for t in range(2):
        publish(lds, 0, t, fragment_of(h, t))
    bm = 4 * TILE_IMG
    for n in range(32):
        lds[(bm + n * 16) // 2: (bm + n * 16) // 2 + DA] = gmu[:, n]
    for wave in range(2):
        acc = np.zeros((64, 16))
        for kb in range(2):                                                 # k-blocks 2 w, 2 w + 1 of the image and of lH
            kbg = 2 * wave + kb
            a = np.zeros((64, 8))
            for lane in range(64):
                i, half = lane & 31, lane >> 5
                for j in range(8):
                    ku = 32 * (kbg >> 1) + frag_unit(8 * (kbg & 1) + j, half)
                    a[lane, j] = Wo[ku, i] if i < DA else 0.0
            b = np.zeros((64, 8))
            for lane in range(64):
                off = wave * TILE_IMG + (kb * 3) * CH + lane_off(lane)     # `own + lH`
                b[lane] = lds[off // 2: off // 2 + 8]
            acc += mfma_32x32x16(a, b)
        want = Wo[32 * wave: 32 * wave + 32].T @ h[32 * wave: 32 * wave + 32]     # [slot][sample], partial over the wave's units
        for lane in range(64):
            n, half = lane & 31, lane >> 5
            for r in range(4):
                k = r + 4 * half
                assert acc[lane, r] == (want[k, n] if k < DA else 0.0)
        # back: gz[unit][sample] = sum_k Wo[unit][k] gmu[k][sample], row tile `wave`
        a = np.zeros((64, 8)); b = np.zeros((64, 8))
        for lane in range(64):
            i, half = lane & 31, lane >> 5
            for j in range(8):
                k = 8 * half + j
                a[lane, j] = Wo[32 * wave + i, k] if k < DA else 0.0
            off = bm + lane_off(lane)
            b[lane] = lds[off // 2: off // 2 + 8]
        got = mfma_32x32x16(a, b)
        assert np.array_equal(got, fragment_of(Wo @ gmu, wave))
