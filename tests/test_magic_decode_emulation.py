"""CPU (-m "not gpu"): the bit algebra of the packed magic-number field decode (csrc/gemv.hip MagicF16<3 / 8>, csrc/gemm.hip
Deq<3 / 8, f16>) restated in numpy, instruction for instruction (v_perm_b32, v_alignbit_b32, shifts, masks, fp16 arithmetic), and
checked against the plain field extraction of the reference layout (qlinear_cuda_old.py:317-344: 32 three-bit values in 3 words,
fields 10 and 21 straddle; :295-316: four 8-bit values per word).  This is what was run before the first GPU session; it pins
  * which field lands in which half of which pair (ka / kb: the k order of the matrix-core slots, which x has to follow),
  * that every intermediate is exactly representable in fp16 (the decode is exact, not merely close),
  * the zero-point range of both conventions (wrap: 0..maxq, no-wrap: 1..maxq + 1)."""
import numpy as np
import pytest

MAGIC = 0x64006400


def perm(s0, s1, sel):
    """v_perm_b32 D, S0, S1, sel: result byte i = byte sel[i] of {S0 (bytes 4-7), S1 (bytes 0-3)}."""
    b = [(s1 >> (8 * i)) & 0xff for i in range(4)] + [(s0 >> (8 * i)) & 0xff for i in range(4)] + [None] * 4 + [0x00]     # selector 12 = constant 0x00
    return sum(b[(sel >> (8 * i)) & 0xff] << (8 * i) for i in range(4))


def alignbit(hi, lo, sh):
    """v_alignbit_b32: low 32 bits of {hi, lo} >> sh."""
    return (((hi << 32) | lo) >> sh) & 0xffffffff


def halves(u):
    return np.array([u & 0xffff, (u >> 16) & 0xffff], dtype=np.uint16).view(np.float16)


def pk_fma(t, mask, scale, c):
    """as_f16x2((t & mask) | magic) * scale + c with ONE rounding (v_pk_fma_f16); asserts the result is exact in fp16."""
    v = halves((t & mask) | MAGIC).astype(np.float64)
    r = v * scale + c
    r16 = r.astype(np.float16)
    assert np.all(r16.astype(np.float64) == r), "intermediate not exact in fp16"
    return r16


def five(t, z):
    t6 = t >> 6
    return [pk_fma(t, 0x00070007, 1.0, -(1024 + z)), pk_fma(t, 0x00380038, 0.125, -(128 + z)), pk_fma(t, 0x01C001C0, 1 / 64, -(16 + z)),
            pk_fma(t6, 0x00380038, 0.125, -(128 + z)), pk_fma(t6, 0x01C001C0, 1 / 64, -(16 + z))]


def gemv_pairs_3bit(w, z):
    """MagicF16<3>::pairs: 16-bit windows of the 96-bit stream at bits 0/15, 30/45, 60/75, 90/93."""
    t0 = perm(w[0] >> 15, w[0], 0x05040100)
    t1 = perm(w[1] >> 13, alignbit(w[1], w[0], 30), 0x05040100)
    t2 = perm(w[2] >> 11, alignbit(w[2], w[1], 28), 0x05040100)
    t3 = perm(w[2] >> 29, w[2] >> 26, 0x05040100)
    return five(t0, z) + five(t1, z) + five(t2, z) + [pk_fma(t3, 0x00070007, 1.0, -(1024 + z))]


def ka3(p):
    return 10 * (p // 5) + p % 5 if p < 15 else 30


def kb3(p):
    return ka3(p) + 5 if p < 15 else 31


@pytest.mark.parametrize("zero_mode", ["wrap", "nowrap"])
def test_3bit_unit_pairs_match_plain_field_extraction(zero_mode):
    rng = np.random.default_rng(3)
    for _ in range(500):
        f = rng.integers(0, 8, 32)
        bits = sum(int(v) << (3 * i) for i, v in enumerate(f))
        w = [(bits >> (32 * j)) & 0xffffffff for j in range(3)]
        zf = int(rng.integers(0, 8))                                  # stored field = zero - 1
        z = ((zf + 1) & 7) if zero_mode == "wrap" else zf + 1       # qlinear_cuda_old.py:301-304 / qlinear_cuda.py:262-264
        for p, pair in enumerate(gemv_pairs_3bit(w, z)):
            assert float(pair[0]) == f[ka3(p)] - z and float(pair[1]) == f[kb3(p)] - z, (p, ka3(p), kb3(p))
    assert sorted([ka3(p) for p in range(16)] + [kb3(p) for p in range(16)]) == list(range(32))      # every k exactly once


def test_x_pair_selectors_pick_the_same_k_as_the_weight_pairs():
    """magic_x_pair: x of a unit sits two values per register in natural order; pair p needs (x[ka], x[kb]) in one register."""
    x = np.arange(32, dtype=np.uint16) + 100                        # value = 100 + k
    regs = [int(x[2 * r]) | (int(x[2 * r + 1]) << 16) for r in range(16)]
    for p in range(16):
        a, b = ka3(p), kb3(p)
        sel = (0x0302 if a & 1 else 0x0100) | ((0x0706 if b & 1 else 0x0504) << 16)
        r = perm(regs[b >> 1], regs[a >> 1], sel)
        assert (r & 0xffff, r >> 16) == (100 + a, 100 + b)
    regs8 = [int(x[0]) | (int(x[1]) << 16), int(x[2]) | (int(x[3]) << 16)]        # 8-bit: pairs (0, 2), (1, 3)
    assert perm(regs8[1], regs8[0], 0x0100 | (0x0504 << 16)) == (100 | (102 << 16))
    assert perm(regs8[1], regs8[0], 0x0302 | (0x0706 << 16)) == (101 | (103 << 16))


@pytest.mark.parametrize("zero_mode", ["wrap", "nowrap"])
def test_8bit_word_pairs(zero_mode):
    rng = np.random.default_rng(8)
    for _ in range(500):
        f = rng.integers(0, 256, 4)
        w = sum(int(v) << (8 * i) for i, v in enumerate(f))
        zf = int(rng.integers(0, 256))
        z = ((zf + 1) & 255) if zero_mode == "wrap" else zf + 1      # up to 256
        p0 = pk_fma(w, 0x00ff00ff, 1.0, -(1024 + z))                 # MagicF16<8>::pairs
        p1 = pk_fma(w >> 8, 0x00ff00ff, 1.0, -(1024 + z))
        assert (float(p0[0]), float(p0[1]), float(p1[0]), float(p1[1])) == (f[0] - z, f[2] - z, f[1] - z, f[3] - z)


def test_gemm_fragments_come_out_in_the_4bit_slot_order():
    """Deq<3, f16> / Deq<8, f16>::frag: a lane's 8 consecutive k of one column -> 4 registers (k0,k4), (k1,k5), (k2,k6), (k3,k7)."""
    rng = np.random.default_rng(11)
    for _ in range(500):
        f = rng.integers(0, 8, 8)
        z = int(rng.integers(0, 9))
        junk = int(rng.integers(0, 1 << 30))                         # the window's upper bits belong to the next fields
        v = (sum(int(x) << (3 * i) for i, x in enumerate(f)) | (junk << 24)) & 0xffffffff
        t = perm(v >> 12, v, 0x05040100)
        t6 = t >> 6
        h = [pk_fma(t, 0x00070007, 1.0, -(1024 + z)), pk_fma(t, 0x00380038, 0.125, -(128 + z)),
             pk_fma(t, 0x01C001C0, 1 / 64, -(16 + z)), pk_fma(t6, 0x00380038, 0.125, -(128 + z))]
        for i in range(4):
            assert (float(h[i][0]), float(h[i][1])) == (f[i] - z, f[i + 4] - z)
        f8 = rng.integers(0, 256, 8)
        z8 = int(rng.integers(0, 257))
        w0 = sum(int(x) << (8 * i) for i, x in enumerate(f8[:4]))
        w1 = sum(int(x) << (8 * i) for i, x in enumerate(f8[4:]))
        lo, hi = perm(w1, w0, 0x05040100), perm(w1, w0, 0x07060302)
        h8 = [pk_fma(lo, 0x00ff00ff, 1.0, -(1024 + z8)), pk_fma(lo >> 8, 0x00ff00ff, 1.0, -(1024 + z8)),
              pk_fma(hi, 0x00ff00ff, 1.0, -(1024 + z8)), pk_fma(hi >> 8, 0x00ff00ff, 1.0, -(1024 + z8))]
        for i in range(4):
            assert (float(h8[i][0]), float(h8[i][1])) == (f8[i] - z8, f8[i + 4] - z8)


def test_zero_point_constants_are_exact_fp16_bit_patterns():
    """setup(): -(1024 + z) is built as an integer add on the fp16 bit pattern 0xE400 (ulp 1 in [1024, 2048)); -(128 + z) and
    -(16 + z) by one exact fp16 addition each."""
    for z in range(0, 257):
        c1 = np.array([0xE400 + z], dtype=np.uint16).view(np.float16)[0]
        assert float(c1) == -(1024 + z)
        if z <= 8:
            c3, c6 = np.float16(c1) + np.float16(896.0), np.float16(c1) + np.float16(1008.0)
            assert float(c3) == -(128 + z) and float(c6) == -(16 + z)


# ---------------------------------------------------------------------------------------------------- round 3: 2-bit, and the windows of gemm_mid_kernel
def four2(t, z):
    """MagicF16<2>::four / Deq2::diffs: fields at bits 0 / 2 / 4 / 6 of both halves."""
    return [pk_fma(t, 0x00030003, 1.0, -(1024 + z)), pk_fma(t, 0x000C000C, 0.25, -(256 + z)), pk_fma(t, 0x00300030, 1 / 16, -(64 + z)),
            pk_fma(t, 0x00C000C0, 1 / 64, -(16 + z))]


@pytest.mark.parametrize("zero_mode", ["wrap", "nowrap"])
def test_2bit_word_pairs(zero_mode):
    """MagicF16<2>::pairs (csrc/gemv.hip): 16 two-bit values per word (qlinear_cuda_old.py:295-316 layout); two v_perm spread bytes 0 / 1 and 2 / 3 over
    the halves of a register; pair p holds fields (ka(p), kb(p)) = (p, p + 4) for p < 4 and (p + 4, p + 8) above; every intermediate exact in fp16."""
    rng = np.random.default_rng(2)
    ka = lambda p: p if p < 4 else p + 4
    for _ in range(500):
        f = rng.integers(0, 4, 16)
        w = sum(int(v) << (2 * i) for i, v in enumerate(f))
        zf = int(rng.integers(0, 4))
        z = ((zf + 1) & 3) if zero_mode == "wrap" else zf + 1        # up to 4
        pairs = four2(perm(w, w, 0x0c010c00), z) + four2(perm(w, w, 0x0c030c02), z)
        for p in range(8):
            assert (float(pairs[p][0]), float(pairs[p][1])) == (f[ka(p)] - z, f[ka(p) + 4] - z), p
    for z in range(0, 5):                                            # setup(): the shifted constants are exact fp16 values
        c1 = np.array([0xE400 + z], dtype=np.uint16).view(np.float16)[0]
        assert [float(np.float16(c1) + np.float16(k)) for k in (768.0, 960.0, 1008.0)] == [-(256 + z), -(64 + z), -(16 + z)]


def test_mid_kernel_lane_windows():
    """gemm_mid_kernel (csrc/gemm_mid.hip): what lane (column, k-octet kg) cuts out of the landing area for its 8 consecutive k of a 32-deep K-step.
    3-bit: 24 bits at bit 24 kg of the column's 96-bit stream (three packed rows) by ONE 64-bit shift of (w0, w1) or (w1, w2); 2-bit: half kg & 1 of word
    kg >> 1; 8-bit: the words of packed rows 2 kg and 2 kg + 1.  Then Deq3 / Deq2 / Deq8 give (k0,k4), (k1,k5), (k2,k6), (k3,k7)."""
    rng = np.random.default_rng(33)
    for _ in range(300):
        f3 = rng.integers(0, 8, 32)
        stream = sum(int(v) << (3 * i) for i, v in enumerate(f3))
        w = [(stream >> (32 * i)) & 0xffffffff for i in range(3)]
        z = int(rng.integers(0, 9))
        for kg in range(4):
            sh = (0, 24, 16, 40)[kg]
            lo, hi = (w[0], w[1]) if kg < 2 else (w[1], w[2])
            v = ((((hi << 32) | lo) >> sh) & 0xffffffff) & 0xFFFFFF
            t = perm(v >> 12, v, 0x05040100)
            t6 = t >> 6
            h = [pk_fma(t, 0x00070007, 1.0, -(1024 + z)), pk_fma(t, 0x00380038, 0.125, -(128 + z)), pk_fma(t, 0x01C001C0, 1 / 64, -(16 + z)),
                 pk_fma(t6, 0x00380038, 0.125, -(128 + z))]
            for i in range(4):
                assert (float(h[i][0]), float(h[i][1])) == (f3[8 * kg + i] - z, f3[8 * kg + i + 4] - z), (kg, i)
        f2 = rng.integers(0, 4, 32)
        w2 = [sum(int(v) << (2 * i) for i, v in enumerate(f2[16 * r:16 * r + 16])) for r in range(2)]
        z2 = int(rng.integers(0, 5))
        for kg in range(4):
            word = w2[kg >> 1]
            v16 = (word >> 16) if (kg & 1) else (word & 0xffff)
            h = four2(perm(v16, v16, 0x0c010c00), z2)
            for i in range(4):
                assert (float(h[i][0]), float(h[i][1])) == (f2[8 * kg + i] - z2, f2[8 * kg + i + 4] - z2), (kg, i)


def test_mid_kernel_3bit_zero_point_run():
    """3-bit zero-points of a 64-column strip are a 24-byte run at byte 24 * strip of the qzeros row -- 16-byte aligned only on even strips.  The kernel's
    table holds three aligned 16-byte pieces from floor16 on; lane j reads 12 bits at bit 8 * zstart + 12 j with two aligned words and a funnel shift."""
    rng = np.random.default_rng(5)
    N = 128 * 3                                                      # the 3-bit plan wants N % 128 == 0, so the row is a multiple of 16 bytes
    zf = rng.integers(0, 8, N)
    bits = sum(int(v) << (3 * i) for i, v in enumerate(zf))
    rowb = N * 3 // 8
    row = bits.to_bytes(rowb, "little")
    for strip in range(N // 64):
        start = 24 * strip
        base16 = start & ~15
        pieces = b""
        for i in range(3):
            pz = base16 + 16 * i
            pieces += row[pz:pz + 16] if pz + 16 <= rowb else row[rowb - 16:rowb]     # a piece beyond the row is never needed
        zstart = start & 15
        words = [int.from_bytes(pieces[4 * i:4 * i + 4], "little") for i in range(12)]
        for j in range(16):
            bit = 8 * zstart + 12 * j
            wi = bit >> 5
            zz = ((((words[wi + 1] if wi + 1 < 12 else 0) << 32) | words[wi]) >> (bit & 31)) & 0xffffffff
            for t in range(4):
                assert (zz >> (3 * t)) & 7 == zf[64 * strip + 4 * j + t], (strip, j, t)


# ---------------------------------------------------------------------------------------------------- round 3: the balanced tail's block -> work mapping
def _xcd_remap(b, n):
    """common.cuh xcd_remap: hardware block b runs on XCD b % 8; consecutive logical ids land on one XCD."""
    q, r = n >> 3, n & 7
    xcd, idx = b & 7, b >> 3
    base = xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q
    return base + idx


def _tile_of(L, nbm, nbn):
    """gemm_kernel: logical tile -> (bm, bn): column blocks 8 tiles wide walked row by row, then the narrower last block."""
    full, per = nbn >> 3, nbm * 8
    if L < full * per:
        cb, r = divmod(L, per)
        return r >> 3, cb * 8 + (r & 7)
    w = nbn & 7
    r = L - full * per
    bm = r // w
    return bm, full * 8 + (r - bm * w)


@pytest.mark.parametrize("nbm,nbn,tail,lg", [(17, 16, 16, 2), (12, 43, 4, 3), (8, 43, 88, 1), (6, 43, 2, 3), (58, 9, 10, 1), (13, 43, 47, 2), (4, 16, 64, 2),
                                             (18, 16, 32, 2), (30, 9, 14, 1)])
def test_balanced_tail_block_mapping_is_a_bijection(nbm, nbn, tail, lg):
    """gemm_kernel<..., TAIL>: blocks [0, whole) are whole tiles, blocks behind them are the K slices of the last `tail` logical tiles.  Every whole
    tile exactly once, every (tail tile, slice) exactly once, every (bm, bn) of the grid covered, the slices of a tile on one XCD whenever whole tiles
    and tail tiles are multiples of 8 (speed only: the exchange is placement-independent), and the flag words of a tile inside its 16."""
    tiles, s = nbm * nbn, 1 << lg
    whole = tiles - tail
    grid = whole + tail * s
    seen_whole, seen_piece, xcds = set(), set(), {}
    for b in range(grid):
        if b < whole:
            L = _xcd_remap(b, whole)
            assert 0 <= L < whole and L not in seen_whole
            seen_whole.add(L)
        else:
            j = _xcd_remap(b - whole, tail << lg)
            L, sl = whole + (j >> lg), j & (s - 1)
            assert whole <= L < tiles and (L, sl) not in seen_piece
            seen_piece.add((L, sl))
            xcds.setdefault(L, set()).add(b & 7)
    assert len(seen_whole) == whole and len(seen_piece) == tail * s
    assert {_tile_of(L, nbm, nbn) for L in range(tiles)} == {(bm, bn) for bm in range(nbm) for bn in range(nbn)}
    if whole % 8 == 0 and tail % 8 == 0:                 # (fewer than 8 tail tiles: an XCD's share of the pieces is less than one tile's slices)
        assert all(len(v) == 1 for v in xcds.values())
    assert all(len(v) <= 2 for v in xcds.values()) or tail < 8
    assert 1 + s <= 16 and 16 * tail * 4 <= 32768
