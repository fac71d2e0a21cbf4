"""CPU model of the register-level dataflow of the D = 64 attention kernels (csrc/flash_attn_m16x.cuh, csrc/flash_attn_m32x.cuh).

The kernels keep P^T in the registers the S^T MFMAs wrote and fetch V^T fragments in the key order those registers have, through
swizzled LDS images and transposing reads. This file restates, in numpy, the lane layouts of the three instructions involved
(`v_mfma_f32_16x16x32_f16`, `v_mfma_f32_32x32x16_f16`, `ds_read_b64_tr_b16`) and the address formulas of the two kernels, runs ONE wave's
KV tile through them and compares with softmax(Q K^T) V computed directly. It is a model of the design (the GPU tests are the parity
tests); it exists so that a change of a fragment formula can be checked here before a GPU is spent on it.
"""
import numpy as np

D, BC, ROW = 64, 128, 128  # head dim, keys per tile, bytes per K / V row


# ---------------------------------------------------------------- instruction models (one wave = 64 lanes)
def mfma_16x16x32(a, b, c):
    """a[lane][8]: A[i = lane & 15][k = 8 (lane >> 4) + e]; b[lane][8]: B[k = 8 (lane >> 4) + e][j = lane & 15];
    c / result [lane][4]: D[i = 4 (lane >> 4) + r][j = lane & 15]."""
    A, B = np.zeros((16, 32)), np.zeros((32, 16))
    for l in range(64):
        A[l & 15, 8 * (l >> 4):8 * (l >> 4) + 8] = a[l]
        B[8 * (l >> 4):8 * (l >> 4) + 8, l & 15] = b[l]
    Dm = A @ B
    out = np.array(c, dtype=np.float64)
    for l in range(64):
        for r in range(4):
            out[l, r] += Dm[4 * (l >> 4) + r, l & 15]
    return out


def mfma_32x32x16(a, b, c):
    """a[lane][8]: A[i = lane & 31][k = 8 (lane >> 5) + e]; b[lane][8]: B[k = 8 (lane >> 5) + e][j = lane & 31];
    c / result [lane][16]: D[i = 8 (r >> 2) + 4 (lane >> 5) + (r & 3)][j = lane & 31]."""
    A, B = np.zeros((32, 16)), np.zeros((16, 32))
    for l in range(64):
        A[l & 31, 8 * (l >> 5):8 * (l >> 5) + 8] = a[l]
        B[8 * (l >> 5):8 * (l >> 5) + 8, l & 31] = b[l]
    Dm = A @ B
    out = np.array(c, dtype=np.float64)
    for l in range(64):
        for r in range(16):
            out[l, r] += Dm[8 * (r >> 2) + 4 * (l >> 5) + (r & 3), l & 31]
    return out


def lds_halves(img, byte_addr, n):
    return img[byte_addr // 2:byte_addr // 2 + n]


def ds_read_b64_tr_b16(img, addr):
    """addr[lane]: byte address of 4 halves. Inside each group of 16 lanes, lane i supplies M[i >> 2][4 (i & 3) .. + 3] and receives
    (M[0][i], M[1][i], M[2][i], M[3][i])."""
    out = np.zeros((64, 4))
    for g in range(4):
        M = np.zeros((4, 16))
        for i in range(16):
            M[i >> 2, 4 * (i & 3):4 * (i & 3) + 4] = lds_halves(img, addr[16 * g + i], 4)
        for i in range(16):
            out[16 * g + i] = M[:, i]
    return out


# ---------------------------------------------------------------- the LDS images the LDS-DMA writes (GeoM16<64, *, 128>)
def swz_k(row):
    return (row >> 1) & 7


def swz_v(row):
    return ((row >> 1) & 3) << 1


def image(tile, swz):
    """LDS[row][chunk c] = tile[row][chunk c ^ swz(row)] (the swizzle is applied on the source side of the DMA)."""
    img = np.zeros(BC * D)
    for row in range(BC):
        for c in range(8):
            img[row * D + 8 * c:row * D + 8 * c + 8] = tile[row, 8 * (c ^ swz(row)):8 * (c ^ swz(row)) + 8]
    return img


def reference(q, k, v):
    s = q @ k.T
    p = np.exp2(s - s.max(axis=1, keepdims=True))
    return (p @ v) / p.sum(axis=1, keepdims=True)


def inputs(seed):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((32, D)), rng.standard_normal((BC, D)), rng.standard_normal((BC, D))


# ---------------------------------------------------------------- flash_attn_m16x.cuh: 16x16x32, 32 query rows per wave
def test_m16x_register_dataflow_reproduces_attention():
    q, k, v = inputs(1)
    kimg, vimg = image(k, swz_k), image(v, swz_v)
    NKB, NKS, NQB, NU, NDB = BC // 16, D // 32, 2, BC // 32, D // 16
    lane = np.arange(64)
    i16, g4 = lane & 15, lane >> 4
    qf = [[np.array([q[qb * 16 + i16[l], ks * 32 + g4[l] * 8:ks * 32 + g4[l] * 8 + 8] for l in range(64)]) for ks in range(NKS)] for qb in range(NQB)]
    kbase = i16 * ROW + ((g4 ^ swz_k(i16)) << 4)
    v_row = 4 * g4 + (i16 >> 2)
    vbase = v_row * ROW + ((((i16 & 3) >> 1) ^ swz_v(v_row)) << 4) + ((i16 & 1) << 3)
    s = np.zeros((NKB, NQB, 64, 4))
    for kb in range(NKB):
        for ks in range(NKS):
            kf = np.array([lds_halves(kimg, (kbase[l] ^ (ks << 6)) + kb * 16 * ROW, 8) for l in range(64)])
            for qb in range(NQB):
                s[kb, qb] = mfma_16x16x32(kf, qf[qb][ks], s[kb, qb])
    # lane (g4, i16), register r of block kb: key 16 kb + 4 g4 + r, query 16 qb + i16
    m = np.zeros((NQB, 16))
    for qb in range(NQB):
        for j in range(16):
            m[qb, j] = max(s[kb, qb, 16 * g + j, r] for kb in range(NKB) for g in range(4) for r in range(4))
    p = np.exp2(s - np.stack([m[qb][i16] for qb in range(NQB)])[None, :, :, None])  # [kb][qb][lane][r], the row maximum of the lane's query
    l_lane = p.sum(axis=(0, 3))  # [qb][lane]: partial row sums
    # P^T k-step u, k-slot 8 g4 + e: e = (kb & 1) * 4 + r of block kb = 2u + (e >> 2)
    pf = np.zeros((NU, NQB, 64, 8))
    for kb in range(NKB):
        for r in range(4):
            pf[kb >> 1, :, :, (kb & 1) * 4 + r] = p[kb, :, :, r]
    ot = np.zeros((NDB, NQB, 64, 4))
    for u in range(NU):
        for db in range(NDB):
            addr = (vbase ^ (db << 5)) + 32 * u * ROW
            vf = np.concatenate([ds_read_b64_tr_b16(vimg, addr), ds_read_b64_tr_b16(vimg, addr + 16 * ROW)], axis=1)
            for qb in range(NQB):
                ot[db, qb] = mfma_16x16x32(vf, pf[u, qb], ot[db, qb])
    out = np.zeros((32, D))
    for qb in range(NQB):
        for j in range(16):
            l_tot = sum(l_lane[qb, 16 * g + j] for g in range(4))
            for db in range(NDB):
                for g in range(4):
                    out[16 * qb + j, 16 * db + 4 * g:16 * db + 4 * g + 4] = ot[db, qb, 16 * g + j] / l_tot
    assert np.abs(out - reference(q, k, v)).max() < 1e-9


# ---------------------------------------------------------------- probe/flash_attn_m32x.cuh: 32x32x16, the same images
def test_m32x_register_dataflow_reproduces_attention():
    q, k, v = inputs(2)
    kimg, vimg = image(k, swz_k), image(v, swz_v)
    NKB, NKS, NDB = BC // 32, D // 16, D // 32
    lane = np.arange(64)
    l31, hi = lane & 31, lane >> 5
    qf = [np.array([q[l31[l], 16 * ks + 8 * hi[l]:16 * ks + 8 * hi[l] + 8] for l in range(64)]) for ks in range(NKS)]
    kbase = l31 * ROW + ((hi ^ swz_k(l31)) << 4)
    i16, dh = lane & 15, (lane >> 4) & 1
    v_row = 4 * hi + (i16 >> 2)
    vbase = v_row * ROW + ((((dh << 1) | ((i16 & 3) >> 1)) ^ swz_v(v_row)) << 4) + ((i16 & 1) << 3)
    s = np.zeros((NKB, 64, 16))
    for t in range(NKB * NKS):  # the kernel's order: pairs of key blocks interleaved k-step by k-step
        kb, ks = 2 * (t // (2 * NKS)) + (t & 1), (t % (2 * NKS)) >> 1
        kf = np.array([lds_halves(kimg, (kbase[l] ^ (ks << 5)) + kb * 32 * ROW, 8) for l in range(64)])
        s[kb] = mfma_32x32x16(kf, qf[ks], s[kb])
    # lane (hi, l31), register r of block kb: key 32 kb + 8 (r >> 2) + 4 hi + (r & 3), query l31
    m = np.array([max(s[kb, 32 * h + j, r] for kb in range(NKB) for h in range(2) for r in range(16)) for j in range(32)])
    p = np.exp2(s - m[l31][None, :, None])
    l_lane = p.sum(axis=(0, 2))
    ot = np.zeros((NDB, 64, 16))
    for kb in range(NKB):
        for u in range(2):
            pf = p[kb, :, 8 * u:8 * u + 8]  # registers 8u .. 8u + 7 in order
            for b in range(NDB):
                addr = (vbase ^ (b << 6)) + (32 * kb + 16 * u) * ROW
                vf = np.concatenate([ds_read_b64_tr_b16(vimg, addr), ds_read_b64_tr_b16(vimg, addr + 8 * ROW)], axis=1)
                ot[b] = mfma_32x32x16(vf, pf, ot[b])
    out = np.zeros((32, D))
    for j in range(32):
        l_tot = l_lane[j] + l_lane[32 + j]
        for b in range(NDB):
            for h in range(2):
                for rq in range(4):
                    out[j, 32 * b + 8 * rq + 4 * h:32 * b + 8 * rq + 4 * h + 4] = ot[b, 32 * h + j, 4 * rq:4 * rq + 4] / l_tot
    assert np.abs(out - reference(q, k, v)).max() < 1e-9


def test_the_model_notices_a_wrong_fragment_formula():
    """The V^T fragment's second transposing read 16 key rows further (the 16x16x32 kernel's distance) instead of 8 is a wrong P^T / V^T
    pairing on 32x32x16: the model must say so."""
    q, k, v = inputs(3)
    vimg = image(v, swz_v)
    lane = np.arange(64)
    hi, i16, dh = lane >> 5, lane & 15, (lane >> 4) & 1
    v_row = 4 * hi + (i16 >> 2)
    vbase = v_row * ROW + ((((dh << 1) | ((i16 & 3) >> 1)) ^ swz_v(v_row)) << 4) + ((i16 & 1) << 3)
    good = np.concatenate([ds_read_b64_tr_b16(vimg, vbase), ds_read_b64_tr_b16(vimg, vbase + 8 * ROW)], axis=1)
    bad = np.concatenate([ds_read_b64_tr_b16(vimg, vbase), ds_read_b64_tr_b16(vimg, vbase + 16 * ROW)], axis=1)
    # lane (hi, l31) must hold V[key 8 (e >> 2) + 4 hi + (e & 3)][d = l31]
    want = np.array([[v[8 * (e >> 2) + 4 * hi[l] + (e & 3), lane[l] & 31] for e in range(8)] for l in range(64)])
    assert np.array_equal(good, want)
    assert not np.array_equal(bad, want)


def test_m16x_transposed_v_fragment_is_the_row_major_fragment():
    """The three *_swizzle_qkv names take V as [B, H, d, N]: the V^T tile image is d rows of BC keys (256-byte rows, chunk swizzle
    row & 15) and a fragment is two PLAIN 8-byte reads two chunks apart -- it must equal the transposing-read fragment of the
    row-major image, k-step by k-step (flash_attn_m16x.cuh, VT = true)."""
    _, _, v = inputs(4)
    vimg = image(v, swz_v)
    RV = BC * 2
    vt = v.T  # [d][key]
    vtimg = np.zeros(D * BC)
    for row in range(D):
        for c in range(BC // 8):
            vtimg[row * BC + 8 * c:row * BC + 8 * c + 8] = vt[row, 8 * (c ^ (row & 15)):8 * (c ^ (row & 15)) + 8]
    lane = np.arange(64)
    i16, g4 = lane & 15, lane >> 4
    v_row = 4 * g4 + (i16 >> 2)
    vbase = v_row * ROW + ((((i16 & 3) >> 1) ^ swz_v(v_row)) << 4) + ((i16 & 1) << 3)
    vtbase = i16 * RV + (((i16 & 15) ^ (g4 >> 1)) << 4) + ((g4 & 1) << 3)
    for u in range(BC // 32):
        for db in range(D // 16):
            addr = (vbase ^ (db << 5)) + 32 * u * ROW
            row_major = np.concatenate([ds_read_b64_tr_b16(vimg, addr), ds_read_b64_tr_b16(vimg, addr + 16 * ROW)], axis=1)
            a0 = (vtbase ^ ((4 * u) << 4)) + 16 * db * RV
            a1 = (vtbase ^ ((4 * u + 2) << 4)) + 16 * db * RV
            transposed = np.array([np.concatenate([lds_halves(vtimg, a0[l], 4), lds_halves(vtimg, a1[l], 4)]) for l in range(64)])
            assert np.array_equal(row_major, transposed), (u, db)
