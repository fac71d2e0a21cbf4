"""CPU model of the LDS images of the 16x16x32 attention kernel (csrc/flash_attn_m16.cuh GeoM16): the claims its comments
make about the layout, checked by enumeration -- no GPU. The formulas below restate GeoM16::swz_k / swz_v, the LDS-DMA
piece mapping, the K / V^T fragment addresses and the P^T key order of that file; the GPU parity tests check the kernel
itself, this file checks WHY the layout is the one it is (conflict-free, no per-piece swizzle term, matching key order)."""
import itertools

import pytest

GEOMS = [(64, 128), (128, 128), (128, 64), (64, 64)]  # (D, BC) forms that are instantiated


def swz_k(row, rowbytes):
    return (row >> 1) & 7 if rowbytes == 128 else row & 15


def swz_v(row, rowbytes):
    return ((row >> 1) & 3) << 1 if rowbytes == 128 else (row & 7) << 1


@pytest.mark.parametrize("D,BC", GEOMS)
def test_dma_source_swizzle_has_no_per_piece_term(D, BC):
    """A wave applies the swizzle of row (widx * RPP + lr) to every piece i it copies; that must equal the swizzle of the
    real row (i*4 + widx) * RPP + lr, for both operands."""
    rowbytes = D * 2
    rpp, cpr = 1024 // rowbytes, rowbytes // 16
    ppw = BC * rowbytes // 1024 // 4
    for i, widx, lane in itertools.product(range(ppw), range(4), range(64)):
        lr = lane // cpr
        row = (i * 4 + widx) * rpp + lr
        assert row < BC
        assert swz_k(row, rowbytes) == swz_k(widx * rpp + lr, rowbytes)
        assert swz_v(row, rowbytes) == swz_v(widx * rpp + lr, rowbytes)


def lds_image(D, BC, operand):
    """(row, position chunk) -> logical chunk the LDS-DMA put there (lane-linear destination, swizzled source)."""
    rowbytes = D * 2
    rpp, cpr = 1024 // rowbytes, rowbytes // 16
    ppw = BC * rowbytes // 1024 // 4
    sw = swz_k if operand == "K" else swz_v
    img = {}
    for i, widx, lane in itertools.product(range(ppw), range(4), range(64)):
        lr, lc = lane // cpr, lane % cpr
        piece = i * 4 + widx
        dst = piece * 1024 + lane * 16
        row, pos = dst // rowbytes, (dst % rowbytes) // 16
        assert row == piece * rpp + lr and pos == lc
        img[(row, pos)] = lc ^ sw(widx * rpp + lr, rowbytes)
    assert len(img) == BC * cpr
    return img


def banks_of(addr, nbytes):
    return [((addr + 4 * k) // 4) % 64 for k in range(nbytes // 4)]


@pytest.mark.parametrize("D,BC", GEOMS)
def test_k_fragment_reads_hit_the_right_chunk_without_bank_conflicts(D, BC):
    """K fragment (kb, ks): lane (i16, g4) reads 16 bytes = d 32*ks + 8*g4 .. +7 of key 16*kb + i16 at
    (kbase ^ (ks << 6)) + kb*16*ROW. The 16 lanes of one group (one b128 pass) touch 64 distinct banks."""
    rowbytes = D * 2
    img = lds_image(D, BC, "K")
    for kb, ks in itertools.product(range(BC // 16), range(D // 32)):
        for g4 in range(4):
            banks = []
            for i16 in range(16):
                kbase = i16 * rowbytes + ((g4 ^ swz_k(i16, rowbytes)) << 4)
                addr = (kbase ^ (ks << 6)) + kb * 16 * rowbytes
                row, pos = addr // rowbytes, (addr % rowbytes) // 16
                assert row == 16 * kb + i16
                assert img[(row, pos)] == 4 * ks + g4  # the logical chunk holding d = 32*ks + 8*g4
                banks += banks_of(addr, 16)
            assert len(set(banks)) == 64


@pytest.mark.parametrize("D,BC", GEOMS)
def test_v_fragment_reads_transposing_layout_key_order_and_banks(D, BC):
    """V^T fragment (u, db): two transposing reads; in each, lane i of a 16-lane group supplies 8 bytes of row
    base + (i >> 2), columns c0 + 4*(i & 3) .. +3, and receives column c0 + i of the 4 rows. The rows must be the keys
    the lane's P^T registers hold for k-slots 8*g4 + j: 32u + 4*g4 + j (j < 4), 32u + 16 + 4*g4 + (j - 4); the 64 lanes of
    one read (512 bytes) must load every bank exactly twice."""
    rowbytes = D * 2
    img = lds_image(D, BC, "V")
    for u, db in itertools.product(range(BC // 32), range(D // 16)):
        for half in range(2):
            bank_load = [0] * 64
            for lane in range(64):
                i16, g4 = lane & 15, lane >> 4
                v_row = 4 * g4 + (i16 >> 2)
                vbase = v_row * rowbytes + ((((i16 & 3) >> 1) ^ swz_v(v_row, rowbytes)) << 4) + ((i16 & 1) << 3)
                addr = (vbase ^ (db << 5)) + 32 * u * rowbytes + half * 16 * rowbytes
                row, pos, within = addr // rowbytes, (addr % rowbytes) // 16, addr % 16
                # rows of this read = the keys of k-slots 8*g4 + 4*half .. +3 of the P^T operand (S^T blocks 2u, 2u+1)
                assert row == 32 * u + 16 * half + 4 * g4 + (i16 >> 2)
                # columns: lane i supplies d = 16*db + 4*(i & 3) .. +3, so the group covers d = 16*db .. +15 and lane i
                # receives d = 16*db + i16 -- the A-operand row (m = i16) of block db
                logical = img[(row, pos)]
                d0 = logical * 8 + within // 2
                assert d0 == 16 * db + 4 * (i16 & 3)
                for b in banks_of(addr, 8):
                    bank_load[b] += 1
            assert max(bank_load) == 2 and min(bank_load) == 2


def test_p_operand_key_order_matches_the_score_registers():
    """S^T block kb: lane (query i16, g4) holds keys 16*kb + 4*g4 + r in register r. The P^T operand of k-step u takes
    the registers of blocks 2u and 2u+1 in order (pf[e] = s[2u + (e >> 2)][e & 3]): k-slot 8*g4 + e is the key below --
    the same keys test_v_fragment_reads_* requires of the two transposing reads."""
    for u, g4, e in itertools.product(range(4), range(4), range(8)):
        kb, r = 2 * u + (e >> 2), e & 3
        key = 16 * kb + 4 * g4 + r
        assert key == 32 * u + (4 * g4 + e if e < 4 else 16 + 4 * g4 + (e - 4))


# ---- the d-split pair form at D = 512 (GeoM16Pair): 1024-byte rows, a DMA piece is ONE row, swizzles with a per-piece term
def pair_image(operand, rowbytes=1024):
    """rowbytes 1024: the D = 512 pair form; 512: the same kernel at D = 256 (PAIR = false, a piece is two rows)."""
    sw = (lambda row: row & 15) if operand == "K" else (lambda row: (row & 15) << 1)
    rpp, cpr = 1024 // rowbytes, rowbytes // 16
    ppw = 32 * rowbytes // 1024 // 4
    img = {}
    for i, widx, lane in itertools.product(range(ppw), range(4), range(64)):
        lr, lc = lane // cpr, lane % cpr
        row = (4 * i + widx) * rpp + lr
        src_chunk = lc ^ sw(widx * rpp + lr) ^ sw(4 * i * rpp)  # what dma_piece computes: src_lane ^ swizzle(4*i*RPP)
        assert src_chunk == lc ^ sw(row) and 0 <= src_chunk < cpr
        dst = (4 * i + widx) * 1024 + lane * 16
        assert dst // rowbytes == row and (dst % rowbytes) // 16 == lc
        img[(row, lc)] = src_chunk
    assert len(img) == 32 * cpr
    return img


def test_pair_form_k_reads():
    img = pair_image("K")
    for part, kb, ks, g4 in itertools.product(range(2), range(2), range(8), range(4)):
        banks = []
        for i16 in range(16):
            kbase = i16 * 1024 + ((g4 ^ (i16 & 15)) << 4) + part * 512
            addr = (kbase ^ (ks << 6)) + kb * 16 * 1024
            row, pos = addr // 1024, (addr % 1024) // 16
            assert row == 16 * kb + i16
            assert img[(row, pos)] == part * 32 + 4 * ks + g4  # d = 256*part + 32*ks + 8*g4
            banks += banks_of(addr, 16)
        assert len(set(banks)) == 64


def test_pair_form_v_reads():
    img = pair_image("V")
    for part, db, half in itertools.product(range(2), range(16), range(2)):
        bank_load = [0] * 64
        for lane in range(64):
            i16, g4 = lane & 15, lane >> 4
            v_row = 4 * g4 + (i16 >> 2)
            vbase = v_row * 1024 + ((((i16 & 3) >> 1) ^ ((v_row & 15) << 1)) << 4) + ((i16 & 1) << 3) + part * 512
            addr = (vbase ^ (db << 5)) + half * 16 * 1024
            row, pos, within = addr // 1024, (addr % 1024) // 16, addr % 16
            assert row == 16 * half + 4 * g4 + (i16 >> 2)
            d0 = img[(row, pos)] * 8 + within // 2
            assert d0 == 256 * part + 16 * db + 4 * (i16 & 3)
            for b in banks_of(addr, 8):
                bank_load[b] += 1
        assert max(bank_load) == 2 and min(bank_load) == 2


def test_d256_form_reads():
    """PAIR = false (D = 256): 512-byte rows, whole d in one wave (no part offset)."""
    imgk, imgv = pair_image("K", 512), pair_image("V", 512)
    for kb, ks, g4 in itertools.product(range(2), range(8), range(4)):
        banks = []
        for i16 in range(16):
            addr = ((i16 * 512 + ((g4 ^ (i16 & 15)) << 4)) ^ (ks << 6)) + kb * 16 * 512
            row, pos = addr // 512, (addr % 512) // 16
            assert row == 16 * kb + i16 and imgk[(row, pos)] == 4 * ks + g4
            banks += banks_of(addr, 16)
        assert len(set(banks)) == 64
    for db, half in itertools.product(range(16), range(2)):
        bank_load = [0] * 64
        for lane in range(64):
            i16, g4 = lane & 15, lane >> 4
            v_row = 4 * g4 + (i16 >> 2)
            vbase = v_row * 512 + ((((i16 & 3) >> 1) ^ ((v_row & 15) << 1)) << 4) + ((i16 & 1) << 3)
            addr = (vbase ^ (db << 5)) + half * 16 * 512
            row, pos, within = addr // 512, (addr % 512) // 16, addr % 16
            assert row == 16 * half + 4 * g4 + (i16 >> 2)
            assert imgv[(row, pos)] * 8 + within // 2 == 16 * db + 4 * (i16 & 3)
            for b in banks_of(addr, 8):
                bank_load[b] += 1
        assert max(bank_load) == 2 and min(bank_load) == 2
