"""Print the XOR-swizzled LDS image of an HGEMM / attention operand tile as the gfx950 kernels lay it out: which LOGICAL 16-byte chunk
sits at every (row, chunk position), and the 64-bank footprint of one fragment read -- the teaching counterpart of the reference's
kernels/hgemm/tools/print_swizzle_layout.py:1-219 (which prints the NVIDIA 32-bank, `col ^ (row / 4)` shared-memory layout of its
`*_swizzle` kernels). The formulas are the ones of the kernels (csrc/hgemm_mfma.cuh kswz / nswz / nswz_bn, csrc/flash_attn_dring.cuh and
flash_attn_dw4.cuh K / V swizzles); tests/test_print_lds_swizzle.py holds this file to the enumeration model of tests/test_lds_layout_model.py.

  python print_lds_swizzle.py --image k64            # A / B^T tile rows of 64 halves (128-byte rows): chunk ^= (row >> 1) & 7
  python print_lds_swizzle.py --image k32 --rows 32  # 32-deep K slots (hgemm_w4s): table {0,2,3,1}[(row >> 2) & 3]
  python print_lds_swizzle.py --image n256           # B tile of the NN layout, 256 columns: chunk ^= (k & 3) << 1 | (k >> 3 & 1) << 3
  python print_lds_swizzle.py --image attn-k --row-bytes 2048   # attention K tile (d = 1024): chunk ^= row & 15
  python print_lds_swizzle.py --image attn-v --row-bytes 2048   # attention V tile: chunk ^= (row & 3) << 2
Why a swizzle at all: LDS images are filled by LDS-DMA (`global_load_lds_dwordx4`), whose destination is lane-linear -- no padding is
possible -- so the bank-conflict fix is applied to the SOURCE address of each lane and, as the same involution, to the fragment read."""
import argparse

BANKS = 64  # 4-byte banks of the gfx950 LDS for ds_read_b64 / b128 / ds_read_b64_tr_b16 (MI355X_MICROARCH.md, LDS table)


def kswz(row, bk):
    """K-contiguous image ([rows][bk] halves): csrc/hgemm_mfma.cuh kswz<BK>."""
    if bk == 64:
        return (row >> 1) & 7
    t = (row >> 2) & 3
    return (((t ^ (t >> 1)) & 1) << 1) | (t >> 1)


def nswz_bn(krow, bn):
    """N-contiguous image of B for the NN layout ([BK][bn] halves): csrc/hgemm_mfma.cuh nswz_bn<BN>."""
    if bn in (192, 64):
        return (((krow >> 1) & 1) | (((krow >> 3) & 1) << 1)) << 1
    if bn == 160:
        return ((krow >> 3) & 1) << 1
    return ((krow & 3) << 1) | (((krow >> 3) & 1) << 3)


def attn_k(row):
    return row & 15


def attn_v(row):
    return (row & 3) << 2


def image(kind, rows, row_bytes):
    """(swizzle(row), chunks per row) of an image kind."""
    if kind == "k64":
        return (lambda r: kswz(r, 64)), 8
    if kind == "k32":
        return (lambda r: kswz(r, 32)), 4
    if kind.startswith("n"):
        bn = int(kind[1:])
        return (lambda r: nswz_bn(r, bn)), bn * 2 // 16
    if kind == "attn-k":
        return attn_k, row_bytes // 16
    if kind == "attn-v":
        return attn_v, row_bytes // 16
    raise ValueError("unknown image kind %r" % kind)


def layout(kind, rows, row_bytes=2048):
    """rows x chunks table: entry [r][pos] = logical chunk stored at chunk position pos of row r."""
    sw, cpr = image(kind, rows, row_bytes)
    return [[pos ^ sw(r) if (pos ^ sw(r)) < cpr else None for pos in range(cpr)] for r in range(rows)], cpr


def banks_of_read(kind, rows_read, logical_chunk, row_bytes=2048, nbytes=16):
    """bank set touched when each row in rows_read reads `nbytes` of logical chunk `logical_chunk` (one lane group of a fragment read)."""
    sw, cpr = image(kind, max(rows_read) + 1, row_bytes)
    rb = cpr * 16
    banks = []
    for r in rows_read:
        addr = r * rb + ((logical_chunk ^ sw(r)) << 4)
        banks += [((addr + 4 * k) // 4) % BANKS for k in range(nbytes // 4)]
    return banks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--image", default="k64", help="k64 | k32 | n64 | n128 | n160 | n192 | n256 | attn-k | attn-v")
    ap.add_argument("--rows", type=int, default=16)
    ap.add_argument("--row-bytes", type=int, default=2048, help="attention images: bytes per K / V row (2 * head dim)")
    ap.add_argument("--max-chunks", type=int, default=16, help="columns printed (a row of 2048 bytes has 128 chunks)")
    ap.add_argument("--no-swizzle", action="store_true", help="print the un-swizzled image beside it for comparison")
    args = ap.parse_args()
    tab, cpr = layout(args.image, args.rows, args.row_bytes)
    shown = min(cpr, args.max_chunks)
    print("LDS image %r: %d rows x %d chunks of 16 bytes (%d-byte rows), 64 banks of 4 bytes; entry = LOGICAL chunk at that position"
          % (args.image, args.rows, cpr, cpr * 16))
    print("row | bank of chunk 0 | " + " ".join("p%-3d" % p for p in range(shown)))
    for r, row in enumerate(tab):
        print("%3d | %15d | %s" % (r, (r * cpr * 4) % BANKS, " ".join("%-4s" % ("-" if c is None else c) for c in row[:shown])))
    # one ds_read_b128 lane group = 16 lanes, each 16 bytes: conflict-free iff the 16 x 4 banks are all distinct
    groups = [list(range(g, g + 16)) for g in range(0, min(args.rows, 64), 16)] if args.image.startswith(("k", "attn-k")) else []
    for g in groups:
        if max(g) >= args.rows:
            continue
        for chunk in (0, 1):
            b_sw = banks_of_read(args.image, g, chunk, args.row_bytes)
            sw0 = len(set(b_sw))
            plain = len({((r * cpr * 16 + (chunk << 4) + 4 * k) // 4) % BANKS for r in g for k in range(4)})
            print("fragment read of logical chunk %d by rows %d..%d (16 lanes x 16 B): %d distinct banks swizzled, %d without the swizzle%s"
                  % (chunk, g[0], g[-1], sw0, plain, "  <- conflict-free" if sw0 == 64 else ""))
    if args.no_swizzle:
        print("un-swizzled: entry [r][p] = p; every row starts at bank (r * %d) %% 64" % (cpr * 4))


if __name__ == "__main__":
    main()
