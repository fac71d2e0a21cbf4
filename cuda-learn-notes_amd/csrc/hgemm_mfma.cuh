// MFMA HGEMM kernels for gfx950: C[M,N] = A[M,K] * B, fp16 in / fp32 accumulate / fp16 out.
//
// Replaces the tensor-core ladders of the reference:
//   kernels/hgemm/mma/basic/hgemm_mma.cu:44,:135          (naive, 1-stage 128x128 tile)
//   kernels/hgemm/mma/basic/hgemm_mma_stage.cu:68,:293,:603,:1044,:1460 (cp.async multi-stage rings)
//   kernels/hgemm/mma/basic/hgemm_mma_stage_tn.cu:70, mma/swizzle/*.cu:121,:148 (TN, smem swizzle)
//   kernels/hgemm/wmma/hgemm_wmma.cu:39-415, wmma/hgemm_wmma_stage.cu:56-721 (WMMA rungs)
//   kernels/hgemm/cutlass/hgemm_mma_stage_tn_cute.cu:26   (CuTe 128x256 TN)
//
// MI355X design (not a translation):
//  * one MFMA shape, v_mfma_f32_16x16x32_f16 (gfx950 2xK form), fp32 accumulators (MFMA has no
//    fp16 accumulator; the reference accumulates in fp16 -- hgemm_mma.cu:37);
//  * operands swapped in the MFMA call (a := B fragment, b := A fragment) so every lane ends up
//    with 4 CONSECUTIVE n of one C row -> 8-byte packed stores without any cross-lane shuffle
//    (the reference needs a 4-lane __shfl gather for that, hgemm_mma_stage.cu:991-1022);
//  * LDS images are lane-linear so they can be filled by LDS-DMA (global_load_lds_dwordx4);
//    the bank-conflict XOR swizzle is applied to the per-lane GLOBAL source address and, with
//    the same involution, to the fragment read address (64-bank / 16-lane-group model of
//    ds_read_b128 and ds_read_b64_tr_b16, MI355X_MICROARCH.md LDS table) -- no padding;
//  * NN layout reads the B fragment with the hardware transpose read ds_read_b64_tr_b16 from a
//    [k][n] image (no ldmatrix.trans on CDNA); TN reads it exactly like A;
//  * multi-stage ring = `stages` LDS buffers, prefetch distance stages-1, ONE raw s_barrier per
//    K tile, counted s_waitcnt vmcnt(N) so DMA stays in flight across the barrier;
//  * block swizzle = XCD-aware remap (8 XCDs, private L2s) + N-band walk of `swizzle_stride`.
#pragma once
#include "common.h"

namespace hgemm {

enum Layout { NN = 0, TN = 1 };

// ---- XOR swizzles (16-byte chunk index within a row of the LDS image) --------------------------
// K-contiguous image ([rows][BK] halves). BK=64: 128-B rows, 8 chunks; BK=32: 64-B rows, 4 chunks.
template <int BK>
__device__ __forceinline__ int kswz(int row) {
  if constexpr (BK == 64) {
    return (row >> 1) & 7;
  } else {
    // BK == 32: table {0,2,3,1}[(row>>2)&3]  (derived for the 4x16-lane ds_read_b128 groups)
    const int t = (row >> 2) & 3;
    return (((t ^ (t >> 1)) & 1) << 1) | (t >> 1);
  }
}
// N-contiguous image of B for the NN layout ([BK][BN] halves, BN in {128,256}).
__device__ __forceinline__ int nswz(int krow) { return ((krow & 3) << 1) | (((krow >> 3) & 1) << 3); }
// Rows that are an ODD multiple of 128 bytes (BN = 64, 192) put consecutive k rows 32 banks apart, so the four even
// (odd) rows a 32-lane group of ds_read_b64_tr_b16 touches share a bank half; they are spread over its four 8-bank
// quarters by XOR-ing the chunk index with 2 * (bit 1 of k | bit 3 of k << 1) -- bits 1..2 only, so a chunk never
// leaves its 128-byte group (8 or 24 chunks per row).
template <int BN>
__device__ __forceinline__ int nswz_bn(int krow) {
  if constexpr (BN == 192 || BN == 64) return ((((krow >> 1) & 1) | (((krow >> 3) & 1) << 1)) << 1);
  // 320-byte rows (BN = 160): consecutive k rows are 16 banks apart, so rows r and r + 8 of a 32-lane group land on the same
  // 16-bank slot (each uses 8 of them): move the second to the other half of the slot
  else if constexpr (BN == 160) return ((krow >> 3) & 1) << 1;
  else return nswz(krow);
}

// Counted wait: leaves N vector-memory ops (LDS-DMA pieces) in flight. vmcnt is a 6-bit field.
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt range");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ---- block index -> (tile_m, tile_n) -----------------------------------------------------------
// swizzle==0: plain row-major walk (the reference's un-swizzled grid).
// swizzle!=0: (1) bijective XCD remap -- hardware places block b on XCD b%8, so give each XCD a
// contiguous run of logical blocks (cdna guide T1); (2) walk N in bands of `band` tiles, M-major
// inside a band (the reference's blockIdx.z N-band, hgemm_mma_stage.cu:608, re-expressed).
__device__ __forceinline__ void tile_coords(int bid, int nblk, int tiles_m, int tiles_n, int swizzle, int band,
                                            int& tm, int& tn) {
  if (!swizzle) {
    tm = bid / tiles_n;
    tn = bid - tm * tiles_n;
    return;
  }
  const int xcd = bid & 7, local = bid >> 3;
  const int q = nblk >> 3, r = nblk & 7;
  const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  if (band <= 0 || band > tiles_n) band = tiles_n;
  const int per_band = tiles_m * band;
  const int b = wg / per_band;
  const int rem = wg - b * per_band;
  const int width = min(band, tiles_n - b * band);
  tm = rem / width;
  tn = b * band + (rem - tm * width);
}

// Walk for problems whose operands do NOT fit the 256 MiB Infinity Cache (round 4; 12544^3 ... 16384^3, the sizes the reference's README
// quotes). With the walk above every XCD streams its OWN contiguous range of the logical order, so the eight XCDs sit in eight distant
// N-bands at once and B (512 MB at 16384^3) is re-read from HBM in every round of tiles. Here the XCDs take the logical order in
// INTERLEAVED chunks of 32 tiles (round r = logical tiles 256 r ... 256 r + 255, XCD x its tiles 32 x ... 32 x + 31): the 256 tiles in
// flight are 256 CONSECUTIVE tiles of the band walk -- with the reference's band of 8 tile columns, 32 rows x 8 columns: the band's 8 B
// panels stay in the cache for the whole band, every A panel is read once per band, and an XCD's chunk is a 4 x 8 sub-block (4 + 8 panels
// per L2, the same fill as before). No padding: ragged tile counts only shorten the last round (its tiles go to the XCDs in launch
// order). (A first form that walked padded 16 x 16 blocks lost 8-13 % on ragged grids -- 12544 = 49 tiles, 10240 = 40 --
// and gained 2-3 % on exact ones: profiles/r04_hgemm_block_walk_probe.log.) Pure schedule change: results are bit-identical.
__device__ __forceinline__ void tile_coords_interleaved(int bid, int nblk, int tiles_m, int tiles_n, int band, int& tm, int& tn) {
  const int full = nblk & ~255;  // tiles in whole rounds of 256
  int wg = bid;
  if (bid < full) {
    const int xcd = bid & 7, local = bid >> 3;
    wg = ((local >> 5) << 8) + (xcd << 5) + (local & 31);
  }
  if (band <= 0 || band > tiles_n) band = tiles_n;
  const int per_band = tiles_m * band;
  const int b = wg / per_band;
  const int rem = wg - b * per_band;
  const int width = min(band, tiles_n - b * band);
  tm = rem / width;
  tn = b * band + (rem - tm * width);
}

// ---- tile configuration ------------------------------------------------------------------------
template <int BM_, int BN_, int BK_, int WM_, int WN_, int STAGES_, int LAYOUT_>
struct Cfg {
  static constexpr int BM = BM_, BN = BN_, BK = BK_, WM = WM_, WN = WN_, STAGES = STAGES_, LAYOUT = LAYOUT_;
  static constexpr int NW = WM * WN, NT = NW * 64;
  static constexpr int WTM = BM / WM, WTN = BN / WN;  // per-wave output tile
  static constexpr int FM = WTM / 16, FN = WTN / 16;  // 16x16 fragments per wave
  static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int A_LOADS = A_BYTES / 1024 / NW, B_LOADS = B_BYTES / 1024 / NW;  // glds per wave per tile
  static constexpr int LOADS = A_LOADS + B_LOADS;
  static constexpr int LDS_BYTES = STAGES * STAGE_BYTES;
  static_assert(BK == 32 || BK == 64, "BK");
  static_assert(A_BYTES % (1024 * NW) == 0 && B_BYTES % (1024 * NW) == 0, "tile must split evenly over waves");
  static_assert(LAYOUT == TN || BN == 64 || BN == 128 || BN == 256, "NN image swizzle derived for BN in {64,128,256}");
  static_assert(WTM % 16 == 0 && WTN % 16 == 0, "wave tile");
};

// LDS-DMA issue hidden from hipcc's waitcnt bookkeeping (cdna guide 5.7): hipcc cannot prove that
// an in-flight DMA does not alias the fragment reads of the OTHER stage buffer and would drain
// vmcnt(0) before the first ds_read of every K tile. Issued from asm, the DMA is ordered only by
// our counted s_waitcnt vmcnt(N) + s_barrier. Source = SGPR base + per-lane 32-bit byte offset,
// LDS destination = M0 (wave-uniform) + lane*16. M0 is saved/restored inside the statement.
__device__ __forceinline__ void glds16_asm(const void* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory");
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
  return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) char*)p;
}

// Per-lane byte offsets for the DMA fill of one operand image (constant over the K loop; the
// K advance is a scalar add on the base pointer).
// K-contiguous image: element (row, kchunk) of the tile; `ld` = leading dimension in halves.
template <typename C, int NLOADS>
struct KFill {
  unsigned voff[NLOADS];
  __device__ __forceinline__ void init(int ld, int wave, int lane) {
    constexpr int LPR = C::BK / 8;  // lanes (16-B chunks) per row
    constexpr int RPI = 64 / LPR;   // rows per wave-instruction
#pragma unroll
    for (int i = 0; i < NLOADS; ++i) {
      const int t = i * C::NW + wave;  // wave-instruction index within the image
      const int row = t * RPI + lane / LPR;
      const int c = lane % LPR;
      voff[i] = ((unsigned)row * (unsigned)ld + ((c ^ kswz<C::BK>(row)) << 3)) * 2u;
    }
  }
};
// N-contiguous B image (NN): element (krow, nchunk).
template <typename C, int NLOADS>
struct NFill {
  unsigned voff[NLOADS];
  __device__ __forceinline__ void init(int N, int wave, int lane) {
    constexpr int LPR = C::BN / 8;
    constexpr int RPI = 64 / LPR;
#pragma unroll
    for (int i = 0; i < NLOADS; ++i) {
      const int t = i * C::NW + wave;
      const int krow = t * RPI + lane / LPR;
      const int c = lane % LPR;
      voff[i] = ((unsigned)krow * (unsigned)N + ((c ^ nswz_bn<C::BN>(krow)) << 3)) * 2u;
    }
  }
};
// Issue one operand image: NLOADS pieces of 1 KiB per wave.
template <int NW, int NLOADS>
__device__ __forceinline__ void issue_image(const char* sbase, const unsigned (&voff)[NLOADS], unsigned lds_img,
                                            int wave) {
#pragma unroll
  for (int i = 0; i < NLOADS; ++i) glds16_asm(sbase, voff[i], lds_img + (unsigned)(i * NW + wave) * 1024u);
}

// Same, but only the pieces whose index bit is set in MASK (compile-time).
template <int NW, int NLOADS, unsigned MASK>
__device__ __forceinline__ void issue_image_masked(const char* sbase, const unsigned (&voff)[NLOADS], unsigned lds_img,
                                                   int wave) {
#pragma unroll
  for (int i = 0; i < NLOADS; ++i)
    if ((MASK >> i) & 1u) glds16_asm(sbase, voff[i], lds_img + (unsigned)(i * NW + wave) * 1024u);
}

// ---- fragment readers --------------------------------------------------------------------------
// K-contiguous image, rows [row0 + 16*f + (lane&15)], k chunk (lane>>4) of k-step kk.
template <int BK>
__device__ __forceinline__ h8 read_kfrag(const char* img, int row, int lane, int kk) {
  const int q = kk * 4 + (lane >> 4);
  return *reinterpret_cast<const h8*>(img + row * (BK * 2) + ((q ^ kswz<BK>(row)) << 4));
}
// NN B image: 16 columns n0w..n0w+15 (tile-relative), k = kk*32 + 8*(lane>>4) .. +7, via two
// transposing reads.
template <int BN>
__device__ __forceinline__ h8 read_nfrag(const char* img, int n0w, int lane, int kk) {
  const int i = lane & 15, g = lane >> 4;
  const int q = (n0w >> 3) + ((i & 3) >> 1);
  const int k_lo = kk * 32 + 8 * g + (i >> 2);
  const int k_hi = k_lo + 4;
  const char* p_lo = img + k_lo * (BN * 2) + ((q ^ nswz_bn<BN>(k_lo)) << 4) + ((i & 1) << 3);
  const char* p_hi = img + k_hi * (BN * 2) + ((q ^ nswz_bn<BN>(k_hi)) << 4) + ((i & 1) << 3);
  return h8_cat(lds_read_tr16(p_lo), lds_read_tr16(p_hi));
}

__device__ __forceinline__ void store_c4(half_t* C, int N, int m, int n, const f4& v) {
  h4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
  *reinterpret_cast<h4*>(C + (size_t)m * N + n) = o;
}

// ---- one K tile of MFMA work for a wave -------------------------------------------------------
template <typename C>
__device__ __forceinline__ void compute_tile(const char* a_img, const char* b_img, int wm, int wn, int lane,
                                             f4 (&acc)[C::FM][C::FN]) {
#pragma unroll
  for (int kk = 0; kk < C::BK / 32; ++kk) {
    h8 af[C::FM], bf[C::FN];
#pragma unroll
    for (int j = 0; j < C::FN; ++j) {
      if constexpr (C::LAYOUT == TN) {
        bf[j] = read_kfrag<C::BK>(b_img, wn * C::WTN + j * 16 + (lane & 15), lane, kk);
      } else {
        bf[j] = read_nfrag<C::BN>(b_img, wn * C::WTN + j * 16, lane, kk);
      }
    }
#pragma unroll
    for (int i = 0; i < C::FM; ++i) af[i] = read_kfrag<C::BK>(a_img, wm * C::WTM + i * 16 + (lane & 15), lane, kk);
#pragma unroll
    for (int i = 0; i < C::FM; ++i)
#pragma unroll
      for (int j = 0; j < C::FN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
  }
}

template <typename C>
__device__ __forceinline__ void store_tile(half_t* Cmat, int N, int m0, int n0, int wm, int wn, int lane,
                                           const f4 (&acc)[C::FM][C::FN]) {
  // swapped-operand MFMA => lane holds C[m = .. + (lane&15)][n = .. + 4*(lane>>4) + r], r = 0..3
#pragma unroll
  for (int i = 0; i < C::FM; ++i)
#pragma unroll
    for (int j = 0; j < C::FN; ++j)
      store_c4(Cmat, N, m0 + wm * C::WTM + i * 16 + (lane & 15), n0 + wn * C::WTN + j * 16 + 4 * (lane >> 4),
               acc[i][j]);
}

// ---- epilogue through LDS: full-line 16-byte stores ---------------------------------------------
// After the K loop the operand ring is dead, so each wave parks its 128x64 fp16 result in a PRIVATE
// LDS region ([64 rows][144 B], two passes of 64 rows) and streams it out as 16 bytes per lane:
// one store instruction = 8 rows x 128 contiguous bytes instead of 16 rows x 4 x 8-byte pieces
// (store tail is issue- and partial-line-bound, cdna guide T21). No barrier: the region is
// wave-private and LDS ops of one wave complete in order.
template <int FM, int FN>
__device__ __forceinline__ void store_tile_via_lds(half_t* Cmat, int N, int row0, int col0, int lane, char* wave_lds,
                                                   const f4 (&acc)[FM][FN]) {
  static_assert(FN == 4, "wave tile N must be 64");
  constexpr int RS = 144;  // row stride: 128 B of data + 16 B pad (keeps 16-B alignment, 2-way write conflict)
#pragma unroll
  for (int h0 = 0; h0 < FM; h0 += 4) {  // passes of up to 64 rows (the last one may be shorter: FM = 6 -> 64 + 32)
    constexpr int NF_FULL = 4;
    const int nf = FM - h0 < NF_FULL ? FM - h0 : NF_FULL;
#pragma unroll
    for (int i = 0; i < NF_FULL; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        if (i < nf) {
          const f4 v = acc[(h0 + i) < FM ? (h0 + i) : 0][j];
          h4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
          *reinterpret_cast<h4*>(wave_lds + (i * 16 + (lane & 15)) * RS + (j * 16 + 4 * (lane >> 4)) * 2) = o;
        }
      }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      if (it < nf * 2) {
        const int r = it * 8 + (lane >> 3);
        const u4 v = *reinterpret_cast<const u4*>(wave_lds + r * RS + (lane & 7) * 16);
        *reinterpret_cast<u4*>(Cmat + (size_t)(row0 + h0 * 16 + r) * N + col0 + (lane & 7) * 8) = v;
      }
    }
  }
}

// ---- multi-stage LDS-DMA ring kernel -----------------------------------------------------------
template <typename C>
__global__ __launch_bounds__(C::NT, (C::NT >= 512 ? 2 : 1)) void hgemm_ring_kernel(
    const half_t* __restrict__ A, const half_t* __restrict__ B, half_t* __restrict__ Cmat, int M, int N, int K,
    int tiles_m, int tiles_n, int swizzle, int band) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / C::WN, wn = wave % C::WN;
  int tm, tn;
  tile_coords(blockIdx.x, gridDim.x, tiles_m, tiles_n, swizzle, band, tm, tn);
  const int m0 = tm * C::BM, n0 = tn * C::BN;

  KFill<C, C::A_LOADS> fa;
  fa.init(K, wave, lane);
  KFill<C, C::B_LOADS> fbt;  // TN
  NFill<C, C::B_LOADS> fbn;  // NN
  if constexpr (C::LAYOUT == TN) fbt.init(K, wave, lane);
  else fbn.init(N, wave, lane);
  // wave-uniform tile origins (bytes); advanced by one K tile per stage() call
  const char* a_src = reinterpret_cast<const char*>(A + (size_t)m0 * K);
  const char* b_src = (C::LAYOUT == TN) ? reinterpret_cast<const char*>(B + (size_t)n0 * K)
                                        : reinterpret_cast<const char*>(B + n0);
  const size_t a_step = (size_t)C::BK * 2;
  const size_t b_step = (C::LAYOUT == TN) ? (size_t)C::BK * 2 : (size_t)C::BK * N * 2;
  const unsigned lds0 = lds_addr_of(smem);

  auto stage = [&](int buf) {
    const unsigned a_img = lds0 + buf * C::STAGE_BYTES;
    issue_image<C::NW, C::A_LOADS>(a_src, fa.voff, a_img, wave);
    if constexpr (C::LAYOUT == TN) issue_image<C::NW, C::B_LOADS>(b_src, fbt.voff, a_img + C::A_BYTES, wave);
    else issue_image<C::NW, C::B_LOADS>(b_src, fbn.voff, a_img + C::A_BYTES, wave);
    a_src += a_step;
    b_src += b_step;
  };

  f4 acc[C::FM][C::FN];
#pragma unroll
  for (int i = 0; i < C::FM; ++i)
#pragma unroll
    for (int j = 0; j < C::FN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

  const int nt = K / C::BK;
  constexpr int S = C::STAGES;
  // prologue: tiles 0 .. S-2 in flight
#pragma unroll
  for (int s = 0; s < S - 1; ++s)
    if (s < nt) stage(s);

  int buf = 0;
  for (int t = 0; t < nt; ++t) {
    // tile t must have landed; tiles t+1 .. t+S-2 may stay in flight
    if (nt - 1 - t >= S - 2) wait_vmcnt<(S - 2) * C::LOADS>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();  // every wave's share of tile t is in LDS; all waves finished tile t-1
    asm volatile("" ::: "memory");   // s_barrier is IntrNoMem: pin the fragment reads below it
    if (t + S - 1 < nt) {
      int pbuf = buf + (S - 1);
      if (pbuf >= S) pbuf -= S;
      stage(pbuf);  // overwrites the buffer read during iteration t-1
    }
    const char* a_img = smem + buf * C::STAGE_BYTES;
    compute_tile<C>(a_img, a_img + C::A_BYTES, wm, wn, lane, acc);
    buf = (buf + 1 == S) ? 0 : buf + 1;
  }
  if constexpr (C::FN == 4 && C::FM % 4 == 0 && C::NW * 64 * 144 <= C::LDS_BYTES) {
    __syncthreads();  // other waves may still be reading fragments of the last K tile
    store_tile_via_lds<C::FM, C::FN>(Cmat, N, m0 + wm * C::WTM, n0 + wn * C::WTN, lane, smem + wave * (64 * 144), acc);
  } else {
    store_tile<C>(Cmat, N, m0, n0, wm, wn, lane, acc);
  }
}

// ---- ping-pong kernel: 256x256x64, 8 waves, the two waves of every SIMD alternate roles -----------
// Waves 0-3 (group 0, M rows 0-127) and waves 4-7 (group 1, rows 128-255) are the two residents of
// SIMD 0-3. Each K tile is cut into 4 quadrant phases of the wave's 128x64 output (64x32 each =
// 16 MFMAs over both k-steps); a phase is  [LDS fragment reads] s_barrier [16 MFMA] s_barrier.
// Group 1 runs ONE barrier behind group 0, so in every barrier-to-barrier slot one wave of a SIMD
// is in its MFMA segment while its partner is in its read (+DMA issue) segment: matrix pipe and
// LDS pipe overlap without any intra-wave software pipelining (cdna guide T3/T5 mechanism, built
// here on a 2-buffer K-tile ring). s_setprio(1) around the MFMA segment lets the scheduler favour
// the computing wave.
//   slot:    0        1        2        3        4        5        6        7      (per K tile)
//   G0:  DMA+L(q0)  M(q0)    L(q1)    M(q1)    L(q2)    M(q2)   wait DMA  M(q3)
//   G1:   M(q3')   DMA+L(q0) M(q0)    L(q1)    M(q1)    L(q2)    M(q2)   wait DMA
// quadrants: q0=(A0,B0) loads A0+B0 (12 frag reads), q1=(A0,B1) loads B1 (4), q2=(A1,B1) loads A1
// (8), q3=(A1,B0) loads nothing (B0 still live).
// Ordering of the LDS-DMA ring (2 K-tile buffers): tile t+1 is issued in slot 0 of tile t into the
// buffer tile t-1 was read from (its last reads, L(q2), ended >= 2 barriers earlier for both
// groups); every wave drains its own DMA with vmcnt(0) BEFORE the barrier that ends slot 6, which
// for group 1 is one rendezvous earlier than group 0's first read of tile t+1.
// BM = 256 (default) or 192: the 192-row form (wave tile 96x64, quadrants of 48x32 = 12 MFMAs) exists for problem
// sizes whose 256x256 tiling leaves CUs idle: 3072^3 is 144 tiles of 256x256 on 256 CUs (56 %) but 192 tiles of 192x256
// (75 %, each 3/4 of the work): one round either way, so the launch is ~25 % shorter.
template <int LAYOUT, int EPI = 0, int SLOTS = 8, int ABL = 0, int SPLIT = 0, int BM = 256>
__global__ __launch_bounds__(512, 2) void hgemm_pp_kernel(const half_t* __restrict__ A, const half_t* __restrict__ B,
                                                          half_t* __restrict__ Cmat, int M, int N, int K,
                                                          int tiles_m, int tiles_n, int swizzle, int band) {
  using C = Cfg<BM, 256, 64, 2, 4, 2, LAYOUT>;
  constexpr int WR = C::WTM, HM = WR / 2, NI = HM / 16;  // wave rows, rows per A half, 16-row fragments per half
  static_assert(BM == 256 || (BM == 192 && SPLIT != 1 && (SLOTS == 4 || SLOTS == 2)), "192-row form: un-split or in-compute DMA");
  static_assert(SLOTS != 2 || SPLIT == 0, "2-slot form: un-split DMA");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;  // wm == group
  int tm, tn;
  tile_coords(blockIdx.x, gridDim.x, tiles_m, tiles_n, swizzle, band, tm, tn);
  const int m0 = tm * C::BM, n0 = tn * C::BN;

  KFill<C, C::A_LOADS> fa;
  fa.init(K, wave, lane);
  KFill<C, C::B_LOADS> fbt;
  NFill<C, C::B_LOADS> fbn;
  if constexpr (LAYOUT == TN) fbt.init(K, wave, lane);
  else fbn.init(N, wave, lane);
  const char* a_src = reinterpret_cast<const char*>(A + (size_t)m0 * K);
  const char* b_src = (LAYOUT == TN) ? reinterpret_cast<const char*>(B + (size_t)n0 * K)
                                     : reinterpret_cast<const char*>(B + n0);
  const size_t a_step = (size_t)C::BK * 2;
  const size_t b_step = (LAYOUT == TN) ? (size_t)C::BK * 2 : (size_t)C::BK * N * 2;
  const unsigned lds0 = lds_addr_of(smem);
  auto stage = [&](int buf) {
    const unsigned a_img = lds0 + buf * C::STAGE_BYTES;
    issue_image<C::NW, C::A_LOADS>(a_src, fa.voff, a_img, wave);
    if constexpr (LAYOUT == TN) issue_image<C::NW, C::B_LOADS>(b_src, fbt.voff, a_img + C::A_BYTES, wave);
    else issue_image<C::NW, C::B_LOADS>(b_src, fbn.voff, a_img + C::A_BYTES, wave);
    a_src += a_step;
    b_src += b_step;
  };

  f4 acc[2 * NI][4];
#pragma unroll
  for (int i = 0; i < 2 * NI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

  h8 af2[2][SLOTS == 2 ? 2 * NI : 1], bf2[2][SLOTS == 2 ? 4 : 1];  // SLOTS == 2: every fragment of a K tile live at once
  h8 af[2][NI];    // A half (NI row-fragments) x 2 k-steps, re-used for A0 then A1
  h8 bf[2][2][2];  // [B half][k-step][2 col-fragments], both halves stay live

  bool first_tile = true;  // ablation builds only
  auto load_a = [&](const char* a_img, int half) {
    if constexpr ((ABL & 1) != 0) { if (!first_tile) return; }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < NI; ++i)
        af[kk][i] = read_kfrag<64>(a_img, wm * WR + half * HM + i * 16 + (lane & 15), lane, kk);
  };
  auto load_b = [&](const char* b_img, int half) {
    if constexpr ((ABL & 1) != 0) { if (!first_tile) return; }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if constexpr (LAYOUT == TN)
          bf[half][kk][j] = read_kfrag<64>(b_img, wn * 64 + half * 32 + j * 16 + (lane & 15), lane, kk);
        else
          bf[half][kk][j] = read_nfrag<256>(b_img, wn * 64 + half * 32 + j * 16, lane, kk);
      }
  };
  auto mma = [&](int ah, int bh) {
    if constexpr ((ABL & 8) == 0) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[ah * NI + i][bh * 2 + j] =
              __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[bh][kk][j], af[kk][i], acc[ah * NI + i][bh * 2 + j], 0, 0, 0);
    if constexpr ((ABL & 8) == 0) __builtin_amdgcn_s_setprio(0);
  };
#define PP_BARRIER()                          \
  do {                                        \
    __builtin_amdgcn_sched_barrier(0);        \
    if constexpr ((ABL & 4) == 0) __builtin_amdgcn_s_barrier(); \
    asm volatile("" ::: "memory");            \
    __builtin_amdgcn_sched_barrier(0);        \
  } while (0)

  const int nt = K / 64;
  stage(0);
  wait_vmcnt<0>();
  PP_BARRIER();                  // tile 0 visible to everyone
  if (wm == 1) PP_BARRIER();     // group 1 runs one slot behind

  for (int t = 0; t < nt; ++t) {
    const char* a_img = smem + (t & 1) * C::STAGE_BYTES;
    const char* b_img = a_img + C::A_BYTES;
    if constexpr ((ABL & 2) == 0 && SPLIT == 0 && SLOTS != 2) { if (t + 1 < nt) stage((t + 1) & 1); }
    if constexpr (SLOTS == 8) {
      load_a(a_img, 0);
      load_b(b_img, 0);
      PP_BARRIER();
      mma(0, 0);
      PP_BARRIER();
      load_b(b_img, 1);
      PP_BARRIER();
      mma(0, 1);
      PP_BARRIER();
      load_a(a_img, 1);
      PP_BARRIER();
      mma(1, 1);
      PP_BARRIER();
      wait_vmcnt<0>();             // own pieces of tile t+1 have landed
      PP_BARRIER();
      mma(1, 0);
      PP_BARRIER();
    } else if constexpr (SPLIT == 0 && SLOTS == 4) {
      // 4 slots per tile, 32 MFMAs per compute slot: half the barriers per tile.
      //   G0:  DMA+L(A0,B0,B1)   M(q0,q1)   L(A1)+wait DMA   M(q2,q3)
      // Reads are drained (lgkmcnt(0)) BEFORE the barrier that ends a read slot: the partner group
      // issues the next tile's DMA into this tile's ring buffer right after that rendezvous.
      load_a(a_img, 0);
      load_b(b_img, 0);
      load_b(b_img, 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      PP_BARRIER();
      mma(0, 0);
      mma(0, 1);
      PP_BARRIER();
      load_a(a_img, 1);
      wait_vmcnt<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      PP_BARRIER();
      mma(1, 1);
      mma(1, 0);
      PP_BARRIER();
    }
    if constexpr (SLOTS == 2) {
      // 2 slots per tile: ONE read slot (next tile's DMA, all 24 fragments of this tile) and ONE compute slot of 64
      // MFMAs -- half the barriers of the 4-slot form, at the price of all fragments live at once (96 registers).
      // Ring: DMA(t+1) goes to the buffer tile t-1 was read from; both groups finished those reads (lgkmcnt(0)
      // before the barrier that ended their read slot) at least one slot earlier. Every wave drains its own DMA
      // before the barrier that ends its READ slot: the pieces had the partner's whole compute slot to land.
      if (t + 1 < nt) stage((t + 1) & 1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int i = 0; i < 2 * NI; ++i)
          af2[kk][i] = read_kfrag<64>(a_img, wm * WR + i * 16 + (lane & 15), lane, kk);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (LAYOUT == TN) bf2[kk][j] = read_kfrag<64>(b_img, wn * 64 + j * 16 + (lane & 15), lane, kk);
          else bf2[kk][j] = read_nfrag<256>(b_img, wn * 64 + j * 16, lane, kk);
        }
      }
      wait_vmcnt<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      PP_BARRIER();
      if constexpr ((ABL & 8) == 0) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 2 * NI; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf2[kk][j], af2[kk][i], acc[i][j], 0, 0, 0);
      if constexpr ((ABL & 8) == 0) __builtin_amdgcn_s_setprio(0);
      PP_BARRIER();
    }
    if constexpr (SLOTS == 4 && SPLIT == 2) {
      // Same 4 slots, but the next tile's DMA is issued INSIDE the first compute slot, one 1-KiB piece behind every
      // fourth MFMA: an LDS-DMA issue costs 60-180 cycles of the issuing wave (MI355X_MICROARCH "LDS-DMA piece"), which
      // in a read slot is exposed (the partner's 32 MFMAs take ~512 cycles, 6-8 pieces + 16 fragment reads more) but
      // behind an MFMA only uses issue slots the matrix pipe leaves free. Ring: the pieces of tile t+1 go to the buffer
      // of tile t-1, whose last reads (slot 2 of tile t-1 of the younger group) ended before this slot began; every wave
      // drains its own pieces at the end of its NEXT read slot (a full slot later), one rendezvous before the older
      // group reads tile t+1.
      const bool more = (t + 1 < nt);
      const unsigned nimg = lds0 + ((t + 1) & 1) * C::STAGE_BYTES;
      auto issue_piece = [&](int n) {  // n < A_LOADS: A piece n, else B piece n - A_LOADS
        if (!more) return;
        if (n < C::A_LOADS) {
          glds16_asm(a_src, fa.voff[n < C::A_LOADS ? n : 0], nimg + (unsigned)(n * C::NW + wave) * 1024u);
        } else {
          const int m = n - C::A_LOADS;
          if constexpr (LAYOUT == TN) glds16_asm(b_src, fbt.voff[m < C::B_LOADS ? m : 0], nimg + C::A_BYTES + (unsigned)(m * C::NW + wave) * 1024u);
          else glds16_asm(b_src, fbn.voff[m < C::B_LOADS ? m : 0], nimg + C::A_BYTES + (unsigned)(m * C::NW + wave) * 1024u);
        }
      };
      auto mma_dma = [&](int ah, int bh, int piece0) {
        if constexpr ((ABL & 8) == 0) __builtin_amdgcn_s_setprio(1);
        int cnt = 0;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              acc[ah * NI + i][bh * 2 + j] =
                  __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[bh][kk][j], af[kk][i], acc[ah * NI + i][bh * 2 + j], 0, 0, 0);
              if ((cnt & 3) == 3) {
                const int pc = piece0 + (cnt >> 2);
                if (pc < C::A_LOADS + C::B_LOADS && (cnt >> 2) < (2 * NI * 2) / 4) issue_piece(pc);
                __builtin_amdgcn_sched_barrier(0);
              }
              ++cnt;
            }
        if constexpr ((ABL & 8) == 0) __builtin_amdgcn_s_setprio(0);
      };
      load_a(a_img, 0);
      load_b(b_img, 0);
      load_b(b_img, 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      PP_BARRIER();
      constexpr int PER_MMA = (2 * NI * 2) / 4, NP = C::A_LOADS + C::B_LOADS;  // pieces behind one quadrant's MFMAs
      mma_dma(0, 0, 0);
      mma_dma(0, 1, PER_MMA);
#pragma unroll
      for (int pc = 2 * PER_MMA; pc < NP; ++pc) issue_piece(pc);  // 192-row form: 7 pieces, 6 slots behind MFMAs
      if (more) {
        a_src += a_step;
        b_src += b_step;
      }
      PP_BARRIER();
      load_a(a_img, 1);
      wait_vmcnt<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      PP_BARRIER();
      mma(1, 1);
      mma(1, 0);
      PP_BARRIER();
    }
    if constexpr (SLOTS == 4 && SPLIT == 1) {
      // Same 4 slots, but the next tile's DMA is split over BOTH read slots so neither is bound by the
      // texture-address path (32 pieces of 1 KiB per group per slot were): slot 0 issues H0(t+1) = the
      // A rows every wave reads first (A0 halves: pieces 0,2) + all of B (6 pieces); slot 2 issues
      // H1(t+1) = the A1 halves (pieces 1,3). Counted waits keep the younger group in flight:
      //   end of slot 2 (t):   vmcnt(2) -> H0(t+1) landed (needed by slot 0 of t+1), H1(t+1) still flying
      //   end of slot 0 (t+1): vmcnt(6) -> H1(t+1) landed (needed by slot 2 of t+1), H0(t+2) still flying
      const bool more = (t + 1 < nt);
      if (more) {
        const unsigned nimg = lds0 + ((t + 1) & 1) * C::STAGE_BYTES;
        issue_image_masked<C::NW, C::A_LOADS, 0x5u>(a_src, fa.voff, nimg, wave);
        if constexpr (LAYOUT == TN) issue_image<C::NW, C::B_LOADS>(b_src, fbt.voff, nimg + C::A_BYTES, wave);
        else issue_image<C::NW, C::B_LOADS>(b_src, fbn.voff, nimg + C::A_BYTES, wave);
        b_src += b_step;
      }
      load_a(a_img, 0);
      load_b(b_img, 0);
      load_b(b_img, 1);
      if (more) wait_vmcnt<6>();
      else wait_vmcnt<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      PP_BARRIER();
      mma(0, 0);
      mma(0, 1);
      PP_BARRIER();
      if (more) {
        const unsigned nimg = lds0 + ((t + 1) & 1) * C::STAGE_BYTES;
        issue_image_masked<C::NW, C::A_LOADS, 0xAu>(a_src, fa.voff, nimg, wave);
        a_src += a_step;
      }
      load_a(a_img, 1);
      if (more) wait_vmcnt<2>();
      else wait_vmcnt<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      PP_BARRIER();
      mma(1, 1);
      mma(1, 0);
      PP_BARRIER();
    }
    first_tile = false;
  }
  if (wm == 0) PP_BARRIER();     // balance the stagger
#undef PP_BARRIER
  if constexpr (EPI == 0) {
    store_tile<C>(Cmat, N, m0, n0, wm, wn, lane, acc);
  } else if constexpr (EPI == 2) {
    // every wave is past its last fragment read of the final K tile (see the slot table above)
    store_tile_via_lds<2 * NI, 4>(Cmat, N, m0 + wm * WR, n0 + wn * 64, lane, smem + wave * (64 * 144), acc);
  } else {  // measurement-only variant: keep the accumulators live, store (almost) nothing
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2 * NI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 123.456f) Cmat[(size_t)m0 * N + n0] = (half_t)s;
  }
}

template <int LAYOUT, int EPI = 0, int SLOTS = 8, int ABL = 0, int SPLIT = 0, int BM = 256>
int launch_pp(const void* a, const void* b, void* c, int M, int N, int K, int swizzle, int swizzle_stride,
              hipStream_t stream) {
  using C = Cfg<BM, 256, 64, 2, 4, 2, LAYOUT>;
  if (M % BM || N % 256 || K % 64) return CLN_ERR_UNSUPPORTED;
  static cln_lds_attr lds_attr;  // per device, thread-safe (common.h)
  if (cln_ensure_lds(lds_attr, reinterpret_cast<const void*>(&hgemm_pp_kernel<LAYOUT, EPI, SLOTS, ABL, SPLIT, BM>), C::LDS_BYTES) != CLN_OK) return CLN_ERR_LAUNCH;
  const int tiles_m = M / BM, tiles_n = N / 256;
  int band = (swizzle && swizzle_stride >= 256) ? swizzle_stride / 256 : tiles_n;
  CLN_LAUNCH((hgemm_pp_kernel<LAYOUT, EPI, SLOTS, ABL, SPLIT, BM>), dim3(tiles_m * tiles_n), dim3(512), C::LDS_BYTES, stream,
                     (const half_t*)a, (const half_t*)b, (half_t*)c, M, N, K, tiles_m, tiles_n, swizzle ? 1 : 0, band);
  return cln_check_launch();
}

// ---- ping-pong over k-halves: 256x256 tile, BK = 32 sub-tiles, 4-deep LDS-DMA ring ---------------
// Every barrier-to-barrier slot is identical: a wave in its READ slot issues the DMA of sub-tile
// s+3 (4 pieces of 1 KiB), reads the 12 fragments of sub-tile s (8 A + 4 B for one 32-deep k-step)
// and drains them; in its COMPUTE slot it issues the 32 MFMAs of that k-step over all 8x4
// accumulators. Group 1 (waves 4-7 = the second resident of each SIMD) runs one barrier behind
// group 0, so a SIMD always has one wave computing and one reading. Versus the quadrant schedule
// above: balanced read slots (the quadrant slots carried 24/8 reads and 8/0 DMA pieces), prefetch
// distance 3 sub-tiles instead of 1 tile, 48 instead of 64 fragment registers.
//   ring invariants (4 buffers of 32 KiB):  DMA(s+3) overwrites buffer (s-1)&3 whose reads were
//   drained (lgkmcnt(0)) before the barrier that ended READ(s-1) of BOTH groups; every wave drains
//   its own DMA(s+1) pieces (counted vmcnt) before the barrier that ends READ(s), which precedes
//   the first read of sub-tile s+1 by either group.
template <int LAYOUT, int EPI = 2>
__global__ __launch_bounds__(512, 2) void hgemm_pp32_kernel(const half_t* __restrict__ A,
                                                            const half_t* __restrict__ B,
                                                            half_t* __restrict__ Cmat, int M, int N, int K,
                                                            int tiles_m, int tiles_n, int swizzle, int band) {
  using C = Cfg<256, 256, 32, 2, 4, 4, LAYOUT>;
  static_assert(C::LOADS == 4, "4 DMA pieces per wave per sub-tile");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  int tm, tn;
  tile_coords(blockIdx.x, gridDim.x, tiles_m, tiles_n, swizzle, band, tm, tn);
  const int m0 = tm * 256, n0 = tn * 256;

  KFill<C, C::A_LOADS> fa;
  fa.init(K, wave, lane);
  KFill<C, C::B_LOADS> fbt;
  NFill<C, C::B_LOADS> fbn;
  if constexpr (LAYOUT == TN) fbt.init(K, wave, lane);
  else fbn.init(N, wave, lane);
  const char* a_src = reinterpret_cast<const char*>(A + (size_t)m0 * K);
  const char* b_src = (LAYOUT == TN) ? reinterpret_cast<const char*>(B + (size_t)n0 * K)
                                     : reinterpret_cast<const char*>(B + n0);
  const size_t a_step = 64;  // 32 halves
  const size_t b_step = (LAYOUT == TN) ? (size_t)64 : (size_t)32 * N * 2;
  const unsigned lds0 = lds_addr_of(smem);
  auto stage = [&](int buf) {
    const unsigned a_img = lds0 + buf * C::STAGE_BYTES;
    issue_image<C::NW, C::A_LOADS>(a_src, fa.voff, a_img, wave);
    if constexpr (LAYOUT == TN) issue_image<C::NW, C::B_LOADS>(b_src, fbt.voff, a_img + C::A_BYTES, wave);
    else issue_image<C::NW, C::B_LOADS>(b_src, fbn.voff, a_img + C::A_BYTES, wave);
    a_src += a_step;
    b_src += b_step;
  };

  f4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

#define PP_BARRIER()                          \
  do {                                        \
    __builtin_amdgcn_sched_barrier(0);        \
    __builtin_amdgcn_s_barrier();             \
    asm volatile("" ::: "memory");            \
    __builtin_amdgcn_sched_barrier(0);        \
  } while (0)

  const int ns = K / 32;
  // prologue: sub-tiles 0..2 in flight, sub-tile 0 landed and visible
  stage(0);
  if (ns > 1) stage(1);
  if (ns > 2) stage(2);
  if (ns > 2) wait_vmcnt<8>();
  else if (ns > 1) wait_vmcnt<4>();
  else wait_vmcnt<0>();
  PP_BARRIER();
  if (wm == 1) PP_BARRIER();  // group 1 runs one slot behind

  for (int s = 0; s < ns; ++s) {
    // ---------------- READ slot
    if (s + 3 < ns) stage((s + 3) & 3);
    const char* a_img = smem + (s & 3) * C::STAGE_BYTES;
    const char* b_img = a_img + C::A_BYTES;
    h8 af[8], bf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (LAYOUT == TN) bf[j] = read_kfrag<32>(b_img, wn * 64 + j * 16 + (lane & 15), lane, 0);
      else bf[j] = read_nfrag<256>(b_img, wn * 64 + j * 16, lane, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) af[i] = read_kfrag<32>(a_img, wm * 128 + i * 16 + (lane & 15), lane, 0);
    // own pieces of sub-tile s+1 must have landed; s+2, s+3 may stay in flight
    if (s + 3 < ns) wait_vmcnt<8>();
    else if (s + 2 < ns) wait_vmcnt<4>();
    else wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PP_BARRIER();
    // ---------------- COMPUTE slot
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    PP_BARRIER();
  }
  if (wm == 0) PP_BARRIER();  // balance the stagger
#undef PP_BARRIER
  if constexpr (EPI == 2) {
    store_tile_via_lds<8, 4>(Cmat, N, m0 + wm * 128, n0 + wn * 64, lane, smem + wave * (64 * 144), acc);
  } else if constexpr (EPI == 0) {
    store_tile<C>(Cmat, N, m0, n0, wm, wn, lane, acc);
  } else {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (t == 123.456f) Cmat[(size_t)m0 * N + n0] = (half_t)t;
  }
}

template <int LAYOUT, int EPI = 2>
int launch_pp32(const void* a, const void* b, void* c, int M, int N, int K, int swizzle, int swizzle_stride,
                hipStream_t stream) {
  using C = Cfg<256, 256, 32, 2, 4, 4, LAYOUT>;
  if (M % 256 || N % 256 || K % 32) return CLN_ERR_UNSUPPORTED;
  static cln_lds_attr lds_attr;  // per device, thread-safe (common.h)
  if (cln_ensure_lds(lds_attr, reinterpret_cast<const void*>(&hgemm_pp32_kernel<LAYOUT, EPI>), C::LDS_BYTES) != CLN_OK) return CLN_ERR_LAUNCH;
  const int tiles_m = M / 256, tiles_n = N / 256;
  int band = (swizzle && swizzle_stride >= 256) ? swizzle_stride / 256 : tiles_n;
  CLN_LAUNCH((hgemm_pp32_kernel<LAYOUT, EPI>), dim3(tiles_m * tiles_n), dim3(512), C::LDS_BYTES, stream,
                     (const half_t*)a, (const half_t*)b, (half_t*)c, M, N, K, tiles_m, tiles_n, swizzle ? 1 : 0, band);
  return cln_check_launch();
}

// ---- ping-pong kernel on the 32x32x16 matrix instruction -------------------------------------------
// Same 256x256x64 tile, 2-buffer ring, split LDS-DMA and two-group stagger as hgemm_pp_kernel<SLOTS=4,
// SPLIT=1>, but the wave's 128x64 result is held as 4x2 tiles of v_mfma_f32_32x32x16_f16: half the MFMA
// instructions per K tile (32 instead of 64, each 8 passes), half the operand-register reads per flop, and
// the 32x32 shape's higher sustained rate (cdna guide section 3 ubench: 2178 vs 1955 TF for f16).
// Fragment maps (operands swapped so a lane owns 4 consecutive n of one C row):
//   "A" slot  <- B fragment: MFMA row  = n (lane&31), k = 8*(lane>>5) .. +7
//   "B" slot  <- A fragment: MFMA col  = m (lane&31), k = 8*(lane>>5) .. +7
//   result reg r: m = lane&31, n = (r&3) + 8*(r>>2) + 4*(lane>>5)
// K-contiguous images keep kswz<64> (conflict-free for the 32-row / 4x16-lane-group b128 pattern as well:
// the 8 even and the 8 odd rows of every lane group get distinct (row>>1)&7). The NN B image gets its own
// chunk swizzle: a 32-lane half reads rows r..r+3 of TWO adjacent 16-column blocks, so rows are spread by
// 4 chunks (64 B) instead of 2.
__device__ __forceinline__ int nswz32(int krow) { return (krow & 3) << 2; }

template <typename C, int NLOADS>
struct NFill32 {
  unsigned voff[NLOADS];
  __device__ __forceinline__ void init(int N, int wave, int lane) {
    constexpr int LPR = C::BN / 8;
    constexpr int RPI = 64 / LPR;
#pragma unroll
    for (int i = 0; i < NLOADS; ++i) {
      const int t = i * C::NW + wave;
      const int krow = t * RPI + lane / LPR;
      const int c = lane % LPR;
      voff[i] = ((unsigned)krow * (unsigned)N + ((c ^ nswz32(krow)) << 3)) * 2u;
    }
  }
};

__device__ __forceinline__ h8 read_kfrag32(const char* img, int row, int lane, int kk) {
  const int q = kk * 2 + (lane >> 5);
  return *reinterpret_cast<const h8*>(img + row * 128 + ((q ^ kswz<64>(row)) << 4));
}
// NN B image [64 k][BN n]: 32 columns n0w..n0w+31 (tile-relative), k = kk*16 + 8*(lane>>5) .. +7.
template <int BN>
__device__ __forceinline__ h8 read_nfrag32(const char* img, int n0w, int lane, int kk) {
  const int i = lane & 15, gp = lane >> 4;
  const int c = ((n0w + 16 * (gp & 1)) >> 3) + ((i & 3) >> 1);
  const int k_lo = kk * 16 + 8 * (gp >> 1) + (i >> 2);
  const int k_hi = k_lo + 4;
  const char* p_lo = img + k_lo * (BN * 2) + ((c ^ nswz32(k_lo)) << 4) + ((i & 1) << 3);
  const char* p_hi = img + k_hi * (BN * 2) + ((c ^ nswz32(k_hi)) << 4) + ((i & 1) << 3);
  return h8_cat(lds_read_tr16(p_lo), lds_read_tr16(p_hi));
}

// LDS-staged epilogue for the 4x2 grid of 32x32 result tiles of one wave (128 rows x 64 columns).
__device__ __forceinline__ void store_tile32_via_lds(half_t* Cmat, int N, int row0, int col0, int lane,
                                                     char* wave_lds, const f16v (&acc)[4][2]) {
  constexpr int RS = 144;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const f16v& v = acc[h * 2 + i][j];
          h4 o = {(half_t)v[rq * 4], (half_t)v[rq * 4 + 1], (half_t)v[rq * 4 + 2], (half_t)v[rq * 4 + 3]};
          *reinterpret_cast<h4*>(wave_lds + (i * 32 + (lane & 31)) * RS + (j * 32 + 8 * rq + 4 * (lane >> 5)) * 2) = o;
        }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int r = it * 8 + (lane >> 3);
      const u4 v = *reinterpret_cast<const u4*>(wave_lds + r * RS + (lane & 7) * 16);
      *reinterpret_cast<u4*>(Cmat + (size_t)(row0 + h * 64 + r) * N + col0 + (lane & 7) * 8) = v;
    }
  }
}

// ABL bits (measurement builds): 1 no fragment reads after tile 0, 2 no DMA after tile 0, 4 no barriers,
// 8 no setprio.  EPI: 2 LDS-staged store, 1 no store (probe).
template <int LAYOUT, int EPI = 2, int ABL = 0>
__global__ __launch_bounds__(512, 2) void hgemm_m32_kernel(const half_t* __restrict__ A, const half_t* __restrict__ B,
                                                           half_t* __restrict__ Cmat, int M, int N, int K,
                                                           int tiles_m, int tiles_n, int swizzle, int band) {
  using C = Cfg<256, 256, 64, 2, 4, 2, LAYOUT>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;  // wm == group
  int tm, tn;
  tile_coords(blockIdx.x, gridDim.x, tiles_m, tiles_n, swizzle, band, tm, tn);
  const int m0 = tm * 256, n0 = tn * 256;

  KFill<C, C::A_LOADS> fa;
  fa.init(K, wave, lane);
  KFill<C, C::B_LOADS> fbt;
  NFill32<C, C::B_LOADS> fbn;
  if constexpr (LAYOUT == TN) fbt.init(K, wave, lane);
  else fbn.init(N, wave, lane);
  const char* a_src = reinterpret_cast<const char*>(A + (size_t)m0 * K);
  const char* b_src = (LAYOUT == TN) ? reinterpret_cast<const char*>(B + (size_t)n0 * K)
                                     : reinterpret_cast<const char*>(B + n0);
  const size_t a_step = 128;
  const size_t b_step = (LAYOUT == TN) ? (size_t)128 : (size_t)64 * N * 2;
  const unsigned lds0 = lds_addr_of(smem);
  auto issue_b = [&](unsigned img) {
    if constexpr (LAYOUT == TN) issue_image<C::NW, C::B_LOADS>(b_src, fbt.voff, img + C::A_BYTES, wave);
    else issue_image<C::NW, C::B_LOADS>(b_src, fbn.voff, img + C::A_BYTES, wave);
  };

  f16v acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  h8 af[4][2];  // [k-step][m-tile of the current 64-row half]
  h8 bf[2][4];  // [n-tile][k-step]

  bool first_tile = true;
  auto load_a = [&](const char* a_img, int half) {
    if constexpr ((ABL & 1) != 0) { if (!first_tile) return; }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        af[kk][i] = read_kfrag32(a_img, wm * 128 + half * 64 + i * 32 + (lane & 31), lane, kk);
  };
  auto load_b = [&](const char* b_img) {
    if constexpr ((ABL & 1) != 0) { if (!first_tile) return; }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if constexpr (LAYOUT == TN) bf[j][kk] = read_kfrag32(b_img, wn * 64 + j * 32 + (lane & 31), lane, kk);
        else bf[j][kk] = read_nfrag32<256>(b_img, wn * 64 + j * 32, lane, kk);
      }
  };
  auto mma = [&](int half) {
    if constexpr ((ABL & 8) == 0) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[half * 2 + i][j] =
              __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[j][kk], af[kk][i], acc[half * 2 + i][j], 0, 0, 0);
    if constexpr ((ABL & 8) == 0) __builtin_amdgcn_s_setprio(0);
  };
#define M32_BARRIER()                                            \
  do {                                                           \
    __builtin_amdgcn_sched_barrier(0);                           \
    if constexpr ((ABL & 4) == 0) __builtin_amdgcn_s_barrier();  \
    asm volatile("" ::: "memory");                               \
    __builtin_amdgcn_sched_barrier(0);                           \
  } while (0)

  const int nt = K / 64;
  {  // tile 0: everything
    issue_image<C::NW, C::A_LOADS>(a_src, fa.voff, lds0, wave);
    issue_b(lds0);
    a_src += a_step;
    b_src += b_step;
  }
  wait_vmcnt<0>();
  M32_BARRIER();
  if (wm == 1) M32_BARRIER();  // group 1 runs one slot behind

  for (int t = 0; t < nt; ++t) {
    const char* a_img = smem + (t & 1) * C::STAGE_BYTES;
    const char* b_img = a_img + C::A_BYTES;
    bool more = (t + 1 < nt);
    if constexpr ((ABL & 2) != 0) more = false;
    const unsigned nimg = lds0 + ((t + 1) & 1) * C::STAGE_BYTES;
    // ---- read slot 0: H0(t+1) = A rows every group reads first + all of B
    if (more) {
      issue_image_masked<C::NW, C::A_LOADS, 0x5u>(a_src, fa.voff, nimg, wave);
      issue_b(nimg);
      b_src += b_step;
    }
    load_a(a_img, 0);
    load_b(b_img);
    if (more) wait_vmcnt<6>();  // H1(t) landed (needed by read slot 1), H0(t+1) may fly
    else wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    M32_BARRIER();
    mma(0);
    M32_BARRIER();
    // ---- read slot 1: H1(t+1) = the second 64-row halves of A
    if (more) {
      issue_image_masked<C::NW, C::A_LOADS, 0xAu>(a_src, fa.voff, nimg, wave);
      a_src += a_step;
    }
    load_a(a_img, 1);
    if (more) wait_vmcnt<2>();  // H0(t+1) landed (needed by read slot 0 of t+1), H1(t+1) may fly
    else wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    M32_BARRIER();
    mma(1);
    M32_BARRIER();
    first_tile = false;
  }
  if (wm == 0) M32_BARRIER();  // balance the stagger
#undef M32_BARRIER
  if constexpr (EPI == 2) {
    store_tile32_via_lds(Cmat, N, m0 + wm * 128, n0 + wn * 64, lane, smem + wave * (64 * 144), acc);
  } else {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 123.456f) Cmat[(size_t)m0 * N + n0] = (half_t)s;
  }
}

template <int LAYOUT, int EPI = 2, int ABL = 0>
int launch_m32(const void* a, const void* b, void* c, int M, int N, int K, int swizzle, int swizzle_stride,
               hipStream_t stream) {
  using C = Cfg<256, 256, 64, 2, 4, 2, LAYOUT>;
  if (M % 256 || N % 256 || K % 64) return CLN_ERR_UNSUPPORTED;
  static cln_lds_attr lds_attr;  // per device, thread-safe (common.h)
  if (cln_ensure_lds(lds_attr, reinterpret_cast<const void*>(&hgemm_m32_kernel<LAYOUT, EPI, ABL>), C::LDS_BYTES) != CLN_OK) return CLN_ERR_LAUNCH;
  const int tiles_m = M / 256, tiles_n = N / 256;
  int band = (swizzle && swizzle_stride >= 256) ? swizzle_stride / 256 : tiles_n;
  CLN_LAUNCH((hgemm_m32_kernel<LAYOUT, EPI, ABL>), dim3(tiles_m * tiles_n), dim3(512), C::LDS_BYTES, stream,
             (const half_t*)a, (const half_t*)b, (half_t*)c, M, N, K, tiles_m, tiles_n, swizzle ? 1 : 0, band);
  return cln_check_launch();
}

// ---- single-stage, register-staged rung (the "1-stage MMA tile" of config C2) ------------------
// Same LDS images and fragment readers; the fill goes global -> VGPR -> ds_write_b128 with two
// barriers per K tile and no overlap, i.e. reference hgemm_mma_m16n8k16_mma2x4_warp4x4
// (hgemm_mma.cu:135-266) re-thought for wave64/MFMA.
template <typename C>
__global__ __launch_bounds__(C::NT) void hgemm_1stage_kernel(const half_t* __restrict__ A,
                                                              const half_t* __restrict__ B,
                                                              half_t* __restrict__ Cmat, int M, int N, int K,
                                                              int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wm = wave / C::WN, wn = wave % C::WN;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int m0 = tm * C::BM, n0 = tn * C::BN;

  KFill<C, C::A_LOADS> fa;
  fa.init(K, wave, lane);
  KFill<C, C::B_LOADS> fbt;
  NFill<C, C::B_LOADS> fbn;
  if constexpr (C::LAYOUT == TN) fbt.init(K, wave, lane);
  else fbn.init(N, wave, lane);
  const char* a_src = reinterpret_cast<const char*>(A + (size_t)m0 * K);
  const char* b_src = (C::LAYOUT == TN) ? reinterpret_cast<const char*>(B + (size_t)n0 * K)
                                        : reinterpret_cast<const char*>(B + n0);
  const size_t a_step = (size_t)C::BK * 2;
  const size_t b_step = (C::LAYOUT == TN) ? (size_t)C::BK * 2 : (size_t)C::BK * N * 2;

  f4 acc[C::FM][C::FN];
#pragma unroll
  for (int i = 0; i < C::FM; ++i)
#pragma unroll
    for (int j = 0; j < C::FN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

  char* a_img = smem;
  char* b_img = smem + C::A_BYTES;
  const int nt = K / C::BK;
  for (int t = 0; t < nt; ++t) {
    u4 ra[C::A_LOADS], rb[C::B_LOADS];
#pragma unroll
    for (int i = 0; i < C::A_LOADS; ++i) ra[i] = *reinterpret_cast<const u4*>(a_src + fa.voff[i]);
#pragma unroll
    for (int i = 0; i < C::B_LOADS; ++i) {
      if constexpr (C::LAYOUT == TN) rb[i] = *reinterpret_cast<const u4*>(b_src + fbt.voff[i]);
      else rb[i] = *reinterpret_cast<const u4*>(b_src + fbn.voff[i]);
    }
    a_src += a_step;
    b_src += b_step;
    __syncthreads();  // previous tile's fragment reads are done
#pragma unroll
    for (int i = 0; i < C::A_LOADS; ++i)
      *reinterpret_cast<u4*>(a_img + (i * C::NW + wave) * 1024 + lane * 16) = ra[i];
#pragma unroll
    for (int i = 0; i < C::B_LOADS; ++i)
      *reinterpret_cast<u4*>(b_img + (i * C::NW + wave) * 1024 + lane * 16) = rb[i];
    __syncthreads();
    compute_tile<C>(a_img, b_img, wm, wn, lane, acc);
  }
  store_tile<C>(Cmat, N, m0, n0, wm, wn, lane, acc);
}

// ---- naive rung: one wave per 16x16 C tile, fragments straight from global memory --------------
// (reference hgemm_mma_m16n8k16_naive hgemm_mma.cu:44-125 and hgemm_wmma_m16n16k16_naive
// hgemm_wmma.cu:39-62: one warp per MMA tile, no reuse). v_mfma_f32_16x16x16_f16.
template <int LAYOUT>
__global__ __launch_bounds__(64) void hgemm_mfma_naive_kernel(const half_t* __restrict__ A,
                                                               const half_t* __restrict__ B,
                                                               half_t* __restrict__ Cmat, int M, int N, int K) {
  const int lane = threadIdx.x;
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
  const int i = lane & 15, g = lane >> 4;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  const bool m_ok = (m0 + i) < M, n_ok = (n0 + i) < N;
  for (int k0 = 0; k0 < K; k0 += 16) {
    const int k = k0 + 4 * g;
    h4 af = {0, 0, 0, 0}, bf = {0, 0, 0, 0};
    if (m_ok && k + 3 < K) af = *reinterpret_cast<const h4*>(A + (size_t)(m0 + i) * K + k);
    if (n_ok && k + 3 < K) {
      if constexpr (LAYOUT == TN) {
        bf = *reinterpret_cast<const h4*>(B + (size_t)(n0 + i) * K + k);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[j] = B[(size_t)(k + j) * N + n0 + i];
      }
    }
    acc = __builtin_amdgcn_mfma_f32_16x16x16f16(bf, af, acc, 0, 0, 0);
  }
  // lane holds C[m0 + i][n0 + 4g + r]
  if (m_ok) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (n0 + 4 * g + r < N) Cmat[(size_t)(m0 + i) * N + n0 + 4 * g + r] = (half_t)acc[r];
  }
}

// ---- host-side launchers -----------------------------------------------------------------------
template <typename C>
int launch_ring(const void* a, const void* b, void* c, int M, int N, int K, int swizzle, int swizzle_stride,
                hipStream_t stream) {
  if (M % C::BM || N % C::BN || K % C::BK) return CLN_ERR_UNSUPPORTED;
  static cln_lds_attr lds_attr;  // per device, thread-safe (common.h)
  if (cln_ensure_lds(lds_attr, reinterpret_cast<const void*>(&hgemm_ring_kernel<C>), C::LDS_BYTES) != CLN_OK) return CLN_ERR_LAUNCH;
  const int tiles_m = M / C::BM, tiles_n = N / C::BN;
  int band = (swizzle && swizzle_stride >= C::BN) ? swizzle_stride / C::BN : tiles_n;
  CLN_LAUNCH((hgemm_ring_kernel<C>), dim3(tiles_m * tiles_n), dim3(C::NT), C::LDS_BYTES, stream,
                     (const half_t*)a, (const half_t*)b, (half_t*)c, M, N, K, tiles_m, tiles_n, swizzle ? 1 : 0, band);
  return cln_check_launch();
}

template <typename C>
int launch_1stage(const void* a, const void* b, void* c, int M, int N, int K, hipStream_t stream) {
  if (M % C::BM || N % C::BN || K % C::BK) return CLN_ERR_UNSUPPORTED;
  const int tiles_m = M / C::BM, tiles_n = N / C::BN;
  CLN_LAUNCH((hgemm_1stage_kernel<C>), dim3(tiles_m * tiles_n), dim3(C::NT), C::STAGE_BYTES, stream,
                     (const half_t*)a, (const half_t*)b, (half_t*)c, M, N, K, tiles_m, tiles_n);
  return cln_check_launch();
}

template <int LAYOUT>
int launch_naive(const void* a, const void* b, void* c, int M, int N, int K, hipStream_t stream) {
  if (K % 4) return CLN_ERR_UNSUPPORTED;
  CLN_LAUNCH((hgemm_mfma_naive_kernel<LAYOUT>), dim3((N + 15) / 16, (M + 15) / 16), dim3(64), 0, stream,
                     (const half_t*)a, (const half_t*)b, (half_t*)c, M, N, K);
  return cln_check_launch();
}

}  // namespace hgemm
