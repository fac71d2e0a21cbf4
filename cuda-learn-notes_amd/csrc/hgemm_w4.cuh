// One-wave-per-SIMD HGEMM: 256x256x64 workgroup tile, 4 waves, each wave owns a 128x128 quadrant.
//
// Same boundary as hgemm_mfma.cuh (replaces reference kernels/hgemm/mma/basic/hgemm_mma_stage.cu:1044,:1460 and the
// TN forms hgemm_mma_stage_tn.cu:70 / cutlass/hgemm_mma_stage_tn_cute.cu:26 at large sizes). Where the ping-pong kernel
// hides LDS latency by alternating TWO waves per SIMD (128x64 wave tiles, 256 registers each), this one gives a single
// wave the whole 512-entry register file: 64 accumulator tiles (256 registers, the MFMA C/D operand -- hipcc places them
// in the AGPR half since no VALU instruction touches them inside the loop), both k-steps' fragments of A and B
// (2 x 64 registers), and software-pipelines ONE instruction stream:
//
//   production schedule (w4_sched(10)), MFMA index n within the 128 of K tile t:
//     n = 1,3,..,31    read the k-step 1 fragments of tile t (the k-step 0 MFMAs n < 64 run meanwhile)
//     n = 36       B1  lgkmcnt(0) + s_barrier: every wave has read all of tile t -> its ring buffer may be overwritten
//     n = 38 + 8p      LDS-DMA piece p of tile t+2 into that buffer, one 1-KiB piece per 8 MFMAs; pieces 12..15 fall
//                      past n = 127 and are issued by tile t+1's body at n = 6, 14, 22, 30 ("late" pieces)
//     n = 102      B2  vmcnt(9) + s_barrier: every wave's pieces of tile t+1 have landed (9 of t+2 stay in flight)
//     n = 103..118     read the k-step 0 fragments of tile t+1
//   The MFMAs are inline asm with the accumulator tied to an AGPR tuple ("+a"): with the builtin hipcc gives C and D
//   different registers and rotates the 64 tiles through ~370 v_accvgpr_mov/read/write per K tile. Inline asm is
//   invisible to hipcc's hazard pass, so the zero-fill and the epilogue reads are fenced by hand (see below) and
//   tests/test_no_spills.py checks the code object: no VALU/accvgpr write between the first and the last MFMA.
//
// Measured (profiles/r02_hgemm_w4_*.log): 4096^3 NN 1430-1470 TF vs 1340-1388 for the ping-pong kernel on the same box,
// 8192^3 1517-1548 vs 1404-1428; the package sits at its ~1390 W cap either way (MFMAs alone 727 W/PF, LDS-DMA traffic
// +143, fragment reads +71, C store +48), so what is left is energy per flop, not schedule.
//
// Versus the 128x64 wave tile: 32 fragment reads per 128 MFMAs instead of 48 (LDS bytes per flop -33 %), 2 barriers per
// 128 MFMAs instead of 4 per 64, no slot in which a SIMD's matrix pipe waits for the partner wave's rendezvous.
#pragma once
#include <type_traits>

#include "hgemm_mfma.cuh"

namespace hgemm {

// Epilogue through LDS for a wave tile of FN x 16 columns (128 or 96): [64 rows][FN*32 + 16 B] wave-private region,
// passes of up to 64 rows; a lane streams 16 bytes, one store instruction = 4 (5) rows x 256 (192) contiguous bytes.
// NT: 1 = non-temporal C stores WHEN the launch-uniform `nt_ok` says the launch's footprint is large (see W4_NT_FOOTPRINT), 2 (probe) =
// write-through (sc0 sc1), 3 (probe) = non-temporal unconditionally
template <int FM, int FN, int NT = 0>
__device__ __forceinline__ void store_wide_tile_via_lds(half_t* Cmat, int N, int row0, int col0, int lane, char* wave_lds,
                                                        const f4 (&acc)[FM][FN], int nt_ok = 1) {
  static_assert(FN == 8 || FN == 6 || FN == 5 || FN == 4, "128-, 96-, 80- or 64-column wave tile");
  constexpr int RS = FN * 32 + 16;  // row stride in bytes
  constexpr int LPR = FN * 2;       // 16-byte lanes per row
  constexpr int RPI = 64 / LPR;     // whole rows per store instruction (4 or 5; lanes >= RPI*LPR idle)
#pragma unroll
  for (int h0 = 0; h0 < FM; h0 += 4) {
    constexpr int NF_FULL = 4;
    const int nf = FM - h0 < NF_FULL ? FM - h0 : NF_FULL;  // 16-row fragments in this pass
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
    const int rows = nf * 16;
#pragma unroll
    for (int it = 0; it < (64 + RPI - 1) / RPI; ++it) {
      const int r = it * RPI + lane / LPR;
      if (it * RPI < rows && lane < RPI * LPR && r < rows) {
        const u4 v = *reinterpret_cast<const u4*>(wave_lds + r * RS + (lane % LPR) * 16);
        u4* dst = reinterpret_cast<u4*>(Cmat + (size_t)(row0 + h0 * 16 + r) * N + col0 + (lane % LPR) * 8);
        if constexpr (NT == 1) {  // (the nt form is inline asm: under a run-time flag hipcc would merge the two stores of the diamond into one plain store)
          if (nt_ok) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
          else asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
        } else if constexpr (NT == 3) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
        else if constexpr (NT == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
        else *dst = v;
      }
    }
  }
}

// The same epilogue in passes of ONE 16-row fragment row: 16 x (FN * 32 + 16) bytes of wave-private LDS (4.25 KiB at FN = 8) instead of 64 rows -- for
// the persistent form (EPI 7), whose ring already holds the next tile's first K tiles while C is stored. Same stores, same order within a row block.
template <int FM, int FN>
__device__ __forceinline__ void store_wide_tile_via_lds_rows16(half_t* Cmat, int N, int row0, int col0, int lane, char* wave_lds, const f4 (&acc)[FM][FN],
                                                               int nt_ok) {
  static_assert(FN == 8, "128-column wave tile");
  constexpr int RS = FN * 32 + 16, LPR = FN * 2, RPI = 64 / LPR;
#pragma unroll
  for (int i = 0; i < FM; ++i) {
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const f4 v = acc[i][j];
      h4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
      *reinterpret_cast<h4*>(wave_lds + (lane & 15) * RS + (j * 16 + 4 * (lane >> 4)) * 2) = o;
    }
#pragma unroll
    for (int it = 0; it < 16 / RPI; ++it) {
      const int r = it * RPI + lane / LPR;
      const u4 v = *reinterpret_cast<const u4*>(wave_lds + r * RS + (lane % LPR) * 16);
      u4* dst = reinterpret_cast<u4*>(Cmat + (size_t)(row0 + i * 16 + r) * N + col0 + (lane % LPR) * 8);
      if (nt_ok) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
      else asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads of this pass precede the writes of the next (same wave-private rows)
  }
}

// Compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>). A `#pragma unroll` loop of 128
// iterations whose body carries every hook exceeds LLVM's pragma-unroll size cap before the hooks are folded away and
// is then left rolled (accumulators indexed dynamically -> scratch memory).
template <int... Ns, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Ns...>, F&& f) {
  (f(std::integral_constant<int, Ns>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// Schedule positions (MFMA index within the NM = 2*FM*FN of one K tile); the numbered ones exist for the probe library.
struct W4Sched {
  int r1_first, r1_step;  // k-step 1 fragment reads: op r at n = r1_first + r1_step * r
  int b1;                 // lgkmcnt(0) + barrier
  int d_first, d_step;    // DMA piece p at n = d_first + d_step * p (positions >= NM: issued by the next tile's body)
  int b2;                 // vmcnt + barrier
  int r0_first, r0_step;  // next tile's k-step 0 fragment reads
};
constexpr W4Sched w4_sched(int var) {  // 256x256 tile (NM = 128, 16 reads, 16 pieces)
  switch (var) {
    case 1: return {1, 2, 36, 38, 4, 94, 96, 2};   // one DMA piece per 4 MFMAs
    case 2: return {1, 2, 40, 42, 3, 94, 96, 2};
    case 3: return {1, 1, 24, 26, 4, 94, 96, 2};   // k-step 1 reads back to back, earlier B1
    case 4: return {1, 2, 36, 38, 4, 102, 103, 1};
    case 5: return {1, 2, 36, 38, 5, 102, 103, 1};
    case 6: return {1, 1, 20, 22, 6, 102, 103, 1};
    case 7: return {1, 2, 36, 38, 4, 94, 95, 1};
    case 8: return {1, 1, 20, 22, 4, 94, 95, 1};
    case 9: return {1, 1, 20, 22, 5, 102, 103, 1};
    case 10: return {1, 2, 36, 38, 8, 102, 103, 1};   // production: one piece per 8 MFMAs, the last four issued by the next tile
    case 11: return {1, 2, 36, 38, 7, 102, 103, 1};
    case 12: return {1, 2, 36, 40, 6, 102, 103, 1};
    case 13: return {1, 1, 36, 38, 4, 102, 103, 1};
    case 14: return {1, 1, 20, 38, 4, 102, 103, 1};
    default: return {1, 2, 40, 42, 2, 94, 96, 2};
  }
}
// Same shape of schedule for any wave tile: reads every other MFMA, B1 five MFMAs after the last read, the DMA from
// there on with `d_step` MFMAs per piece, B2 ten MFMAs before the next tile's reads would run out of tile.
constexpr W4Sched w4_sched_for(int FM, int FN, int d_step) {
  const int NR = FM + FN, NM = 2 * FM * FN, b1 = 1 + 2 * (NR - 1) + 5, b2 = NM - NR - 10;
  return {1, 2, b1, b1 + 2, d_step, b2, b2 + 1, 1};
}

// NN B image for BN = 192: its chunk swizzle is nswz_bn<192> (hgemm_mfma.cuh); the fill below maps lanes to (k row, chunk)
// for rows that are not a power-of-two number of chunks.
template <typename C, int NLOADS>
struct NFillW {
  unsigned voff[NLOADS];
  __device__ __forceinline__ void init(int N, int wave, int lane) {
    constexpr int LPR = C::BN / 8;  // 16-byte chunks per k row (32 or 24)
#pragma unroll
    for (int i = 0; i < NLOADS; ++i) {
      const int L = (i * C::NW + wave) * 64 + lane;  // chunk index within the image, lane-linear
      const int krow = L / LPR, c = L % LPR;
      voff[i] = ((unsigned)krow * (unsigned)N + ((c ^ nswz_bn<C::BN>(krow)) << 3)) * 2u;
    }
  }
};
template <int BN>
__device__ __forceinline__ h8 read_nfrag_w(const char* img, int n0w, int lane, int kk) {
  const int i = lane & 15, g = lane >> 4;
  const int q = (n0w >> 3) + ((i & 3) >> 1);
  const int k_lo = kk * 32 + 8 * g + (i >> 2);
  const int k_hi = k_lo + 4;
  const char* p_lo = img + k_lo * (BN * 2) + ((q ^ nswz_bn<BN>(k_lo)) << 4) + ((i & 1) << 3);
  const char* p_hi = img + k_hi * (BN * 2) + ((q ^ nswz_bn<BN>(k_hi)) << 4) + ((i & 1) << 3);
  return h8_cat(lds_read_tr16(p_lo), lds_read_tr16(p_hi));
}

// Tile configuration without Cfg's BN in {128, 256} restriction.
template <int BM_, int BN_, int LAYOUT_>
struct W4Cfg {
  static constexpr int BM = BM_, BN = BN_, BK = 64, NW = 4, LAYOUT = LAYOUT_;
  static constexpr int WTM = BM / 2, WTN = BN / 2, FM = WTM / 16, FN = WTN / 16;
  static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE_BYTES = A_BYTES + B_BYTES, LDS_BYTES = 2 * STAGE_BYTES;
  static constexpr int A_LOADS = FM, B_LOADS = FN, NP = FM + FN;  // 1-KiB DMA pieces per wave per K tile
  static constexpr int NR = FM + FN, NM = 2 * FM * FN;            // fragment reads per k-step, MFMAs per K tile
  static_assert(((BM == 256 || BM == 192 || BM == 128) && (BN == 256 || BN == 192 || BN == 128) && BM + BN >= 384) ||
                    (BM == 160 && BN == 160),
                "wave tiles of 128, 96, 80 or 64 rows / columns; enough MFMAs per K tile to hang the schedule on");
};

constexpr int W4_TICKET_FLOATS = 1024;  // split-K workspace header: one arrival counter per output tile (<= 768 workgroups per launch), 4 KiB
// VAR: bits 0..3 = schedule number (256x256 only; other shapes take w4_sched_for), bit 4 = boustrophedon MFMA order
// (B fragments walked back and forth, so only ONE operand changes between consecutive MFMAs: ~1 % less power).
// ODD: K / 64 is odd (>= 7): one more whole tile between the loop over tile pairs and the two closing tiles.
template <int LAYOUT, int EPI = 2, int VAR = 0, int ABL = 0, int BM = 256, int BN = 256, bool ODD = false>
__global__ __launch_bounds__(256, 1) void hgemm_w4_kernel(const half_t* __restrict__ A, const half_t* __restrict__ B,
                                                          half_t* __restrict__ Cmat, int M, int N, int K, int tiles_m,
                                                          int tiles_n, int swizzle, int band, float* __restrict__ ws = nullptr) {
  using C = W4Cfg<BM, BN, LAYOUT>;
  constexpr bool SPLITK = EPI == 5 || EPI == 6;  // 5: partials only (+ the reduce launch, probe library); 6: partials + in-kernel fix-up by the last arriver
  constexpr int FM = C::FM, FN = C::FN, NR = C::NR, NP = C::NP, NM = C::NM;
  constexpr W4Sched S = (BM == 256 && BN == 256) ? w4_sched(VAR & 15) : w4_sched_for(FM, FN, (BM == 128 || BN == 128 || BM == 160) ? 2 : (BM == 192 && BN == 192) ? 4 : 5);
  constexpr bool SNAKE = (VAR >> 4) & 1;
  // The DMA of tile t+2 starts after B1 of tile t (position d_first) and may run on into tile t+1 (positions >= NM =
  // "late" pieces, issued by tile t+1's body before its own B1; they only have to land before B2 of tile t+1).
  static_assert(S.r0_first + (NR - 1) * S.r0_step < NM && S.r1_first + (NR - 1) * S.r1_step < S.b1 && S.b1 < S.d_first &&
                    S.b2 < S.r0_first && S.r0_first >= NM / 2 && (NP - 1) * S.d_step < NM && S.d_first + (NP - 1) * S.d_step - NM < S.d_first &&
                    S.d_first < NM && S.b2 >= S.d_first,
                "schedule: reads drained before B1, DMA after B1 and landed before the next B2, late pieces before early ones");
  constexpr int D_EARLY = (NM - 1 - S.d_first) / S.d_step + 1 > NP ? NP : (NM - 1 - S.d_first) / S.d_step + 1;  // pieces issued inside tile t
  // pieces of tile t+2 already issued when B2 is reached: they stay in flight across it
  constexpr int D_BEFORE_B2 = (S.b2 - S.d_first) / S.d_step + 1 > D_EARLY ? D_EARLY : (S.b2 - S.d_first) / S.d_step + 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // EPI 7 (round 5): PERSISTENT tile walk -- the launch has one workgroup per CU and workgroup b runs the tiles b, b + G, b + 2G, ...
  // of the same logical order (G = gridDim.x, a multiple of 8: the XCD of tile b + rG is the XCD of b). Before the C store of a tile the
  // first two K tiles of the NEXT tile are requested into the ring (every wave is past its last fragment read), and the epilogue stages through
  // 17 KiB BESIDE the ring in passes of one 16-row fragment row: the next tile's prologue latency, and the launch of a fresh workgroup, hide under the store.
  constexpr bool PERSIST = EPI == 7;
  const int n_tiles = tiles_m * tiles_n;
  int tm, tn;
  auto coords = [&](int vb, int& tm_, int& tn_) {
    if (swizzle & 4) {  // bit 2: operands larger than the Infinity Cache -- XCDs take the band walk in interleaved chunks (hgemm_mfma.cuh)
      tile_coords_interleaved(vb, PERSIST ? n_tiles : (int)gridDim.x, tiles_m, tiles_n, band, tm_, tn_);
    } else {
      tile_coords(vb, PERSIST ? n_tiles : (int)gridDim.x, tiles_m, tiles_n, swizzle & 1, band, tm_, tn_);  // bit 1 of `swizzle`: non-temporal C stores allowed (launcher)
    }
  };
  int vb = blockIdx.x;
  coords(vb, tm, tn);
  int m0 = tm * BM, n0 = tn * BN;

  // EPI 5 (split-K, hgemm_splitk.cuh): this workgroup multiplies K columns [blockIdx.y * K, (blockIdx.y + 1) * K) of a problem whose leading
  // dimension (the whole K) rides in bits 8.. of `swizzle` in units of 64; every other form: ld == K, k0 == 0 (compile-time: same code as before)
  KFill<C, FM> fa;
  KFill<C, FN> fbt;
  NFillW<C, FN> fbn;
  const char* a_src;
  const char* b_src;
  if constexpr (SPLITK) {
    const int ld = (swizzle >> 8) * 64;
    const size_t k0 = (size_t)blockIdx.y * K;
    fa.init(ld, wave, lane);
    if constexpr (LAYOUT == TN) fbt.init(ld, wave, lane);
    else fbn.init(N, wave, lane);
    a_src = reinterpret_cast<const char*>(A + (size_t)m0 * ld + k0);
    b_src = (LAYOUT == TN) ? reinterpret_cast<const char*>(B + (size_t)n0 * ld + k0) : reinterpret_cast<const char*>(B + k0 * N + n0);
  } else {
    fa.init(K, wave, lane);
    if constexpr (LAYOUT == TN) fbt.init(K, wave, lane);
    else fbn.init(N, wave, lane);
    a_src = reinterpret_cast<const char*>(A + (size_t)m0 * K);
    b_src = (LAYOUT == TN) ? reinterpret_cast<const char*>(B + (size_t)n0 * K) : reinterpret_cast<const char*>(B + n0);
  }
  const size_t a_step = 128;
  const size_t b_step = (LAYOUT == TN) ? (size_t)128 : (size_t)64 * N * 2;
  const unsigned lds0 = lds_addr_of(smem);
  auto voff_of = [&](int p) -> unsigned {
    return p < FM ? fa.voff[p < FM ? p : 0] : (LAYOUT == TN ? fbt.voff[(p - FM) < FN ? (p - FM) : 0] : fbn.voff[(p - FM) < FN ? (p - FM) : 0]);
  };
  // piece p (A pieces 0..FM-1, then B pieces) of the tile whose sources a_src / b_src currently point at; this wave's
  // 16 destinations are 4 KiB apart (A image, then B image, both lane-linear in units of 4 waves x 1 KiB)
  auto piece = [&](int p, unsigned img) { glds16_asm(p < FM ? a_src : b_src, voff_of(p), img + (unsigned)(p * 4 + wave) * 1024u); };
  // In-loop form: M0 walks the destinations -- two instructions per piece instead of six. M0 is ours for the whole K
  // loop: nothing else in it uses M0 (tests/test_no_spills.py checks the code object).
  const char* a_old = a_src;  // sources of the tile whose late pieces are still to be issued (one tile behind a_src)
  const char* b_old = b_src;
  auto piece_m0 = [&](int p, unsigned img, bool late, bool set_m0) {
    // ABL & 8 (probe; garbage results): every piece re-reads the SAME first KiB of A -- the instruction count and the LDS writes of the
    // real kernel, no L2 / fabric traffic: separates the issue cost of the LDS-DMA instructions from the cost of the bytes they move
    const unsigned voff = (ABL & 8) ? (unsigned)(lane * 16) : voff_of(p);
    const char* src = (ABL & 8) ? reinterpret_cast<const char*>(A) : late ? (p < FM ? a_old : b_old) : (p < FM ? a_src : b_src);
    if (set_m0)
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_add_u32 m0, m0, 0x1000"
                   :: "v"(voff), "s"(src), "s"(img + (unsigned)(p * 4 + wave) * 1024u) : "memory", "scc", "m0");
    else
      asm volatile("global_load_lds_dwordx4 %0, %1\n\ts_add_u32 m0, m0, 0x1000" :: "v"(voff), "s"(src) : "memory", "scc", "m0");
  };
  auto advance = [&]() {
    a_old = a_src;
    b_old = b_src;
    a_src += a_step;
    b_src += b_step;
  };

  f4 acc[FM][FN];
  h8 af[2][FM], bf[2][FN];
  bool prefetched = false;  // PERSIST: this tile's first two K tiles were requested before the previous tile's C store
  for (;;) {  // one pass unless PERSIST
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
  // The MFMAs below are inline asm, invisible to hipcc's hazard pass: left alone it sinks each tile's zero-fill
  // (v_accvgpr_mov) to just before the tile's first MFMA, closer than the VALU-write -> MFMA-SrcC wait states allow (seen:
  // NaNs from stale registers). Pin all tiles into their AGPRs HERE, then pad.
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) asm volatile("" : "+a"(acc[i][j]));
  asm volatile("s_nop 7");

  // fragment read op r of k-step kk: r = 0 -> A fragment 0, 1..FN -> B fragments, FN+1.. -> A fragments 1..FM-1
  // (the MFMA order is A-fragment-major: the first FN MFMAs of a k-step need A0 and every B fragment).
  auto read_op = [&](const char* img, int kk, int r) {
    const char* bimg = img + C::A_BYTES;
    if (r == 0 || r > FN) {
      const int i = r == 0 ? 0 : r - FN;
      af[kk][i] = read_kfrag<64>(img, wm * C::WTM + i * 16 + (lane & 15), lane, kk);
    } else {
      const int j = r - 1;
      if constexpr (LAYOUT == TN) bf[kk][j] = read_kfrag<64>(bimg, wn * C::WTN + j * 16 + (lane & 15), lane, kk);
      else bf[kk][j] = read_nfrag_w<BN>(bimg, wn * C::WTN + j * 16, lane, kk);
    }
  };
#define W4_PIN() __builtin_amdgcn_sched_barrier(0)
#define W4_BARRIER()                 \
  do {                               \
    __builtin_amdgcn_s_barrier();    \
    asm volatile("" ::: "memory");   \
  } while (0)

  // One K tile t. DMA: tile t+2 exists (issue its early pieces into this tile's buffer); LATE: tile t+1's late pieces
  // are still to be issued (into the other buffer); NEXT: tile t+1 exists (wait for it, read its k-step 0).
  auto tile = [&](auto dma_c, auto late_c, auto next_c, const char* img, const char* nimg, unsigned img_lds, unsigned nimg_lds) {
    constexpr bool DMA = decltype(dma_c)::value, LATE = decltype(late_c)::value, NEXT = decltype(next_c)::value;
    static_for<NM>([&](auto nc) {
      constexpr int n = decltype(nc)::value;
      constexpr int kk = n / (FM * FN), idx = n % (FM * FN), i = idx / FN, lo0 = idx % FN;
      constexpr int j = (SNAKE && (i & 1)) ? FN - 1 - lo0 : lo0;
      // AGPR-tied accumulator: left to the builtin, hipcc gives C and D different registers and rotates the tiles
      // through ~370 v_accvgpr_mov / read / write per K tile
      asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(bf[kk][j]), "v"(af[kk][i]));
      constexpr bool R1 = n >= S.r1_first && (n - S.r1_first) % S.r1_step == 0 && (n - S.r1_first) / S.r1_step < NR;
      constexpr bool DP = DMA && n >= S.d_first && (n - S.d_first) % S.d_step == 0 && (n - S.d_first) / S.d_step < NP;
      constexpr int gl = n + NM;  // position of a late piece on the previous tile's clock
      constexpr bool LP = LATE && (gl - S.d_first) % S.d_step == 0 && (gl - S.d_first) / S.d_step < NP && (gl - S.d_first) / S.d_step >= D_EARLY;
      constexpr bool R0 = NEXT && n >= S.r0_first && (n - S.r0_first) % S.r0_step == 0 && (n - S.r0_first) / S.r0_step < NR;
      // ABL (probe library only; results are garbage by design): 1 = no fragment reads, 2 = no DMA, 4 = no barriers
      if constexpr (R1 && !(ABL & 1)) read_op(img, 1, (n - S.r1_first) / S.r1_step);
      if constexpr (n == S.b1) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (!(ABL & 4)) W4_BARRIER();
      }
      if constexpr (LP && !(ABL & 2)) piece_m0((gl - S.d_first) / S.d_step, nimg_lds, true, (gl - S.d_first) / S.d_step == D_EARLY);
      if constexpr (DP && !(ABL & 2)) piece_m0((n - S.d_first) / S.d_step, img_lds, false, n == S.d_first);
      if constexpr (NEXT && n == S.b2) {
        if constexpr (DMA && !(ABL & 2)) wait_vmcnt<D_BEFORE_B2>();
        else wait_vmcnt<0>();
        if constexpr (!(ABL & 4)) W4_BARRIER();
      }
      if constexpr (R0 && !(ABL & 1)) read_op(nimg, 0, (n - S.r0_first) / S.r0_step);
      if constexpr (R1 || DP || LP || R0 || n == S.b1 || (NEXT && n == S.b2)) W4_PIN();
    });
    if constexpr (DMA) advance();
  };

  const int nt = K / 64;
  // prologue: tiles 0 and 1 in flight, tile 0 landed, its k-step 0 fragments in registers
  auto prologue_requests = [&]() {
#pragma unroll
    for (int p = 0; p < NP; ++p) piece(p, lds0);
    advance();
#pragma unroll
    for (int p = 0; p < NP; ++p) piece(p, lds0 + C::STAGE_BYTES);
    advance();
  };
  constexpr int C_STORES = FM * FN / 2;  // global stores per lane of one C epilogue (FM x FN x 16 rows x 32 B / 1 KiB per wave-instruction ... / 64 lanes)
  if (PERSIST && prefetched) {
    // both K tiles were requested BEFORE the previous tile's C stores (vmcnt completes in issue order on gfx9, loads and stores alike): once at most
    // the C_STORES younger stores are outstanding, every piece has landed -- the stores themselves are not waited for
    static_assert(C_STORES < 64, "vmcnt is a 6-bit field");
    wait_vmcnt<C_STORES>();
  } else {
    prologue_requests();
    wait_vmcnt<NP>();
  }
  W4_BARRIER();
#pragma unroll
  for (int r = 0; r < NR; ++r) read_op(smem, 0, r);
  W4_PIN();

  // nt even and >= 6, or (ODD) odd and >= 7 (launcher): two peeled tiles, a do-while over tile PAIRS (ring buffer =
  // compile-time constant, every LDS address loop-invariant), [ODD: one more whole tile,] two closing tiles; no
  // control-flow merge that would need accumulator copies.
  const char* img0 = smem;
  const char* img1 = smem + C::STAGE_BYTES;
  const unsigned lds1 = lds0 + C::STAGE_BYTES;
  constexpr std::true_type Y{};
  constexpr std::false_type NO{};
  tile(Y, NO, Y, img0, img1, lds0, lds1);  // tile 0: tile 1 was issued whole by the prologue
  tile(Y, Y, Y, img1, img0, lds1, lds0);
  int t = 2;
  do {
    tile(Y, Y, Y, img0, img1, lds0, lds1);
    tile(Y, Y, Y, img1, img0, lds1, lds0);
    t += 2;
  } while (t + (ODD ? 3 : 2) < nt);
  if constexpr (ODD) {
    tile(Y, Y, Y, img0, img1, lds0, lds1);
    tile(NO, Y, Y, img1, img0, lds1, lds0);
    tile(NO, NO, NO, img0, img1, lds0, lds1);
  } else {
    tile(NO, Y, Y, img0, img1, lds0, lds1);
    tile(NO, NO, NO, img1, img0, lds1, lds0);
  }
#undef W4_PIN
#undef W4_BARRIER
  // Last MFMA results -> v_accvgpr_read: same blindness of the hazard pass. Pad, then re-define every tile AFTER the pad
  // (asm volatile statements keep their order), so no read of an accumulator can be scheduled above it.
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) asm volatile("" : "+a"(acc[i][j]));
  if constexpr (SPLITK) {
    // split-K partial: the fp32 accumulators as they lie in the registers -- the fp32 workspace is [split][tile][wave][fragment][lane] x 16 bytes,
    // one store instruction = 1 KiB contiguous. EPI 5: `Cmat` IS the workspace and the reduce kernel (hgemm_splitk.cuh) sums the splits.
    // EPI 6 (round 5, ONE launch): `ws` = [1024 tickets][partials]; the workgroup that takes the LAST ticket of its tile sums the S partials in
    // ascending split order (its own included: re-read from its own L2 -- the same order, hence the same bits, as the reduce kernel), rounds once
    // and writes C through the ordinary LDS-staged epilogue. Tickets reset themselves: no memset between launches.
    const int tile = tm * tiles_n + tn, tiles = tiles_m * tiles_n;
    float* part = (EPI == 6 ? ws + W4_TICKET_FLOATS : reinterpret_cast<float*>(Cmat));
    float* dst = part + ((size_t)blockIdx.y * tiles + tile) * (BM * BN) + wave * (C::WTM * C::WTN) + lane * 4;
    if constexpr (EPI == 5) {
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) *reinterpret_cast<f4*>(dst + (i * FN + j) * 256) = acc[i][j];
    }
    if constexpr (EPI == 6) {
      // Visibility across the 8 XCD L2s WITHOUT fences: the partials are stored and loaded at AGENT scope (`sc1`: write-through to / coherent
      // read from the memory side) and the ticket is an agent-scope atomic. A __threadfence() pair here -- what the first form of this epilogue
      // did -- writes back AND INVALIDATES the whole L2 of the XCD while the other workgroups are still streaming A / B panels through it:
      // measured +17-45 us per launch (profiles/r05_hgemm_splitk_fused_probe.log, first block).
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst + (i * FN + j) * 256), "v"(acc[i][j]) : "memory");  // (s_nop: the data
          // registers are a temporary copy of AGPRs that hipcc re-uses for the next copy at once -- a store of more than 8 bytes needs one wait state
          // before its data VGPRs are overwritten, and the hazard pass does not see into inline asm: without it the partials were corrupted)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's partial has reached the memory side
      const int S = gridDim.y;
      int* last = reinterpret_cast<int*>(smem + C::LDS_BYTES - 16);  // beyond the four staging regions of the epilogue
      __syncthreads();
      if (threadIdx.x == 0) {
        unsigned* tk = reinterpret_cast<unsigned*>(ws) + tile;
        const unsigned old = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int l = old == (unsigned)(S - 1);
        if (l) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // all S have arrived: the ticket is ready for the next launch
        *last = l;
      }
      __syncthreads();
      if (*last == 0) return;
      const float* src = part + (size_t)tile * (BM * BN) + wave * (C::WTM * C::WTN) + lane * 4;
      const size_t split_stride = (size_t)tiles * (BM * BN);
      // split-major in chunks of 2 fragment rows: 2 x FN agent-scope loads of one split in flight per lane, every element summed in ascending
      // split order (this workgroup's own partial included, re-read: the same order -- hence the same bits -- as the reduce kernel)
#pragma unroll
      for (int i0 = 0; i0 < FM; i0 += 2) {
        for (int ks = 0; ks < S; ++ks) {
          const float* p = src + (size_t)ks * split_stride;
          f4 t[2][FN];
#pragma unroll
          for (int i = i0; i < i0 + 2 && i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(t[i - i0][j]) : "v"(p + (i * FN + j) * 256) : "memory");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
          for (int i = i0; i < i0 + 2 && i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) {
              asm volatile("" : "+v"(t[i - i0][j]));  // (volatile: ordered behind the wait; the add below depends on it)
              acc[i][j] = ks == 0 ? t[i - i0][j] : acc[i][j] + t[i - i0][j];
            }
        }
      }
      store_wide_tile_via_lds<FM, FN, 0>(Cmat, N, m0 + wm * C::WTM, n0 + wn * C::WTN, lane, smem + wave * (64 * (FN * 32 + 16)), acc, 0);
    }
  } else if constexpr (PERSIST) {
    // B1 of the last K tile: every wave is past its last fragment read, no DMA is in flight -> both ring buffers are free for the NEXT tile's prologue,
    // requested here, before this tile's C leaves through the staging rows beside the ring
    const int nvb = vb + (int)gridDim.x;
    const bool more = nvb < n_tiles;
    const int row0 = m0 + wm * C::WTM, col0 = n0 + wn * C::WTN;
    if (more) {
      coords(nvb, tm, tn);
      m0 = tm * BM, n0 = tn * BN;
      a_src = reinterpret_cast<const char*>(A + (size_t)m0 * K);
      b_src = (LAYOUT == TN) ? reinterpret_cast<const char*>(B + (size_t)n0 * K) : reinterpret_cast<const char*>(B + n0);
      prologue_requests();
    }
    // (lane id recomputed here: the epilogue's lane-derived addresses are invariant over the tile loop and would otherwise be hoisted above it and
    // carried -- spilled -- through the K loop, whose register file is full)
    store_wide_tile_via_lds_rows16<FM, FN>(Cmat, N, row0, col0, cln_fresh_lane(), smem + C::LDS_BYTES + wave * (16 * (FN * 32 + 16)), acc, (swizzle >> 1) & 1);
    if (!more) break;
    vb = nvb, prefetched = true;
    continue;
  } else if constexpr (EPI >= 2) {  // EPI 3 / 4 (probe library): the same epilogue with non-temporal / write-through C stores
    // B1 of the last tile: every wave is past its last fragment read; no DMA is in flight
    store_wide_tile_via_lds<FM, FN, EPI - 2>(Cmat, N, m0 + wm * C::WTM, n0 + wn * C::WTN, lane, smem + wave * (64 * (FN * 32 + 16)), acc, (swizzle >> 1) & 1);
  } else {  // measurement-only variant: keep the accumulators live, store (almost) nothing
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 123.456f) Cmat[(size_t)m0 * N + n0] = (half_t)s;
  }
  break;
  }  // for (;;)
}

// K the kernel's peeled structure covers: whole 64-wide tiles, >= 6 of them when their number is even, >= 7 when odd
inline bool w4_k_ok(int K) { return K % 64 == 0 && K >= ((K / 64) & 1 ? 448 : 384); }
// Non-temporal C stores keep the output from displacing the A / B panels other workgroups still read from L2 / MALL: +1.3-1.5 % at 4096^3
// (100.7 MB of operands), +3.4-3.6 % at 8192^3 (profiles/r03_hgemm_c_store_probe.log). A SMALL problem's C is what the next kernel reads
// (chained GEMMs) and fits the 256 MiB Infinity Cache next to A and B: there the streaming hint would push out exactly that data
// (ADVICE r3), so the epilogue takes the hint only from the measured size on: footprint = 2 (MK + KN + MN) bytes >= 96 MB.
constexpr long long W4_NT_FOOTPRINT = 96LL << 20;
inline int w4_nt_ok(int M, int N, int K) { return 2LL * ((long long)M * K + (long long)K * N + (long long)M * N) >= W4_NT_FOOTPRINT ? 1 : 0; }
// The interleaved-chunk walk (tile_coords_interleaved, hgemm_mfma.cuh) when block swizzle is requested and A + B are at least TWICE the 256 MiB
// Infinity Cache: 12544^3 1378-1387 -> 1422-1432 TF (+3-4 %), 15360^3 1354-1360 -> 1412-1423 (+4-5 %), 16384^3 1456-1460 -> 1485-1495 (+2 %), NN;
// 10240^3 (420 MB of operands) -1.5 ... 0 %, 8192^3 0 % -- profiles/r04_hgemm_block_walk_probe.log. Bit-identical results.
constexpr long long W4_INTERLEAVED_OPERANDS = 512LL << 20;
inline int w4_interleaved_walk(int M, int N, int K, int swizzle, int tiles) {
  if (!swizzle || tiles < 512) return 0;
  return (tiles >= 1024 && 2LL * ((long long)M * K + (long long)K * N) >= W4_INTERLEAVED_OPERANDS) ? 1 : 0;
}
// kernel argument `swizzle`: bit 0 block swizzle, bit 1 non-temporal C stores, bit 2 the interleaved-chunk walk
inline int w4_swizzle_arg(int M, int N, int K, int swizzle, int tiles) { return (swizzle ? 1 : 0) | (w4_nt_ok(M, N, K) << 1) | (w4_interleaved_walk(M, N, K, swizzle, tiles) << 2); }
inline int w4_grid(int, int, int, int, int tiles_m, int tiles_n) { return tiles_m * tiles_n; }

template <int LAYOUT, int EPI = 2, int VAR = 0, int ABL = 0, int BM = 256, int BN = 256>
int launch_w4(const void* a, const void* b, void* c, int M, int N, int K, int swizzle, int swizzle_stride,
              hipStream_t stream) {
  using C = W4Cfg<BM, BN, LAYOUT>;
  // the odd-tile-count form exists for the production schedule only (the probe variants keep K % 128 == 0)
  constexpr bool HAS_ODD = ABL == 0 && EPI >= 2 && (VAR == 26 || BM != 256 || BN != 256);
  const bool odd = (K / 64) & 1;
  if (M % BM || N % BN || !w4_k_ok(K) || (odd && !HAS_ODD)) return CLN_ERR_UNSUPPORTED;
  const int tiles_m = M / BM, tiles_n = N / BN;
  int band = (swizzle && swizzle_stride >= BN) ? swizzle_stride / BN : tiles_n;
  if constexpr (HAS_ODD) {
    if (odd) {
      static cln_lds_attr lds_attr_odd;  // per device, thread-safe (common.h)
      if (cln_ensure_lds(lds_attr_odd, reinterpret_cast<const void*>(&hgemm_w4_kernel<LAYOUT, EPI, VAR, ABL, BM, BN, true>), C::LDS_BYTES) != CLN_OK)
        return CLN_ERR_LAUNCH;
      CLN_LAUNCH((hgemm_w4_kernel<LAYOUT, EPI, VAR, ABL, BM, BN, true>), dim3(w4_grid(M, N, K, swizzle, tiles_m, tiles_n)), dim3(256), C::LDS_BYTES, stream,
                 (const half_t*)a, (const half_t*)b, (half_t*)c, M, N, K, tiles_m, tiles_n, w4_swizzle_arg(M, N, K, swizzle, tiles_m * tiles_n), band);
      return cln_check_launch();
    }
  }
  static cln_lds_attr lds_attr;  // per device, thread-safe (common.h)
  if (cln_ensure_lds(lds_attr, reinterpret_cast<const void*>(&hgemm_w4_kernel<LAYOUT, EPI, VAR, ABL, BM, BN>), C::LDS_BYTES) != CLN_OK)
    return CLN_ERR_LAUNCH;
  CLN_LAUNCH((hgemm_w4_kernel<LAYOUT, EPI, VAR, ABL, BM, BN>), dim3(w4_grid(M, N, K, swizzle, tiles_m, tiles_n)), dim3(256), C::LDS_BYTES, stream,
             (const half_t*)a, (const half_t*)b, (half_t*)c, M, N, K, tiles_m, tiles_n, w4_swizzle_arg(M, N, K, swizzle, tiles_m * tiles_n), band);
  return cln_check_launch();
}

// Persistent form (EPI 7, see the kernel): one workgroup per CU walks the tiles b, b + G, ... -- for launches of MORE tiles than CUs.
inline int w4_cu_count() {
  static const int n = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    (void)hipGetLastError();
    return cus > 0 ? cus & ~7 : 256;  // a multiple of 8: tile b + r G stays on the XCD of tile b
  }();
  return n;
}
template <int LAYOUT, int VAR = 26>
int launch_w4_persist(const void* a, const void* b, void* c, int M, int N, int K, int swizzle, int swizzle_stride, hipStream_t stream) {
  constexpr int BM = 256, BN = 256, EPI = 7;
  using C = W4Cfg<BM, BN, LAYOUT>;
  constexpr int LDS = C::LDS_BYTES + 4 * 16 * (C::FN * 32 + 16);
  static_assert(LDS <= 160 * 1024, "ring + epilogue staging rows");
  const bool odd = (K / 64) & 1;
  if (M % BM || N % BN || !w4_k_ok(K)) return CLN_ERR_UNSUPPORTED;
  const int tiles_m = M / BM, tiles_n = N / BN, tiles = tiles_m * tiles_n;
  const int band = (swizzle && swizzle_stride >= BN) ? swizzle_stride / BN : tiles_n;
  const int grid = tiles < w4_cu_count() ? tiles : w4_cu_count();
  if (odd) {
    static cln_lds_attr lds_attr_odd;
    if (cln_ensure_lds(lds_attr_odd, reinterpret_cast<const void*>(&hgemm_w4_kernel<LAYOUT, EPI, VAR, 0, BM, BN, true>), LDS) != CLN_OK) return CLN_ERR_LAUNCH;
    CLN_LAUNCH((hgemm_w4_kernel<LAYOUT, EPI, VAR, 0, BM, BN, true>), dim3(grid), dim3(256), LDS, stream, (const half_t*)a, (const half_t*)b, (half_t*)c, M, N, K,
               tiles_m, tiles_n, w4_swizzle_arg(M, N, K, swizzle, tiles), band);
  } else {
    static cln_lds_attr lds_attr;
    if (cln_ensure_lds(lds_attr, reinterpret_cast<const void*>(&hgemm_w4_kernel<LAYOUT, EPI, VAR, 0, BM, BN>), LDS) != CLN_OK) return CLN_ERR_LAUNCH;
    CLN_LAUNCH((hgemm_w4_kernel<LAYOUT, EPI, VAR, 0, BM, BN>), dim3(grid), dim3(256), LDS, stream, (const half_t*)a, (const half_t*)b, (half_t*)c, M, N, K,
               tiles_m, tiles_n, w4_swizzle_arg(M, N, K, swizzle, tiles), band);
  }
  return cln_check_launch();
}

}  // namespace hgemm
