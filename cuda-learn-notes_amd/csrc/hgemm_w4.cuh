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

// Epilogue through LDS for a 128-column wave tile: [64 rows][272 B] wave-private region, two passes of 64 rows; a lane
// streams 16 bytes, one store instruction = 4 rows x 256 contiguous bytes.
template <int FM, int FN>
__device__ __forceinline__ void store_wide_tile_via_lds(half_t* Cmat, int N, int row0, int col0, int lane, char* wave_lds,
                                                        const f4 (&acc)[FM][FN]) {
  static_assert(FN == 8 && FM % 4 == 0, "128-column wave tile, passes of 64 rows");
  constexpr int RS = 272;
#pragma unroll
  for (int h0 = 0; h0 < FM; h0 += 4) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const f4 v = acc[h0 + i][j];
        h4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
        *reinterpret_cast<h4*>(wave_lds + (i * 16 + (lane & 15)) * RS + (j * 16 + 4 * (lane >> 4)) * 2) = o;
      }
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int r = it * 4 + (lane >> 4);
      const u4 v = *reinterpret_cast<const u4*>(wave_lds + r * RS + (lane & 15) * 16);
      *reinterpret_cast<u4*>(Cmat + (size_t)(row0 + h0 * 16 + r) * N + col0 + (lane & 15) * 8) = v;
    }
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

// Schedule positions (MFMA index within the 128 of one K tile); template so that variants can be probed.
struct W4Sched {
  int r1_first, r1_step;  // k-step 1 fragment reads: op r at n = r1_first + r1_step * r
  int b1;                 // lgkmcnt(0) + barrier
  int d_first, d_step;    // DMA piece p at n = d_first + d_step * p
  int b2;                 // vmcnt + barrier
  int r0_first, r0_step;  // next tile's k-step 0 fragment reads
};
constexpr W4Sched w4_sched(int var) {
  switch (var) {
    case 1: return {1, 2, 36, 38, 4, 94, 96, 2};   // one DMA piece per 4 MFMAs (4 waves x 16 cycles of address path each)
    case 2: return {1, 2, 40, 42, 3, 94, 96, 2};
    case 3: return {1, 1, 24, 26, 4, 94, 96, 2};   // k-step 1 reads back to back, earlier B1
    case 4: return {1, 2, 36, 38, 4, 102, 103, 1};
    case 5: return {1, 2, 36, 38, 5, 102, 103, 1};
    case 6: return {1, 1, 20, 22, 6, 102, 103, 1};
    case 7: return {1, 2, 36, 38, 4, 94, 95, 1};
    case 8: return {1, 1, 20, 22, 4, 94, 95, 1};
    case 9: return {1, 1, 20, 22, 5, 102, 103, 1};
    case 10: return {1, 2, 36, 38, 8, 102, 103, 1};   // one piece per 8 MFMAs: the last five are issued by the next tile
    case 11: return {1, 2, 36, 38, 7, 102, 103, 1};
    case 12: return {1, 2, 36, 40, 6, 102, 103, 1};
    case 13: return {1, 1, 36, 38, 4, 102, 103, 1};   // debugging: back-to-back k-step 1 reads, late B1
    case 14: return {1, 1, 20, 38, 4, 102, 103, 1};   // debugging: early B1, late DMA
    case 15: return {1, 2, 36, 38, 4, 102, 103, 1};   // = 4
    default: return {1, 2, 40, 42, 2, 94, 96, 2};
  }
}

template <int LAYOUT, int EPI = 2, int VAR = 0, int ABL = 0>
__global__ __launch_bounds__(256, 1) void hgemm_w4_kernel(const half_t* __restrict__ A, const half_t* __restrict__ B,
                                                          half_t* __restrict__ Cmat, int M, int N, int K, int tiles_m,
                                                          int tiles_n, int swizzle, int band) {
  using C = Cfg<256, 256, 64, 2, 2, 2, LAYOUT>;
  static_assert(C::A_LOADS == 8 && C::B_LOADS == 8, "16 DMA pieces per wave per K tile");
  constexpr W4Sched S = w4_sched(VAR & 15);
  constexpr int ORDER = VAR >> 4;  // MFMA order inside a k-step: 0 A-fragment-major, 1 same with B walked boustrophedon, 2 B-major, 3 B-major boustrophedon
  // pieces of tile t+2 already issued when B2 is reached: they stay in flight across it
  // The DMA of tile t+2 starts after B1 of tile t (position d_first) and may run on into tile t+1 (positions >= 128 =
  // "late" pieces, issued by tile t+1's body before its own B1 ... they only have to land before B2 of tile t+1).
  static_assert(S.r0_first + 15 * S.r0_step < 128 && S.r1_first + 15 * S.r1_step < S.b1 && S.b1 < S.d_first &&
                    S.b2 < S.r0_first && S.r0_first >= 64 && 15 * S.d_step < 128 && S.d_first + 15 * S.d_step - 128 < S.b2,
                "schedule: reads drained before B1, DMA after B1 and landed before the next B2, late pieces before early ones");
  constexpr int D_BEFORE_B2 = S.b2 < S.d_first ? 0 : ((S.b2 - S.d_first) / S.d_step + 1 > 16 ? 16 : (S.b2 - S.d_first) / S.d_step + 1);
  constexpr int D_EARLY = (127 - S.d_first) / S.d_step + 1 > 16 ? 16 : (127 - S.d_first) / S.d_step + 1;  // pieces issued inside tile t
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int tm, tn;
  tile_coords(blockIdx.x, gridDim.x, tiles_m, tiles_n, swizzle, band, tm, tn);
  const int m0 = tm * 256, n0 = tn * 256;

  KFill<C, 8> fa;
  fa.init(K, wave, lane);
  KFill<C, 8> fbt;
  NFill<C, 8> fbn;
  if constexpr (LAYOUT == TN) fbt.init(K, wave, lane);
  else fbn.init(N, wave, lane);
  const char* a_src = reinterpret_cast<const char*>(A + (size_t)m0 * K);
  const char* b_src = (LAYOUT == TN) ? reinterpret_cast<const char*>(B + (size_t)n0 * K)
                                     : reinterpret_cast<const char*>(B + n0);
  const size_t a_step = 128;
  const size_t b_step = (LAYOUT == TN) ? (size_t)128 : (size_t)64 * N * 2;
  const unsigned lds0 = lds_addr_of(smem);
  // piece p of the tile whose sources a_src / b_src currently point at
  auto piece = [&](int p, unsigned img) {
    if (p < 8) {
      glds16_asm(a_src, fa.voff[p & 7], img + (unsigned)(p * 4 + wave) * 1024u);
    } else {
      const int q = p - 8;
      if constexpr (LAYOUT == TN) glds16_asm(b_src, fbt.voff[q & 7], img + C::A_BYTES + (unsigned)(q * 4 + wave) * 1024u);
      else glds16_asm(b_src, fbn.voff[q & 7], img + C::A_BYTES + (unsigned)(q * 4 + wave) * 1024u);
    }
  };
  // In-loop form: M0 walks the 16 destinations of this wave (4 KiB apart, A image then B image) -- two instructions per
  // piece instead of six. M0 is ours for the whole K loop: nothing else in it uses M0 (tests/test_no_spills.py checks
  // the code object for foreign M0 writes).
  const char* a_old = a_src;  // sources of the tile whose late pieces are still to be issued (one tile behind a_src)
  const char* b_old = b_src;
  auto piece_m0 = [&](int p, unsigned img, bool late, bool set_m0) {
    const unsigned voff = p < 8 ? fa.voff[p & 7] : (LAYOUT == TN ? fbt.voff[(p - 8) & 7] : fbn.voff[(p - 8) & 7]);
    const char* src = late ? (p < 8 ? a_old : b_old) : (p < 8 ? a_src : b_src);
    img += (unsigned)p * 4096u;
    if (set_m0)
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_add_u32 m0, m0, 0x1000"
                   :: "v"(voff), "s"(src), "s"(img + (unsigned)wave * 1024u) : "memory", "scc");  // img already advanced to piece p
    else
      asm volatile("global_load_lds_dwordx4 %0, %1\n\ts_add_u32 m0, m0, 0x1000" :: "v"(voff), "s"(src) : "memory", "scc");
  };
  auto advance = [&]() {
    a_old = a_src;
    b_old = b_src;
    a_src += a_step;
    b_src += b_step;
  };

  f4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
  // The MFMAs below are inline asm, invisible to hipcc's hazard pass: left alone it sinks each tile's zero-fill
  // (v_accvgpr_mov) to just before the tile's first MFMA, closer than the VALU-write -> MFMA-SrcC wait states allow (seen:
  // NaNs from stale registers). Pin all 64 tiles into their AGPRs HERE, then pad.
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("" : "+a"(acc[i][j]));
  asm volatile("s_nop 7");
  h8 af[2][8], bf[2][8];

  // fragment read op r of k-step kk: r = 0 -> A fragment 0, 1..8 -> B fragments 0..7, 9..15 -> A fragments 1..7
  // (the MFMA order is A-fragment-major: the first eight MFMAs of a k-step need A0 and every B fragment).
  auto read_op = [&](const char* img, int kk, int r) {
    const char* bimg = img + C::A_BYTES;
    const bool is_a = (ORDER < 2) ? (r == 0 || r >= 9) : (r >= 1 && r <= 8);
    if (is_a) {
      const int i = (ORDER < 2) ? (r == 0 ? 0 : r - 8) : r - 1;
      af[kk][i] = read_kfrag<64>(img, wm * 128 + i * 16 + (lane & 15), lane, kk);
    } else {
      const int j = (ORDER < 2) ? r - 1 : (r == 0 ? 0 : r - 8);
      if constexpr (LAYOUT == TN) bf[kk][j] = read_kfrag<64>(bimg, wn * 128 + j * 16 + (lane & 15), lane, kk);
      else bf[kk][j] = read_nfrag<256>(bimg, wn * 128 + j * 16, lane, kk);
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
    static_for<128>([&](auto nc) {
      constexpr int n = decltype(nc)::value;
      constexpr int kk = n >> 6, hi = (n & 63) >> 3, lo0 = n & 7, lo = ((ORDER & 1) && (hi & 1)) ? 7 - lo0 : lo0;
      constexpr int i = ORDER < 2 ? hi : lo, j = ORDER < 2 ? lo : hi;
      // AGPR-tied accumulator: left to the builtin, hipcc gives C and D different registers and rotates the 64 tiles
      // through ~370 v_accvgpr_mov / read / write per K tile
      asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(bf[kk][j]), "v"(af[kk][i]));
      constexpr bool R1 = n >= S.r1_first && (n - S.r1_first) % S.r1_step == 0 && (n - S.r1_first) / S.r1_step < 16;
      constexpr bool DP = DMA && n >= S.d_first && (n - S.d_first) % S.d_step == 0 && (n - S.d_first) / S.d_step < 16;
      constexpr int gl = n + 128;  // position of a late piece on the previous tile's clock
      constexpr bool LP = LATE && (gl - S.d_first) % S.d_step == 0 && (gl - S.d_first) / S.d_step < 16 && (gl - S.d_first) / S.d_step >= D_EARLY;
      constexpr bool R0 = NEXT && n >= S.r0_first && (n - S.r0_first) % S.r0_step == 0 && (n - S.r0_first) / S.r0_step < 16;
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
#pragma unroll
  for (int p = 0; p < 16; ++p) piece(p, lds0);
  advance();
#pragma unroll
  for (int p = 0; p < 16; ++p) piece(p, lds0 + C::STAGE_BYTES);
  advance();
  wait_vmcnt<16>();
  W4_BARRIER();
#pragma unroll
  for (int r = 0; r < 16; ++r) read_op(smem, 0, r);
  W4_PIN();

  // nt even and >= 6 (launcher): two peeled tiles, a do-while over tile PAIRS (ring buffer = compile-time constant,
  // every LDS address loop-invariant), two peeled tiles; no control-flow merge that would need accumulator copies.
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
  } while (t + 2 < nt);
  tile(NO, Y, Y, img0, img1, lds0, lds1);
  tile(NO, NO, NO, img1, img0, lds1, lds0);
#undef W4_PIN
#undef W4_BARRIER
  // Last MFMA results -> v_accvgpr_read: same blindness of the hazard pass. Pad, then re-define every tile AFTER the pad
  // (asm volatile statements keep their order), so no read of an accumulator can be scheduled above it.
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("" : "+a"(acc[i][j]));
  if constexpr (EPI == 2) {
    // B1 of the last tile: every wave is past its last fragment read; no DMA is in flight
    store_wide_tile_via_lds<8, 8>(Cmat, N, m0 + wm * 128, n0 + wn * 128, lane, smem + wave * (64 * 272), acc);
  } else if constexpr (EPI == 0) {
    store_tile<C>(Cmat, N, m0, n0, wm, wn, lane, acc);
  } else {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 123.456f) Cmat[(size_t)m0 * N + n0] = (half_t)s;
  }
}

template <int LAYOUT, int EPI = 2, int VAR = 0, int ABL = 0>
int launch_w4(const void* a, const void* b, void* c, int M, int N, int K, int swizzle, int swizzle_stride,
              hipStream_t stream) {
  using C = Cfg<256, 256, 64, 2, 2, 2, LAYOUT>;
  if (M % 256 || N % 256 || K % 128 || K < 384) return CLN_ERR_UNSUPPORTED;
  static cln_lds_attr lds_attr;  // per device, thread-safe (common.h)
  if (cln_ensure_lds(lds_attr, reinterpret_cast<const void*>(&hgemm_w4_kernel<LAYOUT, EPI, VAR, ABL>), C::LDS_BYTES) != CLN_OK)
    return CLN_ERR_LAUNCH;
  const int tiles_m = M / 256, tiles_n = N / 256;
  int band = (swizzle && swizzle_stride >= 256) ? swizzle_stride / 256 : tiles_n;
  CLN_LAUNCH((hgemm_w4_kernel<LAYOUT, EPI, VAR, ABL>), dim3(tiles_m * tiles_n), dim3(256), C::LDS_BYTES, stream,
             (const half_t*)a, (const half_t*)b, (half_t*)c, M, N, K, tiles_m, tiles_n, swizzle ? 1 : 0, band);
  return cln_check_launch();
}

}  // namespace hgemm
