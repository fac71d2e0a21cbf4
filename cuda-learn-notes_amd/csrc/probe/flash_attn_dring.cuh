// FlashAttention-2 forward for head dims 640 / 768 / 1024 (round 3): FOUR waves split the head dim of one 32-row query
// group, K / V stream through two-slot rings of 16-key tiles. Reference rungs: the fine-grained tiling kernels, whose
// head-dim switch goes up to d = 1024 (kernels/flash-attn/mma/basic/flash_attn_mma_tiling_qkv.cu:70, :852-870;
// flash_attn_mma.py:436-506).
//
// Why a new kernel (the predecessor is probe/flash_attn_dwide.cuh, 0.25-0.27 of the fp16 MFMA peak). At these head
// dims the register file holds 64 query rows per CU (O^T alone is 64 x 1024 fp32 = 256 KiB of the 512 KiB), so every 16 keys
// cost 64 KiB of K + V through L2 -> LDS per 4.2 MFLOP: the kernel lives on how well that stream is hidden. The d-wide kernel
// had ONE K tile and ONE V tile of 32 keys in the LDS (2 x 64 KiB): each was refilled during the single phase that did not
// read it, so a phase could not end before its 64 KiB had landed (measured 20 B/clk/CU) and nothing was ever prefetched.
// Here:
//   * tiles of 16 keys: K and V each get a TWO-slot ring (4 x 32 KiB at D = 1024) and every tile is requested 1.5 tiles
//     (three phases) before its first reader; the waits are COUNTED (vmcnt = the two younger requests of the wave);
//   * S^T = K Q^T on v_mfma_f32_16x16x32_f16 (M = 16 keys is all a tile has), O^T += V^T P^T on v_mfma_f32_32x32x16_f16
//     (its contraction length IS 16 keys: no padding in either product);
//   * the partial-S exchange through LDS (needed anyway: four waves hold a quarter of d each) is also the layout change
//     between the two matrix shapes: a wave writes its partial in the 16x16 accumulator layout (4 keys x 16 B per lane) and
//     every wave of the row group reads the four partials in the layout the 32x32x16 B operand wants (8 keys of its own
//     row), sums them in the same order (bit-identical in all four) and runs the same softmax;
//   * D / 4 columns per wave for EVERY head dim: 160 (D = 640: five 32-wide k-steps / output blocks, no padded column, no
//     MFMA on zeros), 192 (768), 256 (1024); always 8 waves = two per SIMD (the d-wide kernel ran D = 768 on 6 waves: two
//     SIMDs carried twice the matrix work of the other two).
// LDS images are lane-linear (LDS-DMA), swizzled on the SOURCE side: K chunk ^= row & 15 (16 rows, 2 KiB-multiple row
// pitch: one transposition-free ds_read_b128 per fragment covers all 64 banks), V chunk ^= (row & 3) << 2 (transposing
// reads), both XORs act on the low four bits of the chunk index IN THE ROW (80 / 96 / 128 chunks: multiples of 16).
#pragma once
#include "flash_attn_dsplit.cuh"

namespace fa2 {

template <int D>
struct GeoRing {
  static_assert(D == 640 || D == 768 || D == 1024, "head dims 640 / 768 / 1024");
  static constexpr int NSP = 4, DH = D / 4, BC = 16, NW = 8, BR = 64, NT = 512;
  static constexpr int ROW = D * 2, TILE = BC * ROW, NP = TILE / 1024;  // 1-KiB DMA pieces per operand tile: 20 / 24 / 32
  static constexpr int PPW = (NP + NW - 1) / NW;                        // at most this many per wave (640: waves 4..7 carry one less)
  static constexpr int SX = NW * 2048;                                  // partial S^T: 32 rows x 16 keys fp32 per wave
  static constexpr int RING = 4 * TILE;                                 // K slot 0, K slot 1, V slot 0, V slot 1
  static constexpr int OS = DH * 2 + 16, EPI = NW * 32 * OS;
  static constexpr int LDS_BYTES = RING + SX > EPI ? RING + SX : EPI;
  static constexpr int NKS = DH / 32, NDB = DH / 32, CPP = DH / 8;      // k-steps, output blocks, 16-byte chunks per part
  static_assert(LDS_BYTES <= 160 * 1024 && (ROW / 16) % 16 == 0, "LDS / swizzle range");
};

template <int N>
__device__ __forceinline__ void dring_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// STAGGER: the two row groups (waves 0-3 / 4-7: one wave of each per SIMD) run ONE PHASE APART, so the exchange + softmax +
// PV phase of one group overlaps with the QK^T phase of the other (the matrix pipe idles through exchange and softmax
// otherwise). Every interval between two workgroup barriers then carries ONE tile request issued by all eight waves
// (even intervals K, odd intervals V), every request is two intervals ahead of its first reader, and the wait before
// every barrier leaves exactly the one younger request in flight.
template <int D, int OPT, bool STAGGER = true, int PRIO = 0, int KPF = 1, int VPF = 1>
__global__ __launch_bounds__(512, 2) void fa2_fwd_dring_kernel(const half_t* __restrict__ Q, const half_t* __restrict__ K,
                                                               const half_t* __restrict__ V, half_t* __restrict__ O,
                                                               int N, int n_qblk, int n_heads, float scale_log2e) {
  using G = GeoRing<D>;
  constexpr int NKS = G::NKS, NDB = G::NDB, PPW = G::PPW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5, i16 = lane & 15, g4 = lane >> 4;
  const int part = wave & 3, rg = wave >> 2;

  int head_i, qb_i;
  {
    const int bid = blockIdx.x;
    if ((OPT & OPT_XCD) && (n_heads & 7) == 0) {  // heads pinned to XCDs: the workgroups of a head share one L2
      const int xcd = bid & 7, k = bid >> 3;
      head_i = (k / n_qblk) * 8 + xcd;
      qb_i = k - (k / n_qblk) * n_qblk;
    } else {
      head_i = bid / n_qblk;
      qb_i = bid - head_i * n_qblk;
    }
  }
  const size_t head = (size_t)head_i * N * D;
  const int q_row0 = qb_i * G::BR + rg * 32;
  const unsigned lds0 = hgemm::lds_addr_of(smem);
  const char* Kh = reinterpret_cast<const char*>(K + head);
  const char* Vh = reinterpret_cast<const char*>(V + head);

  // ---- LDS-DMA: piece p = i * 8 + wave of an operand tile is the lane-linear KiB p of its image: byte o = p * 1024 +
  // lane * 16 = chunk c of row r; it is fetched from source chunk c ^ swizzle(r) of the same row (the tile's rows are
  // contiguous in memory: pitch ROW). Per-lane source offsets are loop constants (PPW each for K and V).
  const bool short_wave = G::NP % G::NW != 0 && wave >= G::NP % G::NW;  // D = 640: 20 pieces over 8 waves
  unsigned k_voff[PPW], v_voff[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int o = (i * G::NW + wave) * 1024 + lane * 16;
    const int r = o / G::ROW, c = (o % G::ROW) >> 4;
    k_voff[i] = (unsigned)(r * G::ROW + ((c ^ (r & 15)) << 4));
    v_voff[i] = (unsigned)(r * G::ROW + ((c ^ ((r & 3) << 2)) << 4));
  }
  auto dma_tile = [&](const char* base, const unsigned (&voff)[PPW], int jt, unsigned dst) __attribute__((always_inline)) {
    const char* s = base + (size_t)jt * G::TILE;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      if (i * G::NW + G::NW <= G::NP) hgemm::glds16_asm(s, voff[i], dst + (unsigned)(i * G::NW + wave) * 1024u);
      else if (!short_wave) hgemm::glds16_asm(s, voff[i], dst + (unsigned)(i * G::NW + wave) * 1024u);  // wave-uniform branch
    }
  };
  // ---- Q fragments (B operand of S^T = K Q^T on 16x16x32): query 16*qb + i16, d = part*DH + 32*ks + 8*g4 .. +7
  h8 qf[2][NKS];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const half_t* qp = Q + head + (size_t)(q_row0 + qb * 16 + i16) * D + part * G::DH + g4 * 8;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[qb][ks] = *reinterpret_cast<const h8*>(qp + ks * 32);
  }
  f16v ot[NDB];
#pragma unroll
  for (int b = 0; b < NDB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[b][r] = 0.f;
  float m_run = -1.0e30f, l_run = 0.f;

  const int T = N / G::BC;
  __builtin_assume(T > 0);
  auto clampt = [&](int t) __attribute__((always_inline)) { return t < T ? t : T - 1; };  // past the end: refill a dead slot (keeps the counts uniform)
  // OPT_1STAGE = the `stages = 1` form: every tile request is waited for where it is issued (no load runs under compute)
  auto req_k = [&](int t) __attribute__((always_inline)) {
    if constexpr ((OPT & OPT_ABL_DMA) != 0) return;  // probe ablation: no K / V traffic at all after the prologue
    dma_tile(Kh, k_voff, clampt(t), lds0 + (t & 1) * G::TILE);
    if constexpr ((OPT & OPT_1STAGE) != 0) hgemm::wait_vmcnt<0>();
  };
  auto req_v = [&](int t) __attribute__((always_inline)) {
    if constexpr ((OPT & OPT_ABL_DMA) != 0) return;
    dma_tile(Vh, v_voff, clampt(t), lds0 + 2 * G::TILE + (t & 1) * G::TILE);
    if constexpr ((OPT & OPT_1STAGE) != 0) hgemm::wait_vmcnt<0>();
  };
  // OPT_SPREAD: the PPW pieces of a tile request go out one at a time between the MFMAs of the phase instead of back to back at its
  // head (an LDS-DMA instruction holds the wave's issue for 60-180 clocks; four in a row stall its MFMAs that long)
  auto piece_of = [&](bool is_v, int t, int i) __attribute__((always_inline)) {
    if constexpr ((OPT & OPT_ABL_DMA) != 0) return;
    const char* src = (is_v ? Vh : Kh) + (size_t)clampt(t) * G::TILE;
    const unsigned dst = lds0 + (is_v ? 2 * G::TILE : 0) + (t & 1) * G::TILE + (unsigned)(i * G::NW + wave) * 1024u;
    const unsigned vo = is_v ? v_voff[i < PPW ? i : 0] : k_voff[i < PPW ? i : 0];
    if (i * G::NW + G::NW <= G::NP) hgemm::glds16_asm(src, vo, dst);
    else if (!short_wave) hgemm::glds16_asm(src, vo, dst);
    if constexpr ((OPT & OPT_1STAGE) != 0) hgemm::wait_vmcnt<0>();
  };
  auto wait_young = [&](bool two) __attribute__((always_inline)) {  // leave this wave's one / two youngest tile requests in flight
    if constexpr (G::NP % G::NW == 0) {
      if (two) dring_wait_vm<2 * PPW>();
      else dring_wait_vm<PPW>();
    } else {
      if (short_wave) { if (two) dring_wait_vm<2 * (PPW - 1)>(); else dring_wait_vm<PPW - 1>(); }
      else { if (two) dring_wait_vm<2 * PPW>(); else dring_wait_vm<PPW>(); }
    }
  };
  // lock-step request order of a wave: K0 V0 K1 | V1 K2 | V2 K3 | ...   staggered: K0 V0 | K1 | V1 | K2 | V2 | ...
  req_k(0);
  req_v(0);
  if constexpr (!STAGGER) req_k(1);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), compiler-visible: the first tiles and the Q loads (prologue only)
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+v"(qf[qb][ks]));  // keep the Q loads out of the KV loop
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // ---- fragment offsets. K fragment ks (A operand, 16 keys x 32 d): row i16, logical chunk part*CPP + 4*ks + g4.
  // V^T fragment b (A operand of the 32x32x16 PV step, 32 d x 16 keys): two transposing reads, rows v_row and v_row + 8;
  // lane half `hi` therefore holds keys 4*hi .. +3 and 8 + 4*hi .. +3 -- the P^T fragment is built in the same key order.
  // POW2 (D = 1024: a part starts on a 16-chunk boundary): the swizzle XOR only touches bits the fragment index does
  // not carry into, so fragment i = (one lane constant) ^ (i & 3) << 6, + (i >> 2) * 256 -- two address registers instead
  // of sixteen (the register file is full at this head dim). Otherwise one register per fragment.
  constexpr bool POW2 = G::CPP % 16 == 0;
  const int v_row = 4 * hi + (i16 >> 2);
  const int v_w = ((lane >> 4) & 1) * 2 + ((i16 & 3) >> 1);
  auto k_off_of = [&](int ks) { return (unsigned)(i16 * G::ROW + (((part * G::CPP + 4 * ks + g4) ^ i16) << 4)); };
  auto v_off_of = [&](int b) {
    return (unsigned)(2 * G::TILE + v_row * G::ROW + (((part * G::CPP + 4 * b + v_w) ^ ((v_row & 3) << 2)) << 4) + ((i16 & 1) << 3));
  };
  unsigned koff[POW2 ? 1 : NKS], voffs[POW2 ? 1 : NDB];
#pragma unroll
  for (int ks = 0; ks < (POW2 ? 1 : NKS); ++ks) koff[ks] = k_off_of(ks);
#pragma unroll
  for (int b = 0; b < (POW2 ? 1 : NDB); ++b) voffs[b] = v_off_of(b);
  if constexpr (POW2) asm volatile("" : "+v"(koff[0]), "+v"(voffs[0]));  // opaque: hipcc would otherwise hoist all sixteen
  auto k_addr = [&](int ks) __attribute__((always_inline)) -> unsigned {
    if constexpr (POW2) return (koff[0] ^ (unsigned)((ks & 3) << 6)) + (unsigned)((ks >> 2) * 256);
    else return koff[ks];
  };
  auto v_addr = [&](int b) __attribute__((always_inline)) -> unsigned {
    if constexpr (POW2) return (voffs[0] ^ (unsigned)((b & 3) << 6)) + (unsigned)((b >> 2) * 256);
    else return voffs[b];
  };
  // partial-S exchange image of a wave: [32 rows][16 keys] fp32, 64-byte rows, 16-byte chunk ^= (row >> 2) & 3
  // chunk swizzle of a row (rows repeat every 16): sw = bit 2 | (bit 3 ^ bit 1) << 1. It has to be conflict-free under TWO rules:
  // the ds_read_b128 of the 32x32 layout is served in 16-lane groups over 256 B -- (row & 3, sw) must be distinct over 16 rows --
  // and the ds_write_b128 of the 16x16 layout in 8-lane groups over 128 B -- (row & 1, sw) distinct over 8 consecutive rows. The first
  // form, sw = (row >> 2) & 3, met only the read rule: every write was a 2-way conflict (SQ_LDS_BANK_CONFLICT 135 k cycles per CU and
  // launch at [1,16,4096,1024], 8 per write; 4 k with this one -- profiles/r03_fa_dring_lds_counters.log)
  auto sx_sw = [](int row) { return ((row >> 2) & 1) | ((((row >> 3) ^ (row >> 1)) & 1) << 1); };
  char* sx_w = smem + G::RING + wave * 2048 + i16 * 64 + ((g4 ^ sx_sw(i16)) << 4);  // + qb * 1024
  const char* sx_r = smem + G::RING + rg * 4 * 2048 + l31 * 64;                      // + p * 2048 + chunk
  const int sw_r = sx_sw(l31 & 15);
  const int sx_c0 = (hi ^ sw_r) << 4, sx_c1 = ((2 + hi) ^ sw_r) << 4;

  if constexpr (STAGGER) {
    if (rg == 1) {  // group 1 runs one interval behind: its share of the first interval's request, then the barrier
      req_k(1);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  }

  for (int j = 0; j < T; ++j) {
    const int slot = j & 1;
    // ================= phase 1: partial S^T over this wave's quarter of d
    if constexpr (PRIO != 0) __builtin_amdgcn_s_setprio(0);
    {
      f4 s[2];
      s[0] = f4{0.f, 0.f, 0.f, 0.f}, s[1] = f4{0.f, 0.f, 0.f, 0.f};
      const char* kb = smem + slot * G::TILE;
      // KPF K fragments in flight: with one (the first form of this kernel) every k-step paid a whole LDS round trip between its
      // read and its MFMAs -- "R w MM R w MM ..." in the ISA -- and the time of the fragment reads was ADDED to the matrix time
      // (ablations in profiles/r03_fa_dring_lds_counters.log)
      constexpr int KD = KPF < NKS ? KPF : NKS;
      auto rd_k = [&](int ks) __attribute__((always_inline)) -> h8 {
        if constexpr ((OPT & OPT_ABL_K) != 0) return qf[1][ks];
        else return *reinterpret_cast<const h8*>(kb + k_addr(ks));
      };
      h8 kf[KD];
#pragma unroll
      for (int i = 0; i < KD; ++i) kf[i] = rd_k(i);
      if constexpr (KD > 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        s[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[ks % KD], qf[0][ks], s[0], 0, 0, 0);
        s[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[ks % KD], qf[1][ks], s[1], 0, 0, 0);
        cln_mfma_keep(s[0], kf[ks % KD], qf[0][ks]);  // destinations disjoint from the operands (common.h)
        cln_mfma_keep(s[1], kf[ks % KD], qf[1][ks]);
        if (ks + KD < NKS) kf[ks % KD] = rd_k(ks + KD);
        if constexpr ((OPT & OPT_SPREAD) != 0) {
          constexpr int STRIDE = NKS / PPW > 0 ? NKS / PPW : 1;
          if (ks % STRIDE == 0 && ks / STRIDE < PPW) {
            if constexpr (!STAGGER) piece_of(true, j + 1, ks / STRIDE);
            else if (rg == 0) piece_of(false, j + 1, ks / STRIDE);
            else piece_of(true, j + 1, ks / STRIDE);
          }
        } else if (ks == 0) {
          if constexpr (!STAGGER) req_v(j + 1);
          else if (rg == 0) req_k(j + 1);
          else req_v(j + 1);
        }
        if constexpr (KD > 1) __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr ((OPT & OPT_ABL_XW) == 0) {
        *reinterpret_cast<f4*>(sx_w) = s[0];
        *reinterpret_cast<f4*>(sx_w + 1024) = s[1];
      } else {
        asm volatile("" ::"v"(s[0]), "v"(s[1]));
      }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the partial is in LDS
    wait_young(!STAGGER);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // ================= phase 2: S = sum of the four partials, softmax, O^T += V^T P^T
    if constexpr (PRIO != 0) __builtin_amdgcn_s_setprio(1);
    {
      const char* vb = smem + slot * G::TILE;
      constexpr int VD = VPF < NDB ? VPF : NDB;
      auto rd_v = [&](int b) __attribute__((always_inline)) -> h8 {
        if constexpr ((OPT & OPT_ABL_V) != 0) return qf[0][b % NKS];
        else {
          const char* vp = vb + v_addr(b);
          return h8_cat(lds_read_tr16(vp), lds_read_tr16(vp + 8 * G::ROW));
        }
      };
      h8 vf[VD];
      if constexpr (VD > 1) {  // the first V fragments fly under the softmax
#pragma unroll
        for (int i = 0; i < VD; ++i) vf[i] = rd_v(i);
      }
      float s8[8];
      {
        f4 a0 = *reinterpret_cast<const f4*>(sx_r + sx_c0), a1 = *reinterpret_cast<const f4*>(sx_r + sx_c1);
#pragma unroll
        for (int p = 1; p < ((OPT & OPT_ABL_XR) != 0 ? 1 : 4); ++p) {
          const f4 t0 = *reinterpret_cast<const f4*>(sx_r + p * 2048 + sx_c0), t1v = *reinterpret_cast<const f4*>(sx_r + p * 2048 + sx_c1);
          a0 += t0, a1 += t1v;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) s8[e] = a0[e], s8[4 + e] = a1[e];
      }
      // lock-step: the K slot of tile j is free since the last barrier; staggered: see the interval table above
      if constexpr ((OPT & OPT_SPREAD) == 0) {
        if constexpr (!STAGGER) req_k(j + 2);
        else if (rg == 0) req_v(j + 1);
        else req_k(j + 2);
      }
      float mx = s8[0];
#pragma unroll
      for (int e = 1; e < 8; ++e) mx = fmaxf(mx, s8[e]);
      {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
      }
      const float mxs = mx * scale_log2e;
      bool grow;
      if constexpr ((OPT & OPT_DEFER) != 0) grow = (mxs - m_run) > 8.0f;
      else grow = mxs > m_run;
      if (__builtin_amdgcn_ballot_w64(grow) != 0) {
        const float m_new = fmaxf(m_run, mxs);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int b = 0; b < NDB; ++b)
#pragma unroll
          for (int r = 0; r < 16; r += 4) {  // serialise the register round trips of the rescale
            float t0 = ot[b][r], t1v = ot[b][r + 1], t2 = ot[b][r + 2], t3 = ot[b][r + 3];
            asm volatile("" : "+v"(t0), "+v"(t1v), "+v"(t2), "+v"(t3));
            ot[b][r] = t0 * alpha, ot[b][r + 1] = t1v * alpha, ot[b][r + 2] = t2 * alpha, ot[b][r + 3] = t3 * alpha;
          }
      }
      h8 pf;
      {
        const float nm = -m_run;
        float psum = 0.f;
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const float a0 = __builtin_amdgcn_exp2f(fmaf(s8[e], scale_log2e, nm));
          const float a1 = __builtin_amdgcn_exp2f(fmaf(s8[e + 1], scale_log2e, nm));
          psum += a0 + a1;
          const h2 a = __builtin_convertvector(f2{a0, a1}, h2);
          pf[e] = a[0], pf[e + 1] = a[1];
        }
        l_run += psum;
      }
      if constexpr (VD == 1) vf[0] = rd_v(0);
      else __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b = 0; b < NDB; ++b) {
        ot[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[b % VD], pf, ot[b], 0, 0, 0);
        cln_mfma_keep(ot[b], vf[b % VD], pf);
        if (b + VD < NDB) vf[b % VD] = rd_v(b + VD);
        if constexpr ((OPT & OPT_SPREAD) != 0) {
          constexpr int STRIDE = NDB / PPW > 0 ? NDB / PPW : 1;
          if (b % STRIDE == 0 && b / STRIDE < PPW) {
            if constexpr (!STAGGER) piece_of(false, j + 2, b / STRIDE);
            else if (rg == 0) piece_of(true, j + 1, b / STRIDE);
            else piece_of(false, j + 2, b / STRIDE);
          }
        }
        if constexpr (VD > 1) __builtin_amdgcn_sched_barrier(0);
      }
    }
    wait_young(!STAGGER);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  if constexpr (STAGGER) {
    if (rg == 0) {  // group 1's last phase 2: keep the barrier count equal and the rings intact until it is done
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  }
  hgemm::wait_vmcnt<0>();  // the dead refills of the last tiles: nothing may land in the staging area below
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // ---- epilogue: O = O^T / l, staged through LDS (wave-private rows)
  float l_tot;
  {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
    l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  }
  const float inv = 1.0f / l_tot;
  char* ob = smem + wave * (32 * G::OS);
  const int lane_e = cln_fresh_lane(), l31_e = lane_e & 31, hi_e = lane_e >> 5;
#pragma unroll
  for (int b = 0; b < NDB; ++b) {
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      h4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (half_t)(ot[b][rq * 4 + e] * inv);
      *reinterpret_cast<h4*>(ob + l31_e * G::OS + (b * 32 + rq * 8 + hi_e * 4) * 2) = o;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  constexpr int LPR = G::DH / 8;  // 16-byte segments per row of this wave's column block
  half_t* og = O + head + (size_t)q_row0 * D + part * G::DH;
  for (int idx = lane_e; idx < 32 * LPR; idx += 64) {
    const int row = idx / LPR, c = idx % LPR;
    *reinterpret_cast<u4*>(og + (size_t)row * D + c * 8) = *reinterpret_cast<const u4*>(ob + row * G::OS + c * 16);
  }
}

template <int D, int OPT, bool STAGGER = true, int PRIO = 0, int KPF = 1, int VPF = 1>
int launch_dring(const void* q, const void* k, const void* v, void* o, int B, int H, int N, hipStream_t stream) {
  using G = GeoRing<D>;
  if (N % G::BR != 0) return CLN_ERR_UNSUPPORTED;
  static cln_lds_attr lds_attr;  // per device, thread-safe (common.h)
  if (cln_ensure_lds(lds_attr, reinterpret_cast<const void*>(&fa2_fwd_dring_kernel<D, OPT, STAGGER, PRIO, KPF, VPF>), G::LDS_BYTES) != CLN_OK) return CLN_ERR_LAUNCH;
  const float scale_log2e = 1.4426950408889634f / sqrtf((float)D);
  const int n_qblk = N / G::BR;
  CLN_LAUNCH((fa2_fwd_dring_kernel<D, OPT, STAGGER, PRIO, KPF, VPF>), dim3(n_qblk * B * H), dim3(G::NT), G::LDS_BYTES, stream, (const half_t*)q,
             (const half_t*)k, (const half_t*)v, (half_t*)o, N, n_qblk, B * H, scale_log2e);
  return cln_check_launch();
}

}  // namespace fa2
