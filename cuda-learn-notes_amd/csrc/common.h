// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernel library.
//
// One header instead of the reference's per-file macro blocks
// (reference: kernels/*/**.cu top-of-file macro sections, e.g.
// kernels/elementwise/elementwise.cu:12-18, kernels/reduce/block_all_reduce.cu:12-18).
// Everything here assumes wave64 (gfx950) -- there is no 32-lane path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>

// ---------------------------------------------------------------- C-ABI status codes
// Reference bindings throw std::runtime_error (e.g. kernels/hgemm/naive/hgemm.cu:772-782);
// the C-ABI returns an int and the Python host maps it back to RuntimeError.
#define CLN_OK 0
#define CLN_ERR_BAD_ARG (-1)        // null pointer / non-positive dim / misaligned pointer
#define CLN_ERR_UNSUPPORTED (-2)    // shape outside the supported set ("headdim not support!", K dispatch lists)
#define CLN_ERR_LAUNCH (-3)         // hipGetLastError() != hipSuccess after the launch
#define CLN_ERR_VENDOR (-4)         // rocBLAS row: handle missing / rocblas status != success

#define CLN_API extern "C" __attribute__((visibility("default")))

// Launch with a clean per-thread error slot: hipGetLastError() is sticky across unrelated runtime calls of
// the host process (e.g. torch's allocator polls events -> hipErrorNotReady), which must not be reported as
// a failure of OUR launch.
#define CLN_LAUNCH(...)            \
  do {                             \
    (void)hipGetLastError();       \
    hipLaunchKernelGGL(__VA_ARGS__); \
  } while (0)

static inline int cln_check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) fprintf(stderr, "[cln_amd] launch failed: %s\n", hipGetErrorString(e));
  return e == hipSuccess ? CLN_OK : CLN_ERR_LAUNCH;
}
// hipFuncSetAttribute wrapper with the same diagnostics
static inline int cln_set_lds(const void* fn, int bytes) {
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    fprintf(stderr, "[cln_amd] hipFuncSetAttribute(%d B LDS) failed: %s\n", bytes, hipGetErrorString(e));
    (void)hipGetLastError();
    return CLN_ERR_LAUNCH;
  }
  return CLN_OK;
}

// "Raise MaxDynamicSharedMemorySize once" -- once PER DEVICE (the attribute lives in the per-device function
// object: a second GPU used from the same process needs its own call) and safe from several host threads: one bit
// per HIP device in an atomic mask; two racing threads both issue the (idempotent) attribute call.
// (reference re-issues cudaFuncSetAttribute on every call: kernels/hgemm/mma/basic/hgemm_mma_stage.cu:2333.)
struct cln_lds_attr {
  std::atomic<unsigned long long> done{0};
};
static inline int cln_ensure_lds(cln_lds_attr& st, const void* fn, int bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    return CLN_ERR_LAUNCH;
  }
  if (dev >= 64) return cln_set_lds(fn, bytes);  // beyond the mask: always set
  const unsigned long long bit = 1ull << dev;
  if (st.done.load(std::memory_order_acquire) & bit) return CLN_OK;
  if (cln_set_lds(fn, bytes) != CLN_OK) return CLN_ERR_LAUNCH;
  st.done.fetch_or(bit, std::memory_order_release);
  return CLN_OK;
}

static inline bool cln_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
// every packed rung issues `bytes`-wide vector accesses (bytes = pack width, a power of two <= 16): a sliced view with
// a storage offset must be rejected, not loaded misaligned
static inline bool cln_aligned(const void* p, size_t bytes) { return (reinterpret_cast<uintptr_t>(p) & (bytes - 1)) == 0; }

// ---------------------------------------------------------------- vector types
typedef _Float16 half_t;
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned short us8 __attribute__((ext_vector_type(8)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

#define CLN_WAVE 64
#ifndef CLN_STREAM_WGS_PER_CU
#define CLN_STREAM_WGS_PER_CU 32
#endif

// Streaming stores. A launch whose tensors fill the 256 MB MALL together (or more) writes its output with non-temporal
// stores (global_store ... nt): the write stream then leaves the MALL to the read stream. Measured on y = 2x over f32x4
// (tools/ubench/stream_nt.hip, profiles/r03_stream_nt_ubench.log): 256 MB in + 256 MB out 4.1-4.8 -> 6.2-6.8 TB/s; no
// difference at 64 + 64 MB (everything is MALL-resident) and at >= 512 + 512 MB (nothing is); non-temporal LOADS on top
// of it lose the gain again. On the kernels (profiles/r03_bw_streaming_stores_probe.log): 8192^2 f32 softmax 5.45 -> 6.88
// TB/s, rope 5.27 -> 6.85, three-tensor f16 add 5.22 -> 7.20, embedding 65536 x 1024 6.18 -> 6.94; at exactly 256 MB (the
// 8192^2 f16 rows) +0.5-5 %.
static inline int cln_stream_nt(long long footprint_bytes) { return footprint_bytes >= (256LL << 20) ? 1 : 0; }

// Grid sizing for HBM-bound streaming kernels: enough workgroups to cover all
// 256 CUs several times over, grid-stride for the rest (cdna guide G11).
// Round 6: a launch that moves >= 512 MB (`traffic_bytes`) is NOT capped -- one trip per thread, waves that retire are replaced instead of looping
// in lockstep: y = 2x over f32x4 at 512 + 512 MB 96.8 -> 88.6 us (+9 %), c = a + b at 3 x 256 MB 155 -> 136 us (+14 %); below that the capped
// grid-stride form is level or ahead (134 + 134 MB: 23.7 vs 24.1 us) -- tools/ubench/stream_forms.hip, profiles/r06_stream_forms_ubench.log.
static inline int cln_stream_grid(long long work_items, int block, long long traffic_bytes = 0) {
  long long g = (work_items + block - 1) / block;
  const long long cap = traffic_bytes >= (512LL << 20) ? 0x7fffffffLL : 256LL * CLN_STREAM_WGS_PER_CU;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

#ifdef __HIPCC__
template <int BYTES> struct cln_raw;
template <> struct cln_raw<16> { typedef u4 type; };
template <> struct cln_raw<8> { typedef u2 type; };
template <> struct cln_raw<4> { typedef unsigned int type; };
template <> struct cln_raw<2> { typedef unsigned short type; };
template <> struct cln_raw<1> { typedef unsigned char type; };
// *dst = v, as a non-temporal store when `nt` (a launch-uniform flag: cln_stream_nt of the launch's footprint). The nt form is
// inline asm: with __builtin_nontemporal_store under a run-time flag hipcc merges the two stores of the diamond into ONE
// plain store (the merge drops the nt bit). The trailing s_nop covers the "store of > 64 bits, then a write to its data
// registers" wait state, which the hazard pass cannot see inside an asm statement.
template <typename P>
__device__ __forceinline__ void cln_store_stream(P* dst, const P& v, int nt) {
  typedef typename cln_raw<sizeof(P)>::type R;
  if (nt) {
    const R raw = __builtin_bit_cast(R, v);
    if constexpr (sizeof(P) == 16) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(dst), "v"(raw) : "memory");
    else if constexpr (sizeof(P) == 8) asm volatile("global_store_dwordx2 %0, %1, off nt\n\ts_nop 1" ::"v"(dst), "v"(raw) : "memory");
    else if constexpr (sizeof(P) == 4) asm volatile("global_store_dword %0, %1, off nt" ::"v"(dst), "v"(raw) : "memory");
    else if constexpr (sizeof(P) == 2) asm volatile("global_store_short %0, %1, off nt" ::"v"(dst), "v"((unsigned int)raw) : "memory");
    else asm volatile("global_store_byte %0, %1, off nt" ::"v"(dst), "v"((unsigned int)raw) : "memory");
  } else {
    *dst = v;
  }
}

// ---------------------------------------------------------------- wave64 reductions
// All-lanes reductions on the VALU only (no LDS round trips): four DPP steps inside a 16-lane row
// (quad_perm xor 1, xor 2, row_half_mirror, row_mirror), then v_permlane16_swap / v_permlane32_swap (gfx950)
// for the cross-row steps. ~12 VALU instructions instead of six dependent ds_bpermute (~100+ cycles each):
// the row kernels (softmax / norms, one workgroup per row) are latency-bound on exactly this chain.
// (reference: warp_reduce_sum_f32 with WARP_SIZE 32, kernels/reduce/block_all_reduce.cu:30-37 -- re-derived for
// 64 lanes, not translated.)
template <int CTRL>
__device__ __forceinline__ float cln_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int cln_dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
#define CLN_WAVE_REDUCE(v, OP, DPPF, TOBITS, FROMBITS)                                              \
  v = OP(v, DPPF<0xB1>(v));  /* quad_perm [1,0,3,2] */                                               \
  v = OP(v, DPPF<0x4E>(v));  /* quad_perm [2,3,0,1] */                                               \
  v = OP(v, DPPF<0x141>(v)); /* row_half_mirror */                                                   \
  v = OP(v, DPPF<0x140>(v)); /* row_mirror */                                                        \
  {                                                                                                  \
    const auto s16 = __builtin_amdgcn_permlane16_swap(TOBITS(v), TOBITS(v), false, false);           \
    v = OP(FROMBITS(s16[0]), FROMBITS(s16[1]));                                                      \
    const auto s32 = __builtin_amdgcn_permlane32_swap(TOBITS(v), TOBITS(v), false, false);           \
    v = OP(FROMBITS(s32[0]), FROMBITS(s32[1]));                                                      \
  }
__device__ __forceinline__ float cln_addf(float a, float b) { return a + b; }
__device__ __forceinline__ int cln_addi(int a, int b) { return a + b; }
__device__ __forceinline__ unsigned cln_i2u(int a) { return (unsigned)a; }
__device__ __forceinline__ int cln_u2i(unsigned a) { return (int)a; }
__device__ __forceinline__ float wave_sum(float v) {
  CLN_WAVE_REDUCE(v, cln_addf, cln_dpp, __float_as_uint, __uint_as_float)
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  CLN_WAVE_REDUCE(v, fmaxf, cln_dpp, __float_as_uint, __uint_as_float)
  return v;
}
__device__ __forceinline__ int wave_sum_i32(int v) {
  CLN_WAVE_REDUCE(v, cln_addi, cln_dpp_i, cln_i2u, cln_u2i)
  return v;
}

// Block-wide sum for NT threads (NT multiple of 64, <= 1024). `scratch` must hold
// 16 floats. Every thread gets the result. Two barriers.
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* scratch) {
  constexpr int NW = NT / 64;
  v = wave_sum(v);
  if constexpr (NW == 1) return v;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();  // protect scratch reuse across consecutive calls
  if (lane == 0) scratch[w] = v;
  __syncthreads();
  float t = (lane < NW) ? scratch[lane] : 0.0f;
#pragma unroll
  for (int m = 8; m >= 1; m >>= 1) t += __shfl_xor(t, m, 64);  // NW <= 16
  return __shfl(t, 0, 64);
}
template <int NT>
__device__ __forceinline__ float block_max(float v, float* scratch) {
  constexpr int NW = NT / 64;
  v = wave_max(v);
  if constexpr (NW == 1) return v;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) scratch[w] = v;
  __syncthreads();
  float t = (lane < NW) ? scratch[lane] : -3.402823466e+38f;
#pragma unroll
  for (int m = 8; m >= 1; m >>= 1) t = fmaxf(t, __shfl_xor(t, m, 64));
  return __shfl(t, 0, 64);
}

// Runtime-block-size variants (blockDim.x multiple of 64, <= 1024).
__device__ __forceinline__ float block_sum_rt(float v, float* scratch) {
  const int nw = blockDim.x >> 6;
  v = wave_sum(v);
  if (nw == 1) return v;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) scratch[w] = v;
  __syncthreads();
  float t = (lane < nw) ? scratch[lane] : 0.0f;
#pragma unroll
  for (int m = 8; m >= 1; m >>= 1) t += __shfl_xor(t, m, 64);
  return __shfl(t, 0, 64);
}
__device__ __forceinline__ float block_max_rt(float v, float* scratch) {
  const int nw = blockDim.x >> 6;
  v = wave_max(v);
  if (nw == 1) return v;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) scratch[w] = v;
  __syncthreads();
  float t = (lane < nw) ? scratch[lane] : -3.402823466e+38f;
#pragma unroll
  for (int m = 8; m >= 1; m >>= 1) t = fmaxf(t, __shfl_xor(t, m, 64));
  return __shfl(t, 0, 64);
}

// ---------------------------------------------------------------- LDS / global address-space casts
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_cvoid;

// 16-byte async global->LDS copy (LDS-DMA). LDS destination is wave-uniform base +
// lane*16 (cdna guide section 5 caveat) -- callers pass the wave-uniform base.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gbl_cvoid*)gsrc, (lds_void*)lds_wave_base, 16, 0, 0);
}

// LDS transpose read: within each 16-lane group the lanes supply the 8-byte pieces of a
// [4][16] b16 block (lane i -> row i>>2, cols 4*(i&3)..+3) and lane i receives column i
// (4 rows). (ds_read_b64_tr_b16, cdna guide section 2 / T10.)
typedef __fp16 fp16x4_tr __attribute__((__vector_size__(4 * sizeof(__fp16))));
__device__ __forceinline__ h4 lds_read_tr16(const void* lds_addr) {
  fp16x4_tr t = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
      (__attribute__((address_space(3))) fp16x4_tr*)(lds_addr));
  h4 r;
  __builtin_memcpy(&r, &t, 8);
  return r;
}

// The same reads by LDS BYTE ADDRESS (an integer, as lds_addr_of() returns it) instead of a pointer derived from the `extern __shared__` symbol:
// an address formed as `smem + offset` costs one v_add_u32 per access even when the symbol resolves to 0 (hipcc adds the relocated symbol to every
// computed offset), and on gfx950 a plain VALU instruction is never free under MFMAs (matrix time and VALU time add). Callers fold the
// symbol's address into their per-lane base ONCE, outside the loop.
template <class T>
__device__ __forceinline__ T lds_ld(unsigned lds_byte_addr) {
  return *reinterpret_cast<const __attribute__((address_space(3))) T*>(lds_byte_addr);
}
template <class T>
__device__ __forceinline__ void lds_st(unsigned lds_byte_addr, const T& v) {
  *reinterpret_cast<__attribute__((address_space(3))) T*>(lds_byte_addr) = v;
}
__device__ __forceinline__ h4 lds_read_tr16_at(unsigned lds_byte_addr) {
  fp16x4_tr t = __builtin_amdgcn_ds_read_tr16_b64_v4f16(reinterpret_cast<__attribute__((address_space(3))) fp16x4_tr*>(lds_byte_addr));
  h4 r;
  __builtin_memcpy(&r, &t, 8);
  return r;
}

// Lane id recomputed at the point of use (two VALU instructions). The attention kernels run their KV loop with a full
// register file; lane-derived epilogue addresses computed at kernel entry would be carried -- i.e. spilled -- across
// it. Opaque to hipcc so it is neither hoisted nor merged with the entry-time lane id.
__device__ __forceinline__ int cln_fresh_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

// hipcc (ROCm 7.2) puts no early-clobber on an MFMA destination: the LAST MFMA that reads a dying A / B fragment may be given
// that fragment's registers as its destination (`v_mfma_f32_16x16x32_f16 v[66:69], v[66:69], v[34:37], 0`). On MI355X such
// an instruction intermittently returns wrong values when another wave's MFMAs interleave with it on the same SIMD: round 3
// traced the D = 256 attention kernel's ~1 % wrong launches and the D = 512 probe's 100 % to it (tools/mfma_overlap_scan.py,
// profiles/r03_fa_mfma_overlap_bisect.log; the failure rate follows the number of such instructions in the loop). This
// empty statement reads the result AND both operands right behind the MFMA, so the operands are alive across it and the
// register allocator must keep them disjoint from the destination. It emits no instruction and draws no hazard padding.
template <typename R, typename A, typename B>
__device__ __forceinline__ void cln_mfma_keep(const R& r, const A& a, const B& b) {
#if defined(__HIP_DEVICE_COMPILE__)  // (the host pass of hipcc parses device functions too: "v" is no x86 constraint)
  asm volatile("" ::"v"(r), "v"(a), "v"(b));
#else
  (void)r, (void)a, (void)b;
#endif
}

__device__ __forceinline__ h8 h8_cat(h4 lo, h4 hi) {
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
#endif  // __HIPCC__
