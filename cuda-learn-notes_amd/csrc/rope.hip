// Rotary position embedding on x[seq_len, hidden] fp32, interleaved pairs (x[2i], x[2i+1]),
// theta = 10000. Replaces reference kernels/rope/rope.cu:20-67 (kernels) / :80-120 (bindings).
//
// Semantics: the TORCH ORACLE the reference script prints beside its kernels (naive_rope,
// kernels/rope/rope.py:68-88): pair i of token t is rotated by t * theta^(-2i/hidden).
// The reference CUDA kernels compute the exponent with an INTEGER division
// (`token_idx / (N * 2)`, rope.cu:26, :41, :55-56) which is always 0, so every pair is rotated by
// t radians; `ref_quirk != 0` reproduces that behaviour bit-for-bit in structure for users who
// depend on it. HBM-bound (1 read + 1 write); sincos is computed in-kernel because the API passes
// no table (two v_sin/v_cos per pair hide under the 8 B/pair of traffic at ~6 TB/s); the frequency (one pow per
// column) is formed once per thread, which owns a column and walks the rows.
#include "common.h"

namespace {

__device__ __forceinline__ void rotate(float x1, float x2, float ang, float& o1, float& o2) {
  // angles reach seq_len radians: two-constant Cody-Waite reduction to [-pi, pi] (exact to ~1e-7
  // for |ang| < 2^15), then the hardware sin/cos, whose argument is in revolutions.
  const float k = rintf(ang * 0.15915494309189535f);
  float r = fmaf(-k, 6.28318548202514648f, ang);   // 2*pi rounded to fp32
  r = fmaf(-k, -1.74845553e-07f, r);                // 2*pi - fp32(2*pi)
  const float rev = r * 0.15915494309189535f;
  const float s = __builtin_amdgcn_sinf(rev), c = __builtin_amdgcn_cosf(rev);
  o1 = x1 * c - x2 * s;
  o2 = x1 * s + x2 * c;
}

// PAIRS pairs per thread: 1 -> 8-byte accesses (f32 / f32_v2 rungs), 2 -> 16-byte (f32x4_pack).
// A thread owns ONE column unit (PAIRS adjacent pairs) and walks rows blockIdx.y, + gridDim.y, ...: its rotation
// frequencies are formed once (PAIRS calls of pow), exactly as the script forms them (rope.py:77):
//   1.0 / (theta ** (float(2 i) / dim)), every step in fp32
// -- the angle t * freq amplifies a 1-ulp difference in freq by the token index (t = 8192: 1e-3 rad), so an algebraically
// equal exp2() form is NOT close enough at long sequences, and pow() per ELEMENT (the first version of this fix) cost more
// than the memory traffic (43 us vs 20 us at 4096 x 4096). A row of threads still reads a row of x contiguously.
template <int PAIRS>
__global__ __launch_bounds__(256) void rope_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                   int seq_len, int half_hidden, int ref_quirk, int stream_nt) {
  const int units_per_row = half_hidden / PAIRS;
  const int u = blockIdx.x * 256 + threadIdx.x;
  if (u >= units_per_row) return;
  float freq[PAIRS];
#pragma unroll
  for (int p = 0; p < PAIRS; ++p)
    freq[p] = ref_quirk ? 1.0f : 1.0f / powf(10000.0f, (float)(2 * (u * PAIRS + p)) / (float)(2 * half_hidden));
  // (round 5: an explicit four-rows-per-trip form -- loads of four rows issued together, the streaming / plain store choice a template parameter --
  // measured SLOWER: 4096^2 26.4 -> 28.9 us, 8192^2 97 -> 111 us; the waves of 16 resident workgroups already overlap their single loads)
#pragma unroll 4  // independent rows (x and out are __restrict__)
  for (int t = blockIdx.y; t < seq_len; t += gridDim.y) {
    const size_t off = ((size_t)t * half_hidden + (size_t)u * PAIRS) * 2;
    float v[2 * PAIRS], o[2 * PAIRS];
    if constexpr (PAIRS == 2) {
      *reinterpret_cast<f4*>(v) = *reinterpret_cast<const f4*>(x + off);
    } else {
      *reinterpret_cast<f2*>(v) = *reinterpret_cast<const f2*>(x + off);
    }
#pragma unroll
    for (int p = 0; p < PAIRS; ++p) rotate(v[2 * p], v[2 * p + 1], (float)t * freq[p], o[2 * p], o[2 * p + 1]);
    if constexpr (PAIRS == 2) {
      cln_store_stream(reinterpret_cast<f4*>(out + off), *reinterpret_cast<const f4*>(o), stream_nt);
    } else {
      cln_store_stream(reinterpret_cast<f2*>(out + off), *reinterpret_cast<const f2*>(o), stream_nt);
    }
  }
}

template <int PAIRS>
int launch_rope(const void* x, void* out, int seq_len, int hidden, int ref_quirk, hipStream_t st) {
  if (!x || !out || seq_len <= 0 || hidden <= 0) return CLN_ERR_BAD_ARG;
  if (hidden % (2 * PAIRS)) return CLN_ERR_UNSUPPORTED;
  if (!cln_aligned(x, 8 * PAIRS) || !cln_aligned(out, 8 * PAIRS)) return CLN_ERR_BAD_ARG;
  const int half_hidden = hidden / 2;
  const int gx = (half_hidden / PAIRS + 255) / 256;
  // round 5 (profiles/r05_rope_grid_probe.log): at most 16384 workgroups and at least 4 rows per thread instead of 4096 / 16 -- 8192^2 96.9 -> 90.4 us
  // (5.54 -> 5.94 TB/s), 4096 x 2048 16.6 -> 14.3 us, 64 rows 5.4 -> 3.7 us, 4096^2 26.4 -> 26.0: a thread has ONE 16-byte load in flight, so the bytes in
  // flight are the resident waves; the pow per column unit stays cheap beside four rows of sin / cos.
  constexpr int cap_wg = 16384, min_rows = 4;
  int gy = cap_wg / gx;
  if (gy > (seq_len + min_rows - 1) / min_rows) gy = (seq_len + min_rows - 1) / min_rows;
  if (gy < 1) gy = 1;
  if (gy > 65535) gy = 65535;
  CLN_LAUNCH((rope_kernel<PAIRS>), dim3(gx, gy), dim3(256), 0, st, (const float*)x, (float*)out, seq_len,
                     half_hidden, ref_quirk, cln_stream_nt(8LL * seq_len * hidden));
  return cln_check_launch();
}

}  // namespace

// (x, out, seq_len, hidden, ref_quirk, stream) -- reference `void rope_*(Tensor x, Tensor out)`
CLN_API int rope_f32(const void* x, void* out, int seq_len, int hidden, int ref_quirk, void* stream) {
  return launch_rope<1>(x, out, seq_len, hidden, ref_quirk, (hipStream_t)stream);
}
CLN_API int rope_f32_v2(const void* x, void* out, int seq_len, int hidden, int ref_quirk, void* stream) {
  return launch_rope<1>(x, out, seq_len, hidden, ref_quirk, (hipStream_t)stream);
}
CLN_API int rope_f32x4_pack(const void* x, void* out, int seq_len, int hidden, int ref_quirk, void* stream) {
  return launch_rope<2>(x, out, seq_len, hidden, ref_quirk, (hipStream_t)stream);
}
