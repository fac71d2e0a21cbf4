// Round 6: the vendor yardsticks beat two of the library's streams -- torch.add(out=) runs [8192,8192] f32 in 135 us where
// elementwise_add_f32x4 takes 148 (0.91), rocprim::reduce is 2-6 % ahead of block_all_reduce_sum_* -- so which FORM of the stream is it?
// c = a + b over f32x4 and sum(x) over f32x4 / f16x8, each over ROTATING buffer sets (> 1 GiB in all: HBM, not the Infinity Cache):
//   add:    G  = the library's form: capped grid (256 CUs x 32 workgroups of 256), grid-stride, one 16-byte load pair per trip      [nt store: Gn]
//           O  = one pack per thread, no loop (grid = packs / 256)
//           Bk = block-contiguous, k packs per thread (a workgroup owns 256 k consecutive packs: loads first, then stores), no loop  [nt store: Bkn]
//           Uk = grid-stride with k load pairs in flight per trip                                                                     [nt store: Ukn]
//   reduce: P  = the library's form: 256 workgroups x 1024 threads, grid-stride, 8 loads in flight, one atomic per workgroup
//           Ck = W workgroups x 256 threads, each workgroup walks consecutive chunks of 256 k packs (k loads in flight), one atomic per workgroup
//   hipcc --offload-arch=gfx950 -O3 stream_forms.hip -o stream_forms && ./stream_forms
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdio.h>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void st(f4* p, f4 v, bool nt) {
  if (nt) __builtin_nontemporal_store(v, p);
  else *p = v;
}

template <bool NT>
__global__ __launch_bounds__(256) void add_G(const f4* __restrict__ a, const f4* __restrict__ b, f4* __restrict__ c, long long n) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) st(c + i, a[i] + b[i], NT);
}
template <bool NT>
__global__ __launch_bounds__(256) void add_O(const f4* __restrict__ a, const f4* __restrict__ b, f4* __restrict__ c, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) st(c + i, a[i] + b[i], NT);
}
template <int K, bool NT>
__global__ __launch_bounds__(256) void add_B(const f4* __restrict__ a, const f4* __restrict__ b, f4* __restrict__ c, long long n) {
  const long long base = (long long)blockIdx.x * (256 * K) + threadIdx.x;
  f4 x[K], y[K];
#pragma unroll
  for (int k = 0; k < K; ++k) x[k] = a[base + k * 256];
#pragma unroll
  for (int k = 0; k < K; ++k) y[k] = b[base + k * 256];
#pragma unroll
  for (int k = 0; k < K; ++k) st(c + base + k * 256, x[k] + y[k], NT);
}
template <int K, bool NT>
__global__ __launch_bounds__(256) void add_U(const f4* __restrict__ a, const f4* __restrict__ b, f4* __restrict__ c, long long n) {
  const long long stride = (long long)gridDim.x * 256;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  for (; i + (K - 1) * stride < n; i += K * stride) {
    f4 x[K], y[K];
#pragma unroll
    for (int k = 0; k < K; ++k) x[k] = a[i + k * stride], y[k] = b[i + k * stride];
#pragma unroll
    for (int k = 0; k < K; ++k) st(c + i + k * stride, x[k] + y[k], NT);
  }
  for (; i < n; i += stride) st(c + i, a[i] + b[i], NT);
}

// y = f(x), 1 read + 1 write (the activation / row-kernel traffic): the same three walks
template <bool NT>
__global__ __launch_bounds__(256) void sc_G(const f4* __restrict__ a, f4* __restrict__ c, long long n) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) st(c + i, a[i] * 2.0f, NT);
}
template <int K, bool NT>
__global__ __launch_bounds__(256) void sc_B(const f4* __restrict__ a, f4* __restrict__ c, long long n) {
  const long long base = (long long)blockIdx.x * (256 * K) + threadIdx.x;
  f4 x[K];
#pragma unroll
  for (int k = 0; k < K; ++k) x[k] = a[base + k * 256];
#pragma unroll
  for (int k = 0; k < K; ++k) st(c + base + k * 256, x[k] * 2.0f, NT);
}

template <typename V>
__device__ __forceinline__ float vsum(const V& v);
template <>
__device__ __forceinline__ float vsum<f4>(const f4& v) { return (v[0] + v[1]) + (v[2] + v[3]); }
template <>
__device__ __forceinline__ float vsum<h8>(const h8& v) {
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) s += (float)v[e];
  return s;
}
template <int NT_>
__device__ __forceinline__ void block_finish(float s, float* y) {
  __shared__ float sc[NT_ / 64];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
  if ((threadIdx.x & 63) == 0) sc[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < NT_ / 64; ++w) t += sc[w];
    atomicAdd(y, t);
  }
}
template <typename V>
__global__ __launch_bounds__(1024) void red_P(const V* __restrict__ x, float* __restrict__ y, long long n) {
  const long long stride = (long long)gridDim.x * 1024;
  long long i = (long long)blockIdx.x * 1024 + threadIdx.x;
  float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  for (; i + 7 * stride < n; i += 8 * stride) {
    V p[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) p[u] = x[i + u * stride];
    s0 += vsum(p[0]) + vsum(p[4]), s1 += vsum(p[1]) + vsum(p[5]), s2 += vsum(p[2]) + vsum(p[6]), s3 += vsum(p[3]) + vsum(p[7]);
  }
  for (; i < n; i += stride) s0 += vsum(x[i]);
  block_finish<1024>((s0 + s1) + (s2 + s3), y);
}
// Dk: NT_ threads per workgroup, consecutive chunks of NT_ * K packs per trip, k loads in flight, the partial STORED to y[blockIdx] (no atomic: the stream alone)
template <typename V, int K, int NT_>
__global__ __launch_bounds__(NT_) void red_D(const V* __restrict__ x, float* __restrict__ y, long long n) {
  const long long chunk = NT_ * K, stride = (long long)gridDim.x * chunk;
  float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  for (long long base = (long long)blockIdx.x * chunk + threadIdx.x; base + (K - 1) * NT_ < n; base += stride) {
    V p[K];
#pragma unroll
    for (int k = 0; k < K; ++k) p[k] = x[base + k * NT_];
#pragma unroll
    for (int k = 0; k < K; k += 4) s0 += vsum(p[k]), s1 += vsum(p[k + 1]), s2 += vsum(p[k + 2]), s3 += vsum(p[k + 3]);
  }
  float s = (s0 + s1) + (s2 + s3);
  __shared__ float sc[NT_ / 64];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
  if ((threadIdx.x & 63) == 0) sc[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < NT_ / 64; ++w) t += sc[w];
    y[blockIdx.x] = t;
  }
}
// Sk: the library's STRIDED walk (pack i + u * grid * NT_), the partial stored to y[blockIdx] -- the same finish as Dk
template <typename V, int K, int NT_>
__global__ __launch_bounds__(NT_) void red_S(const V* __restrict__ x, float* __restrict__ y, long long n) {
  const long long stride = (long long)gridDim.x * NT_;
  long long i = (long long)blockIdx.x * NT_ + threadIdx.x;
  float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  for (; i + (K - 1) * stride < n; i += K * stride) {
    V p[K];
#pragma unroll
    for (int u = 0; u < K; ++u) p[u] = x[i + u * stride];
#pragma unroll
    for (int k = 0; k < K; k += 4) s0 += vsum(p[k]), s1 += vsum(p[k + 1]), s2 += vsum(p[k + 2]), s3 += vsum(p[k + 3]);
  }
  float s = (s0 + s1) + (s2 + s3);
  __shared__ float sc[NT_ / 64];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
  if ((threadIdx.x & 63) == 0) sc[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < NT_ / 64; ++w) t += sc[w];
    y[blockIdx.x] = t;
  }
}
template <typename V, int K>
__global__ __launch_bounds__(256) void red_C(const V* __restrict__ x, float* __restrict__ y, long long n) {
  const long long chunk = 256 * K, stride = (long long)gridDim.x * chunk;
  float s0 = 0, s1 = 0;
  for (long long base = (long long)blockIdx.x * chunk + threadIdx.x; base + (K - 1) * 256 < n; base += stride) {
    V p[K];
#pragma unroll
    for (int k = 0; k < K; ++k) p[k] = x[base + k * 256];
#pragma unroll
    for (int k = 0; k < K; k += 2) s0 += vsum(p[k]), s1 += vsum(p[k + 1]);
  }
  block_finish<256>(s0 + s1, y);
}

struct Timer {
  hipEvent_t e0, e1;
  Timer() { (void)hipEventCreate(&e0), (void)hipEventCreate(&e1); }
  template <typename F>
  float us(F&& launch_rotation, int nsets, int target_launches) {  // launch_rotation(set index)
    for (int i = 0; i < 2 * nsets; ++i) launch_rotation(i % nsets);
    (void)hipDeviceSynchronize();
    const int reps = (target_launches + nsets - 1) / nsets;
    (void)hipEventRecord(e0);
    for (int r = 0; r < reps; ++r)
      for (int s = 0; s < nsets; ++s) launch_rotation(s);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / (reps * nsets);
  }
};

int main() {
  Timer T;
  char* pool;
  const size_t POOL = 3ull << 30;  // 3 GiB: three rotating sets of the largest problem
  if (hipMalloc(&pool, POOL) != hipSuccess) return 1;
  (void)hipMemset(pool, 0x3c, POOL);
  float* y;
  (void)hipMalloc(&y, 1 << 20);
  (void)hipMemset(y, 0, 1 << 20);
  for (long long side : {2048, 4096, 8192}) {
    const long long n = side * side / 4, bytes = n * 16;  // f32x4 packs
    const int nsets = (int)(POOL / (3 * bytes)) > 24 ? 24 : (int)(POOL / (3 * bytes));
    auto A = [&](int s) { return (const f4*)(pool + (size_t)s * 3 * bytes); };
    auto B = [&](int s) { return (const f4*)(pool + (size_t)s * 3 * bytes + bytes); };
    auto C = [&](int s) { return (f4*)(pool + (size_t)s * 3 * bytes + 2 * bytes); };
    const int cap = 256 * 32, gG = (int)((n + 255) / 256 > cap ? cap : (n + 255) / 256);
    const int L = side >= 8192 ? 60 : 300;
    printf("FORMS add f32x4 [%lld,%lld] (%d rotating sets, %.0f MB each): ", side, side, nsets, 3.0 * bytes / 1e6);
#define ROW(tag, launch) { const float us = T.us([&](int s) { launch; }, nsets, L); printf(" %s %.1f us %.0f GB/s |", tag, us, 3.0 * bytes / us * 1e-3); }
    ROW("G", (add_G<false><<<gG, 256>>>(A(s), B(s), C(s), n)))
    ROW("Gn", (add_G<true><<<gG, 256>>>(A(s), B(s), C(s), n)))
    ROW("O", (add_O<false><<<(int)(n / 256), 256>>>(A(s), B(s), C(s), n)))
    ROW("On", (add_O<true><<<(int)(n / 256), 256>>>(A(s), B(s), C(s), n)))
    ROW("B2", (add_B<2, false><<<(int)(n / 512), 256>>>(A(s), B(s), C(s), n)))
    ROW("B4", (add_B<4, false><<<(int)(n / 1024), 256>>>(A(s), B(s), C(s), n)))
    ROW("B4n", (add_B<4, true><<<(int)(n / 1024), 256>>>(A(s), B(s), C(s), n)))
    ROW("B8", (add_B<8, false><<<(int)(n / 2048), 256>>>(A(s), B(s), C(s), n)))
    ROW("B8n", (add_B<8, true><<<(int)(n / 2048), 256>>>(A(s), B(s), C(s), n)))
    ROW("U2", (add_U<2, false><<<gG, 256>>>(A(s), B(s), C(s), n)))
    ROW("U4", (add_U<4, false><<<gG, 256>>>(A(s), B(s), C(s), n)))
    ROW("U4n", (add_U<4, true><<<gG, 256>>>(A(s), B(s), C(s), n)))
    ROW("U4 grid 2048", (add_U<4, false><<<2048, 256>>>(A(s), B(s), C(s), n)))
    ROW("G grid 4096", (add_G<false><<<4096, 256>>>(A(s), B(s), C(s), n)))
    ROW("G grid 16384", (add_G<false><<<(n / 256 > 16384 ? 16384 : (int)(n / 256)), 256>>>(A(s), B(s), C(s), n)))
    printf("\n");
#undef ROW
  }
  for (long long side : {2048, 4096, 8192}) {
    const long long n = side * side / 4, bytes = n * 16;
    const int nsets = (int)(POOL / (2 * bytes)) > 24 ? 24 : (int)(POOL / (2 * bytes));
    auto A = [&](int s) { return (const f4*)(pool + (size_t)s * 2 * bytes); };
    auto C = [&](int s) { return (f4*)(pool + (size_t)s * 2 * bytes + bytes); };
    const int cap = 256 * 32, gG = (int)((n + 255) / 256 > cap ? cap : (n + 255) / 256);
    const int L = side >= 8192 ? 80 : 300;
    printf("FORMS scale f32x4 1R+1W [%lld,%lld] (%d rotating sets, %.0f MB each): ", side, side, nsets, 2.0 * bytes / 1e6);
#define ROW(tag, launch) { const float us = T.us([&](int s) { launch; }, nsets, L); printf(" %s %.1f us %.0f GB/s |", tag, us, 2.0 * bytes / us * 1e-3); }
    ROW("G", (sc_G<false><<<gG, 256>>>(A(s), C(s), n)))
    ROW("Gn", (sc_G<true><<<gG, 256>>>(A(s), C(s), n)))
    ROW("B1", (sc_B<1, false><<<(int)(n / 256), 256>>>(A(s), C(s), n)))
    ROW("B1n", (sc_B<1, true><<<(int)(n / 256), 256>>>(A(s), C(s), n)))
    ROW("B2", (sc_B<2, false><<<(int)(n / 512), 256>>>(A(s), C(s), n)))
    ROW("B4", (sc_B<4, false><<<(int)(n / 1024), 256>>>(A(s), C(s), n)))
    ROW("B4n", (sc_B<4, true><<<(int)(n / 1024), 256>>>(A(s), C(s), n)))
    ROW("B8", (sc_B<8, false><<<(int)(n / 2048), 256>>>(A(s), C(s), n)))
    ROW("hipMemcpyDtoD", ((void)hipMemcpyDtoDAsync((hipDeviceptr_t)C(s), (hipDeviceptr_t)A(s), bytes, 0)))
    printf("\n");
#undef ROW
  }
  for (int half = 0; half < 2; ++half)
    for (long long side : {4096, 8192}) {
      const long long bytes = side * side * (half ? 2 : 4), n = bytes / 16;
      const int nsets = (int)(POOL / bytes) > 32 ? 32 : (int)(POOL / bytes);
      const int L = side >= 8192 ? 100 : 400;
      printf("FORMS reduce %s [%lld,%lld] (%d rotating sets, %.0f MB each): ", half ? "f16x8" : "f32x4", side, side, nsets, bytes / 1e6);
#define ROW(tag, KERN, grid, block) { const float us = half ? T.us([&](int s) { KERN<h8><<<grid, block>>>((const h8*)(pool + (size_t)s * bytes), y, n); }, nsets, L) \
                                                        : T.us([&](int s) { KERN<f4><<<grid, block>>>((const f4*)(pool + (size_t)s * bytes), y, n); }, nsets, L); \
                                  printf(" %s %.1f us %.0f GB/s |", tag, us, bytes / us * 1e-3); }
#define ROWC(tag, K, grid) { const float us = half ? T.us([&](int s) { red_C<h8, K><<<grid, 256>>>((const h8*)(pool + (size_t)s * bytes), y, n); }, nsets, L) \
                                                  : T.us([&](int s) { red_C<f4, K><<<grid, 256>>>((const f4*)(pool + (size_t)s * bytes), y, n); }, nsets, L); \
                             printf(" %s %.1f us %.0f GB/s |", tag, us, bytes / us * 1e-3); }
      ROW("P 256x1024", red_P, 256, 1024)
      ROW("P 512x1024", red_P, 512, 1024)
      ROWC("C4 x1024", 4, 1024) ROWC("C4 x2048", 4, 2048) ROWC("C8 x1024", 8, 1024) ROWC("C8 x2048", 8, 2048) ROWC("C8 x4096", 8, 4096)
      ROWC("C16 x1024", 16, 1024) ROWC("C16 x2048", 16, 2048)
      printf("\n");
      printf("FORMS reduce %s [%lld,%lld] partial stored, no atomic: ", half ? "f16x8" : "f32x4", side, side);
#define ROWD(tag, KERN, K, NTT, grid) { const float us = half ? T.us([&](int s) { KERN<h8, K, NTT><<<grid, NTT>>>((const h8*)(pool + (size_t)s * bytes), y, n); }, nsets, L) \
                                                              : T.us([&](int s) { KERN<f4, K, NTT><<<grid, NTT>>>((const f4*)(pool + (size_t)s * bytes), y, n); }, nsets, L); \
                                        printf(" %s %.1f us %.0f GB/s |", tag, us, bytes / us * 1e-3); }
      ROWD("S8 256x1024", red_S, 8, 1024, 256) ROWD("S4 256x1024", red_S, 4, 1024, 256) ROWD("S8 512x512", red_S, 8, 512, 512) ROWD("S8 1024x256", red_S, 8, 256, 1024) ROWD("S16 256x1024", red_S, 16, 1024, 256)
      ROWD("D8 256x1024", red_D, 8, 1024, 256) ROWD("D4 256x1024", red_D, 4, 1024, 256) ROWD("D8 512x512", red_D, 8, 512, 512) ROWD("D8 1024x256", red_D, 8, 256, 1024) ROWD("D4 1024x256", red_D, 4, 256, 1024)
      ROWD("D16 256x1024", red_D, 16, 1024, 256) ROWD("D8 2048x256", red_D, 8, 256, 2048) ROWD("D8 512x1024", red_D, 8, 1024, 512) ROWD("D4 2048x256", red_D, 4, 256, 2048) ROWD("D4 4096x256", red_D, 4, 256, 4096)
      printf("\n");
#undef ROWD
#undef ROW
#undef ROWC
    }
  return 0;
}
