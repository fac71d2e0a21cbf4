// Past the 256 MB MALL: do non-temporal loads / stores help a read+write stream? (gfx950)
// The bandwidth rungs whose working set exceeds the MALL (8192^2 f32 softmax / rope, the three-tensor f16 add: >= 400 MB
// per launch) sit at 0.65-0.68 of 8 TB/s, the ones below it at 0.79-0.86 (profiles/r03_bw_rocprof.json). y = 2x over f32x4,
// grid-stride, UNROLL independent 16-byte loads in flight per lane, four forms of the memory instructions:
//   plain | nt loads | nt stores | both        (__builtin_nontemporal_load / _store -> global_load / global_store ... nt)
//   hipcc --offload-arch=gfx950 -O3 stream_nt.hip -o stream_nt && ./stream_nt
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NTL, bool NTS, int UNROLL>
__global__ __launch_bounds__(256) void scale_k(const f4* __restrict__ x, f4* __restrict__ y, long long nvec) {
  const long long stride = (long long)gridDim.x * 256;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < nvec; i += UNROLL * stride) {
    f4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = NTL ? __builtin_nontemporal_load(x + i + u * stride) : x[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const f4 r = v[u] * 2.0f;
      if (NTS) __builtin_nontemporal_store(r, y + i + u * stride);
      else y[i + u * stride] = r;
    }
  }
  for (; i < nvec; i += stride) y[i] = x[i] * 2.0f;
}

template <bool NTL, bool NTS, int UNROLL>
float run(const f4* x, f4* y, long long nvec, int grid, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) scale_k<NTL, NTS, UNROLL><<<grid, 256>>>(x, y, nvec);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) scale_k<NTL, NTS, UNROLL><<<grid, 256>>>(x, y, nvec);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / iters;
}

int main() {
  f4 *x, *y;
  const long long maxb = 1ll << 30;
  (void)hipMalloc(&x, maxb), (void)hipMalloc(&y, maxb);
  (void)hipMemset(x, 0x3c, maxb), (void)hipMemset(y, 0, maxb);
  const long long per_tensor_mb[] = {64, 128, 256, 512, 1024};
  for (long long mb : per_tensor_mb) {
    const long long bytes = mb << 20, nvec = bytes / 16;
    for (int grid : {256 * 8, 256 * 16, 256 * 32}) {
      const int iters = mb >= 512 ? 50 : 200;
      const float t[4] = {run<false, false, 4>(x, y, nvec, grid, iters), run<true, false, 4>(x, y, nvec, grid, iters),
                          run<false, true, 4>(x, y, nvec, grid, iters), run<true, true, 4>(x, y, nvec, grid, iters)};
      const float t8 = run<false, false, 8>(x, y, nvec, grid, iters), t8nt = run<true, true, 8>(x, y, nvec, grid, iters);
      printf("NT %4lld MB in + %4lld MB out, grid %5d: plain %7.2f us %6.0f GB/s | nt-load %6.0f | nt-store %6.0f | both %6.0f | 8 in flight: plain %6.0f both %6.0f\n",
             mb, mb, grid, t[0], 2.0 * bytes / t[0] * 1e-3, 2.0 * bytes / t[1] * 1e-3, 2.0 * bytes / t[2] * 1e-3, 2.0 * bytes / t[3] * 1e-3,
             2.0 * bytes / t8 * 1e-3, 2.0 * bytes / t8nt * 1e-3);
    }
  }
  return 0;
}
