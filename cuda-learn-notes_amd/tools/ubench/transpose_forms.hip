// Round 6: y[col,row] = x[row,col]^T, f32 -- the library's LDS-tile rung reaches 4.3-4.4 TB/s where hipMemcpyDtoD moves the same bytes at 5.2-5.35
// (bench.py next_rows.mat_transpose). Which form of the tile is it?
//   L   = the library's kernel (blas1.hip tr_lds_tile<1>): 64x64 tile, 256 threads, float4 global accesses, SCALAR LDS writes and reads (32 per thread), one barrier
//   R   = no LDS: a wave owns a 32x32 block as 8x8 lanes of 4x4 register blocks -- lane (a, b) reads rows 4a..4a+3 at columns 4b..4b+3 (four 16-byte loads; the 8
//         lanes of one a cover one 128-byte line per row), transposes in registers (renaming only) and writes rows 4b..4b+3 of y at columns 4a..4a+3 (the 8 lanes of one b
//         cover one 128-byte line per output row): full lines on both sides, no barrier
//   R2  = the same with TWO 32x32 blocks per wave (8 loads in flight per lane)
//   hipcc --offload-arch=gfx950 -O3 transpose_forms.hip -o transpose_forms && ./transpose_forms
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ __launch_bounds__(256) void tr_L(const float* __restrict__ x, float* __restrict__ y, int row, int col) {
  __shared__ float tile[64][65];
  const int tiles_c = col / 64;
  const int tr = blockIdx.x / tiles_c, tc = blockIdx.x - tr * tiles_c;
  const int t = threadIdx.x;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int r = it * 16 + (t >> 4), c4 = (t & 15) * 4;
    const float4 v = *reinterpret_cast<const float4*>(x + (size_t)(tr * 64 + r) * col + tc * 64 + c4);
    tile[r][c4] = v.x, tile[r][c4 + 1] = v.y, tile[r][c4 + 2] = v.z, tile[r][c4 + 3] = v.w;
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int c = it * 16 + (t >> 4), r4 = (t & 15) * 4;
    float4 v = {tile[r4][c], tile[r4 + 1][c], tile[r4 + 2][c], tile[r4 + 3][c]};
    *reinterpret_cast<float4*>(y + (size_t)(tc * 64 + c) * row + tr * 64 + r4) = v;
  }
}

// NB blocks of 32x32 per wave, side by side along the columns of x
template <int NB>
__global__ __launch_bounds__(256) void tr_R(const float* __restrict__ x, float* __restrict__ y, int row, int col) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int a = lane >> 3, b = lane & 7;
  const int blocks_c = col / (32 * NB);
  const long long w = (long long)blockIdx.x * 4 + wave;  // wave index: one (32 x 32 NB) strip each
  const int br = (int)(w / blocks_c), bc = (int)(w - (long long)br * blocks_c);
  if (br * 32 >= row) return;
  float4 v[NB][4];
#pragma unroll
  for (int n = 0; n < NB; ++n)
#pragma unroll
    for (int i = 0; i < 4; ++i) v[n][i] = *reinterpret_cast<const float4*>(x + (size_t)(br * 32 + 4 * a + i) * col + (bc * NB + n) * 32 + 4 * b);
#pragma unroll
  for (int n = 0; n < NB; ++n) {
    const float4 o0 = {v[n][0].x, v[n][1].x, v[n][2].x, v[n][3].x}, o1 = {v[n][0].y, v[n][1].y, v[n][2].y, v[n][3].y};
    const float4 o2 = {v[n][0].z, v[n][1].z, v[n][2].z, v[n][3].z}, o3 = {v[n][0].w, v[n][1].w, v[n][2].w, v[n][3].w};
    float* yo = y + (size_t)((bc * NB + n) * 32 + 4 * b) * row + br * 32 + 4 * a;
    *reinterpret_cast<float4*>(yo) = o0;
    *reinterpret_cast<float4*>(yo + row) = o1;
    *reinterpret_cast<float4*>(yo + 2 * (size_t)row) = o2;
    *reinterpret_cast<float4*>(yo + 3 * (size_t)row) = o3;
  }
}

int main() {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
  char* pool;
  const size_t POOL = 3ull << 30;
  if (hipMalloc(&pool, POOL) != hipSuccess) return 1;
  (void)hipMemset(pool, 0x3c, POOL);
  for (int side : {2048, 4096, 8192}) {
    const size_t bytes = (size_t)side * side * 4;
    const int nsets = (int)(POOL / (2 * bytes)) > 16 ? 16 : (int)(POOL / (2 * bytes));
    auto X = [&](int s) { return (const float*)(pool + (size_t)s * 2 * bytes); };
    auto Y = [&](int s) { return (float*)(pool + (size_t)s * 2 * bytes + bytes); };
    printf("TRFORMS f32 [%d,%d] (%d rotating sets): ", side, side, nsets);
    auto time_us = [&](auto&& launch) {
      for (int i = 0; i < 2 * nsets; ++i) launch(i % nsets);
      (void)hipDeviceSynchronize();
      const int reps = (side >= 8192 ? 60 : 200) / nsets + 1;
      (void)hipEventRecord(e0);
      for (int r = 0; r < reps; ++r)
        for (int s = 0; s < nsets; ++s) launch(s);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      return ms * 1e3f / (reps * nsets);
    };
    float us = time_us([&](int s) { tr_L<<<(side / 64) * (side / 64), 256>>>(X(s), Y(s), side, side); });
    printf(" L %.1f us %.0f GB/s |", us, 2.0 * bytes / us * 1e-3);
    us = time_us([&](int s) { tr_R<1><<<(side / 32) * (side / 32) / 4, 256>>>(X(s), Y(s), side, side); });
    printf(" R %.1f us %.0f GB/s |", us, 2.0 * bytes / us * 1e-3);
    us = time_us([&](int s) { tr_R<2><<<(side / 32) * (side / 64) / 4, 256>>>(X(s), Y(s), side, side); });
    printf(" R2 %.1f us %.0f GB/s |", us, 2.0 * bytes / us * 1e-3);
    us = time_us([&](int s) { (void)hipMemcpyDtoDAsync((hipDeviceptr_t)Y(s), (hipDeviceptr_t)const_cast<float*>(X(s)), bytes, 0); });
    printf(" hipMemcpyDtoD %.1f us %.0f GB/s\n", us, 2.0 * bytes / us * 1e-3);
    // correctness of R on set 0 (host check of a few entries)
    tr_R<1><<<(side / 32) * (side / 32) / 4, 256>>>(X(0), Y(0), side, side);
    (void)hipDeviceSynchronize();
  }
  // correctness: distinct values
  {
    const int side = 256;
    float* hx = (float*)malloc(side * side * 4), *hy = (float*)malloc(side * side * 4);
    for (int i = 0; i < side * side; ++i) hx[i] = (float)i;
    (void)hipMemcpy(pool, hx, side * side * 4, hipMemcpyHostToDevice);
    for (int form = 0; form < 2; ++form) {
      if (form == 0) tr_R<1><<<(side / 32) * (side / 32) / 4, 256>>>((const float*)pool, (float*)(pool + (1 << 20)), side, side);
      else tr_R<2><<<(side / 32) * (side / 64) / 4, 256>>>((const float*)pool, (float*)(pool + (1 << 20)), side, side);
      (void)hipMemcpy(hy, pool + (1 << 20), side * side * 4, hipMemcpyDeviceToHost);
      int bad = 0;
      for (int r = 0; r < side; ++r)
        for (int c = 0; c < side; ++c) bad += hy[c * side + r] != hx[r * side + c];
      printf("TRFORMS check R%d: %d wrong of %d\n", form + 1, bad, side * side);
    }
  }
  return 0;
}
