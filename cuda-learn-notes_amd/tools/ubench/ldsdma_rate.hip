// How fast can ONE CU pull L2-resident data into its LDS? (gfx950) The one-wave-per-SIMD HGEMM moves 64 KiB per K tile and CU
// in 1.4 us (~22 B/clk/CU), the ring attention kernel for D = 1024 64 KiB per 16 keys in ~2800 clocks (~23 B/clk/CU): is
// that a ceiling of the L2 -> LDS path or of those kernels? Every workgroup (one per CU) streams the same SPAN bytes of
// its own / of its XCD's region into its LDS again and again:
//   mode 0: global_load_lds_dwordx4 (LDS-DMA, 1 KiB per wave-instruction), IN_FLIGHT pieces per wave outstanding
//   mode 1: global_load_dwordx4 into registers (same path up to the L1), results xor-ed
//   mode 2: LDS-DMA with the K-tile source swizzle of the attention kernels (16-byte chunks permuted inside 256 B)
// Prints bytes / clock / CU from s_memtime-free wall time at the measured effective clock (hipEvents + GRBM not
// available here: uses wall time and reports GB/s per CU as well).
//   hipcc --offload-arch=gfx950 -O3 ldsdma_rate.hip -o ldsdma_rate && ./ldsdma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// NW waves per workgroup; each iteration the workgroup fetches SPAN bytes (SPAN / 1024 / NW pieces per wave)
template <int MODE, int NW, int SPAN>
__global__ __launch_bounds__(NW * 64) void k(const char* __restrict__ src, unsigned* out, int iters, int share) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // share = 1: all workgroups of an XCD (blockIdx & 7) read the same region; 0: every workgroup its own
  const char* base = src + (size_t)(share ? (blockIdx.x & 7) : blockIdx.x) * SPAN;
  constexpr int PPW = SPAN / 1024 / NW;
  const unsigned lds0 = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) char*)smem;
  unsigned voff = lane * 16;
  if (MODE == 2) voff = (unsigned)((((lane & 15) ^ ((lane >> 4) * 5 + wave)) & 15) << 4) + (lane >> 4) * 256;
  u4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int piece = i * NW + wave;
      if (MODE == 1) {
        const u4 v = *reinterpret_cast<const u4*>(base + piece * 1024 + voff);
        acc ^= v;
      } else {
        glds16(base + piece * 1024, voff, lds0 + piece * 1024);
      }
    }
    if (MODE != 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");  // the previous sweep has landed, this one is in flight
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3] ^ *reinterpret_cast<unsigned*>(smem + 64);
}

template <int MODE, int NW, int SPAN>
void run(const char* tag, const char* src, unsigned* out, int share) {
  const int iters = 2000, grid = 256;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE, NW, SPAN>), hipFuncAttributeMaxDynamicSharedMemorySize, SPAN);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NW, SPAN>), dim3(grid), dim3(NW * 64), SPAN, 0, src, out, iters, share);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes_per_cu = (double)iters * SPAN;
  printf("%-34s share=%d  %7.3f ms  %6.1f GB/s per CU  %5.1f B/clk/CU at 2.1 GHz  chip %5.2f TB/s\n", tag, share, ms,
         bytes_per_cu / ms * 1e-6, bytes_per_cu / (ms * 1e-3) / 2.1e9, bytes_per_cu * grid / ms * 1e-9);
}

int main() {
  char* src;
  unsigned* out;
  hipMalloc(&src, (size_t)256 * 131072);
  hipMemset(src, 1, (size_t)256 * 131072);
  hipMalloc(&out, 4096);
  for (int share = 1; share >= 0; --share) {
    run<0, 4, 65536>("lds-dma 4 waves 64 KiB", src, out, share);
    run<0, 8, 65536>("lds-dma 8 waves 64 KiB", src, out, share);
    run<0, 8, 131072>("lds-dma 8 waves 128 KiB", src, out, share);
    run<0, 4, 32768>("lds-dma 4 waves 32 KiB", src, out, share);
    run<0, 16, 65536>("lds-dma 16 waves 64 KiB", src, out, share);
    run<2, 8, 65536>("lds-dma swizzled src 8 waves 64 KiB", src, out, share);
    run<1, 8, 65536>("global_load x4 -> VGPR 8 waves", src, out, share);
    run<1, 16, 65536>("global_load x4 -> VGPR 16 waves", src, out, share);
  }
  return 0;
}
