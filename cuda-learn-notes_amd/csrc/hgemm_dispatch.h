// Host-side dispatch into the explicit instantiations of hgemm::hgemm_ring_kernel.
#pragma once
#include "common.h"
namespace hgemm {
enum TileId { T128 = 0, T256 = 1, T256x128 = 2, T128x256 = 3, T256W4 = 4, T128W8 = 5, T64x128 = 6 };  // T256W4: 256x256 tile, 4 waves x 128x128
// stages in [2,5]; BK (64 or 32) is chosen so that stages * stage_bytes fits the 160 KiB LDS and
// K % BK == 0. Returns CLN_ERR_UNSUPPORTED when M/N/K do not divide the tile.
int ring_dispatch_nn(int tile, const void* a, const void* b, void* c, int M, int N, int K, int stages, int swizzle,
                     int swizzle_stride, hipStream_t stream);
int ring_dispatch_tn(int tile, const void* a, const void* b, void* c, int M, int N, int K, int stages, int swizzle,
                     int swizzle_stride, hipStream_t stream);
// direct (tile, BK, stages) selection for the tuning harness; bk in {32,64}
int ring_exact_nn(int tile, int bk, int stages, const void* a, const void* b, void* c, int M, int N, int K,
                  int swizzle, int swizzle_stride, hipStream_t stream);
int ring_exact_tn(int tile, int bk, int stages, const void* a, const void* b, void* c, int M, int N, int K,
                  int swizzle, int swizzle_stride, hipStream_t stream);
}  // namespace hgemm
