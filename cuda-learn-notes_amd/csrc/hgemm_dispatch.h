// Host-side dispatch into the explicit instantiations of hgemm::hgemm_ring_kernel.
#pragma once
#include "common.h"
namespace hgemm {
enum TileId { T128 = 0, T256 = 1, T256x128 = 2, T128x256 = 3, T256W4 = 4, T128W8 = 5, T64x128 = 6, T64x64 = 7, T64x64W2 = 8 };  // T256W4: 256x256 tile, 4 waves x 128x128
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
// (BK, stages) the ring dispatcher runs for a BM x BN tile: BK = 64 when K allows it and `stages` buffers fit the
// 160 KiB LDS, else BK = 32 with the stage count trimmed to fit. One definition for the launcher (hgemm_ring_impl.inc)
// and for cln_describe().
inline void ring_pick(int BM, int BN, int K, int& S, int& BK) {
  // reference bindings fall back to 2 stages for an unknown count (hgemm_mma_stage.cu:2380-2454 default case)
  if (S < 2 || S > 5) S = 2;
  // The 4-wave tiles (128x128, 64x128, 64x64) run two or more workgroups per CU; a 64-deep ring that grows past 80 KiB leaves ONE: 128x128 at
  // 3 / 4 / 5 stages 842 / 826 / 825 TF against 1005 at 2 (4096^3), 64x128 at 4 / 5 stages 630 / 615 against 787-832. The 32-deep ring of the
  // same depth keeps two on the CU: 128x128 937 / 940 / 935, 64x128 645 / 627 at 2048^3 where the 64-deep one has 571 / 560
  // (profiles/r04_hgemm_ring_bk_stages_probe.log).
  const bool keep_two_per_cu = BM + BN <= 256 && S * (BM + BN) * 64 * 2 > 80 * 1024;
  if (K % 64 == 0 && !keep_two_per_cu && S * (BM + BN) * 64 * 2 <= 160 * 1024) {
    BK = 64;
    return;
  }
  while (S > 2 && S * (BM + BN) * 32 * 2 > 160 * 1024) --S;
  BK = 32;
}
inline void tile_dims(int tile, int& BM, int& BN, int& waves) {
  switch (tile) {
    case T256: BM = 256, BN = 256, waves = 8; break;
    case T256x128: BM = 256, BN = 128, waves = 8; break;
    case T128x256: BM = 128, BN = 256, waves = 8; break;
    case T256W4: BM = 256, BN = 256, waves = 4; break;
    case T128W8: BM = 128, BN = 128, waves = 8; break;
    case T64x128: BM = 64, BN = 128, waves = 4; break;
    case T64x64: BM = 64, BN = 64, waves = 4; break;
    case T64x64W2: BM = 64, BN = 64, waves = 2; break;
    default: BM = 128, BN = 128, waves = 4; break;
  }
}
}  // namespace hgemm
