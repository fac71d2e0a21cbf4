// Vendor comparator for the attention rungs: AMD's composable-kernel "ck_tile" FMHA forward kernels, instantiated from the
// headers the ROCm image ships (/opt/rocm/include/ck_tile/ops/fmha). This is the kernel family FlashAttention-2-ROCm (the
// `flash_attn` package's CK backend) and aiter dispatch to on MI300 / MI355X, i.e. the comparator north_star names
// ("FA2 fwd D=64 >= FlashAttention-2-ROCm") and the reference script times as `flash_attn_func`
// (kernels/flash-attn/flash_attn_mma.py:10, :591) -- neither Python package is importable in the image, the C++ templates are.
// NOT a product path and NOT a reference name: it lives in libcln_amd_vendor.so beside the rocBLAS / hipBLASLt rows.
//
// Instances (the tile configurations of ck_tile's own codegen tables for fp16, batch mode, no mask / bias / dropout / LSE):
//   D = 64 : FmhaFwdKernel + BlockFmhaPipelineQRKSVSAsync, block tile 128 x 64 x 32 | 64 x 32 (bk0max 64), 4 waves, 32x32x16 MFMA
//   D = 128: (a) the same pipeline with the 128 x 128 x 32 | 128 x 32 (bk0max 128) tile, 4 waves
//            (b) FmhaFwdV3Kernel + BlockFmhaFwdV3Pipeline, 256 x 32 x 128 | 128 x 32 x 128, 8 waves -- the gfx950 kernel
//                aiter ships (ck_tile/ops/fmha_fwd_v3_impl.hpp names its origin), `variant` = 3
#define CK_TILE_FMHA_FWD_FAST_EXP2 1
#include <hip/hip_runtime.h>

#include <utility>
#include <variant>

#include "ck_tile/core.hpp"
#include "ck_tile/host/kernel_launch.hpp"
#include "ck_tile/host/stream_config.hpp"
#include "ck_tile/ops/epilogue.hpp"
#include "ck_tile/ops/fmha.hpp"

#include "common.h"

namespace {

using half = ck_tile::half_t;

template <int HDIM>
struct TileOf;
template <>
struct TileOf<64> {
  using type = ck_tile::sequence<128, 64, 32, 64, 32, 64>;
};
template <>
struct TileOf<128> {
  using type = ck_tile::sequence<128, 128, 32, 128, 32, 128>;
};

template <int HDIM>
struct AsyncInstance {
  using Shape = ck_tile::TileFmhaShape<typename TileOf<HDIM>::type, ck_tile::sequence<4, 1, 1>, ck_tile::sequence<32, 32, 16>,
                                       ck_tile::sequence<4, 1, 1>, ck_tile::sequence<32, 32, 16>, true /* V row-major */>;
  // the async pipeline is generated with seqlen_q / head-dim padding on (its loads are bounds-checked buffer loads)
  using Traits = ck_tile::TileFmhaTraits<true, false, true, true, false /* soft cap */, ck_tile::BlockAttentionBiasEnum::NO_BIAS,
                                         false, false /* LSE */, false /* dropout */, false /* fp8 quant */, -1, false>;
  using Variant = ck_tile::ComposedAttention<0, CK_TILE_FMHA_FWD_FAST_EXP2>;
  using Mask = ck_tile::SimplifiedGenericAttentionMask<false>;
  using Problem = ck_tile::BlockFmhaPipelineProblem<half, half, half, float, float, half, uint8_t, float, half, float, half, Shape,
                                                    false /* batch mode */, Variant, Mask, false /* tr-load */, Traits>;
  using Pipeline = ck_tile::BlockFmhaPipelineQRKSVSAsync<Problem>;
  using Epilogue = ck_tile::Default2DEpilogue<ck_tile::Default2DEpilogueProblem<float, half, true, true>>;
  using Kernel = ck_tile::FmhaFwdKernel<Pipeline, Epilogue>;
};

template <int HDIM>
int run_async(const void* q, const void* k, const void* v, void* o, int B, int H, int N, hipStream_t stream) {
  using Kernel = typename AsyncInstance<HDIM>::Kernel;
  const float scale = 1.0f / sqrtf((float)HDIM);
  const ck_tile::index_t row = HDIM, head = (ck_tile::index_t)N * HDIM;
  const long long batch_ll = (long long)H * N * HDIM;
  if (batch_ll * B > 0x7fffffffLL) return CLN_ERR_UNSUPPORTED;  // 32-bit element strides in this kernel's arguments
  const ck_tile::index_t batch = (ck_tile::index_t)batch_ll;
  auto kargs = Kernel::MakeKargsImpl(q, k, v, nullptr, nullptr, nullptr, o, N, N, HDIM, HDIM, H, 1, scale, 1.0f, 1.0f, 0.0f,
                                     row, row, row, 0, 0, row, head, head, head, 0, 0, 0, head, batch, batch, batch, 0, 0, 0, batch,
                                     -1, -1, 0, 0.0f, false, std::make_pair<uint64_t, uint64_t>(0, 0));
  const dim3 grids = Kernel::GridSize(B, H, N, HDIM, false);
  const dim3 blocks = Kernel::BlockSize();
  constexpr ck_tile::index_t kBlockPerCu = Kernel::kBlockPerCu;
  ck_tile::launch_kernel(ck_tile::stream_config{stream, false}, ck_tile::make_kernel<kBlockPerCu>(Kernel{}, grids, blocks, 0, kargs));
  return hipGetLastError() == hipSuccess ? CLN_OK : CLN_ERR_LAUNCH;
}

// ---- the gfx950 "v3" kernel (head dim 128 only): the configuration of ck_tile/ops/fmha_fwd_v3_impl.hpp
struct V3Instance {
  using Shape = ck_tile::TileFmhaShape<ck_tile::sequence<256, 32, 128, 128, 32, 128>, ck_tile::sequence<8, 1, 1>,
                                       ck_tile::sequence<32, 32, 16>, ck_tile::sequence<8, 1, 1>, ck_tile::sequence<32, 32, 16>, true>;
  using Traits = ck_tile::TileFmhaFwdV3Traits<true, true, false, false, false /* LSE */, -1>;
  using Mask = ck_tile::GenericAttentionMask<false, false>;
  using Problem = ck_tile::BlockFmhaFwdV3PipelineProblem<half, half, half, float, float, float, half, float, half, Shape,
                                                         false /* fixed seqlen */, Mask, Traits>;
  using Pipeline = ck_tile::BlockFmhaFwdV3Pipeline<Problem>;
  using Epilogue = ck_tile::Default2DEpilogue<ck_tile::Default2DEpilogueProblem<float, half, true, true, true>>;
  using Kernel = ck_tile::FmhaFwdV3Kernel<Pipeline, Epilogue>;
};

int run_v3(const void* q, const void* k, const void* v, void* o, int B, int H, int N, hipStream_t stream) {
  using Kernel = V3Instance::Kernel;
  constexpr int HDIM = 128;
  const float scale = 1.0f / sqrtf((float)HDIM);
  const ck_tile::index_t row = HDIM, head = (ck_tile::index_t)N * HDIM;
  const long long batch_ll = (long long)H * N * HDIM;
  if (batch_ll * B > 0x7fffffffLL) return CLN_ERR_UNSUPPORTED;
  const ck_tile::index_t batch = (ck_tile::index_t)batch_ll;
  auto kargs = Kernel::MakeKargs(q, k, v, nullptr, o, N, N, HDIM, HDIM, H, 1, scale, row, row, row, row, head, head, head, 0, head,
                                 batch, batch, batch, 0, batch, -1, -1, 0 /* no mask */, 2 /* remap_opt, as aiter */, nullptr, nullptr);
  const dim3 grids = Kernel::GridSize(B, H, N, HDIM);
  constexpr dim3 blocks = Kernel::BlockSize();
  constexpr ck_tile::index_t kBlockPerCu = Kernel::kBlockPerCu;
  ck_tile::launch_kernel(ck_tile::stream_config{stream, false}, ck_tile::make_kernel<kBlockPerCu>(Kernel{}, grids, blocks, 0, kargs));
  return hipGetLastError() == hipSuccess ? CLN_OK : CLN_ERR_LAUNCH;
}

}  // namespace

// (q, k, v, o: fp16 [B, H, N, D] contiguous; variant 0 = the classic async pipeline, 3 = the gfx950 v3 kernel (D = 128 only); stream)
CLN_API int cln_fa2_ck_tile_fwd(const void* q, const void* k, const void* v, void* o, int B, int H, int N, int D, int variant,
                                void* stream) {
  if (!q || !k || !v || !o || B <= 0 || H <= 0 || N <= 0) return CLN_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (variant == 3) return D == 128 ? run_v3(q, k, v, o, B, H, N, s) : CLN_ERR_UNSUPPORTED;
  if (variant != 0) return CLN_ERR_BAD_ARG;
  if (D == 64) return run_async<64>(q, k, v, o, B, H, N, s);
  if (D == 128) return run_async<128>(q, k, v, o, B, H, N, s);
  return CLN_ERR_UNSUPPORTED;
}
