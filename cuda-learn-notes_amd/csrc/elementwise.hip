// c = a + b, six access-width rungs. HBM-bound: 3 streams, 3*sizeof(T) bytes per element.
//
// Replaces the kernels + bindings of reference kernels/elementwise/elementwise.cu:24-108
// (kernels) and :122-177 (TORCH_BINDING_ELEM_ADD launch-shape macro + PYBIND11_MODULE).
// Design for gfx950: the rung name fixes only the per-lane access width (4 B, 16 B, 2 B, 4 B, 4x4 B, 16 B), exactly what the reference ladder
// teaches. Walk (round 6): BLOCK-CONTIGUOUS, no loop -- workgroup b owns the 256 K consecutive packs from 256 K b on, every lane issues its K
// load pairs, then its K stores; K = 4 below 512 MB of traffic, 1 above. Rounds 1-5 ran a grid-stride loop over a capped grid (256 CUs x 32
// workgroups): with torch.add(out=) as the yardstick beside it that form was 9 % BEHIND at [8192,8192] f32 (148 vs 135 us); on the same box
// (tools/ubench/stream_forms.hip, profiles/r06_stream_forms_ubench.log) grid-stride 155 us, one pack per thread 136, four per thread 138; at
// [4096,4096] 36.6 / 36.1 / 35.2, at [2048,2048] 11.0 / 11.0 / 10.6. Every wave of a grid-stride launch alternates loads and stores in lockstep
// with every other wave; waves that retire and are replaced do not. Non-temporal stores change none of these numbers (kept: outputs of launches
// that fill the MALL do not displace their inputs, common.h cln_store_stream).
#include "common.h"

namespace {

template <typename VT, int K>
__global__ __launch_bounds__(256) void add_vec_kernel(const VT* __restrict__ a, const VT* __restrict__ b,
                                                      VT* __restrict__ c, long long nvec, int stream_nt) {
  const long long base = (long long)blockIdx.x * (256 * K) + threadIdx.x;
  if (base + (K - 1) * 256 < nvec) {  // every pack of this lane exists (all workgroups but the last)
    VT x[K], y[K];
#pragma unroll
    for (int k = 0; k < K; ++k) x[k] = a[base + k * 256];
#pragma unroll
    for (int k = 0; k < K; ++k) y[k] = b[base + k * 256];
#pragma unroll
    for (int k = 0; k < K; ++k) cln_store_stream(c + base + k * 256, (VT)(x[k] + y[k]), stream_nt);
  } else {
#pragma unroll
    for (int k = 0; k < K; ++k)
      if (base + k * 256 < nvec) c[base + k * 256] = a[base + k * 256] + b[base + k * 256];
  }
}

// scalar tail (n not a multiple of the vector width)
template <typename T>
__global__ void add_tail_kernel(const T* a, const T* b, T* c, long long start, long long n) {
  long long i = start + (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) c[i] = a[i] + b[i];
}

// "f16x8" rung (eight halves per thread moved as four separate 4-byte half2 accesses, reference elementwise.cu:62-86; the "_pack" rung moves them as
// one 16-byte access): the half2 kernel (sixteen packs per lane) -- access c of a lane is its pack in the c-th 256-pack row of the
// workgroup's block, so every access instruction of a wave covers 256 contiguous bytes. Rounds 1-5 gave a lane eight CONSECUTIVE halves as the reference
// does: four instructions that each touch 4 of every 16 bytes (22.3 us against 20.1 for the f16x2 rung at [4096,4096], tools/rung_survey.py).
template <typename T, typename VT, int VEC>
int launch_add(const void* a, const void* b, void* c, long long n, hipStream_t stream) {
  if (!a || !b || !c || n < 0) return CLN_ERR_BAD_ARG;
  if (n == 0) return CLN_OK;
  if (VEC * sizeof(T) >= 16 && !(cln_aligned16(a) && cln_aligned16(b) && cln_aligned16(c))) return CLN_ERR_BAD_ARG;
  const long long nvec = n / VEC;
  if (nvec > 0) {
    const long long traffic = 3LL * n * (long long)sizeof(T);
    // packs per lane: 64 bytes per operand in flight per lane (4 packs of 16 bytes ... 16 of 4 or 2 bytes), ONE for the 16-byte rungs above 512 MB
    constexpr int AB = (int)sizeof(VT);
    constexpr int KB = AB >= 16 ? 4 : (64 / AB > 16 ? 16 : 64 / AB);
    if ((AB >= 16 && traffic >= (512LL << 20)) || nvec < 1024 * KB) {
      CLN_LAUNCH((add_vec_kernel<VT, 1>), dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, stream, (const VT*)a, (const VT*)b, (VT*)c, nvec, cln_stream_nt(traffic));
    } else {
      CLN_LAUNCH((add_vec_kernel<VT, KB>), dim3((unsigned)((nvec + 256 * KB - 1) / (256 * KB))), dim3(256), 0, stream, (const VT*)a, (const VT*)b, (VT*)c, nvec, cln_stream_nt(traffic));
    }
  }
  const long long done = nvec * VEC;
  if (done < n) {
    CLN_LAUNCH((add_tail_kernel<T>), dim3(1), dim3(64), 0, stream, (const T*)a, (const T*)b, (T*)c, done, n);
  }
  return cln_check_launch();
}

}  // namespace

// (a, b, c, n_elements, stream) -- replaces `void elementwise_add_*(torch::Tensor a, b, c)`
// reference kernels/elementwise/elementwise.cu:163-168.
CLN_API int elementwise_add_f32(const void* a, const void* b, void* c, long long n, void* stream) {
  return launch_add<float, float, 1>(a, b, c, n, (hipStream_t)stream);
}
CLN_API int elementwise_add_f32x4(const void* a, const void* b, void* c, long long n, void* stream) {
  return launch_add<float, f4, 4>(a, b, c, n, (hipStream_t)stream);
}
CLN_API int elementwise_add_f16(const void* a, const void* b, void* c, long long n, void* stream) {
  return launch_add<half_t, half_t, 1>(a, b, c, n, (hipStream_t)stream);
}
CLN_API int elementwise_add_f16x2(const void* a, const void* b, void* c, long long n, void* stream) {
  return launch_add<half_t, h2, 2>(a, b, c, n, (hipStream_t)stream);
}
CLN_API int elementwise_add_f16x8(const void* a, const void* b, void* c, long long n, void* stream) {
  return launch_add<half_t, h2, 2>(a, b, c, n, (hipStream_t)stream);
}
CLN_API int elementwise_add_f16x8_pack(const void* a, const void* b, void* c, long long n, void* stream) {
  return launch_add<half_t, h8, 8>(a, b, c, n, (hipStream_t)stream);
}
