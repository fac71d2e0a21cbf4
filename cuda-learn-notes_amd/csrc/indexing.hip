// Bit-exact indexing kernels named by north_star's tolerance clause:
//   histogram_i32 / histogram_i32x4     reference kernels/histogram/histogram.cu:19-37, bindings :56-80
//   embedding_{f32,f32x4,f32x4_pack,f16,f16x8,f16x8_pack}   reference kernels/embedding/embedding.cu:16-79,
//                                                           bindings :99-133
// gfx950 design (integer / byte work: HBM + atomic bound, nothing here goes near the matrix cores):
//  * histogram: the reference issues one global atomic per element. Here each workgroup counts into an LDS
//    histogram (ds_add_u32, no return) when the bin count fits 32 KiB, streams its grid-stride share of the
//    input with the access width the rung name states (4 B or 16 B per lane), and flushes only its non-zero
//    bins with global atomics -- device-scope atomics drop from N to <= grid x bins. Bin counts above the
//    LDS budget fall back to the reference's direct global atomics. Values outside [0, nbins) are ignored
//    (the reference reads/writes out of bounds for them).
//  * embedding: out[i, :] = weight[idx[i], :]. A row is a contiguous run, so the copy is flattened over
//    (row, pack) pairs: lane-consecutive packs of one row coalesce into full lines; the index word is
//    read once per pack (L1 broadcast). Pack width per the rung name: 4 B, 4 x 4 B, 16 B (f32), 2 B,
//    8 x 2 B, 16 B (f16). Rows whose index is outside [0, vocab) are written as zeros.
#include "common.h"

namespace {

constexpr int HIST_LDS_BINS = 8192;  // 32 KiB of LDS counters

template <int VEC, int NT>
__global__ __launch_bounds__(NT) void histogram_lds_kernel(const int* __restrict__ a, int* __restrict__ y,
                                                            long long n, int nbins) {
  __shared__ int h[HIST_LDS_BINS];
  for (int i = threadIdx.x; i < nbins; i += NT) h[i] = 0;
  __syncthreads();
  const long long nvec = n / VEC;
  const long long stride = (long long)gridDim.x * NT;
  auto count = [&](int v) {
    if ((unsigned)v < (unsigned)nbins) atomicAdd(&h[v], 1);
  };
  long long i = (long long)blockIdx.x * NT + threadIdx.x;
  if constexpr (VEC == 4) {
    // four independent 16-byte loads in flight per lane: with one, a CU holds ~20 KB of requests against the
    // ~60 KB that HBM latency x per-CU bandwidth asks for, and the kernel ran at 2.3 TB/s whatever the atomics did
    for (; i + 3 * stride < nvec; i += 4 * stride) {
      const int4 v0 = *reinterpret_cast<const int4*>(a + i * 4);
      const int4 v1 = *reinterpret_cast<const int4*>(a + (i + stride) * 4);
      const int4 v2 = *reinterpret_cast<const int4*>(a + (i + 2 * stride) * 4);
      const int4 v3 = *reinterpret_cast<const int4*>(a + (i + 3 * stride) * 4);
      count(v0.x), count(v0.y), count(v0.z), count(v0.w);
      count(v1.x), count(v1.y), count(v1.z), count(v1.w);
      count(v2.x), count(v2.y), count(v2.z), count(v2.w);
      count(v3.x), count(v3.y), count(v3.z), count(v3.w);
    }
    for (; i < nvec; i += stride) {
      const int4 v = *reinterpret_cast<const int4*>(a + i * 4);
      count(v.x), count(v.y), count(v.z), count(v.w);
    }
  } else {
    for (; i + 3 * stride < nvec; i += 4 * stride) {
      const int v0 = a[i], v1 = a[i + stride], v2 = a[i + 2 * stride], v3 = a[i + 3 * stride];
      count(v0), count(v1), count(v2), count(v3);
    }
    for (; i < nvec; i += stride) count(a[i]);
  }
  if (VEC > 1 && blockIdx.x == 0) {  // ragged tail
    for (long long t = nvec * VEC + threadIdx.x; t < n; t += NT) {
      count(a[t]);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nbins; i += NT) {
    const int c = h[i];
    if (c) atomicAdd(&y[i], c);
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void histogram_global_kernel(const int* __restrict__ a, int* __restrict__ y,
                                                               long long n, int nbins) {
  const long long nvec = n / VEC;
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) {
    if constexpr (VEC == 4) {
      const int4 v = *reinterpret_cast<const int4*>(a + i * 4);
      if ((unsigned)v.x < (unsigned)nbins) atomicAdd(&y[v.x], 1);
      if ((unsigned)v.y < (unsigned)nbins) atomicAdd(&y[v.y], 1);
      if ((unsigned)v.z < (unsigned)nbins) atomicAdd(&y[v.z], 1);
      if ((unsigned)v.w < (unsigned)nbins) atomicAdd(&y[v.w], 1);
    } else {
      const int v = a[i];
      if ((unsigned)v < (unsigned)nbins) atomicAdd(&y[v], 1);
    }
  }
  if (VEC > 1 && blockIdx.x == 0) {
    for (long long i = nvec * VEC + threadIdx.x; i < n; i += 256) {
      const int v = a[i];
      if ((unsigned)v < (unsigned)nbins) atomicAdd(&y[v], 1);
    }
  }
}

template <int VEC>
int launch_hist(const void* a, void* y, long long n, int nbins, hipStream_t st) {
  if (!a || !y || n < 0 || nbins <= 0) return CLN_ERR_BAD_ARG;
  if (n == 0) return CLN_OK;
  if (VEC == 4 && !cln_aligned16(a)) return CLN_ERR_BAD_ARG;
  long long g = (n / VEC + 255) / 256;
  if (nbins <= HIST_LDS_BINS) {
    // The flush costs grid x (non-zero bins) device-scope atomics at ~45 per ns chip-wide: 1024 workgroups x 1024
    // bins was 23 us of a 30 us launch. One 1024-thread workgroup per CU (the reduce kernel's shape) keeps the
    // streaming rate and cuts the flush to <= 256 x bins.
    constexpr int NT = 1024;
    long long gw = (n / VEC + NT - 1) / NT;
    const int grid = (int)(gw < 1 ? 1 : (gw > 256 ? 256 : gw));
    CLN_LAUNCH((histogram_lds_kernel<VEC, NT>), dim3(grid), dim3(NT), 0, st, (const int*)a, (int*)y, n, nbins);
  } else {
    const int grid = (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
    CLN_LAUNCH((histogram_global_kernel<VEC>), dim3(grid), dim3(256), 0, st, (const int*)a, (int*)y, n, nbins);
  }
  return cln_check_launch();
}

// ---- embedding -----------------------------------------------------------------------------------
// T: element type, VEC: elements per lane, PACKED: one 16-byte (VEC*sizeof(T)) access vs VEC scalar accesses.
// Walk (round 6, as elementwise.hip): block-contiguous, no loop -- workgroup b owns the 256 KP consecutive packs from 256 KP b on; a lane loads its
// KP indices, then its KP gathers, then stores (KP = 4). Rounds 1-5 walked a capped grid-stride loop with ONE index -> gather -> store chain in
// flight per lane: the scalar rungs sat at 4.1 (f32) / 2.3 (f16) TB/s on [65536,1024] while the 16-byte rungs reached 7.2 / 6.8.
template <typename T, int VEC, bool PACKED, int KP>
__global__ __launch_bounds__(256) void embedding_kernel(const int* __restrict__ idx, const T* __restrict__ weight,
                                                        T* __restrict__ out, long long n, int emb, int vocab, int stream_nt) {
  typedef T vec_t __attribute__((ext_vector_type(VEC)));
  const int ppr = emb / VEC;  // packs per row
  const long long total = n * ppr;
  const long long base = (long long)blockIdx.x * (256 * KP) + threadIdx.x;
  long long row[KP];
  int col[KP], id[KP];
  bool live[KP], ok[KP];
#pragma unroll
  for (int k = 0; k < KP; ++k) {
    const long long t = base + k * 256;
    live[k] = t < total;
    row[k] = live[k] ? t / ppr : 0;
    col[k] = (int)((live[k] ? t : 0) - row[k] * ppr) * VEC;
  }
#pragma unroll
  for (int k = 0; k < KP; ++k) id[k] = idx[row[k]];
  vec_t v[KP];
#pragma unroll
  for (int k = 0; k < KP; ++k) {
    ok[k] = (unsigned)id[k] < (unsigned)vocab;
    const T* src = weight + (size_t)(ok[k] ? id[k] : 0) * emb + col[k];
    if constexpr (PACKED) {
      v[k] = *reinterpret_cast<const vec_t*>(src);
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) v[k][e] = src[e];
    }
  }
#pragma unroll
  for (int k = 0; k < KP; ++k) {
    if (!live[k]) continue;
    if (!ok[k]) v[k] = vec_t(0);
    T* dst = out + (size_t)row[k] * emb + col[k];
    if constexpr (PACKED) {
      cln_store_stream(reinterpret_cast<vec_t*>(dst), v[k], stream_nt);
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) cln_store_stream(dst + e, (T)v[k][e], stream_nt);
    }
  }
}

template <typename T, int VEC, bool PACKED>
int launch_emb(const void* idx, const void* weight, void* out, long long n, int emb, int vocab, hipStream_t st) {
  if (!idx || !weight || !out || n < 0 || emb <= 0 || vocab <= 0) return CLN_ERR_BAD_ARG;
  if (emb % VEC != 0) return CLN_ERR_UNSUPPORTED;
  if (PACKED && (!cln_aligned16(weight) || !cln_aligned16(out))) return CLN_ERR_BAD_ARG;
  if (n == 0) return CLN_OK;
  const long long total = n * (emb / VEC), traffic = 2LL * n * emb * (long long)sizeof(T);  // gathered rows + output
  // 16-byte rungs above 512 MB of traffic: one pack per lane (as elementwise.hip; [65536,1024] f32x4_pack 74.8 us against 77.7 with four)
  const int kp = (VEC * sizeof(T) >= 16 && traffic >= (512LL << 20)) ? 1 : 4;
  const long long grid = (total + 256 * kp - 1) / (256 * kp);
  if (grid > 0x7FFFFFFFLL) return CLN_ERR_UNSUPPORTED;
  if (kp == 1)
    CLN_LAUNCH((embedding_kernel<T, VEC, PACKED, 1>), dim3((unsigned)grid), dim3(256), 0, st, (const int*)idx, (const T*)weight, (T*)out, n, emb, vocab, cln_stream_nt(traffic));
  else
    CLN_LAUNCH((embedding_kernel<T, VEC, PACKED, 4>), dim3((unsigned)grid), dim3(256), 0, st, (const int*)idx, (const T*)weight, (T*)out, n, emb, vocab, cln_stream_nt(traffic));
  return cln_check_launch();
}

}  // namespace

// (a int32[n], y int32[nbins] zeroed by the caller, n, nbins, stream)
CLN_API int histogram_i32(const void* a, void* y, long long n, int nbins, void* stream) {
  return launch_hist<1>(a, y, n, nbins, (hipStream_t)stream);
}
CLN_API int histogram_i32x4(const void* a, void* y, long long n, int nbins, void* stream) {
  return launch_hist<4>(a, y, n, nbins, (hipStream_t)stream);
}

// (idx int32[n], weight T[vocab, emb], out T[n, emb], n, emb, vocab, stream)
#define CLN_EMB(name, T, VEC, PACKED)                                                                   \
  CLN_API int name(const void* idx, const void* weight, void* out, long long n, int emb, int vocab,    \
                   void* stream) {                                                                      \
    return launch_emb<T, VEC, PACKED>(idx, weight, out, n, emb, vocab, (hipStream_t)stream);            \
  }
CLN_EMB(embedding_f32, float, 1, false)
CLN_EMB(embedding_f32x4, float, 4, false)
CLN_EMB(embedding_f32x4_pack, float, 4, true)
CLN_EMB(embedding_f16, half_t, 1, false)
CLN_EMB(embedding_f16x8, half_t, 8, false)
CLN_EMB(embedding_f16x8_pack, half_t, 8, true)
