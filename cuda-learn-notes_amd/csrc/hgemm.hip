// C-ABI entry points of the HGEMM library: one symbol per function exported by the reference's
// pybind module (kernels/hgemm/pybind/hgemm.cc:58-107), same names, same argument meaning.
//
//   G3:  int name(a, b, c, M, N, K, stream)
//   G6:  int name(a, b, c, M, N, K, stages, swizzle, swizzle_stride, stream)
//
// a: [M,K] row-major fp16; b: [K,N] row-major fp16 (NN) or storage [N,K] (TN, reference
// as_col_major kernels/hgemm/tools/utils.py:135-140); c: [M,N] row-major fp16.
// Which names are distinct gfx950 kernels and which are aliases is recorded in
// cuda-learn-notes_amd/manifest.py (generated table in DESIGN.md).
#include "hgemm_dispatch.h"
#include "hgemm_mfma.cuh"
#include "hgemm_w4.cuh"
#include "hgemm_w4s.cuh"
#include "hgemm_splitk.cuh"
#include "hgemm_valu.cuh"
#include "stream_scratch.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <cstdint>
#include <mutex>
#include <vector>

using namespace hgemm;

namespace {

int check_args(const void* a, const void* b, const void* c, int M, int N, int K) {
  if (!a || !b || !c) return CLN_ERR_BAD_ARG;
  if (M <= 0 || N <= 0 || K <= 0) return CLN_ERR_BAD_ARG;
  if (!cln_aligned16(a) || !cln_aligned16(b) || !cln_aligned16(c)) return CLN_ERR_BAD_ARG;
  return CLN_OK;
}

// "best" policy shared by the reference's top rungs (warp4x4x2 family): pick the tile shape with the best estimated
// throughput for (M, N) on 256 CUs. score = eff x util x (1 + (1 - util) / 2):
//   util = tiles / (rounds x slots)   -- slots = 256 workgroups in flight for the 8-wave kernels (one per CU), 512 for
//                                        the 4-wave 64x128 / 128x128 rings (two per CU);
//   eff  = measured rate at full occupancy relative to the one-wave-per-SIMD 256x256 kernel (same box):
//          hgemm_w4 256x256 1.00 | 192x256 0.96 | 256x192 0.95 | 192x192 0.93 | 128x256 0.82 | 256x128 0.81 | 160x160 0.88 (all need K % 64 == 0 and >= 6 K tiles, >= 7 when their number is odd) |
//          ping-pong 256x256 0.94 | 192x256 0.91 | ring 128x256 0.79 | ring 64x128 0.545 | ring 128x128 0.53;
//   the last factor: with CUs idle the busy ones clock higher (measured 1.2-1.3x at util 0.4-0.55).
// It reproduces the measured winner at every size of profiles/r02_hgemm_midsize_probe.log and
// r02_hgemm_w4_shapes_probe.log (1024..8192): ring 64x64 up to 1536, ring 64x128 at 1792 / 2048, w4 192x192 at 2304 / 3072 (890 / 1320 TF vs 769 /
// 1066 for the previous policy, rocBLAS TN 894 / 1185), w4 160x160 at 2560 / 3200 (1065-1126 / 1080 TF vs 1015-1039 for 128x256 and 857 for the 64x128 ring; rocBLAS TN 900-1006 / 915), w4 192x256 at 4608 / 6144 (1356 / 1502 vs
// 1146 / 1266, rocBLAS TN 1216 / 1388), w4 256x256 at 3584 / 4096 / 7680 / 8192.
enum BestPlan { PLAN_PP256 = 0, PLAN_PP192, PLAN_R128x256, PLAN_R64x128, PLAN_R128, PLAN_W256, PLAN_W192x256, PLAN_W256x192, PLAN_W192, PLAN_W128x256, PLAN_W256x128, PLAN_R64x64, PLAN_W160 };
int best_plan(int M, int N, int K, double* best_score = nullptr) {
  auto score = [](long long tiles, int slots, double eff) {
    if (tiles <= 0) return 0.0;
    const long long rounds = (tiles + slots - 1) / slots;
    const double util = (double)tiles / (double)(rounds * slots);
    return eff * util * (1.0 + 0.5 * (1.0 - util));
  };
  // Small problems are latency-bound (launch ramp + one L2 round trip per K tile), not throughput-bound: up to 640
  // tiles of 64x64 (1536^2, 1600^2) the smallest tile wins because it puts a workgroup on every CU (1024^3: 6.8 us vs 9.1 us for
  // 64x128 on half the CUs and 8.4 / 10.0 us for rocBLAS TN / NN; profiles/r02_hgemm_small_probe.log); from 1792^2 on the
  // throughput model below takes over. At EVERY K: round 3 had limited the shortcut to K <= 2048 on one unmeasured shape; measured in round 4
  // (profiles/r04_hgemm_small_mn_long_k_probe.log, C-ABI timed) the 64x64 ring wins all of 1024^2 / 1152^2 / 1280^2 / 1536^2 / 1600^2 at
  // K = 4096 ... 16384 by 7-27 % over what the model picked (1600^2 x 16384: 669 vs 593 TF for hgemm_w4<160x160>, 100 tiles on 256 CUs;
  // 1280^2 x 4096: 434 vs 330) -- the model prices a tile against a FULL chip and these grids do not fill it.
  if (best_score) *best_score = 0.0;
  if (M % 64 == 0 && N % 64 == 0 && K % 64 == 0 && (long long)(M / 64) * (N / 64) <= 640) return PLAN_R64x64;
  double best = -1.0;
  int plan = PLAN_R128;
  auto offer = [&](int p, double sc) {
    if (sc > best) best = sc, plan = p;
  };
  // eff = measured rate at full occupancy relative to the one-wave-per-SIMD 256x256 kernel (round 2, same box:
  // profiles/r02_hgemm_w4_shapes_probe.log, r02_hgemm_midsize_probe.log)
  if (M % 64 == 0 && N % 64 == 0 && K % 64 == 0) offer(PLAN_R64x64, score((long long)(M / 64) * (N / 64), 1024, 0.44));
  if (M % 128 == 0 && N % 128 == 0) offer(PLAN_R128, score((long long)(M / 128) * (N / 128), 512, 0.53));
  if (M % 64 == 0 && N % 128 == 0) offer(PLAN_R64x128, score((long long)(M / 64) * (N / 128), 512, 0.545));
  if (M % 128 == 0 && N % 256 == 0) offer(PLAN_R128x256, score((long long)(M / 128) * (N / 256), 256, 0.79));
  if (K % 64 == 0 && N % 256 == 0) {
    if (M % 192 == 0) offer(PLAN_PP192, score((long long)(M / 192) * (N / 256), 256, 0.91));
    if (M % 256 == 0) offer(PLAN_PP256, score((long long)(M / 256) * (N / 256), 256, 0.94));
  }
  if (w4_k_ok(K)) {
    if (M % 160 == 0 && N % 160 == 0) offer(PLAN_W160, score((long long)(M / 160) * (N / 160), 256, 0.88));
    if (M % 256 == 0 && N % 128 == 0) offer(PLAN_W256x128, score((long long)(M / 256) * (N / 128), 256, 0.81));
    if (M % 128 == 0 && N % 256 == 0) offer(PLAN_W128x256, score((long long)(M / 128) * (N / 256), 256, 0.82));
    if (M % 192 == 0 && N % 192 == 0) offer(PLAN_W192, score((long long)(M / 192) * (N / 192), 256, 0.93));
    if (M % 256 == 0 && N % 192 == 0) offer(PLAN_W256x192, score((long long)(M / 256) * (N / 192), 256, 0.95));
    if (M % 192 == 0 && N % 256 == 0) offer(PLAN_W192x256, score((long long)(M / 192) * (N / 256), 256, 0.96));
    if (M % 256 == 0 && N % 256 == 0) offer(PLAN_W256, score((long long)(M / 256) * (N / 256), 256, 1.00));
  }
  if (best_score) *best_score = best;
  return plan;
}
// ---- split-K (hgemm_splitk.cuh): few output tiles, long K ------------------------------------------------------------------------------
// WHEN: K >= 4096 and M N <= 2048^2 (K >= 5120 above 1536^2). Measured over 28 shapes x every (tile, splits) candidate against the policy
// above (profiles/r04_hgemm_splitk_probe.log): inside that region split-K wins by 1.09-6.7x (1024^2 x 16384: 310 -> 823 TF, 128 x 8192 x 8192:
// 267 -> 541, 768^2 x 12288: 174 -> 513; rocBLAS TN 603 / 471 / 339); outside it loses or ties (2048^3 0.79x, 1024^2 x 2048 0.65x, 2048^2 x 4096
// 0.97x, 2560^2 / 3072^2 x 8192 0.97x / 0.82x, 1536^2 x 3072 0.99x).
// WHICH (tile, S): the minimum of a time model fitted to the same sweep (rms 7 %; its pick is the measured best at 27 of the 28 shapes, 7 % off at one):
//   t = rounds x (K / (64 S) x tau + phi) + rho,   rounds = ceil(tiles S / 256)
//   tau = 2 BM BN 64 flop / (eff x 5.86 TF), shortened by 0.34 (1 - fill) when fewer than 256 workgroups run (idle CUs: higher clocks, no contention);
//         eff: 256x256 1.25 | 192x256 1.19 | 192x192 1.08 | 128x256 0.97 | 160x160 0.94 (a workgroup alone on its CU, no C epilogue through LDS)
//   phi = 3.9 us x sqrt(BM BN / 256^2): launch ramp, prologue, partial store;   rho = 5.2 us + (4 S M N + 2 M N) bytes / 3.6 TB/s: the reduce launch
// The workspace (S M N floats <= 256 MiB) is per stream, allocated on first use and only ever grown (an outgrown buffer is retired, not freed);
// under stream capture a launch that would have to allocate takes the single-pass plan instead.
struct SplitK {
  int bm = 0, bn = 0, S = 0;
};
constexpr size_t SPLITK_WS_MAX = 256u << 20;
// $CLN_AMD_NO_SPLITK=1 (read once): every shape single-pass -- for callers that need eager, captured and workspace-less runs of one shape to be
// bit-identical (the split forms differ from the single-pass kernel in the fp32 summation order)
bool splitk_disabled() {
  static const bool v = [] {
    const char* e = getenv("CLN_AMD_NO_SPLITK");
    return e && e[0] == '1';
  }();
  return v;
}
SplitK splitk_plan(int M, int N, int K) {
  SplitK best;
  if (splitk_disabled()) return best;
  const double mn = (double)M * (double)N;
  if (K < 4096 || K % 64 || mn > 2048.0 * 2048.0 || (mn > 1536.0 * 1536.0 && K < 5120)) return best;
  static const struct { int bm, bn; double eff; } shapes[] = {{256, 256, 1.249}, {192, 256, 1.188}, {192, 192, 1.078}, {128, 256, 0.974}, {160, 160, 0.942}};
  double t_best = 1e30;
  for (const auto& sh : shapes) {
    if (M % sh.bm || N % sh.bn) continue;
    const long long tiles = (long long)(M / sh.bm) * (N / sh.bn);
    for (int S = 2; S <= 32; ++S) {
      if (!w4_splitk_ok(K, S) || tiles * S > 768 || (double)S * mn * 4.0 > (double)SPLITK_WS_MAX) continue;
      const long long n = tiles * S, rounds = (n + 255) / 256;
      const double fill = n >= 256 ? 1.0 : (double)n / 256.0;
      const double tau = 2.0 * sh.bm * sh.bn * 64.0 / (5.86e6 * sh.eff) * (1.0 - 0.338 * (1.0 - fill));
      double t = rounds * ((double)(K / S / 64) * tau + 3.88 * sqrt(sh.bm * sh.bn / 65536.0)) + 5.18 + (4.0 * S * mn + 2.0 * mn) / 3.615e6;
      // the one-launch form (S <= splitk_fused_max_s: the tile's last workgroup reduces) measured 0.5-1.5 us under the two-launch cost the model was fitted on
      // (512 x 8192^2 68.1 -> 67.3 us, 2048^2 x 8192 70.6 -> 70.1, 640 x 5120^2 43.4 -> 41.9: profiles/r05_hgemm_splitk_fused_probe.log) -- ADVICE r5
      if (S <= 2) t -= 1.0;
      if (t < t_best) t_best = t, best.bm = sh.bm, best.bn = sh.bn, best.S = S;
    }
  }
  return best;
}

// ---- split-K workspace (round 5: caller-visible, bounded, freeable) -------------------------------
// One workspace per (device, stream): W4_TICKET_FLOATS zeroed arrival counters + the fp32 partials. The CALLER's (cln_hgemm_set_workspace: the
// library never allocates for that stream; a shape whose workspace does not fit takes the single-pass plan) -- and that is the ONLY kind by
// default (round 6, SURVEY 8(b) "no hidden workspace"): a stream nobody gave a region runs every shape single-pass. host.py hands each stream a
// tensor from torch's caching allocator; a C caller without an allocator of its own opts in to library-owned buffers with
// cln_hgemm_library_workspace(1) (allocated on first use, partial area a power of two from 16 MiB up to SPLITK_WS_MAX).
//   * g_ws_mu is held from the lookup to the END of the launch sequence that uses the workspace (1-2 launches): two host threads calling hgemm
//     on one stream can no longer interleave their partial / reduce launches (ctypes drops the GIL around the C call). On one
//     stream the launches of consecutive calls run in order, so one workspace per stream is enough.
//   * every use records an event; a buffer is freed only after its event has completed (growth, LRU eviction, cln_release_workspaces) -- the
//     event stays valid after its stream is destroyed, so a process that cycles streams does not leak: at most SPLITK_OWNED_MAX library-owned
//     workspaces exist at a time (least recently used evicted first).
//   * under stream capture the library never allocates, frees or records: a stream without a workspace takes the single-pass plan, and a
//     captured graph holds the pointer of the workspace it was captured with -- replay it on the capture stream (or give each graph its own
//     region through cln_hgemm_set_workspace), include/cln_amd.h says so. A library-owned workspace that a capture has used is PINNED: neither
//     the LRU eviction nor growth frees it behind the graph's back (a larger shape on that stream runs single-pass); only
//     cln_release_workspaces() -- the caller's statement that no such graph will be replayed -- does.
struct SplitKWs {
  int dev = 0;
  hipStream_t stream = nullptr;
  float* p = nullptr;
  size_t bytes = 0;
  bool user = false;          // caller-owned region (cln_hgemm_set_workspace)
  hipEvent_t ev = nullptr;    // completion of the last launch that used the region (library-owned only)
  unsigned long long used = 0;
  bool pinned = false;        // a stream capture has used the region: a graph holds its address, so the library never frees it on its own
};
constexpr size_t SPLITK_OWNED_MAX = 8;
bool g_lib_ws = false;  // library-owned workspaces allowed (cln_hgemm_library_workspace); guarded by g_ws_mu
std::mutex g_ws_mu;
std::vector<SplitKWs> g_ws;
unsigned long long g_ws_clock = 0;

bool stream_capturing(hipStream_t stream) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cs) != hipSuccess) return (void)hipGetLastError(), true;
  return cs != hipStreamCaptureStatusNone;
}
void ws_free_entry(SplitKWs& w) {  // g_ws_mu held; library-owned entries only
  if (w.ev) {
    (void)hipEventSynchronize(w.ev);  // the launches that used the buffer are done (valid even if the stream is gone)
    (void)hipEventDestroy(w.ev);
  }
  if (w.p) (void)hipFree(w.p);
  (void)hipGetLastError();
  w.p = nullptr, w.ev = nullptr, w.bytes = 0;
}
// g_ws_mu held. The stream's workspace with room for `bytes` (tickets included), or nullptr: caller-owned region too small, allocation
// failed, or the stream is being captured and the workspace would have to be allocated now.
SplitKWs* ws_acquire(hipStream_t stream, size_t bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return (void)hipGetLastError(), nullptr;
  SplitKWs* e = nullptr;
  for (auto& w : g_ws)
    if (w.dev == dev && w.stream == stream) e = &w;
  if (e && e->user) return e->bytes >= bytes ? (e->used = ++g_ws_clock, e) : nullptr;
  if (!g_lib_ws) return nullptr;  // no caller-owned region and the library may not allocate: single-pass plan
  if (e && e->bytes >= bytes) {
    if (!e->pinned && stream_capturing(stream)) e->pinned = true;
    return e->used = ++g_ws_clock, e;
  }
  if (bytes > SPLITK_WS_MAX + W4_TICKET_FLOATS * 4 || (e && e->pinned) || stream_capturing(stream)) return nullptr;
  size_t want = 16u << 20;  // the PARTIAL area is the power of two (header + 2^k bytes asked for 2^(k+1)); the ticket header rides on top
  while (want < bytes - W4_TICKET_FLOATS * 4) want <<= 1;
  want += W4_TICKET_FLOATS * 4;
  if (e) {
    ws_free_entry(*e);  // grow: waits for the stream's earlier split-K launches (rare: sizes double)
  } else {
    size_t owned = 0;
    for (auto& w : g_ws) owned += (w.user || w.pinned) ? 0 : 1;
    while (owned >= SPLITK_OWNED_MAX) {  // evict the least recently used library-owned workspace (never a pinned one: see above)
      size_t lru = g_ws.size();
      for (size_t i = 0; i < g_ws.size(); ++i)
        if (!g_ws[i].user && !g_ws[i].pinned && (lru == g_ws.size() || g_ws[i].used < g_ws[lru].used)) lru = i;
      ws_free_entry(g_ws[lru]);
      g_ws.erase(g_ws.begin() + lru);
      --owned;
    }
    g_ws.push_back(SplitKWs());
    e = &g_ws.back();
    e->dev = dev, e->stream = stream;
  }
  float* fresh = nullptr;
  if (hipMalloc(&fresh, want) != hipSuccess || hipMemsetAsync(fresh, 0, W4_TICKET_FLOATS * 4, stream) != hipSuccess) {
    (void)hipGetLastError();
    if (fresh) (void)hipFree(fresh);
    for (size_t i = 0; i < g_ws.size(); ++i)
      if (&g_ws[i] == e) { g_ws.erase(g_ws.begin() + i); break; }
    return nullptr;
  }
  if (hipEventCreateWithFlags(&e->ev, hipEventDisableTiming) != hipSuccess) e->ev = nullptr, (void)hipGetLastError();
  e->p = fresh, e->bytes = want, e->used = ++g_ws_clock;
  return e;
}
void ws_mark_used(SplitKWs* e, hipStream_t stream) {  // g_ws_mu held, after the launches
  if (e && !e->user && e->ev && !stream_capturing(stream)) (void)hipEventRecord(e->ev, stream), (void)hipGetLastError();
}
// splits up to which the ONE-launch form (EPI 6: the last-arriving workgroup of a tile reduces) is taken. Measured (profiles/r05_hgemm_splitk_fused_probe.log,
// every shape of the round-4 split-K / tail probes under 0 = never, the default, 64 = always): at 2 splits one launch is 1-4 % faster than partial +
// reduce launch (512 x 8192^2 68.1 -> 67.3 us, 2048^2 x 8192 70.6 -> 70.1, 640 x 5120^2 43.4 -> 41.9); at 4 splits the tile's reduction on ONE CU
// (1 MiB of partials at one CU's load rate) costs 3-5 us more than the reduce launch it saves (256 x 4096^2 23.4 -> 26.6 us, 4352^3 138.9 -> 149.6),
// at 8 and more 5-6 us more. (The probe library launches either form at any S -- kind 17 two launches, kind 20 one: tools/hg_splitk_fused_probe.py.)
constexpr int splitk_fused_max_s() { return 2; }
// A CAPTURED launch never takes the one-launch form (round 6): its arrival tickets are mutable state in the workspace header, a graph replays
// on whatever stream is current, and a ticket lost or doubled by an overlapping launch never recovers. Partial + reduce launch keeps no state
// between calls (same bits).
template <int LAYOUT>
int splitk_dispatch(const SplitK& sk, const void* a, const void* b, void* c, float* ws, int M, int N, int K, hipStream_t st, bool fused_ok) {
  // `ws` = tickets + partials; the two-launch form uses the partial area only
#define CLN_SK(BMM, BNN)                                                                                                        \
  if (sk.bm == BMM && sk.bn == BNN)                                                                                             \
    return fused_ok && sk.S <= splitk_fused_max_s() ? launch_w4_splitk_fused<LAYOUT, 26, BMM, BNN>(a, b, c, ws, M, N, K, sk.S, st)         \
                                        : launch_w4_splitk<LAYOUT, 26, BMM, BNN>(a, b, c, ws + W4_TICKET_FLOATS, M, N, K, sk.S, st);
  CLN_SK(256, 256) CLN_SK(192, 256) CLN_SK(192, 192) CLN_SK(128, 256) CLN_SK(160, 160)
#undef CLN_SK
  return CLN_ERR_UNSUPPORTED;
}

// ---- tail split (hgemm_splitk.cuh launch_w4_tail_split): a count of 256 x 256 tiles just past whole rounds of 256 ---------------------------
// 4352^3 = 289 tiles, 5888^3 = 529, 7168^3 = 784 (sizes of the reference's sweep): the last round runs 16-33 tiles on 256 CUs and still costs
// 0.55-0.7 of a full one. The last r tile rows go to split-K (all CUs busy for K / S), rows above them fill whole rounds:
// profiles/r04_hgemm_tail_probe.log -- 4352^3 1032 -> 1223 TF (rocBLAS TN 1188), 5888^3 1162 -> 1337, 7168^3 1270 -> 1438, 9216^3 1381 -> 1450;
// it loses where another tile shape already fills the rounds (4608^3: 192 x 256 tiles 1308 vs 1228) or the tail is most of a round (5120^3, 8704^3).
// Decision by time estimates: a full round of 256 x 256 tiles T = 1.42 us x K / 64 + 5 us (95 us at 4096^3, 182 us per round at 8192^3), a last
// round of fraction f costs (0.55 + 0.45 f) T (measured 0.54-0.70 for f = 0.06-0.44), the split-K part by splitk_plan's model; taken when the
// single-pass 256 x 256 score times t_single / t_tail beats best_plan's best score by 3 %.
struct TailSplit {
  int m_split = 0, S = 0;
};
TailSplit tail_plan(int M, int N, int K) {
  TailSplit out;
  if (splitk_disabled() || M % 256 || N % 256 || !w4_k_ok(K)) return out;
  const long long tm = M / 256, tn = N / 256, tiles = tm * tn;
  if (tiles <= 256 || tiles % 256 == 0) return out;
  const double T = 1.42 * (K / 64) + 5.0;
  auto rounds_time = [&](long long n) {
    const long long full = n / 256, rest = n % 256;
    return T * ((double)full + (rest ? 0.55 + 0.45 * (double)rest / 256.0 : 0.0));
  };
  const double t_single = rounds_time(tiles);
  double t_best = 1e30;
  for (long long r = 1; r < tm && r * tn <= 256; ++r) {
    const long long tiles_a = (tm - r) * tn, tiles_b = r * tn;
    if (tiles_a < 224) break;  // the rows above must still (nearly) fill a round
    for (int S = 2; S <= 8; ++S) {
      if (!w4_splitk_ok(K, S) || tiles_b * S > 512 || (double)S * (double)(r * 256) * (double)N * 4.0 > (double)SPLITK_WS_MAX) continue;
      const long long n = tiles_b * S, rounds = (n + 255) / 256;
      const double fill = n >= 256 ? 1.0 : (double)n / 256.0;
      const double tau = 2.0 * 256 * 256 * 64.0 / (5.86e6 * 1.249) * (1.0 - 0.338 * (1.0 - fill));
      const double mn_b = (double)(r * 256) * (double)N;
      const double t_b = rounds * ((double)(K / S / 64) * tau + 3.88) + 5.18 + (4.0 * S * mn_b + 2.0 * mn_b) / 3.615e6;
      const double t = rounds_time(tiles_a) + t_b;
      if (t < t_best) t_best = t, out.m_split = (int)((tm - r) * 256), out.S = S;
    }
  }
  if (out.S == 0) return out;
  double best_score = 0.0;
  (void)best_plan(M, N, K, &best_score);
  const long long rounds = (tiles + 255) / 256;
  const double util = (double)tiles / (double)(rounds * 256);
  const double score_tail = util * (1.0 + 0.5 * (1.0 - util)) * t_single / t_best;
  if (score_tail < 1.03 * best_score) out = TailSplit();
  return out;
}

int plan_tile(int plan) {
  switch (plan) {
    case PLAN_R128x256: return T128x256;
    case PLAN_R64x128: return T64x128;
    case PLAN_R64x64: return T64x64;
    case PLAN_PP256: case PLAN_PP192: case PLAN_W256: return T256;
    default: return T128;
  }
}

// Top rungs (reference warp4x4x2 family): the tile shape comes from best_plan; where that is a 256x256 tile the
// `stages` knob selects a distinct pipeline structure: 2 -> one-wave-per-SIMD kernel (hgemm_w4.cuh; best_plan offers it
// when w4_k_ok(K): whole 64-wide K tiles, >= 6 of them, >= 7 when their number is odd -- otherwise the quadrant
// ping-pong over a 2 x 64-deep ring with split DMA, which is also what an out-of-range `stages` gets), 4 -> k-half
// ping-pong over a 4 x 32-deep ring, 3/5 -> plain multi-stage ring. The other hgemm_w4 tile forms (192 / 160 / 128-wide)
// have ONE pipeline: `stages` is ignored there and cln_describe says so.
// EPI 3: C through the wave-private LDS staging, then NON-TEMPORAL 16-byte stores -- the output does not displace the A / B panels the other
// workgroups (and, back to back, the next launch) still read from L2 / MALL: +1.3-1.5 % at 4096^3, +3.4-3.6 % at 8192^3 over plain stores,
// same bits (profiles/r03_hgemm_c_store_probe.log; write-through `sc0 sc1` stores: +1.4 % / -0.2 %)
// (The persistent tile walk -- hgemm_w4.cuh EPI 7: one workgroup per CU walks its tiles, the next tile's first K tiles requested before the C store --
// was built and measured in round 5: 8192^3 NN 1549 -> 1539 / TN 1550 -> 1557 TF, 8960^3 1465 -> 1436 / 1466 -> 1464, 10240^3 1469 -> 1475 / 1470 -> 1488,
// 16384^3 1491 -> 1483, bit-identical: +-1 %, inside the box noise. It lives in the probe library, kind 19; profiles/r05_hgemm_persist_probe.log.)
constexpr int W4_EPILOGUE = 3;
constexpr int W4_PRODUCTION = 26;  // schedule 10 (one DMA piece per 8 MFMAs, running on into the next tile), boustrophedon MFMA order
template <int LAYOUT>
int best_dispatch(const void* a, const void* b, void* c, int M, int N, int K, int stages, int swizzle, int stride,
                  hipStream_t st) {
  const SplitK sk = splitk_plan(M, N, K);
  if (sk.S >= 2) {  // few tiles, long K: K split over S workgroups per tile, fp32 partials; S <= splitk_fused_max_s(): ONE launch (`stages` ignored: one pipeline)
    std::lock_guard<std::mutex> lock(g_ws_mu);  // held across the launch(es): see the workspace notes above
    SplitKWs* w = ws_acquire(st, w4_splitk_ws_bytes(M, N, sk.S));
    if (w) {
      const int rc = splitk_dispatch<LAYOUT>(sk, a, b, c, w->p, M, N, K, st, !stream_capturing(st));
      ws_mark_used(w, st);
      return rc;
    }
  }
  const TailSplit ts = tail_plan(M, N, K);
  if (ts.S >= 2) {  // a few tiles past whole rounds: the last tile rows split over K (`stages` ignored)
    std::lock_guard<std::mutex> lock(g_ws_mu);
    SplitKWs* w = ws_acquire(st, w4_splitk_ws_bytes(M - ts.m_split, N, ts.S));
    if (w) {
      const int rc = ts.S <= splitk_fused_max_s() && !stream_capturing(st)
                         ? launch_w4_tail_split<LAYOUT, W4_EPILOGUE, W4_PRODUCTION, 256, 256, true>(a, b, c, w->p, M, N, K, ts.m_split, ts.S, swizzle, stride, st)
                         : launch_w4_tail_split<LAYOUT, W4_EPILOGUE, W4_PRODUCTION, 256, 256, false>(a, b, c, w->p + W4_TICKET_FLOATS, M, N, K, ts.m_split, ts.S, swizzle, stride, st);
      ws_mark_used(w, st);
      return rc;
    }
  }
  int plan = best_plan(M, N, K);
  if (plan == PLAN_W192) return launch_w4<LAYOUT, W4_EPILOGUE, W4_PRODUCTION, 0, 192, 192>(a, b, c, M, N, K, swizzle, stride, st);
  if (plan == PLAN_W192x256) return launch_w4<LAYOUT, W4_EPILOGUE, W4_PRODUCTION, 0, 192, 256>(a, b, c, M, N, K, swizzle, stride, st);
  if (plan == PLAN_W256x192) return launch_w4<LAYOUT, W4_EPILOGUE, W4_PRODUCTION, 0, 256, 192>(a, b, c, M, N, K, swizzle, stride, st);
  if (plan == PLAN_W160) return launch_w4<LAYOUT, W4_EPILOGUE, W4_PRODUCTION, 0, 160, 160>(a, b, c, M, N, K, swizzle, stride, st);
  if (plan == PLAN_W128x256) return launch_w4<LAYOUT, W4_EPILOGUE, W4_PRODUCTION, 0, 128, 256>(a, b, c, M, N, K, swizzle, stride, st);
  if (plan == PLAN_W256x128) return launch_w4<LAYOUT, W4_EPILOGUE, W4_PRODUCTION, 0, 256, 128>(a, b, c, M, N, K, swizzle, stride, st);
  if (plan == PLAN_W256) {
    if (stages == 2) return launch_w4<LAYOUT, W4_EPILOGUE, W4_PRODUCTION>(a, b, c, M, N, K, swizzle, stride, st);
    // stages 3 / 4 / 5: the same one-wave-per-SIMD structure over a ring of `stages` 32-deep K slots (hgemm_w4s.cuh), bit-identical
    if (stages == 3 && w4s_k_ok(K, 3)) return launch_w4s<LAYOUT, 3, W4_EPILOGUE>(a, b, c, M, N, K, swizzle, stride, st);
    if (stages == 4 && w4s_k_ok(K, 4)) return launch_w4s<LAYOUT, 4, W4_EPILOGUE>(a, b, c, M, N, K, swizzle, stride, st);
    if (stages == 5 && w4s_k_ok(K, 5)) return launch_w4s<LAYOUT, 5, W4_EPILOGUE>(a, b, c, M, N, K, swizzle, stride, st);
    plan = PLAN_PP256;
  }
  if (plan == PLAN_PP192) return launch_pp<LAYOUT, 2, 4, 0, 0, 192>(a, b, c, M, N, K, swizzle, stride, st);
  if (plan == PLAN_PP256) {
    if (stages == 2 || stages < 2 || stages > 5) return launch_pp<LAYOUT, 2, 4, 0, 1>(a, b, c, M, N, K, swizzle, stride, st);
    if (stages == 4 && K % 32 == 0) return launch_pp32<LAYOUT, 2>(a, b, c, M, N, K, swizzle, stride, st);
  }
  const int tile = plan_tile(plan);
  return LAYOUT == TN ? ring_dispatch_tn(tile, a, b, c, M, N, K, stages, swizzle, stride, st)
                      : ring_dispatch_nn(tile, a, b, c, M, N, K, stages, swizzle, stride, st);
}

// what best_dispatch / ring_dispatch run for a shape, as text (cln_describe)
int describe_ring(int tile, int layout, int M, int N, int K, int stages, char* buf, int len) {
  int BM, BN, waves, BK = 32;
  tile_dims(tile, BM, BN, waves);
  if (M % BM || N % BN) return CLN_ERR_UNSUPPORTED;
  ring_pick(BM, BN, K, stages, BK);
  if (K % BK) return CLN_ERR_UNSUPPORTED;
  return snprintf(buf, len, "mfma_ring<%dx%dx%d,%d waves,stages=%d,%s>", BM, BN, BK, waves, stages, layout == TN ? "TN" : "NN");
}
int describe_w4(int BM, int BN, int layout, char* buf, int len, bool stages_ignored = false) {
  return snprintf(buf, len, "hgemm_w4<%dx%dx64,4 waves,%dx%d wave tiles,cross-tile LDS-DMA,LDS epilogue,%s>%s", BM, BN, BM / 2,
                  BN / 2, layout == TN ? "TN" : "NN", stages_ignored ? " [stages ignored: one pipeline]" : "");
}
int describe_best(int layout, int M, int N, int K, int stages, char* buf, int len) {
  int plan = best_plan(M, N, K);
  const char* l = layout == TN ? "TN" : "NN";
  const SplitK sk = splitk_plan(M, N, K);
  if (sk.S >= 2)
    return snprintf(buf, len, "hgemm_w4<%dx%dx64,4 waves,%dx%d wave tiles,cross-tile LDS-DMA,%s> split-K x %d (K %d per workgroup, fp32 partials in "
                              "register layout) + %s [stages ignored: one pipeline]", sk.bm, sk.bn, sk.bm / 2, sk.bn / 2, l, sk.S, K / sk.S,
                    sk.S <= splitk_fused_max_s() ? "in-kernel fix-up by the last-arriving workgroup (one launch)" : "hgemm_splitk_reduce");
  const TailSplit ts = tail_plan(M, N, K);
  if (ts.S >= 2)
    return snprintf(buf, len, "hgemm_w4<256x256x64,4 waves,128x128 wave tiles,cross-tile LDS-DMA,LDS epilogue,%s> on rows [0, %d) + the last %d tile rows as "
                              "split-K x %d (K %d per workgroup) + %s [tail split; stages ignored: one pipeline]", l, ts.m_split,
                    (M - ts.m_split) / 256, ts.S, K / ts.S, ts.S <= splitk_fused_max_s() ? "in-kernel fix-up" : "hgemm_splitk_reduce");
  if (plan == PLAN_W192) return describe_w4(192, 192, layout, buf, len, stages != 2);
  if (plan == PLAN_W192x256) return describe_w4(192, 256, layout, buf, len, stages != 2);
  if (plan == PLAN_W256x192) return describe_w4(256, 192, layout, buf, len, stages != 2);
  if (plan == PLAN_W160) return describe_w4(160, 160, layout, buf, len, stages != 2);
  if (plan == PLAN_W128x256) return describe_w4(128, 256, layout, buf, len, stages != 2);
  if (plan == PLAN_W256x128) return describe_w4(256, 128, layout, buf, len, stages != 2);
  if (plan == PLAN_W256) {
    if (stages == 2) return describe_w4(256, 256, layout, buf, len);
    if (stages >= 3 && stages <= 5 && w4s_k_ok(K, stages))
      return snprintf(buf, len, "hgemm_w4s<256x256,ring of %d x 32-deep K slots,4 waves,128x128 wave tiles,LDS-DMA %d slots ahead,LDS epilogue,%s>", stages,
                      stages - 1, l);
    plan = PLAN_PP256;
  }
  if (plan == PLAN_PP192) return snprintf(buf, len, "hgemm_pp<192x256x64,8 waves,4 slots,LDS epilogue,%s>", l);
  if (plan == PLAN_PP256) {
    if (stages == 2 || stages < 2 || stages > 5)
      return snprintf(buf, len, "hgemm_pp<256x256x64,8 waves,4 slots,split DMA,LDS epilogue,%s>", l);
    if (stages == 4 && K % 32 == 0) return snprintf(buf, len, "hgemm_pp32<256x256,BK=32 sub-tiles,4-deep ring,%s>", l);
  }
  return describe_ring(plan_tile(plan), layout, M, N, K, stages, buf, len);
}

using C1S_128_NN = Cfg<128, 128, 32, 2, 2, 1, NN>;
using C1S_64x128_NN = Cfg<64, 128, 32, 1, 2, 1, NN>;
using C1S_64_NN = Cfg<64, 64, 64, 2, 2, 1, NN>;
// The 1-stage rungs (register-staged, two barriers per K tile, no overlap -- config C2's "naive 1-stage MFMA tile"): every K
// tile exposes a full global-load latency, so a small problem wants (a) a workgroup on every CU and (b) fewer, deeper K
// tiles: below 256 tiles of 128x128 the 64x64x64 form of the same kernel runs (1024^3: 64 -> 256 workgroups, 32 -> 16 K
// tiles).
inline int launch_1stage_128_or_64(const void* a, const void* b, void* c, int M, int N, int K, hipStream_t stream) {
  if ((long long)(M / 128) * (N / 128) < 256 && M % 64 == 0 && N % 64 == 0 && K % 64 == 0)
    return launch_1stage<C1S_64_NN>(a, b, c, M, N, K, stream);
  return launch_1stage<C1S_128_NN>(a, b, c, M, N, K, stream);
}

}  // namespace

#define CLN_G3(name, expr)                                                                        \
  CLN_API int name(const void* a, const void* b, void* c, int M, int N, int K, void* stream_) {  \
    int rc = check_args(a, b, c, M, N, K);                                                        \
    if (rc != CLN_OK) return rc;                                                                  \
    hipStream_t stream = (hipStream_t)stream_;                                                    \
    return (expr);                                                                                \
  }
#define CLN_G6(name, expr)                                                                        \
  CLN_API int name(const void* a, const void* b, void* c, int M, int N, int K, int stages,       \
                   int swizzle, int swizzle_stride, void* stream_) {                              \
    int rc = check_args(a, b, c, M, N, K);                                                        \
    if (rc != CLN_OK) return rc;                                                                  \
    hipStream_t stream = (hipStream_t)stream_;                                                    \
    return (expr);                                                                                \
  }

// ---- VALU rungs (reference kernels/hgemm/naive/hgemm.cu:784-998, hgemm_async.cu:734-908) -------
CLN_G3(hgemm_naive_f16, launch_valu_naive(a, b, c, M, N, K, stream))
CLN_G3(hgemm_sliced_k_f16, launch_valu_sliced_k(a, b, c, M, N, K, stream))
CLN_G3(hgemm_t_8x8_sliced_k_f16x4, (launch_valu_tile<8, 8, false, false>(a, b, c, M, N, K, stream)))
CLN_G3(hgemm_t_8x8_sliced_k_f16x4_pack, (launch_valu_tile<8, 8, false, false>(a, b, c, M, N, K, stream)))
CLN_G3(hgemm_t_8x8_sliced_k_f16x4_bcf, (launch_valu_tile<8, 8, false, false>(a, b, c, M, N, K, stream)))
CLN_G3(hgemm_t_8x8_sliced_k_f16x4_pack_bcf, (launch_valu_tile<8, 8, false, false>(a, b, c, M, N, K, stream)))
CLN_G3(hgemm_t_8x8_sliced_k_f16x8_pack_bcf, (launch_valu_tile<8, 8, false, false>(a, b, c, M, N, K, stream)))
CLN_G3(hgemm_t_8x8_sliced_k_f16x8_pack_bcf_dbuf, (launch_valu_tile<8, 8, true, false>(a, b, c, M, N, K, stream)))
CLN_G3(hgemm_t_8x8_sliced_k16_f16x8_pack_dbuf, (launch_valu_tile<16, 8, true, false>(a, b, c, M, N, K, stream)))
CLN_G3(hgemm_t_8x8_sliced_k16_f16x8_pack_dbuf_async,
       (launch_valu_tile<16, 8, true, true>(a, b, c, M, N, K, stream)))
CLN_G3(hgemm_t_8x8_sliced_k32_f16x8_pack_dbuf, (launch_valu_tile<32, 8, true, false>(a, b, c, M, N, K, stream)))
CLN_G3(hgemm_t_8x8_sliced_k32_f16x8_pack_dbuf_async,
       (launch_valu_tile<32, 8, true, true>(a, b, c, M, N, K, stream)))
CLN_G3(hgemm_t_16x8_sliced_k32_f16x8_pack_dbuf, (launch_valu_tile<32, 16, true, false>(a, b, c, M, N, K, stream)))
CLN_G3(hgemm_t_16x8_sliced_k32_f16x8_pack_dbuf_async,
       (launch_valu_tile<32, 16, true, true>(a, b, c, M, N, K, stream)))

// Rungs whose NAME fixes the block tile (reference: 256x256 `mma4x4_warp4x4`, 256x128 `mma4x2_warp4x4`, CuTe 128x256):
// at stages = 2 they run the one-wave-per-SIMD kernel of that tile when the shape allows it (w4_k_ok: K % 64 == 0, >= 6 K tiles, >= 7 when odd),
// the multi-stage ring of the same tile otherwise (other stage counts, other K).
template <int LAYOUT, int BM, int BN>
static bool fixed_tile_runs_w4(int M, int N, int K, int stages) {
  return stages == 2 && M % BM == 0 && N % BN == 0 && w4_k_ok(K);
}
template <int LAYOUT, int BM, int BN>
static int fixed_tile_dispatch(int ring_tile, const void* a, const void* b, void* c, int M, int N, int K, int stages,
                               int swizzle, int swizzle_stride, hipStream_t stream) {
  if (fixed_tile_runs_w4<LAYOUT, BM, BN>(M, N, K, stages))
    return launch_w4<LAYOUT, W4_EPILOGUE, W4_PRODUCTION, 0, BM, BN>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);
  if constexpr (BM == 256 && BN == 256) {  // stages 3 / 4 / 5 on the 256x256 tile: the ring-of-slots form of the same kernel
    if (M % 256 == 0 && N % 256 == 0 && stages >= 3 && stages <= 5 && w4s_k_ok(K, stages)) {
      if (stages == 3) return launch_w4s<LAYOUT, 3, W4_EPILOGUE>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);
      if (stages == 4) return launch_w4s<LAYOUT, 4, W4_EPILOGUE>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);
      return launch_w4s<LAYOUT, 5, W4_EPILOGUE>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);
    }
  }
  if constexpr (LAYOUT == TN) return ring_dispatch_tn(ring_tile, a, b, c, M, N, K, stages, swizzle, swizzle_stride, stream);
  else return ring_dispatch_nn(ring_tile, a, b, c, M, N, K, stages, swizzle, swizzle_stride, stream);
}

// ---- matrix-core rungs, no `stages` argument ----------------------------------------------------
// reference kernels/hgemm/wmma/hgemm_wmma.cu:594-758, kernels/hgemm/mma/basic/hgemm_mma.cu:270-336
CLN_G3(hgemm_wmma_m16n16k16_naive, launch_naive<NN>(a, b, c, M, N, K, stream))
CLN_G3(hgemm_mma_m16n8k16_naive, launch_naive<NN>(a, b, c, M, N, K, stream))
CLN_G3(hgemm_wmma_m16n16k16_mma4x2, launch_1stage<C1S_64x128_NN>(a, b, c, M, N, K, stream))
CLN_G3(hgemm_wmma_m16n16k16_mma4x2_warp2x4, launch_1stage_128_or_64(a, b, c, M, N, K, stream))
CLN_G3(hgemm_mma_m16n8k16_mma2x4_warp4x4, launch_1stage_128_or_64(a, b, c, M, N, K, stream))
CLN_G3(hgemm_wmma_m16n16k16_mma4x2_warp2x4_dbuf_async,
       ring_exact_nn(T128, (K % 64 == 0) ? 64 : 32, 2, a, b, c, M, N, K, 0, 1, stream))
CLN_G3(hgemm_wmma_m32n8k16_mma2x4_warp2x4_dbuf_async,
       ring_exact_nn(T128, 32, 2, a, b, c, M, N, K, 0, 1, stream))

// ---- multi-stage rings --------------------------------------------------------------------------
// reference kernels/hgemm/wmma/hgemm_wmma_stage.cu:1001-1464
CLN_G6(hgemm_wmma_m16n16k16_mma4x2_warp2x4_stages,
       ring_dispatch_nn(T128, a, b, c, M, N, K, stages, swizzle, swizzle_stride, stream))
CLN_G6(hgemm_wmma_m16n16k16_mma4x2_warp2x4_stages_dsmem,
       ring_dispatch_nn(T128, a, b, c, M, N, K, stages, swizzle, swizzle_stride, stream))
CLN_G6(hgemm_wmma_m16n16k16_mma4x2_warp4x4_stages_dsmem,
       (fixed_tile_dispatch<NN, 256, 128>(T256x128, a, b, c, M, N, K, stages, swizzle, swizzle_stride, stream)))
CLN_G6(hgemm_wmma_m16n16k16_mma4x4_warp4x4_stages_dsmem,
       (fixed_tile_dispatch<NN, 256, 256>(T256, a, b, c, M, N, K, stages, swizzle, swizzle_stride, stream)))
// reference kernels/hgemm/mma/basic/hgemm_mma_stage.cu:2124-2717, mma/swizzle/hgemm_mma_stage_swizzle.cu:757
CLN_G6(hgemm_mma_m16n8k16_mma2x4_warp4x4_stages,
       ring_dispatch_nn(T128, a, b, c, M, N, K, stages, swizzle, swizzle_stride, stream))
CLN_G6(hgemm_mma_m16n8k16_mma2x4_warp4x4_stages_dsmem,
       ring_dispatch_nn(T128, a, b, c, M, N, K, stages, swizzle, swizzle_stride, stream))
CLN_G6(hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem,
       best_dispatch<NN>(a, b, c, M, N, K, stages, swizzle, swizzle_stride, stream))
CLN_G6(hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_x4,
       best_dispatch<NN>(a, b, c, M, N, K, stages, swizzle, swizzle_stride, stream))
CLN_G6(hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_rr,
       best_dispatch<NN>(a, b, c, M, N, K, stages, swizzle, swizzle_stride, stream))
CLN_G6(hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_swizzle,
       best_dispatch<NN>(a, b, c, M, N, K, stages, swizzle, swizzle_stride, stream))
// TN family: reference hgemm_mma_stage_tn.cu:517, hgemm_mma_stage_tn_swizzle_x4.cu:860,
// cutlass/hgemm_mma_stage_tn_cute.cu:521 (128x256 tile)
CLN_G6(hgemm_mma_m16n8k16_mma2x4_warp4x4_stages_dsmem_tn,
       ring_dispatch_tn(T128, a, b, c, M, N, K, stages, swizzle, swizzle_stride, stream))
CLN_G6(hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_tn_swizzle_x4,
       best_dispatch<TN>(a, b, c, M, N, K, stages, swizzle, swizzle_stride, stream))
CLN_G6(hgemm_mma_stages_block_swizzle_tn_cute,
       ((N % 256 == 0) ? fixed_tile_dispatch<TN, 128, 256>(T128x256, a, b, c, M, N, K, stages, swizzle, swizzle_stride, stream)
                       : ring_dispatch_tn(T128, a, b, c, M, N, K, stages, swizzle, swizzle_stride, stream)))

// ---- workspace entry points (include/cln_amd.h; not part of the reference surface: its bindings take only a, b, c --
// kernels/hgemm/mma/basic/hgemm_mma_stage.cu:2380-2413 -- and never allocate, kernels/hgemm/pybind/hgemm.cc:58-107: neither does this library
// unless asked to, cln_hgemm_library_workspace)
// bytes the split-K / tail-split plan of the best-dispatch names needs for (M, N, K); 0 = the shape runs single-pass and never touches a workspace
CLN_API size_t cln_hgemm_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const SplitK sk = splitk_plan(M, N, K);
  if (sk.S >= 2) return w4_splitk_ws_bytes(M, N, sk.S);
  const TailSplit ts = tail_plan(M, N, K);
  if (ts.S >= 2) return w4_splitk_ws_bytes(M - ts.m_split, N, ts.S);
  return 0;
}
// Caller-owned workspace for the launches on `stream` of the current device: `ptr` (device memory, 16-byte aligned, `bytes` long) replaces the
// library-owned buffer until it is withdrawn with ptr = NULL. Its first 4 KiB (arrival counters) are zeroed ON THE STREAM by this call; the
// caller keeps the region alive and untouched while launches that may use it are queued. A shape that needs more than `bytes` runs single-pass.
CLN_API int cln_hgemm_set_workspace(void* ptr, size_t bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return (void)hipGetLastError(), CLN_ERR_LAUNCH;
  if (ptr && (!cln_aligned16(ptr) || bytes < (size_t)W4_TICKET_FLOATS * 4 + 16)) return CLN_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lock(g_ws_mu);
  for (size_t i = 0; i < g_ws.size(); ++i)
    if (g_ws[i].dev == dev && g_ws[i].stream == st) {
      if (!g_ws[i].user && g_ws[i].pinned) {
        // a captured graph still holds this library-owned region: it stays allocated (and counted by cln_hgemm_workspace_held) under a key no stream
        // has, until cln_release_workspaces()
        g_ws[i].stream = reinterpret_cast<hipStream_t>(~(uintptr_t)0 - g_ws_clock++);
        break;
      }
      if (!g_ws[i].user) ws_free_entry(g_ws[i]);
      g_ws.erase(g_ws.begin() + i);
      break;
    }
  if (!ptr) return CLN_OK;
  if (hipMemsetAsync(ptr, 0, W4_TICKET_FLOATS * 4, st) != hipSuccess) return (void)hipGetLastError(), CLN_ERR_LAUNCH;
  SplitKWs w;
  w.dev = dev, w.stream = st, w.p = (float*)ptr, w.bytes = bytes, w.user = true, w.used = ++g_ws_clock;
  g_ws.push_back(w);
  return CLN_OK;
}
// Library-owned workspaces on (1) / off (0, the default): the opt-in of a C caller that has no allocator of its own. On: a stream without a caller-owned
// region gets a library buffer on its first split-K shape (at most SPLITK_OWNED_MAX at a time, least recently used freed after its last launch's
// event; never allocated or freed under stream capture; one a capture has used is pinned until cln_release_workspaces()). Off: such a stream runs
// single-pass; buffers already held stay held until cln_release_workspaces(). Returns the previous setting.
CLN_API int cln_hgemm_library_workspace(int enable) {
  std::lock_guard<std::mutex> lock(g_ws_mu);
  const int was = g_lib_ws ? 1 : 0;
  g_lib_ws = enable != 0;
  return was;
}
// Frees every library-owned workspace (after the launches that used it have completed) and forgets the caller-owned ones. Returns the bytes freed.
CLN_API size_t cln_release_workspaces(void) {
  std::lock_guard<std::mutex> lock(g_ws_mu);
  size_t freed = 0;
  for (auto& w : g_ws)
    if (!w.user) freed += w.bytes, ws_free_entry(w);
  g_ws.clear();
  return freed + cln_stream_scratch_release();  // + the scratch slabs of the scalar-result kernels (stream_scratch.h)
}
// bytes of library-owned workspace currently held by this process (all devices, all streams)
CLN_API size_t cln_hgemm_workspace_held(void) {
  std::lock_guard<std::mutex> lock(g_ws_mu);
  size_t held = 0;
  for (auto& w : g_ws)
    if (!w.user) held += w.bytes;
  return held;
}

// describe hook of this library group (see cln_describe in describe.hip): the kernel a G6 name runs for (M, N, K,
// stages); CLN_ERR_BAD_ARG when `name` is not one of the run-time dispatched HGEMM names (every other HGEMM name is
// one fixed kernel: manifest.py `impl`).
int cln_hgemm_describe(const char* name, int M, int N, int K, int stages, char* buf, int len) {
  struct Row { const char* name; int kind; int tile; int layout; };  // kind 0: ring_dispatch(tile), 1: best_dispatch, 2: fixed_tile_dispatch
  static const Row rows[] = {
      {"hgemm_wmma_m16n16k16_mma4x2_warp2x4_stages", 0, T128, NN},
      {"hgemm_wmma_m16n16k16_mma4x2_warp2x4_stages_dsmem", 0, T128, NN},
      {"hgemm_wmma_m16n16k16_mma4x2_warp4x4_stages_dsmem", 2, T256x128, NN},
      {"hgemm_wmma_m16n16k16_mma4x4_warp4x4_stages_dsmem", 2, T256, NN},
      {"hgemm_mma_m16n8k16_mma2x4_warp4x4_stages", 0, T128, NN},
      {"hgemm_mma_m16n8k16_mma2x4_warp4x4_stages_dsmem", 0, T128, NN},
      {"hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem", 1, 0, NN},
      {"hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_x4", 1, 0, NN},
      {"hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_rr", 1, 0, NN},
      {"hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_swizzle", 1, 0, NN},
      {"hgemm_mma_m16n8k16_mma2x4_warp4x4_stages_dsmem_tn", 0, T128, TN},
      {"hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_tn_swizzle_x4", 1, 0, TN},
      {"hgemm_mma_stages_block_swizzle_tn_cute", 2, -1, TN},
  };
  if (M <= 0 || N <= 0 || K <= 0) return CLN_ERR_BAD_ARG;
  for (const Row& r : rows) {
    if (strcmp(r.name, name) != 0) continue;
    if (r.kind == 1) return describe_best(r.layout, M, N, K, stages, buf, len);
    const int tile = r.tile >= 0 ? r.tile : ((N % 256 == 0) ? T128x256 : T128);
    if (r.kind == 2 && tile != T128) {
      int BM, BN, waves;
      tile_dims(tile, BM, BN, waves);
      if (stages == 2 && M % BM == 0 && N % BN == 0 && w4_k_ok(K))
        return describe_w4(BM, BN, r.layout, buf, len);
      if (tile == T256 && stages >= 3 && stages <= 5 && M % 256 == 0 && N % 256 == 0 && w4s_k_ok(K, stages))
        return snprintf(buf, len, "hgemm_w4s<256x256,ring of %d x 32-deep K slots,4 waves,128x128 wave tiles,LDS-DMA %d slots ahead,LDS epilogue,%s>", stages,
                        stages - 1, r.layout == TN ? "TN" : "NN");
    }
    return describe_ring(tile, r.layout, M, N, K, stages, buf, len);
  }
  return CLN_ERR_BAD_ARG;
}
