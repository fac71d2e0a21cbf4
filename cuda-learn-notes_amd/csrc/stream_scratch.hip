// stream_scratch.h: the host side (slot table, slabs).
#include "stream_scratch.h"
#include <mutex>
#include <vector>

namespace {
constexpr int SLOTS = 64, MAX_DEV = 64;
struct Slot {
  hipStream_t stream = nullptr;
  bool live = false;
  unsigned long long used = 0;
};
struct DevSlab {
  ClnScratch* p = nullptr;
  Slot slot[SLOTS];
};
std::mutex g_mu;
DevSlab g_slab[MAX_DEV];
unsigned long long g_clock = 0;

bool capturing(hipStream_t st) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess) return (void)hipGetLastError(), true;
  return cs != hipStreamCaptureStatusNone;
}
}  // namespace

ClnScratch* cln_stream_scratch(hipStream_t stream) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return (void)hipGetLastError(), nullptr;
  // A captured launch NEVER gets a slot (round 6): a graph replays on whatever stream is current, so a slot baked into it could be hit by an
  // eager launch of its capture stream -- or by a second replay -- at the same time, and a lost or doubled ticket never recovers. The caller's
  // fallback (memset node + one atomicAdd per block into y) has no state outside the call.
  if (capturing(stream)) return nullptr;
  std::lock_guard<std::mutex> lock(g_mu);
  DevSlab& d = g_slab[dev];
  if (d.p) {
    for (int i = 0; i < SLOTS; ++i)
      if (d.slot[i].live && d.slot[i].stream == stream) return d.slot[i].used = ++g_clock, d.p + i;
  }
  if (!d.p) {
    ClnScratch* fresh = nullptr;
    if (hipMalloc(&fresh, sizeof(ClnScratch) * SLOTS) != hipSuccess || hipMemset(fresh, 0, sizeof(ClnScratch) * SLOTS) != hipSuccess) {
      (void)hipGetLastError();
      if (fresh) (void)hipFree(fresh);
      return nullptr;
    }
    d.p = fresh;
  }
  int pick = -1;
  for (int i = 0; i < SLOTS && pick < 0; ++i)
    if (!d.slot[i].live) pick = i;
  if (pick < 0) {  // all 64 in use: take the least recently used one once everything queued on the device is done (its last launch left it zeroed)
    for (int i = 0; i < SLOTS; ++i)
      if (pick < 0 || d.slot[i].used < d.slot[pick].used) pick = i;
    if (hipDeviceSynchronize() != hipSuccess) return (void)hipGetLastError(), nullptr;
  }
  d.slot[pick].stream = stream, d.slot[pick].live = true, d.slot[pick].used = ++g_clock;
  return d.p + pick;
}

size_t cln_stream_scratch_release() {
  std::lock_guard<std::mutex> lock(g_mu);
  size_t freed = 0;
  int cur = 0;
  (void)hipGetDevice(&cur);
  for (int dev = 0; dev < MAX_DEV; ++dev) {
    DevSlab& d = g_slab[dev];
    if (!d.p) continue;
    if (hipSetDevice(dev) == hipSuccess) {
      (void)hipDeviceSynchronize();
      (void)hipFree(d.p);
      freed += sizeof(ClnScratch) * SLOTS;
    }
    d = DevSlab();
  }
  (void)hipSetDevice(cur);
  (void)hipGetLastError();
  return freed;
}
