// Per-(device, stream) scratch words for the scalar-result kernels (block_all_reduce_sum_*, dot_prod_*): round 5, VERDICT r4 #4.
//
// The reference bindings allocate the result with torch::zeros (kernels/reduce/block_all_reduce.cu:737-738, dot_product.cu:236-238) and the
// kernels add one atomic per block into it: every call is TWO dispatches (fill + kernel), 8.4 us of host time against torch.sum's 4.
// Here the blocks add into a library-owned, zero-initialised scratch word; the block that takes the LAST ticket moves the total into y
// (overwriting it -- y no longer has to be zeroed) and leaves sum and ticket at zero for the next launch: one dispatch, no fill.
//   * one 256-byte slot per (device, stream): launches on one stream run in order, so they can share it; 64 slots per device, the least
//     recently used one re-assigned after a hipDeviceSynchronize() when a 65th stream shows up (a process that cycles streams never leaks);
//   * nullptr when the slot cannot be had without touching the device (first use or eviction while the stream is being captured, allocation
//     failure): the caller then zeroes y on the stream (a memset node under capture) and the kernel adds into y directly, as before;
//   * a captured launch holds its capture stream's slot: replay graphs on the capture stream (include/cln_amd.h, workspace notes).
#pragma once
#include "common.h"

struct ClnScratch {
  float sum;            // fp32 or int32 bits
  unsigned pad0[31];
  unsigned ticket;      // blocks arrived
  unsigned pad1[31];
};
static_assert(sizeof(ClnScratch) == 256, "one slot = 256 bytes");

ClnScratch* cln_stream_scratch(hipStream_t stream);
size_t cln_stream_scratch_release();  // frees every slab (cln_release_workspaces); returns the bytes freed

#if defined(__HIPCC__)
// Device side: `t` is this block's partial (one thread per block calls it). Returns nothing; the last block writes *y.
template <typename O>
__device__ __forceinline__ void cln_scratch_finish(ClnScratch* sc, O* y, O t, unsigned nblocks) {
  O* sum = reinterpret_cast<O*>(&sc->sum);
  O old = __hip_atomic_fetch_add(sum, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // the add has been PERFORMED at the point of coherence once its return value is here; only then is the ticket taken -- no L2 write-back /
  // invalidate (what a release fence would cost every block), just one returning atomic
  asm volatile("" : "+v"(old)::"memory");
  const unsigned tk = __hip_atomic_fetch_add(&sc->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tk == nblocks - 1) {  // every other block's add precedes its ticket, all tickets precede this one
    *y = __hip_atomic_exchange(sum, (O)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&sc->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
#endif
