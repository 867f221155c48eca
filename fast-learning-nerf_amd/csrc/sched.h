// sched.h -- dynamic tile scheduling shared by the persistent forward / dX kernels of mlp_fwd / mlp_bwd_*.hip and mlp_bf16.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>

// Dynamic tile scheduling of the persistent forward / dX kernels: the two workgroups of a CU do not run at the same
// speed (the first-dispatched one wins the arbitration: 198 k vs 243 k cycles per tile, tools/trace_fwd.py), so a static
// round-robin leaves the slower half ~4 tiles behind at the end.  Every workgroup takes tile blockIdx.x first and then
// draws tickets from a counter in global memory; the last workgroup to leave resets the counter pair, so a launch never
// depends on host-side state (graph replay safe).  Concurrent launches (different streams) must not share a pair: the
// host hands out pairs round-robin from a pool of BSCHED_SLOTS.
#define BSCHED_SLOTS 64
static inline unsigned* b_sched_pair() {
  static unsigned* pool[16] = {};
  static unsigned next[16] = {};
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  if (!pool[dev]) {
    unsigned* p = nullptr;
    if (hipMalloc(&p, BSCHED_SLOTS * 2 * sizeof(unsigned)) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, BSCHED_SLOTS * 2 * sizeof(unsigned)) != hipSuccess) return nullptr;
    pool[dev] = p;
  }
  const unsigned k = next[dev]++ % BSCHED_SLOTS;
  return pool[dev] + 2 * k;
}
// end of a tile: thread 0 draws the next ticket into the LDS word `slot` (a place nobody reads or writes around the
// tile boundary), the tile's closing barrier publishes it
__device__ __forceinline__ int64_t b_next_tile(unsigned* sched, volatile int* slot, int tid) {
  if (tid == 0) *slot = (int)(atomicAdd(sched, 1u) + gridDim.x);
  __syncthreads();
  return (int64_t)__builtin_amdgcn_readfirstlane(*slot);
}
__device__ __forceinline__ void b_sched_exit(unsigned* sched, int tid) {
  if (tid == 0) {
    __threadfence();
    if (atomicAdd(sched + 1, 1u) == gridDim.x - 1) {   // everybody else has drawn its last (failing) ticket
      sched[0] = 0u; sched[1] = 0u;
      __threadfence();
    }
  }
}

