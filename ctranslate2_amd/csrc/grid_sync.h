// Device-wide phase barrier and cross-workgroup hand-over WITHOUT release / acquire fences.
//
// On gfx950 an agent-scope release or acquire fence writes back / invalidates the whole L2 of the
// XCD that executes it (48 us per round for 256 workgroups, profiles/r02_xcd_barrier_probe.txt; 2x
// on the decode step when every workgroup of a launch executed one). The forms below use RELAXED
// device-scope atomics only - they are performed at the memory side, which is the one point all
// eight XCDs agree on - and were measured by tools/relaxed_barrier_probe.cpp
// (profiles/r04_relaxed_barrier_probe.txt): 2.0 / 2.9 / 4.4 us per round at 32 / 128 / 256
// workgroups, no stale value in 400 rounds. The fused greedy sampler (gemm_s8.hip, ARGMAX form) is
// built on the same rules.
//
// Rules (each one was a bug or a measurement once):
//   * data another workgroup will read inside the SAME launch is published with grid_publish (an
//     atomic store: write-through) and read with grid_observe (an atomic load: never served from a
//     stale L2 line). Plain stores / loads are only coherent across a kernel boundary.
//   * the publisher drains its stores (`s_waitcnt vmcnt(0)`) BEFORE it arrives at the barrier.
//   * polls are bounded: a lost wake-up must surface as an error flag, never as a hung GPU.
//   * every workgroup that waits must be resident: use it with grids of at most one workgroup per
//     CU-slot the kernel's resources allow (persistent kernels), or let only workgroups wait that
//     have finished their own work (the "last ticket finishes" form needs no residency at all).
//
// Round 5: grid_publish / grid_observe are the stores and loads of the decoder-stack launches'
// hand-over (decoder_cluster.hip: 64-bit {tag, value} slots, or plain lines behind an arrival
// counter); the device-wide barrier below was measured (4.4 us at 256 workgroups) and NOT built
// on - the launches meet in clusters of eight, never device-wide.
#pragma once

#include <hip/hip_runtime.h>

namespace ct2x {

template <typename T>
__device__ __forceinline__ void grid_publish(T* p, T v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename T>
__device__ __forceinline__ T grid_observe(const T* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// All threads of the workgroup call. `counter` counts arrivals monotonically (never reset inside a
// launch); `target` = arrivals that complete this phase (phase index x workgroups). Returns false
// when the bounded poll ran out (the caller records it and carries on: results are then wrong but
// the device is not hung).
__device__ __forceinline__ bool grid_barrier_relaxed(unsigned* counter, unsigned target,
                                                     unsigned spin_limit = 1u << 22) {
  __shared__ int ok;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's published stores have left
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    int good = 1;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > spin_limit) {
        good = 0;
        break;
      }
    }
    ok = good;
  }
  __syncthreads();
  return ok != 0;
}

// "Last ticket finishes": every workgroup calls it once, after its published stores / atomics; the
// one that returns true is the last to arrive and may read what all the others published (with
// grid_observe / atomics). It must reset *ticket to 0 before the launch ends if the launch is
// replayed from a graph. No workgroup waits, so no residency requirement.
__device__ __forceinline__ bool grid_last_arrival(unsigned* ticket, unsigned workgroups) {
  __shared__ unsigned mine;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) mine = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  return mine == workgroups - 1;
}

}  // namespace ct2x
