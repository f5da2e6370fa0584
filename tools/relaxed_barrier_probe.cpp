// What a device-wide phase barrier costs on the MI355X when it is built from RELAXED device-scope
// atomics and atomic (write-through) data stores instead of release / acquire fences (which write
// back and invalidate an XCD's whole L2: 48 us per round for 256 workgroups,
// profiles/r02_xcd_barrier_probe.txt). Every round each workgroup publishes a value with an
// agent-scope atomic store, arrives with one relaxed atomicAdd, polls the counter, then reads its
// neighbour's value with an agent-scope atomic load and checks it. Bounded polls: a lost wake-up
// shows up as a timeout, never as a hang.
//
//   hipcc --offload-arch=gfx950 -O3 -o build/tools/relaxed_barrier_probe tools/relaxed_barrier_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); exit(1);} } while (0)

struct Ctl {
  unsigned counter;
  unsigned errors;    // STALE values: the hand-over was not coherent
  unsigned timeouts;
  unsigned future;    // the peer had already published the next round (not an error: it ran ahead)
};

constexpr unsigned kSpinLimit = 1u << 20;

// payload: floats per workgroup published every round (the hand-over a fused kernel would do)
template <int PAYLOAD>
__global__ __launch_bounds__(256) void barrier_rounds(Ctl* ctl, unsigned* data, int rounds, unsigned base) {
  const unsigned wg = blockIdx.x, n = gridDim.x;
  for (int r = 0; r < rounds; ++r) {
    const unsigned tag = base + static_cast<unsigned>(r) + 1u;
    for (int i = threadIdx.x; i < PAYLOAD; i += 256)
      __hip_atomic_store(data + static_cast<size_t>(wg) * PAYLOAD + i, tag, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    // the stores have left when vmcnt drains (write-through stores are acknowledged by the memory side)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(&ctl->counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (base + static_cast<unsigned>(r) + 1u) * n;
      unsigned spins = 0;
      while (__hip_atomic_load(&ctl->counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > kSpinLimit) { atomicAdd(&ctl->timeouts, 1u); break; }
      }
    }
    __syncthreads();
    const unsigned peer = (wg + 1u) % n;
    for (int i = threadIdx.x; i < PAYLOAD; i += 256) {
      const unsigned v = __hip_atomic_load(data + static_cast<size_t>(peer) * PAYLOAD + i, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
      if (v < tag) atomicAdd(&ctl->errors, 1u);
      else if (v > tag) atomicAdd(&ctl->future, 1u);
    }
  }
}

template <int PAYLOAD>
static void run(int grid, int rounds, Ctl* ctl, unsigned* data, const char* label) {
  CK(hipMemset(ctl, 0, sizeof(Ctl)));
  CK(hipMemset(data, 0, sizeof(unsigned) * 1024 * 4096));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  hipLaunchKernelGGL(barrier_rounds<PAYLOAD>, dim3(grid), dim3(256), 0, 0, ctl, data, 4, 0u);  // warm-up
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(barrier_rounds<PAYLOAD>, dim3(grid), dim3(256), 0, 0, ctl, data, rounds, 4u);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  Ctl h;
  CK(hipMemcpy(&h, ctl, sizeof(h), hipMemcpyDeviceToHost));
  printf("%-58s grid %4d: %7.3f us/round  stale=%u timeouts=%u (peer already a round ahead: %u)\n", label, grid,
         1e3 * ms / rounds, h.errors, h.timeouts, h.future);
}

int main() {
  Ctl* ctl;
  unsigned* data;
  CK(hipMalloc(&ctl, sizeof(Ctl)));
  CK(hipMalloc(&data, sizeof(unsigned) * 1024 * 4096));
  const int rounds = 400;
  for (int grid : {32, 64, 128, 256, 512}) run<1>(grid, rounds, ctl, data, "relaxed barrier, 4 B published per workgroup");
  for (int grid : {128, 256}) run<256>(grid, rounds, ctl, data, "relaxed barrier, 1 KB published per workgroup");
  for (int grid : {128, 256}) run<4096>(grid, rounds, ctl, data, "relaxed barrier, 16 KB published per workgroup");
  return 0;
}
