// Two measurements that decide the shape of the round-5 persistent decoder-stack kernel
// (DESIGN.md 8.000):
//
//  A. What ONE compute unit pulls through its XCD's L2 when G workgroups (one per CU) all walk the
//     SAME weight set - the cost model of a "row-resident" decoder stack where a workgroup owns a
//     batch row for all layers (no device-wide barrier at all, but every workgroup streams every
//     weight byte): us per 3.7 MB layer and GB/s per CU at 32 / 64 / 128 / 256 workgroups, with
//     the int8 dot products (v_dot4, activation chunk broadcast from LDS) in the loop.
//
//  B. What a hand-over between the workgroups of a small CLUSTER costs when all of them sit on the
//     same XCD (they share one L2): C workgroups publish 256 B each, arrive, wait, read their
//     peers' lines. Variants of the memory instructions (stale reads are COUNTED, polls bounded):
//       0  agent-scope relaxed atomic store / load (sc1: performed at the memory side) - round 4's form
//       1  L2 read-modify-write atomics without sc1 for the flag AND the data (fetch_add 0 = load)
//       2  plain stores + vmcnt(0), flag by L2 RMW, data by sc0 ("workgroup scope") atomic loads
//       3  plain stores + vmcnt(0), flag by L2 RMW, then buffer_inv sc1 and plain loads
//     Workgroups find their cluster from HW_REG_XCC_ID and a per-XCD ticket, not from blockIdx.
//
//   hipcc --offload-arch=gfx950 -O3 -o build/tools/row_resident_probe tools/experiments/row_resident_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); exit(1);} } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

// ---------------------------------------------------------------- A: per-CU stream through the L2
// weights: [layers][bytes_per_layer], packed so that a wave's 64 lanes read 1 KB contiguous.
// Every workgroup walks all of it; lane accumulates dot4 against a 16-byte activation chunk that
// all lanes read from the same LDS address (broadcast).
template <int THREADS, int INFLIGHT>
__global__ __launch_bounds__(THREADS) void stream_all_kernel(const v4i* __restrict__ w, size_t n16_per_layer,
                                                             int layers, int reps, int* sink) {
  __shared__ v4i xq[64];
  if (threadIdx.x < 64) xq[threadIdx.x] = v4i{(int)threadIdx.x * 0x01010101, 0x01020304, 0x7f7f7f7f, -1};
  __syncthreads();
  int acc = 0;
  for (int r = 0; r < reps; ++r) {
    for (int l = 0; l < layers; ++l) {
      const v4i* base = w + static_cast<size_t>(l) * n16_per_layer;
      size_t i = threadIdx.x;
      for (; i + (INFLIGHT - 1) * THREADS < n16_per_layer; i += INFLIGHT * THREADS) {
        v4i t[INFLIGHT];
#pragma unroll
        for (int u = 0; u < INFLIGHT; ++u) t[u] = base[i + u * THREADS];
#pragma unroll
        for (int u = 0; u < INFLIGHT; ++u) {
          const v4i x = xq[(i / THREADS + u) & 63];
          acc = __builtin_amdgcn_sdot4(t[u].x, x.x, acc, false);
          acc = __builtin_amdgcn_sdot4(t[u].y, x.y, acc, false);
          acc = __builtin_amdgcn_sdot4(t[u].z, x.z, acc, false);
          acc = __builtin_amdgcn_sdot4(t[u].w, x.w, acc, false);
        }
      }
    }
  }
  if (acc == 0x12345678) sink[0] = acc;
}

template <int THREADS, int INFLIGHT>
static void run_stream(int grid, const v4i* w, size_t bytes_per_layer, int layers, int* sink) {
  const int reps = 4;
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  hipLaunchKernelGGL((stream_all_kernel<THREADS, INFLIGHT>), dim3(grid), dim3(THREADS), 0, 0, w, bytes_per_layer / 16,
                     layers, 1, sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((stream_all_kernel<THREADS, INFLIGHT>), dim3(grid), dim3(THREADS), 0, 0, w, bytes_per_layer / 16,
                     layers, reps, sink);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  const double us_layer = 1e3 * ms / (reps * layers);
  printf("A  grid %3d x %4d threads, %d x 16 B in flight: %7.2f us per %.2f MB layer  = %6.1f GB/s per CU, %6.2f TB/s L2->CU total\n",
         grid, THREADS, INFLIGHT, us_layer, bytes_per_layer / 1048576.0, bytes_per_layer / us_layer * 1e-3,
         grid * bytes_per_layer / us_layer * 1e-6);
}

// ---------------------------------------------------------------- B: cluster hand-over on one XCD
struct Ctl {
  unsigned xcd_ticket[16];
  unsigned errors, timeouts, future, clusters;
};

constexpr unsigned kSpinLimit = 1u << 18;
constexpr int kLine = 64;  // unsigned per workgroup and round (256 B)

template <int MODE>
__device__ __forceinline__ void put(unsigned* p, unsigned v) {
  if (MODE == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else if (MODE == 1) __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else *reinterpret_cast<volatile unsigned*>(p) = v;
}
template <int MODE>
__device__ __forceinline__ unsigned get(unsigned* p) {
  if (MODE == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (MODE == 1) return __hip_atomic_fetch_add(p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (MODE == 2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return *reinterpret_cast<volatile unsigned*>(p);
}
template <int MODE>
__device__ __forceinline__ unsigned poll(unsigned* p) {
  if (MODE == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __hip_atomic_fetch_add(p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <int MODE>
__device__ __forceinline__ void arrive(unsigned* p) {
  if (MODE == 0) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// counters: [clusters_total] unsigned, 64-byte apart; data: [clusters_total][C][kLine]
template <int MODE>
__global__ __launch_bounds__(256) void cluster_rounds(Ctl* ctl, unsigned* counters, unsigned* data, int C, int rounds,
                                                      unsigned base, int per_xcd) {
  __shared__ unsigned s_slot;
  if (threadIdx.x == 0) {
    const unsigned x = xcc_id();
    const unsigned t = __hip_atomic_fetch_add(&ctl->xcd_ticket[x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_slot = (t < static_cast<unsigned>(per_xcd)) ? (x * per_xcd + t) : 0xffffffffu;
  }
  __syncthreads();
  const unsigned slot = s_slot;
  if (slot == 0xffffffffu) return;  // this XCD got more workgroups than planned: they sit out
  const unsigned cluster = slot / C, me = slot % C;
  unsigned* counter = counters + static_cast<size_t>(cluster) * 16;
  unsigned* mine = data + (static_cast<size_t>(cluster) * C + me) * kLine;
  for (int r = 0; r < rounds; ++r) {
    const unsigned tag = base + static_cast<unsigned>(r) + 1u;
    if (threadIdx.x < kLine) put<MODE>(mine + threadIdx.x, tag);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      arrive<MODE>(counter);
      const unsigned target = tag * C;
      unsigned spins = 0;
      while (poll<MODE>(counter) < target) {
        if (++spins > kSpinLimit) { atomicAdd(&ctl->timeouts, 1u); break; }
      }
    }
    __syncthreads();
    if (MODE == 3) asm volatile("buffer_inv sc1" ::: "memory");
    // read every peer's line (C - 1 lines of 256 B), 64 lanes per line
    for (int idx = threadIdx.x; idx < (C - 1) * kLine; idx += 256) {
      const unsigned peer = (me + 1 + idx / kLine) % C;
      const unsigned v = get<MODE>(data + (static_cast<size_t>(cluster) * C + peer) * kLine + (idx % kLine));
      if (v < tag) atomicAdd(&ctl->errors, 1u);
      else if (v > tag) atomicAdd(&ctl->future, 1u);
    }
  }
}

template <int MODE>
static void run_cluster(int C, int per_xcd, Ctl* ctl, unsigned* counters, unsigned* data, const char* label) {
  const int rounds = 400;
  const int grid = 8 * per_xcd;
  CK(hipMemset(counters, 0, 4096 * 64));
  CK(hipMemset(data, 0, sizeof(unsigned) * 256 * kLine));
  CK(hipMemset(ctl, 0, sizeof(Ctl)));
  hipLaunchKernelGGL(cluster_rounds<MODE>, dim3(grid), dim3(256), 0, 0, ctl, counters, data, C, 4, 0u, per_xcd);
  CK(hipDeviceSynchronize());
  Ctl h0;
  CK(hipMemcpy(&h0, ctl, sizeof(h0), hipMemcpyDeviceToHost));
  CK(hipMemset(ctl, 0, sizeof(Ctl)));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(cluster_rounds<MODE>, dim3(grid), dim3(256), 0, 0, ctl, counters, data, C, rounds, 4u, per_xcd);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  Ctl h;
  CK(hipMemcpy(&h, ctl, sizeof(h), hipMemcpyDeviceToHost));
  unsigned lo = 1u << 30, hi = 0;
  for (int x = 0; x < 8; ++x) { lo = h.xcd_ticket[x] < lo ? h.xcd_ticket[x] : lo; hi = h.xcd_ticket[x] > hi ? h.xcd_ticket[x] : hi; }
  printf("B  mode %d (%s) cluster of %2d, %2d workgroups per XCD: %7.3f us/round  stale=%u timeouts=%u ahead=%u  (per-XCD arrivals %u..%u; warm-up stale=%u timeouts=%u)\n",
         MODE, label, C, per_xcd, 1e3 * ms / rounds, h.errors, h.timeouts, h.future, lo, hi, h0.errors, h0.timeouts);
}

int main() {
  // ---- A
  const size_t layer_bytes = 3735552;  // Transformer-base decoder layer: 3.56 MiB of int8 weights
  const int layers = 6;
  v4i* w;
  int* sink;
  CK(hipMalloc(&w, layer_bytes * layers));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(w, 1, layer_bytes * layers));
  for (int grid : {8, 32, 64, 128, 256}) run_stream<512, 8>(grid, w, layer_bytes, layers, sink);
  for (int grid : {32, 128}) run_stream<1024, 4>(grid, w, layer_bytes, layers, sink);
  for (int grid : {32, 128}) run_stream<1024, 8>(grid, w, layer_bytes, layers, sink);
  for (int grid : {32, 128}) run_stream<256, 16>(grid, w, layer_bytes, layers, sink);
  // ---- B
  Ctl* ctl;
  unsigned *counters, *data;
  CK(hipMalloc(&ctl, sizeof(Ctl)));
  CK(hipMalloc(&counters, 4096 * 64));
  CK(hipMalloc(&data, sizeof(unsigned) * 256 * kLine));
  const char* names[4] = {"sc1 atomics        ", "L2 RMW, no sc1     ", "plain st, sc0 ld   ", "plain st, inv, ld  "};
  for (int C : {2, 4, 8}) {
    for (int per_xcd : {8, 32}) {
      if (per_xcd % C) continue;
      run_cluster<0>(C, per_xcd, ctl, counters, data, names[0]);
      run_cluster<1>(C, per_xcd, ctl, counters, data, names[1]);
      run_cluster<2>(C, per_xcd, ctl, counters, data, names[2]);
      run_cluster<3>(C, per_xcd, ctl, counters, data, names[3]);
    }
  }
  run_cluster<0>(32, 32, ctl, counters, data, names[0]);
  run_cluster<1>(32, 32, ctl, counters, data, names[1]);
  run_cluster<3>(32, 32, ctl, counters, data, names[3]);
  return 0;
}
