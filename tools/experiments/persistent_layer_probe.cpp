// NOT RUN YET (written with the round's GPU budget spent; compile-checked only). The experiment
// DESIGN.md §8.00 names as the next one: how much of a Llama-shaped decode layer's wall time comes
// back when the layer is ONE persistent launch whose workgroups keep the weight stream busy across
// the row-wide phases, instead of nine dependent launches during five of which HBM idles.
//
// Model of a layer (rocprofv3 of config.l8, profiles/r04_final_kernel_stats_bench_l8.csv):
//   norm (latency chain, ~3 us of dependent work on 16 workgroups)  -> QKV stream   25 MB
//   attention (modelled as a 6 us latency phase)                    -> quantise (latency, ~1 us)
//   -> out-proj stream 17 MB -> norm (latency) -> gated pair stream 117 MB -> quantise (latency)
//   -> down-proj stream 59 MB
// Form A ("launches"): every phase is its own launch, the layers of a step replayed from a hipGraph.
// Form B ("persistent"): one launch per layer; the phases are separated by the relaxed-atomic
//   barrier of tools/relaxed_barrier_probe.cpp (4.4 us at 256 workgroups); during a latency phase the
//   workgroups that have no row to work on PREFETCH the head of the next stream into LDS (a
//   64 KB ring per workgroup), so the stream phase starts with its first tile in hand and the
//   memory system never drains.
// Both forms read every weight byte once (grid-stride 16-byte loads, folded into a checksum that is
// compared between the forms) from a 1.7 GB region walked layer by layer, so the MALL does not help.
//
//   hipcc --offload-arch=gfx950 -O3 -o build/tools/persistent_layer_probe tools/experiments/persistent_layer_probe.cpp
//   build/tools/persistent_layer_probe            # prints us per layer for both forms
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); exit(1);} } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int kWgs = 256, kThreads = 1024;
constexpr int kStreams = 4;
static const size_t kStreamBytes[kStreams] = {25u << 20, 17u << 20, 117u << 20, 59u << 20};
constexpr int kLayers = 8;  // 8 x 218 MB = 1.7 GB walked once per "step"
constexpr unsigned kSpinLimit = 1u << 22;
constexpr int kRingBytes = 64 << 10;  // LDS bytes a workgroup prefetches during a latency phase

struct Ctl {
  unsigned counter;
  unsigned timeouts;
  unsigned long long checksum;
  unsigned pad[2];
};

// ~`steps` dependent FMAs on every lane: the stand-in for the ordered RMSNorm chain (512 steps at
// depth 4096 = ~3 us) or the attention phase
__device__ __forceinline__ float latency_chain(float seed, int steps) {
  float v = seed;
  for (int i = 0; i < steps; ++i) v = fmaf(v, 1.0000001f, 1e-9f);
  return v;
}

__device__ __forceinline__ unsigned long long stream_chunk(const v4i* __restrict__ w, size_t n16, size_t first,
                                                            size_t stride) {
  v4i acc = {0, 0, 0, 0};
  // eight 16-byte loads in flight per lane
  size_t i = first;
  for (; i + 7 * stride < n16; i += 8 * stride) {
    v4i t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = __builtin_nontemporal_load(w + i + u * stride);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc ^= t[u];
  }
  for (; i < n16; i += stride) acc ^= __builtin_nontemporal_load(w + i);
  return static_cast<unsigned long long>(static_cast<unsigned>(acc.x ^ acc.z)) |
         (static_cast<unsigned long long>(static_cast<unsigned>(acc.y ^ acc.w)) << 32);
}

// ---- form A: one launch per phase ------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void stream_kernel(const v4i* w, size_t n16, Ctl* ctl) {
  const size_t tid = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x;
  unsigned long long c = stream_chunk(w, n16, tid, static_cast<size_t>(gridDim.x) * kThreads);
  // (one atomic per wave keeps the checksum from dominating)
  for (int off = 32; off >= 1; off >>= 1) c ^= __shfl_xor(c, off, 64);
  if ((threadIdx.x & 63) == 0) atomicXor(&ctl->checksum, c);
}

__global__ __launch_bounds__(256) void latency_kernel(float* sink, int steps) {
  const float v = latency_chain(static_cast<float>(threadIdx.x), steps);
  if (v == 12345.678f) sink[blockIdx.x] = v;  // never true: keeps the chain alive
}

// ---- form B: one launch per layer --------------------------------------------------------------
__device__ __forceinline__ void grid_barrier(Ctl* ctl, unsigned target) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(&ctl->counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(&ctl->counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > kSpinLimit) { atomicAdd(&ctl->timeouts, 1u); break; }
    }
  }
  __syncthreads();
}

struct LayerArgs {
  const v4i* w[kStreams];
  size_t n16[kStreams];
  int latency_steps[5];  // norm, attention, quantise, norm, quantise
  int prefetch;          // 1: idle workgroups fill their LDS ring during the latency phases
};

__global__ __launch_bounds__(kThreads) void persistent_layer_kernel(LayerArgs a, Ctl* ctl, float* sink,
                                                                    unsigned barrier_base) {
  extern __shared__ __attribute__((aligned(16))) v4i ring[];
  const unsigned n = gridDim.x;
  const size_t tid = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x;
  const size_t stride = static_cast<size_t>(n) * kThreads;
  unsigned phase = 0;
  unsigned long long c = 0;
  constexpr int kRing16 = kRingBytes / 16;        // 16-byte slots per workgroup
  constexpr int kPer = kRing16 / kThreads;        // slots per thread (4)
  auto latency_phase = [&](int steps, int next_stream) {
    // the head of the next stream is REQUESTED first (this thread's first kPer grid-stride
    // elements), the chain runs while the requests are in flight, then they are parked in LDS
    const bool pf = a.prefetch && next_stream >= 0;
    v4i head[kPer];
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const size_t i = tid + static_cast<size_t>(u) * stride;
      head[u] = v4i{0, 0, 0, 0};
      if (pf && i < a.n16[next_stream]) head[u] = __builtin_nontemporal_load(a.w[next_stream] + i);
    }
    // the rows live on the first 16 workgroups (one wave each would do; the chain is per lane)
    if (blockIdx.x < 16) {
      const float v = latency_chain(static_cast<float>(threadIdx.x), steps);
      if (v == 12345.678f) sink[blockIdx.x] = v;
    }
    if (pf) {
#pragma unroll
      for (int u = 0; u < kPer; ++u) ring[u * kThreads + threadIdx.x] = head[u];
    }
    grid_barrier(ctl, (barrier_base + ++phase) * n);
  };
  auto stream_phase = [&](int s, bool prefetched) {
    size_t first = tid;
    if (a.prefetch && prefetched) {
      v4i acc = {0, 0, 0, 0};
#pragma unroll
      for (int u = 0; u < kPer; ++u) {
        const size_t i = tid + static_cast<size_t>(u) * stride;
        if (i < a.n16[s]) acc ^= ring[u * kThreads + threadIdx.x];
      }
      c ^= static_cast<unsigned long long>(static_cast<unsigned>(acc.x ^ acc.z)) |
           (static_cast<unsigned long long>(static_cast<unsigned>(acc.y ^ acc.w)) << 32);
      first = tid + static_cast<size_t>(kPer) * stride;
    }
    c ^= stream_chunk(a.w[s], a.n16[s], first, stride);
    grid_barrier(ctl, (barrier_base + ++phase) * n);
  };
  latency_phase(a.latency_steps[0], 0);  // norm            | prefetch QKV
  stream_phase(0, true);                 // QKV
  latency_phase(a.latency_steps[1], -1); // attention
  latency_phase(a.latency_steps[2], 1);  // quantise        | prefetch out-proj
  stream_phase(1, true);                 // out-proj
  latency_phase(a.latency_steps[3], 2);  // norm            | prefetch gated pair
  stream_phase(2, true);                 // gated pair
  latency_phase(a.latency_steps[4], 3);  // quantise        | prefetch down-proj
  stream_phase(3, true);                 // down-proj
  for (int off = 32; off >= 1; off >>= 1) c ^= __shfl_xor(c, off, 64);
  if ((threadIdx.x & 63) == 0) atomicXor(&ctl->checksum, c);
}

constexpr unsigned kPhasesPerLayer = 9;

int main() {
  size_t layer_bytes = 0;
  for (int s = 0; s < kStreams; ++s) layer_bytes += kStreamBytes[s];
  char* weights = nullptr;
  CK(hipMalloc(&weights, layer_bytes * kLayers));
  {
    // cheap non-constant fill (the checksum must depend on every byte)
    std::vector<uint32_t> host(1 << 22);
    for (size_t i = 0; i < host.size(); ++i) host[i] = static_cast<uint32_t>(i * 2654435761u);
    for (size_t off = 0; off < layer_bytes * kLayers; off += host.size() * 4) {
      const size_t nb = std::min(host.size() * 4, layer_bytes * kLayers - off);
      CK(hipMemcpy(weights + off, host.data(), nb, hipMemcpyHostToDevice));
    }
  }
  Ctl* ctl;
  float* sink;
  CK(hipMalloc(&ctl, sizeof(Ctl)));
  CK(hipMalloc(&sink, 4096));
  // chain lengths: ~3 us norm, ~6 us attention, ~1 us quantise at ~5 ns per dependent FMA
  const int lat[5] = {600, 1200, 200, 600, 200};
  hipStream_t stream;
  CK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int reps = 5;  // odd: see the checksum comparison

  auto layer_args = [&](int l, int prefetch) {
    LayerArgs a;
    size_t off = static_cast<size_t>(l) * layer_bytes;
    for (int s = 0; s < kStreams; ++s) {
      a.w[s] = reinterpret_cast<const v4i*>(weights + off);
      a.n16[s] = kStreamBytes[s] / 16;
      off += kStreamBytes[s];
    }
    for (int i = 0; i < 5; ++i) a.latency_steps[i] = lat[i];
    a.prefetch = prefetch;
    return a;
  };

  // ---- form A -----------------------------------------------------------------------------------
  unsigned long long sum_a = 0;
  {
    CK(hipMemset(ctl, 0, sizeof(Ctl)));
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
    for (int l = 0; l < kLayers; ++l) {
      const LayerArgs a = layer_args(l, 0);
      const int order[9] = {-1, 0, -2, -3, 1, -4, 2, -5, 3};  // negative: latency phase (1-based)
      for (int p = 0; p < 9; ++p) {
        if (order[p] < 0)
          hipLaunchKernelGGL(latency_kernel, dim3(16), dim3(256), 0, stream, sink, lat[-order[p] - 1]);
        else
          hipLaunchKernelGGL(stream_kernel, dim3(kWgs), dim3(kThreads), 0, stream, a.w[order[p]],
                             a.n16[order[p]], ctl);
      }
    }
    CK(hipStreamEndCapture(stream, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, stream));  // warm-up
    CK(hipStreamSynchronize(stream));
    CK(hipMemset(ctl, 0, sizeof(Ctl)));
    CK(hipEventRecord(e0, stream));
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, stream));
    CK(hipEventRecord(e1, stream));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    Ctl h;
    CK(hipMemcpy(&h, ctl, sizeof(h), hipMemcpyDeviceToHost));
    sum_a = h.checksum;
    printf("launch per phase (9 launches per layer, hipGraph): %8.2f us per layer  (%.2f TB/s of weights)\n",
           1e3 * ms / (reps * kLayers), layer_bytes / (1e9 * ms / (reps * kLayers)));
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
  }

  // ---- form B -----------------------------------------------------------------------------------
  for (int prefetch = 0; prefetch <= 1; ++prefetch) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(persistent_layer_kernel),
                           hipFuncAttributeMaxDynamicSharedMemorySize, kRingBytes));
    CK(hipMemset(ctl, 0, sizeof(Ctl)));
    unsigned base = 0;
    auto run_step = [&]() {
      for (int l = 0; l < kLayers; ++l) {
        hipLaunchKernelGGL(persistent_layer_kernel, dim3(kWgs), dim3(kThreads), kRingBytes, stream,
                           layer_args(l, prefetch), ctl, sink, base);
        base += kPhasesPerLayer;
      }
    };
    run_step();  // warm-up
    CK(hipStreamSynchronize(stream));
    CK(hipMemset(ctl, 0, sizeof(Ctl)));  // (counter and barrier targets restart together)
    base = 0;
    Ctl h;
    CK(hipEventRecord(e0, stream));
    for (int r = 0; r < reps; ++r) run_step();
    CK(hipEventRecord(e1, stream));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(&h, ctl, sizeof(h), hipMemcpyDeviceToHost));
    // (an odd number of identical passes xor-ed together leaves one pass: the same value in both forms)
    printf("one persistent launch per layer, prefetch %d:          %8.2f us per layer  (%.2f TB/s)  timeouts=%u  "
           "checksum %s\n",
           prefetch, 1e3 * ms / (reps * kLayers), layer_bytes / (1e9 * ms / (reps * kLayers)), h.timeouts,
           h.checksum == sum_a ? "equal to the launch form" : "DIFFERS");
  }
  return 0;
}
