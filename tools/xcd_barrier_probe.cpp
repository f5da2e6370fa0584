// Probe for a persistent decode kernel confined to ONE XCD of the MI355X (32 CUs sharing one 4 MB
// L2): how workgroups map to XCDs, what a grid barrier costs when producer and consumer share an L2
// (no L2 write-back / invalidate: stores complete in the L2, `buffer_inv sc0` drops the reader's
// L1), whether the hand-over is coherent, and what a "stage" (barrier + weight tile from MALL/HBM)
// costs next to the ~5 us of a dependent kernel launch in a hipGraph.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/xcdp tools/xcd_barrier_probe.cpp && /tmp/xcdp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}
__device__ __forceinline__ unsigned hw_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
  return v;
}

__global__ void map_kernel(unsigned* out) {
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = xcc_id();
    out[2 * blockIdx.x + 1] = hw_id();
  }
}

constexpr unsigned kSpinLimit = 1u << 18;

struct Ctl {
  unsigned counter;   // arrivals (monotonic)
  unsigned errors;    // stale / wrong hand-over values
  unsigned timeouts;  // bounded spins that ran out
  unsigned xcc_mismatch;
};

// mode bits: 0 = L2-scope barrier (stores done in L2 + L1 invalidate), 1 = agent-scope barrier
template <int SCOPE>
__device__ __forceinline__ void grid_barrier(Ctl* ctl, unsigned target) {
  if (SCOPE == 0) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(&ctl->counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      unsigned spins = 0;
      while (__hip_atomic_fetch_add(&ctl->counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
        if (++spins > kSpinLimit) { atomicAdd(&ctl->timeouts, 1u); break; }
      }
    }
    __syncthreads();
    asm volatile("buffer_inv sc0" ::: "memory");
  } else if (SCOPE == 2) {
    // as SCOPE 0, but the poll is a real read-modify-write executed in the L2
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(&ctl->counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      unsigned spins = 0, seen = 0;
      const unsigned zero = 0;
      unsigned* ptr = &ctl->counter;
      do {
        asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)"
                     : "=v"(seen) : "v"(ptr), "v"(zero) : "memory");
        if (++spins > kSpinLimit) { atomicAdd(&ctl->timeouts, 1u); break; }
      } while (seen < target);
    }
    __syncthreads();
    asm volatile("buffer_inv sc0" ::: "memory");
  } else {
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(&ctl->counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      unsigned spins = 0;
      while (__hip_atomic_load(&ctl->counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (++spins > kSpinLimit) { atomicAdd(&ctl->timeouts, 1u); break; }
      }
    }
    __syncthreads();
    __threadfence();
  }
}

// SCOPE as above. stride: participating workgroups are blockIdx.x % stride == 0 (stride 8 on a
// 256-workgroup grid = one XCD if the dispatcher deals workgroups round robin). payload: bytes of
// "weights" every participant streams per round (distinct per round, rotating through wbytes).
template <int SCOPE>
__global__ void __launch_bounds__(256)
rounds_kernel(Ctl* ctl, float* data /*[2][P][256]*/, const uint4* weights, size_t wbytes, int stride,
              int rounds, int payload, unsigned expect_xcc, float* sink) {
  if (blockIdx.x % stride != 0) return;
  const int p = blockIdx.x / stride;
  const int P = (gridDim.x + stride - 1) / stride;
  if (threadIdx.x == 0 && expect_xcc < 16 && xcc_id() != expect_xcc) atomicAdd(&ctl->xcc_mismatch, 1u);
  float acc = 0.f;
  const size_t wvec = wbytes / 16;
  for (int r = 1; r <= rounds; ++r) {
    // "stage body": stream this workgroup's weight tile, then publish one activation line
    if (payload) {
      const int nvec = payload / 16 / 256;  // per thread
      size_t base = (size_t(r) * P + p) * (payload / 16) % (wvec - payload / 16);
      uint4 v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (i < nvec) v[i] = weights[base + size_t(i) * 256 + threadIdx.x];
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (i < nvec) acc += __uint_as_float((v[i].x ^ v[i].y ^ v[i].z ^ v[i].w) & 0x3f800000u);
    }
    float* mine = data + (size_t(r & 1) * P + p) * 256;
    mine[threadIdx.x] = float(r) + acc * 0.f;
    grid_barrier<SCOPE>(ctl, unsigned(r) * P);
    if (__hip_atomic_load(&ctl->timeouts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;  // never hang the box
    // consumer side: read two other workgroups' lines (last seen two rounds ago in this L1)
    const float a = data[(size_t(r & 1) * P + (p + 1) % P) * 256 + threadIdx.x];
    const float b = data[(size_t(r & 1) * P + (p + 7) % P) * 256 + threadIdx.x];
    if (a != float(r) || b != float(r)) atomicAdd(&ctl->errors, 1u);
    acc += a - b;
  }
  if (acc == 12345.f) sink[0] = acc;
}

int main() {
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  // ---- 1. workgroup -> XCD map
  for (int grid : {256, 512}) {
    unsigned* d; CK(hipMalloc(&d, 8 * grid));
    hipLaunchKernelGGL(map_kernel, dim3(grid), dim3(256), 0, s, d);
    CK(hipStreamSynchronize(s));
    std::vector<unsigned> h(2 * grid);
    CK(hipMemcpy(h.data(), d, 8 * grid, hipMemcpyDeviceToHost));
    int rr = 0;
    for (int i = 0; i < grid; ++i) rr += (h[2 * i] == unsigned(i % 8));
    printf("map grid=%d: xcc == blockIdx %% 8 for %d / %d workgroups; first 16 xcc:", grid, rr, grid);
    for (int i = 0; i < 16; ++i) printf(" %u", h[2 * i]);
    printf("\n");
    // distinct (xcc, cu) among blockIdx % 8 == 0
    std::vector<unsigned> seen;
    for (int i = 0; i < grid; i += 8) {
      unsigned hw = h[2 * i + 1];
      unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
      unsigned key = (se << 8) | (sh << 4) | cu;
      bool f = false; for (unsigned k : seen) f |= (k == key);
      if (!f) seen.push_back(key);
    }
    printf("   blockIdx %% 8 == 0: %zu distinct (se, sh, cu) among %d workgroups\n", seen.size(), grid / 8);
    CK(hipFree(d));
  }
  // ---- 2. barrier cost / coherence
  const size_t wbytes = size_t(96) << 20;
  uint4* w; CK(hipMalloc(&w, wbytes)); CK(hipMemset(w, 1, wbytes));
  float* data; CK(hipMalloc(&data, sizeof(float) * 2 * 256 * 256)); CK(hipMemset(data, 0, sizeof(float) * 2 * 256 * 256));
  float* sink; CK(hipMalloc(&sink, 4));
  Ctl* ctl; CK(hipMalloc(&ctl, sizeof(Ctl)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct Case { const char* name; int scope, grid, stride, payload; unsigned xcc; };
  const Case cases[] = {
      {"one XCD, 32 WG, L2-scope barrier          ", 0, 256, 8, 0, 0},
      {"one XCD, 32 WG, L2-scope barrier, RMW poll", 2, 256, 8, 0, 0},
      {"one XCD, 32 WG, agent-scope barrier       ", 1, 256, 8, 0, 0},
      {"8 XCDs, 32 WG, agent-scope barrier        ", 1, 32, 1, 0, 99},
      {"8 XCDs, 256 WG, agent-scope barrier       ", 1, 256, 1, 0, 99},
      {"one XCD, 64 WG, L2-scope barrier          ", 0, 512, 8, 0, 0},
      {"one XCD, 32 WG, L2 barrier + 16 KB/WG     ", 0, 256, 8, 16384, 0},
      {"one XCD, 32 WG, L2 barrier + 64 KB/WG     ", 0, 256, 8, 65536, 0},
      {"one XCD, 64 WG, L2 barrier + 16 KB/WG     ", 0, 512, 8, 16384, 0},
      {"one XCD, 64 WG, L2 barrier + 64 KB/WG     ", 0, 512, 8, 65536, 0},
      {"8 XCDs, 256 WG, agent barrier + 16 KB/WG  ", 1, 256, 1, 16384, 99},
  };
  const int rounds = 4000;
  for (const Case& c : cases) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipMemsetAsync(ctl, 0, sizeof(Ctl), s));
      CK(hipMemsetAsync(data, 0, sizeof(float) * 2 * 256 * 256, s));
      CK(hipEventRecord(e0, s));
      if (c.scope == 2)
        hipLaunchKernelGGL(rounds_kernel<2>, dim3(c.grid), dim3(256), 0, s, ctl, data, w, wbytes, c.stride, rounds, c.payload, c.xcc, sink);
      else if (c.scope == 0)
        hipLaunchKernelGGL(rounds_kernel<0>, dim3(c.grid), dim3(256), 0, s, ctl, data, w, wbytes, c.stride, rounds, c.payload, c.xcc, sink);
      else
        hipLaunchKernelGGL(rounds_kernel<1>, dim3(c.grid), dim3(256), 0, s, ctl, data, w, wbytes, c.stride, rounds, c.payload, c.xcc, sink);
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      Ctl h; CK(hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
      if (rep == 1)
        printf("%s: %.3f us/round  errors=%u timeouts=%u xcc_mismatch=%u\n", c.name, ms * 1e3 / rounds,
               h.errors, h.timeouts, h.xcc_mismatch);
    }
  }
  // ---- 3. the same payload as dependent launches in a hipGraph (the status quo)
  return 0;
}
