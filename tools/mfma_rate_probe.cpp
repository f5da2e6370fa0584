// Sustained rate of the two int8 matrix instructions of gfx950 - v_mfma_i32_16x16x64_i8 and
// v_mfma_i32_32x32x32_i8 - in a register-only loop (no LDS, no memory), by operand data:
// zero, uniform int8, a quantised-Gaussian fill (|q| mostly < 40: what per-row absmax
// quantisation of activations / N(0, sigma) weights produces) and +-3. The chip clocks to its
// power budget (MI355X_MICROARCH.md, DVFS give-back), so the rate a GEMM can reach depends on
// the instruction AND on the bits it toggles; this probe prices both for the roofline denominator
// of bench.py's prefill_gemm leg.
// Build: hipcc --offload-arch=gfx950 -O3 -o build/tools/mfma_rate_probe tools/mfma_rate_probe.cpp
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define CHECK(x)                                                        \
  do {                                                                  \
    hipError_t e_ = (x);                                                \
    if (e_ != hipSuccess) {                                             \
      printf("%s: %s\n", #x, hipGetErrorString(e_));                    \
      exit(1);                                                          \
    }                                                                   \
  } while (0)

// FORM 0: 16x16x64, 32 accumulators of 4 (8 A x 4 B fragments, as the 128 x 64 wave tile)
// FORM 1: 32x32x32,  8 accumulators of 16 (2 A x 4 B fragments, x 2 K-steps)
template <int FORM>
__global__ __launch_bounds__(512) void probe(const v4i* __restrict__ src, int steps, int* out,
                                             uint64_t* cycles) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  v4i fa[8], fb[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    fa[i] = src[(blockIdx.x * 16 + i) * 64 + lane];
    fb[i] = src[(blockIdx.x * 16 + 8 + i) * 64 + lane];
  }
  int sum = 0;
  const uint64_t t0 = __builtin_readcyclecounter();
  if (FORM == 0) {
    v4i acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = v4i{0, 0, 0, 0};
    for (int t = 0; t < steps; ++t) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  } else {
    v16i acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;
    for (int t = 0; t < steps; ++t) {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb[s * 2 + j], fa[s * 4 + i], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) sum += acc[i][j][e];
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  if (lane == 0) cycles[blockIdx.x * 8 + wave] = t1 - t0;
  if (sum == 0x7fffffff) out[threadIdx.x] = sum;
}

template <int FORM>
void run(const v4i* src, int steps, int threads, int* out, uint64_t* cyc, const char* fill) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int it = 0; it < 3; ++it) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe<FORM>), dim3(256), dim3(threads), 0, 0, src, steps, out, cyc);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const int waves = threads / 64;
  uint64_t h[2048];
  CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
  double sum = 0;
  for (int b = 0; b < 256; ++b)
    for (int w = 0; w < waves; ++w) sum += static_cast<double>(h[b * 8 + w]);
  const double cyc_per_step = sum / (256 * waves) / steps;
  const double ops_per_step = FORM == 0 ? 32.0 * 16 * 16 * 64 * 2 : 16.0 * 32 * 32 * 32 * 2;
  const double tops = 256.0 * waves * steps * ops_per_step / (best * 1e-3) / 1e12;
  printf("%-10s %s %d waves/SIMD  %7.1f cyc/step (%5.2f per instruction)  %7.3f ms  %7.1f TOP/s  (%.2f GHz)\n",
         fill, FORM == 0 ? "16x16x64" : "32x32x32", waves / 4, cyc_per_step,
         cyc_per_step / (FORM == 0 ? 32 : 16), best, tops, cyc_per_step * steps / (best * 1e-3) / 1e9);
}

int main(int argc, char** argv) {
  const int steps = argc > 1 ? atoi(argv[1]) : 4096;
  const size_t n = 256 * 16 * 64 * 16;  // bytes
  int8_t* h = static_cast<int8_t*>(malloc(n));
  v4i* src;
  int* out;
  uint64_t* cyc;
  CHECK(hipMalloc(&src, n));
  CHECK(hipMalloc(&out, 4096));
  CHECK(hipMalloc(&cyc, 2048 * 8));
  const char* fills[] = {"zero", "uniform", "gauss32", "pm3"};
  for (int f = 0; f < 4; ++f) {
    uint32_t s = 12345;
    for (size_t i = 0; i < n; ++i) {
      s = s * 1664525u + 1013904223u;
      const uint32_t r = s >> 8;
      int v = 0;
      if (f == 1) v = static_cast<int>(r % 255u) - 127;
      if (f == 2) {  // sum of 4 uniforms ~ Gaussian, sigma ~ 32, clipped to +-127
        const double u = ((r & 255) + ((r >> 8) & 255) + ((r >> 16) & 255)) / 255.0 - 1.5;
        v = static_cast<int>(lrint(u * 64.0));
        if (v > 127) v = 127;
        if (v < -127) v = -127;
      }
      if (f == 3) v = static_cast<int>(r % 7u) - 3;
      h[i] = static_cast<int8_t>(v);
    }
    CHECK(hipMemcpy(src, h, n, hipMemcpyHostToDevice));
    for (int threads : {256, 512}) {
      run<0>(src, steps, threads, out, cyc, fills[f]);
      run<1>(src, steps, threads, out, cyc, fills[f]);
    }
  }
  return 0;
}
