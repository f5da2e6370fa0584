// Where do the ~4.3 us of a [32, 512] LayerNorm + Quantize launch go? The body of norm_vec_kernel
// (ctranslate2_amd/csrc/rowops.hip) with s_memtime stamps between its phases; 32 workgroups of
// 128 threads like the real launch. Prints core-clock cycles per phase for workgroup 0 / wave 0.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -x hip tools/norm_probe.cpp \
//         -Ictranslate2_amd/csrc -Iinclude -o /tmp/np
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "device_utils.h"
#include "row_norm.h"
using namespace ct2x;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s\n", hipGetErrorString(e)); return 1; } } while (0)

constexpr int kChunks = 1;
__global__ void probe(const float* __restrict__ x, const float* __restrict__ gamma,
                      const float* __restrict__ beta, int64_t depth, float eps, int quant_mode,
                      int8_t* __restrict__ q, float* __restrict__ scales, long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) float row_buf[];
  __shared__ float scratch[32];
  long long t[8];
  t[0] = clock64();
  const int64_t row = blockIdx.x;
  const int n4 = static_cast<int>(depth / 4);
  const float4* xr = reinterpret_cast<const float4*>(x + row * depth);
  const float4* g4 = reinterpret_cast<const float4*>(gamma);
  const float4* b4 = reinterpret_cast<const float4*>(beta);
  float4 v[kChunks], gv[kChunks], bv[kChunks];
#pragma unroll
  for (int u = 0; u < kChunks; ++u) {
    const int i = threadIdx.x + u * blockDim.x;
    v[u] = gv[u] = bv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4) { v[u] = xr[i]; gv[u] = g4[i]; bv[u] = b4[i]; }
  }
#pragma unroll
  for (int u = 0; u < kChunks; ++u) {
    const int i = threadIdx.x + u * blockDim.x;
    if (i < n4) reinterpret_cast<float4*>(row_buf)[i] = v[u];
  }
  t[1] = clock64();   // loads landed (the LDS write needs them)
  __syncthreads();
  t[2] = clock64();
  float sum, sumsq;
  ordered_row_stats(row_buf, depth, sum, sumsq);
  asm volatile("" :: "v"(sum), "v"(sumsq));
  t[3] = clock64();   // chains + tree
  const NormStats st = norm_stats(sum, sumsq, depth, eps, true);
  float amax = 0.f;
#pragma unroll
  for (int u = 0; u < kChunks; ++u) {
    const int c = 4 * (threadIdx.x + u * blockDim.x);
    v[u].x = norm_apply(v[u].x, gv[u].x, bv[u].x, st, true, false, c);
    v[u].y = norm_apply(v[u].y, gv[u].y, bv[u].y, st, true, false, c + 1);
    v[u].z = norm_apply(v[u].z, gv[u].z, bv[u].z, st, true, false, c + 2);
    v[u].w = norm_apply(v[u].w, gv[u].w, bv[u].w, st, true, false, c + 3);
    amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[u].x), fabsf(v[u].y)), fmaxf(fabsf(v[u].z), fabsf(v[u].w))));
  }
  asm volatile("" :: "v"(amax));
  t[4] = clock64();   // stats (rsqrt, div) + apply
  amax = block_max(amax, scratch);
  t[5] = clock64();   // block max (two barriers)
  const float scale = (amax != 0.f) ? 127.f / amax : 1.f;
  if (threadIdx.x == 0) scales[row] = scale;
  char4* qr = reinterpret_cast<char4*>(q + row * depth);
#pragma unroll
  for (int u = 0; u < kChunks; ++u) {
    const int i = threadIdx.x + u * blockDim.x;
    if (i < n4) {
      char4 o;
      o.x = to_s8(v[u].x, scale, quant_mode); o.y = to_s8(v[u].y, scale, quant_mode);
      o.z = to_s8(v[u].z, scale, quant_mode); o.w = to_s8(v[u].w, scale, quant_mode);
      qr[i] = o;
    }
  }
  t[6] = clock64();
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (int i = 0; i < 7; ++i) stamps[i] = t[i];
}

int main() {
  const int rows = 32, depth = 512;
  float *x, *g, *b, *sc; int8_t* q; long long* st;
  CK(hipMalloc(&x, rows * depth * 4)); CK(hipMalloc(&g, depth * 4)); CK(hipMalloc(&b, depth * 4));
  CK(hipMalloc(&sc, rows * 4)); CK(hipMalloc(&q, rows * depth)); CK(hipMalloc(&st, 64));
  std::vector<float> h(rows * depth);
  for (size_t i = 0; i < h.size(); ++i) h[i] = ((i * 2654435761u) % 2001) / 1000.f - 1.f;
  CK(hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(g, h.data(), depth * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(b, h.data() + depth, depth * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(probe, dim3(rows), dim3(128), depth * 4, 0, x, g, b, (int64_t)depth, 1e-5f, 3, q, sc, st);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    long long hs[7]; CK(hipMemcpy(hs, st, sizeof hs, hipMemcpyDeviceToHost));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("rep %d: event bracket %.2f us | cycles: loads+park %lld  barrier %lld  chains+tree %lld  stats+apply %lld  block_max %lld  quantize+store %lld  | total %lld\n",
           rep, ms * 1e3, hs[1] - hs[0], hs[2] - hs[1], hs[3] - hs[2], hs[4] - hs[3], hs[5] - hs[4], hs[6] - hs[5], hs[6] - hs[0]);
  }
  return 0;
}
