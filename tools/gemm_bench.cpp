// Throughput of the int8 projection GEMM (ct2x_gemm_s8_dequant / ct2x_gemm_s8s8s32) at MFMA-bound
// shapes (encoder / cross-KV projection / decoder-only prefill) against the dense int8 MFMA peak,
// with an on-device exact check of the int32 result against a one-thread-per-output kernel.
// Build on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -x hip tools/gemm_bench.cpp -Iinclude \
//         -Lctranslate2_amd -lct2_mi355 -Wl,-rpath,$PWD/ctranslate2_amd -o /tmp/gb
// Usage: /tmp/gb [M N K]...   (default: the Transformer-base encoder and Llama-3-8B prefill shapes)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ct2_mi355.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
#define CX(x) do { if ((x) != 0) { printf("ERR %s: %s\n", #x, ct2x_last_error()); exit(1);} } while (0)

__global__ void fill_s8(int8_t* p, size_t n, unsigned seed) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned h = (unsigned)(i * 2654435761u) ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    p[i] = (int8_t)((int)(h % 255u) - 127);
  }
}
__global__ void fill_f32(float* p, size_t n, float v) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
// Exact check of sampled outputs: one thread per sample.
__global__ void check_samples(const int8_t* A, const int8_t* B, const int32_t* C, int M, int N, int K,
                              int samples, unsigned long long* mismatches) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= samples) return;
  unsigned h = s * 2654435761u + 12345u;
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
  const int row = (s < 4096) ? ((s & 1) ? M - 1 - (s >> 1) % M : (s >> 1) % M) : h % M;
  const int col = (h >> 7) % N;
  int acc = 0;
  for (int k = 0; k < K; ++k) acc += (int)A[(size_t)row * K + k] * (int)B[(size_t)col * K + k];
  if (acc != C[(size_t)row * N + col]) atomicAdd(mismatches, 1ULL);
}

int main(int argc, char** argv) {
  struct Shape { int m, n, k; };
  std::vector<Shape> shapes;
  for (int i = 1; i + 2 < argc; i += 3) shapes.push_back({atoi(argv[i]), atoi(argv[i + 1]), atoi(argv[i + 2])});
  if (shapes.empty())
    shapes = {{1024, 512, 512}, {1024, 1536, 512}, {1024, 2048, 512}, {1024, 512, 2048},
              {4096, 2048, 512}, {2048, 4096, 4096}, {2048, 6144, 4096}, {2048, 14336, 4096},
              {2048, 4096, 14336}, {4096, 4096, 4096}, {8192, 8192, 8192}, {1000, 1111, 384}};
  CX(ct2x_set_device(0));
  hipStream_t s = (hipStream_t)ct2x_thread_stream();
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  unsigned long long* d_mis; CK(hipMalloc(&d_mis, 8));
  for (auto sh : shapes) {
    const size_t M = sh.m, N = sh.n, K = sh.k;
    int8_t *a, *b; int32_t* c; float *y, *as, *ws, *bias;
    CK(hipMalloc(&a, M * K)); CK(hipMalloc(&b, N * K)); CK(hipMalloc(&c, M * N * 4));
    CK(hipMalloc(&y, M * N * 4)); CK(hipMalloc(&as, M * 4)); CK(hipMalloc(&ws, N * 4)); CK(hipMalloc(&bias, N * 4));
    hipLaunchKernelGGL(fill_s8, dim3(2048), dim3(256), 0, s, a, M * K, 1u);
    hipLaunchKernelGGL(fill_s8, dim3(2048), dim3(256), 0, s, b, N * K, 2u);
    hipLaunchKernelGGL(fill_f32, dim3((M + 255) / 256), dim3(256), 0, s, as, M, 40.f);
    hipLaunchKernelGGL(fill_f32, dim3((N + 255) / 256), dim3(256), 0, s, ws, N, 300.f);
    hipLaunchKernelGGL(fill_f32, dim3((N + 255) / 256), dim3(256), 0, s, bias, N, 0.01f);
    CK(hipMemsetAsync(d_mis, 0, 8, s));
    CX(ct2x_gemm_s8s8s32(a, b, M, N, K, c, s));
    const int samples = 16384;
    hipLaunchKernelGGL(check_samples, dim3(samples / 256), dim3(256), 0, s, a, b, c, sh.m, sh.n, sh.k, samples, d_mis);
    unsigned long long mis = 0;
    CK(hipMemcpyAsync(&mis, d_mis, 8, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    const double ops = 2.0 * M * N * K;
    const int reps = ops > 1e11 ? 20 : 100;
    // Decode shapes (M <= 128) stream weights that are never cache resident in the real loop
    // (7.5 GB per step for the Llama shape): rotate through > 512 MB of weight copies so the
    // 256 MB Infinity Cache cannot flatter the number.
    int copies = 1;
    int8_t* brot = nullptr;
    if (M <= 128 && !getenv("GB_NO_ROTATE")) {
      copies = (int)((600ull << 20) / (N * K)) + 1;
      CK(hipMalloc(&brot, (size_t)copies * N * K));
      hipLaunchKernelGGL(fill_s8, dim3(4096), dim3(256), 0, s, brot, (size_t)copies * N * K, 3u);
      CK(hipStreamSynchronize(s));
    }
    int rot = 0;
    for (int mode = 0; mode < 2; ++mode) {
      auto launch = [&] {
        const int8_t* bb = brot ? brot + (size_t)(rot++ % copies) * N * K : b;
        if (mode == 0) CX(ct2x_gemm_s8s8s32(a, bb, M, N, K, c, s));
        else CX(ct2x_gemm_s8_dequant(a, as, bb, ws, bias, nullptr, M, N, K, 0, y, 0, s));
      };
      for (int i = 0; i < 3; ++i) launch();
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < reps; ++i) launch();
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / reps;
      const double bytes = (double)M * K + (double)N * K + 4.0 * M * N;
      printf("%-22s M=%5d N=%5d K=%5d  %9.2f us  %8.1f TOPS (%5.1f%% of 5000)  %7.1f GB/s  mismatches=%llu/%d\n",
             mode == 0 ? "gemm_s8s8s32" : "gemm_s8_dequant+relu", sh.m, sh.n, sh.k, us, ops / us / 1e6,
             ops / us / 1e6 / 50.0, bytes / us / 1e3, mis, samples);
    }
    if (brot) CK(hipFree(brot));
    CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(c)); CK(hipFree(y)); CK(hipFree(as)); CK(hipFree(ws)); CK(hipFree(bias));
  }
  return 0;
}
