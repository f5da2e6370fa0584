// Throughput of the int8 projection GEMM (ct2x_gemm_s8_dequant / ct2x_gemm_s8s8s32) at MFMA-bound
// shapes (encoder / cross-KV projection / decoder-only prefill) against the dense int8 MFMA peak,
// with an on-device exact check of the int32 result against a one-thread-per-output kernel.
// Build on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -x hip tools/gemm_bench.cpp -Iinclude \
//         -Lctranslate2_amd -lct2_mi355 -Wl,-rpath,$PWD/ctranslate2_amd -o /tmp/gb
// Usage: /tmp/gb [--fill uniform|gauss] [--act N] [M N K]...
//   (default: the Transformer-base encoder and Llama-3-8B prefill shapes; --fill gauss = operands
//   as per-row absmax quantisation of Gaussian data produces them, |q| mostly < 40; --act = the
//   ops::ActivationType of the fused form, default 0 = ReLU)
// Every int32 result is compared in full with a one-thread-per-output kernel when M N K <= 7e10
// (sampled otherwise), and the fused fp32 result bit for bit with the reference's formula applied
// to that int32 result (ReLU / no activation only).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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
// Quantised-Gaussian fill: sum of three uniforms, sigma ~ 32, clipped to +-127.
__global__ void fill_s8_gauss(int8_t* p, size_t n, unsigned seed) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned h = (unsigned)(i * 2654435761u) ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    const float u = ((h & 255) + ((h >> 8) & 255) + ((h >> 16) & 255)) / 255.f - 1.5f;
    int v = (int)rintf(u * 64.f);
    p[i] = (int8_t)(v > 127 ? 127 : v < -127 ? -127 : v);
  }
}
__global__ void full_check(const int8_t* A, const int8_t* B, const int32_t* C, int M, int N, int K,
                           unsigned long long* mismatches) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x, row = blockIdx.y;
  if (col >= N) return;
  int acc = 0;
  const int* a = (const int*)(A + (size_t)row * K);
  const int* b = (const int*)(B + (size_t)col * K);
  for (int k = 0; k < K / 4; ++k) {
    const int x = a[k], y = b[k];
    acc += (int)(int8_t)x * (int)(int8_t)y + (int)(int8_t)(x >> 8) * (int)(int8_t)(y >> 8) +
           (int)(int8_t)(x >> 16) * (int)(int8_t)(y >> 16) + (x >> 24) * (y >> 24);
  }
  if (acc != C[(size_t)row * N + col]) atomicAdd(mismatches, 1ULL);
}
// y = act(c * (1 / a_scale) / w_scale + bias), the reference's CPU operand order.
__global__ void check_dequant(const int32_t* C, const float* y, const float* as, const float* ws,
                              const float* bias, size_t M, size_t N, int relu,
                              unsigned long long* mismatches) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= M * N) return;
  const size_t row = i / N, col = i % N;
  float v = (float)C[i] * (1.f / as[row]);
  v = v / ws[col];
  v = v + bias[col];
  if (relu) v = fmaxf(v, 0.f);
  if (__float_as_uint(v) != __float_as_uint(y[i])) atomicAdd(mismatches, 1ULL);
}
// Per-row / per-column operands that differ from index to index (a swapped row / column map in
// an epilogue must not pass the check).
__global__ void fill_f32(float* p, size_t n, float v, float step, int period) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) p[i] = v + step * (float)((i * 7 + i / period) % period);
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
  bool gauss = false;
  int act = 0;
  int argi = 1;
  while (argi + 1 < argc && argv[argi][0] == '-') {
    if (!strcmp(argv[argi], "--fill")) gauss = !strcmp(argv[argi + 1], "gauss");
    if (!strcmp(argv[argi], "--act")) act = atoi(argv[argi + 1]);
    argi += 2;
  }
  for (int i = argi; i + 2 < argc; i += 3) shapes.push_back({atoi(argv[i]), atoi(argv[i + 1]), atoi(argv[i + 2])});
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
    if (gauss) {
      hipLaunchKernelGGL(fill_s8_gauss, dim3(2048), dim3(256), 0, s, a, M * K, 1u);
      hipLaunchKernelGGL(fill_s8_gauss, dim3(2048), dim3(256), 0, s, b, N * K, 2u);
    } else {
      hipLaunchKernelGGL(fill_s8, dim3(2048), dim3(256), 0, s, a, M * K, 1u);
      hipLaunchKernelGGL(fill_s8, dim3(2048), dim3(256), 0, s, b, N * K, 2u);
    }
    hipLaunchKernelGGL(fill_f32, dim3((M + 255) / 256), dim3(256), 0, s, as, M, 20.7f, 0.37f, 97);
    hipLaunchKernelGGL(fill_f32, dim3((N + 255) / 256), dim3(256), 0, s, ws, N, 150.3f, 1.7f, 89);
    hipLaunchKernelGGL(fill_f32, dim3((N + 255) / 256), dim3(256), 0, s, bias, N, -0.06f, 0.01f, 13);
    CK(hipMemsetAsync(d_mis, 0, 8, s));
    CX(ct2x_gemm_s8s8s32(a, b, M, N, K, c, s));
    int samples = 16384;
    const bool full = (double)M * N * K <= 7e10 && K % 4 == 0;
    if (full) {
      hipLaunchKernelGGL(full_check, dim3((N + 255) / 256, M), dim3(256), 0, s, a, b, c, sh.m, sh.n, sh.k, d_mis);
      samples = -1;
    } else {
      hipLaunchKernelGGL(check_samples, dim3(samples / 256), dim3(256), 0, s, a, b, c, sh.m, sh.n, sh.k, samples, d_mis);
    }
    unsigned long long mis = 0;
    CK(hipMemcpyAsync(&mis, d_mis, 8, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    // fused form against the formula on the (checked) int32 result
    unsigned long long mis_f = 0;
    if (act <= 0) {
      CK(hipMemsetAsync(d_mis, 0, 8, s));
      CX(ct2x_gemm_s8_dequant(a, as, b, ws, bias, nullptr, M, N, K, act, y, 0, s));
      hipLaunchKernelGGL(check_dequant, dim3((M * N + 255) / 256), dim3(256), 0, s, c, y, as, ws, bias, M, N, act == 0 ? 1 : 0, d_mis);
      CK(hipMemcpyAsync(&mis_f, d_mis, 8, hipMemcpyDeviceToHost, s));
      CK(hipStreamSynchronize(s));
    }
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
        else CX(ct2x_gemm_s8_dequant(a, as, bb, ws, bias, nullptr, M, N, K, act, y, 0, s));
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
             mode == 0 ? "gemm_s8s8s32" : "gemm_s8_dequant+act", sh.m, sh.n, sh.k, us, ops / us / 1e6,
             ops / us / 1e6 / 50.0, bytes / us / 1e3, mode == 0 ? mis : mis_f, samples);
    }
    if (brot) CK(hipFree(brot));
    CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(c)); CK(hipFree(y)); CK(hipFree(as)); CK(hipFree(ws)); CK(hipFree(bias));
  }
  return 0;
}
