// GPU-side cost of each hot-path kernel at the Transformer-base decode shapes, measured the way
// the decode loop experiences it: N launches enqueued behind a long "plug" kernel so the host is
// out of the picture, timed with HIP events on the library's stream. Build on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -x hip tools/kernel_bench.cpp -Iinclude \
//         -Lctranslate2_amd -lct2_mi355 -Wl,-rpath,$PWD/ctranslate2_amd -o /tmp/kb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>
#include "ct2_mi355.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
#define CX(x) do { if ((x) != 0) { printf("ERR %s: %s\n", #x, ct2x_last_error()); exit(1);} } while (0)

__global__ void plug(float* y, int iters) {
  float v = threadIdx.x;
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  y[blockIdx.x * blockDim.x + threadIdx.x] = v;
}

template <typename T> T* dalloc(size_t n, bool random = true) {
  T* p; CK(hipMalloc(&p, n * sizeof(T)));
  std::vector<T> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = random ? (T)((rand() % 255) - 127) : (T)0;
  CK(hipMemcpy(p, h.data(), n * sizeof(T), hipMemcpyHostToDevice));
  return p;
}
float* dallocf(size_t n, float scale = 1.f) {
  float* p; CK(hipMalloc(&p, n * sizeof(float)));
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = scale * ((rand() % 2001) - 1000) / 1000.f;
  CK(hipMemcpy(p, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
  return p;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 32;
  const int N_LAUNCH = 400;
  CX(ct2x_set_device(0));
  hipStream_t s = (hipStream_t)ct2x_thread_stream();
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float* plug_buf; CK(hipMalloc(&plug_buf, 4096 * 256 * 4));
  const int d = 512, ffn = 2048, V = 32000, H = 8, dh = 64, T = 64;
  float* x = dallocf((size_t)M * ffn);
  float* y = dallocf((size_t)M * V);
  float* res = dallocf((size_t)M * ffn);
  float* gamma = dallocf(ffn), *beta = dallocf(ffn), *bias = dallocf(V, 0.01f);
  float* wscale = dallocf(V, 50.f); 
  { std::vector<float> h(V, 300.f); CK(hipMemcpy(wscale, h.data(), V * 4, hipMemcpyHostToDevice)); }
  int8_t* q = dalloc<int8_t>((size_t)M * ffn);
  float* qs = dallocf(M); { std::vector<float> h(M, 40.f); CK(hipMemcpy(qs, h.data(), M * 4, hipMemcpyHostToDevice)); }
  int8_t* w = dalloc<int8_t>((size_t)V * d);
  int8_t* w2 = dalloc<int8_t>((size_t)ffn * ffn);
  float* kc = dallocf((size_t)M * H * T * dh), *vc = dallocf((size_t)M * H * T * dh);
  float* qv = dallocf((size_t)M * H * dh), *ctx = dallocf((size_t)M * H * dh);
  float* topv = dallocf(M * 8); int32_t* topi = dalloc<int32_t>(M * 8, false);

  auto run = [&](const char* name, double bytes, std::function<void()> launch) {
    for (int i = 0; i < 5; ++i) launch();
    CK(hipStreamSynchronize(s));
    hipLaunchKernelGGL(plug, dim3(4096), dim3(256), 0, s, plug_buf, 100000);
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < N_LAUNCH; ++i) launch();
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / N_LAUNCH;
    printf("%-44s %7.2f us/launch  %8.1f GB/s\n", name, us, bytes > 0 ? bytes / us / 1e3 : 0.0);
  };

  run("quantize_s8 [M,512]", M * 512.0 * 5, [&] { CX(ct2x_quantize_s8(x, 0, M, d, 3, q, qs, s)); });
  run("quantize_s8 [M,2048]", M * 2048.0 * 5, [&] { CX(ct2x_quantize_s8(x, 0, M, ffn, 3, q, qs, s)); });
  run("norm_quantize (LN) [M,512]", M * 512.0 * 5, [&] { CX(ct2x_norm_quantize_s8(x, gamma, beta, M, d, 1e-5f, 0, 3, nullptr, q, qs, s)); });
  run("layer_norm [M,512]", M * 512.0 * 8, [&] { CX(ct2x_layer_norm(x, gamma, beta, M, d, 1e-5f, y, 0, s)); });
  struct { int n, k; } shapes[] = {{512, 512}, {1536, 512}, {2048, 512}, {512, 2048}, {32000, 512}};
  for (auto sh : shapes) {
    char name[128]; snprintf(name, sizeof name, "gemm_s8_dequant [M=%d,N=%d,K=%d]", M, sh.n, sh.k);
    const double bytes = (double)M * sh.k + (double)sh.n * sh.k + 4.0 * sh.n + 4.0 * M + 4.0 * M * sh.n;
    run(name, bytes, [&] { CX(ct2x_gemm_s8_dequant(q, qs, sh.k == 2048 ? w2 : w, wscale, bias, nullptr, M, sh.n, sh.k, -1, y, 0, s)); });
  }
  {
    // Same GEMM, but cycling through 152 distinct 256 KB weight matrices (38 MB, the per-step
    // weight footprint of Transformer-base): what the decode loop sees, where a weight byte is
    // never L2-resident from the previous step.
    int8_t* wbig = dalloc<int8_t>((size_t)152 * 512 * 512);
    int idx = 0;
    run("gemm_s8_dequant [M,512,512] rotating 38MB weights", (double)M * 512 + 512.0 * 512 + 4.0 * M * 512,
        [&] { CX(ct2x_gemm_s8_dequant(q, qs, wbig + (size_t)(idx++ % 152) * 512 * 512, wscale, bias, nullptr, M, 512, 512, -1, y, 0, s)); });
    int8_t* wbig2 = dalloc<int8_t>((size_t)40 * 512 * 2048);
    idx = 0;
    run("gemm_s8_dequant [M,2048,512] rotating 40MB weights", (double)M * 512 + 2048.0 * 512 + 4.0 * M * 2048,
        [&] { CX(ct2x_gemm_s8_dequant(q, qs, wbig2 + (size_t)(idx++ % 40) * 512 * 2048, wscale, bias, nullptr, M, 2048, 512, -1, y, 0, s)); });
  }
  for (int n : {512, 1536, 2048}) {
    char name[128]; snprintf(name, sizeof name, "norm_gemm_s8_dequant (LN prologue) [M,%d,512]", n);
    run(name, M * 512.0 * 4 + n * 512.0 + 4.0 * M * n,
        [&] { CX(ct2x_norm_gemm_s8_dequant(x, gamma, beta, 1e-5f, 0, 3, w, wscale, bias, nullptr, M, n, 512, -1, y, s)); });
  }
  run("gemm_s8_dequant+residual [M,512,512]", 0, [&] { CX(ct2x_gemm_s8_dequant(q, qs, w, wscale, bias, res, M, 512, 512, -1, y, 0, s)); });
  for (int t : {1, 16, 32, 64}) {
    char name[128]; snprintf(name, sizeof name, "attention_decode rows=%d H=8 dh=64 T=%d", M, t);
    run(name, 2.0 * M * H * t * dh * 4, [&] { CX(ct2x_attention_decode(qv, kc, vc, nullptr, t, M, H, H, dh, T, 1, 0.125f, ctx, s)); });
  }
  run("topk k=1 [M,32000]", M * 32000.0 * 4, [&] { CX(ct2x_topk(y, 0, M, V, 1, topv, topi, s)); });
  run("softmax log [M,32000]", M * 32000.0 * 8, [&] { CX(ct2x_softmax(y, nullptr, M, V, 1, y, 0, s)); });
  return 0;
}
