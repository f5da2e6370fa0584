// What an (almost) empty launch costs as a function of its dynamic LDS size and kernarg size:
// hot back-to-back launches of 128 workgroups x 512 threads, HIP events.
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { char b[256]; };
__global__ __launch_bounds__(512) void k_empty(float* out, Big a) {
  extern __shared__ float smem[];
  if (threadIdx.x == 0 && a.b[0] == 77) { smem[0] = 1.f; out[blockIdx.x] = smem[0]; }
}
int main() {
  float* out; hipMalloc(&out, 4096);
  hipFuncSetAttribute((const void*)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  Big a{};
  for (size_t lds : {0ul, 16384ul, 65536ul, 135168ul, 160ul * 1024}) {
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, 0);
      for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_empty, dim3(128), dim3(512), lds, 0, out, a);
      hipEventRecord(e1, 0);
      hipDeviceSynchronize();
      hipEventElapsedTime(&ms, e0, e1);
    }
    printf("empty launch, %zu B dynamic LDS: %.2f us per launch\n", lds, ms * 1e3 / 20);
  }
  return 0;
}
