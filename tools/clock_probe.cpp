// What shader clock does a launch-bound stream of small kernels actually run at?
// A 1-wave kernel runs a dependent v_fma chain of known length (4 cycles per dependent fma on
// gfx950); wall time per kernel minus the empty-kernel time gives the effective clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__global__ void chain(float* y, int iters, int blocks_active) {
  float v = threadIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i) v = fmaf(v, 1.0001f, 0.5f);
  if (v == 12345.f) y[0] = v;
}
__global__ void plug(float* y, int iters) {
  float v = threadIdx.x;
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  y[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
int main() {
  float* y; CK(hipMalloc(&y, 1 << 24));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int N = 500;
  for (int grid : {1, 32, 256, 2048}) {
    float t[2];
    int its[2] = {0, 4000};
    for (int k = 0; k < 2; ++k) {
      hipLaunchKernelGGL(plug, dim3(4096), dim3(256), 0, s, y, 100000);
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(chain, dim3(grid), dim3(64), 0, s, y, its[k], grid);
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      CK(hipEventElapsedTime(&t[k], e0, e1));
    }
    const double us0 = t[0] * 1e3 / N, us1 = t[1] * 1e3 / N;
    const double cycles = 4000.0 * 4.0;  // dependent fma latency ~4 cycles
    printf("grid %5d: empty %.2f us, chain(4000 fma) %.2f us -> %.2f us of work => ~%.0f MHz effective\n",
           grid, us0, us1, us1 - us0, cycles / (us1 - us0));
  }
  // one long kernel for the steady-state clock
  float tl;
  CK(hipEventRecord(e0, s));
  hipLaunchKernelGGL(chain, dim3(2048), dim3(64), 0, s, y, 4000000, 0);
  CK(hipEventRecord(e1, s));
  CK(hipStreamSynchronize(s));
  CK(hipEventElapsedTime(&tl, e0, e1));
  printf("long kernel: 4M dependent fma in %.2f ms => ~%.0f MHz effective\n", tl, 4e6 * 4.0 / (tl * 1e3));
  return 0;
}
