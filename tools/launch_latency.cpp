// Microbenchmark: per-kernel cost of back-to-back dependent launches on one stream
// (eager and hipGraph), to calibrate the launch-bound decode step against the guide's
// 1.45 us "dependent kernel boundary" figure.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void tiny(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) *p = *p + 1; }
__global__ void rowk(const float* x, float* y, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = x[i] * 2.f + 1.f;
}
__global__ void heavy(float* y, int iters) {
  float v = threadIdx.x;
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  y[blockIdx.x * blockDim.x + threadIdx.x] = v;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  int n_launch = 2000;
  int* d; float *x, *y;
  CK(hipMalloc(&d, 4)); CK(hipMemset(d, 0, 4));
  CK(hipMalloc(&x, 1 << 22)); CK(hipMalloc(&y, 1 << 24));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    // eager tiny
    CK(hipStreamSynchronize(s));
    double t0 = now();
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < n_launch; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    double t1 = now();
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("eager tiny 1 WG      : %.2f us/kernel (events) %.2f us (wall)\n", ms * 1e3 / n_launch, (t1 - t0) * 1e6 / n_launch);
    // eager 32-WG row kernel
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < n_launch; ++i) hipLaunchKernelGGL(rowk, dim3(32), dim3(512), 0, s, x, y, 32 * 512);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("eager 32 WG x 512    : %.2f us/kernel\n", ms * 1e3 / n_launch);
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < n_launch; ++i) hipLaunchKernelGGL(rowk, dim3(256), dim3(256), 0, s, x, y, 256 * 256);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("eager 256 WG x 256   : %.2f us/kernel\n", ms * 1e3 / n_launch);
    // graph of 100 tiny kernels, launched 20 times
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(rowk, dim3(32), dim3(512), 0, s, x, y, 32 * 512);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("graph 100x(32 WG)    : %.2f us/kernel\n", ms * 1e3 / 2000);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    // after warming clocks with a heavy kernel
    hipLaunchKernelGGL(heavy, dim3(4096), dim3(256), 0, s, y, 200000);
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < n_launch; ++i) hipLaunchKernelGGL(rowk, dim3(32), dim3(512), 0, s, x, y, 32 * 512);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("eager 32 WG after heavy: %.2f us/kernel\n", ms * 1e3 / n_launch);
    // null stream
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < n_launch; ++i) hipLaunchKernelGGL(rowk, dim3(32), dim3(512), 0, 0, x, y, 32 * 512);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("eager 32 WG null stream: %.2f us/kernel\n", ms * 1e3 / n_launch);
  }
  return 0;
}
