// Do small dependent-kernel chains on DIFFERENT streams overlap on this GPU/runtime?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__global__ void work(float* y, int iters) {
  float v = threadIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i) v = fmaf(v, 1.0001f, 0.5f);
  y[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 200;   // ~3 us of work per kernel
  const int N = 2000;
  for (int ns : {1, 2, 4, 8}) {
    std::vector<hipStream_t> s(ns);
    std::vector<float*> y(ns);
    for (int i = 0; i < ns; ++i) { CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking)); CK(hipMalloc(&y[i], 1 << 20)); }
    // eager, one host thread per stream
    auto run = [&](bool graph) {
      CK(hipDeviceSynchronize());
      double t0 = now();
      std::vector<std::thread> th;
      for (int i = 0; i < ns; ++i) th.emplace_back([&, i] {
        CK(hipSetDevice(0));
        if (!graph) {
          for (int k = 0; k < N; ++k) hipLaunchKernelGGL(work, dim3(32), dim3(256), 0, s[i], y[i], iters);
        } else {
          hipGraph_t g; hipGraphExec_t ge;
          CK(hipStreamBeginCapture(s[i], hipStreamCaptureModeThreadLocal));
          for (int k = 0; k < 100; ++k) hipLaunchKernelGGL(work, dim3(32), dim3(256), 0, s[i], y[i], iters);
          CK(hipStreamEndCapture(s[i], &g));
          CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
          for (int k = 0; k < N / 100; ++k) CK(hipGraphLaunch(ge, s[i]));
          CK(hipStreamSynchronize(s[i]));
          CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
        CK(hipStreamSynchronize(s[i]));
      });
      for (auto& t : th) t.join();
      double t1 = now();
      printf("streams %d %s: %.2f us per kernel per stream, aggregate %.2f kernels/us\n", ns, graph ? "graph" : "eager",
             (t1 - t0) * 1e6 / N, ns * N / ((t1 - t0) * 1e6));
    };
    run(false); run(false); run(true); run(true);
  }
  return 0;
}
