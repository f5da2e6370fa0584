// Cost model probe for the ordered (sequential) fp32 sums of the exact LayerNorm: cycles per
// element of (a) one dependent v_add chain, (b) add + fma chains interleaved (registers only),
// (c) the same fed from LDS broadcasts (compiler schedule), (d) hand-pipelined LDS reads,
// (e) v_readlane feeding. One wave; s_memtime stamps.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s\n", hipGetErrorString(e)); return 1; } } while (0)

__global__ void probe(const float* x, float* out, long long* cyc, int n) {
  __shared__ __attribute__((aligned(16))) float row[4096 + 256];
  for (int i = threadIdx.x; i < n; i += blockDim.x) row[i] = x[i];
  __syncthreads();
  const int lane = threadIdx.x;
  float r[8];
  for (int i = 0; i < 8; ++i) r[i] = x[lane * 8 + i];
  long long t0, t1;
  // (a) single dependent add chain on register data (same 8 values re-used: n adds)
  float s = 0.f;
  t0 = clock64();
  for (int i = 0; i < n / 8; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) s = s + r[u];
  }
  t1 = clock64();
  if (lane == 0) cyc[0] = t1 - t0;
  // (b) add + fma interleaved
  float s2 = 0.f, ss = 0.f;
  t0 = clock64();
  for (int i = 0; i < n / 8; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) { ss = fmaf(r[u], r[u], ss); s2 = s2 + r[u]; }
  }
  t1 = clock64();
  if (lane == 0) cyc[1] = t1 - t0;
  // (c) LDS broadcast, compiler schedule
  float s3 = 0.f, ss3 = 0.f;
  const v4f* r4 = reinterpret_cast<const v4f*>(row);
  t0 = clock64();
#pragma unroll 8
  for (int i = 0; i < n / 4; ++i) {
    const v4f v = r4[i];
    ss3 = fmaf(v.x, v.x, ss3); s3 = s3 + v.x; ss3 = fmaf(v.y, v.y, ss3); s3 = s3 + v.y;
    ss3 = fmaf(v.z, v.z, ss3); s3 = s3 + v.z; ss3 = fmaf(v.w, v.w, ss3); s3 = s3 + v.w;
  }
  t1 = clock64();
  if (lane == 0) cyc[2] = t1 - t0;
  // (d) hand-pipelined, 16 reads per batch, no per-element guards (n4 multiple of 32)
  float s4 = 0.f, ss4 = 0.f;
  {
    constexpr int B = 16;
    v4f a[B], b[B];
    const int n4 = n / 4;
    t0 = clock64();
#pragma unroll
    for (int u = 0; u < B; ++u) a[u] = r4[u];
    for (int base = 0; base < n4; base += 2 * B) {
#pragma unroll
      for (int u = 0; u < B; ++u) b[u] = r4[base + B + u];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < B; ++u) {
        ss4 = fmaf(a[u].x, a[u].x, ss4); s4 = s4 + a[u].x; ss4 = fmaf(a[u].y, a[u].y, ss4); s4 = s4 + a[u].y;
        ss4 = fmaf(a[u].z, a[u].z, ss4); s4 = s4 + a[u].z; ss4 = fmaf(a[u].w, a[u].w, ss4); s4 = s4 + a[u].w;
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < B; ++u) a[u] = r4[base + 2 * B + u];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < B; ++u) {
        ss4 = fmaf(b[u].x, b[u].x, ss4); s4 = s4 + b[u].x; ss4 = fmaf(b[u].y, b[u].y, ss4); s4 = s4 + b[u].y;
        ss4 = fmaf(b[u].z, b[u].z, ss4); s4 = s4 + b[u].z; ss4 = fmaf(b[u].w, b[u].w, ss4); s4 = s4 + b[u].w;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    t1 = clock64();
    if (lane == 0) cyc[3] = t1 - t0;
  }
  // (e) readlane feeding: lane l holds r[0..7]; element j = 8*l + u
  float s5 = 0.f, ss5 = 0.f;
  t0 = clock64();
  for (int l = 0; l < n / 8 && l < 64; ++l) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r[u]), l));
      ss5 = fmaf(v, v, ss5); s5 = s5 + v;
    }
  }
  t1 = clock64();
  if (lane == 0) cyc[4] = t1 - t0;
  out[lane] = s + s2 + ss + s3 + ss3 + s4 + ss4 + s5 + ss5;
}

int main() {
  const int n = 512;
  float *x, *out; long long* cyc;
  CK(hipMalloc(&x, 8192 * 4)); CK(hipMalloc(&out, 64 * 4)); CK(hipMalloc(&cyc, 8 * 8));
  std::vector<float> h(8192); for (int i = 0; i < 8192; ++i) h[i] = (i % 37) * 0.013f - 0.2f;
  CK(hipMemcpy(x, h.data(), 8192 * 4, hipMemcpyHostToDevice));
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, x, out, cyc, n);
    CK(hipDeviceSynchronize());
    long long c[8]; CK(hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost));
    printf("n=%d  (a) single add chain %lld  (b) add+fma regs %lld  (c) LDS compiler %lld  (d) LDS pipelined %lld  (e) readlane %lld   [s_memtime ticks; /n = per element]\n",
           n, c[0], c[1], c[2], c[3], c[4]);
  }
  return 0;
}
