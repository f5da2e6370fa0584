// fp32 (strided-batched) GEMM with the BiasAdd / activation / residual epilogue fused in:
// the float arm of ops::Gemm (src/ops/gemm.cc:11-25,64-76; src/cuda/primitives.cu:485-517)
// and ops::MatMul's gemm_batch_strided (src/cuda/primitives.cu:599-683).
//
// Only float32 *models* and generic MatMul calls route here; the int8 decoder path uses
// gemm_s8.hip. LDS-tiled 64x64x16 with a 4x4 register micro-tile per lane and explicit fmaf
// (the CPU oracle's sgemm is FMA based).

#include "device_utils.h"
#include "kernels.h"

namespace ct2x {

namespace {

constexpr int TM = 64, TN = 64, TK = 16;

template <bool kTransB>
__global__ __launch_bounds__(256) void gemm_f32_kernel(
    const float* __restrict__ A, const float* __restrict__ B, int M, int N, int K, float alpha,
    int64_t stride_a, int64_t stride_b, int64_t stride_c, const float* __restrict__ bias,
    const float* __restrict__ residual, int act, float* __restrict__ C) {
  __shared__ float As[TK][TM + 4];
  __shared__ float Bs[TK][TN + 4];
  const int batch = blockIdx.z;
  A += batch * stride_a;
  B += batch * stride_b;
  C += batch * stride_c;
  if (residual) residual += batch * stride_c;
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
  const int tid = threadIdx.x;
  const int tx = tid % 16, ty = tid / 16;  // 16 x 16 lanes, each a 4x4 micro-tile

  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

  for (int k0 = 0; k0 < K; k0 += TK) {
    // A tile: TM x TK, A is [M, K] row-major.
    for (int e = tid; e < TM * TK; e += 256) {
      const int r = e / TK, kk = e % TK;
      const int gr = m0 + r, gk = k0 + kk;
      As[kk][r] = (gr < M && gk < K) ? A[static_cast<int64_t>(gr) * K + gk] : 0.f;
    }
    // B tile: TK x TN. trans_b: B is [N, K] row-major; otherwise [K, N].
    for (int e = tid; e < TN * TK; e += 256) {
      int c, kk;
      if (kTransB) {
        c = e / TK;
        kk = e % TK;
      } else {
        kk = e / TN;
        c = e % TN;
      }
      const int gc = n0 + c, gk = k0 + kk;
      float v = 0.f;
      if (gc < N && gk < K)
        v = kTransB ? B[static_cast<int64_t>(gc) * K + gk] : B[static_cast<int64_t>(gk) * N + gc];
      Bs[kk][c] = v;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < TK; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[kk][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bs[kk][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = m0 + ty * 4 + i;
    if (row >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = n0 + tx * 4 + j;
      if (col >= N) continue;
      float v = acc[i][j];
      if (alpha != 1.f) v = v * alpha;
      // apply_bias_and_activation: act(x + bias (+ residual)) -- the reference adds the
      // residual BEFORE the activation in the float path (src/ops/bias_add_gpu.cu:46-102,
      // src/ops/gemm.cc:11-25).
      if (bias) v = v + bias[col];
      if (residual) v = v + residual[static_cast<int64_t>(row) * N + col];
      v = apply_activation(v, act);
      C[static_cast<int64_t>(row) * N + col] = v;
    }
  }
}

}  // namespace

void launch_gemm_f32(const float* a, const float* b, int64_t batch, int64_t m, int64_t n,
                     int64_t k, bool trans_b, float alpha, int64_t stride_a, int64_t stride_b,
                     int64_t stride_c, const float* bias, const float* residual, int act,
                     float* c, hipStream_t stream) {
  if (batch == 0 || m == 0 || n == 0) return;
  dim3 grid((n + TN - 1) / TN, (m + TM - 1) / TM, batch);
  if (trans_b)
    hipLaunchKernelGGL(gemm_f32_kernel<true>, grid, dim3(256), 0, stream, a, b,
                       static_cast<int>(m), static_cast<int>(n), static_cast<int>(k), alpha,
                       stride_a, stride_b, stride_c, bias, residual, act, c);
  else
    hipLaunchKernelGGL(gemm_f32_kernel<false>, grid, dim3(256), 0, stream, a, b,
                       static_cast<int>(m), static_cast<int>(n), static_cast<int>(k), alpha,
                       stride_a, stride_b, stride_c, bias, residual, act, c);
}

}  // namespace ct2x
