// Row-owning fused blocks for decode steps (rows <= 32) on gfx950.
//
// (An experiment that lost - see rowblock_disabled() - kept for the record and for comparison
// runs.)
// At batch * beam = 32 a decode step is a chain of ~76 tiny dependent launches, each costing a
// kernel boundary (~1.5 us) plus a memory round trip for data the previous kernel just wrote.
// The column-split skinny GEMM (gemm_s8.hip) is the right shape for streaming a weight ONCE, but
// it scatters an output row over 32 workgroups, so everything row-wise that follows - the residual
// stream's LayerNorm and its Quantize (src/layers/common.cc:339-401, 464-472) - needs another
// launch. For the d x d output projections of the attention blocks the other split pays: ONE
// WORKGROUP PER ROW computes the whole output row (the 256 KB weight is re-read by every
// workgroup, but from the XCD's L2, where 4 MB hold it after the first touch: ~2 us of L2
// bandwidth per CU instead of a launch + a round trip), so the row stays in the workgroup and
//
//   Dense(out) = Gemm(s8.s8->s32) -> Dequantize(+bias) -> Add(residual)
//   -> LayerNorm / RMSNorm of the NEXT sub-layer -> Quantize
//
// become one launch (rowgemv_ln_kernel). Arithmetic is unchanged element by element: exact int32
// dot products (v_mfma_i32_16x16x64_i8 with one live row), the CPU formula order of the
// Dequantize epilogue (gemm_s8.hip store_epilogue), the reference's compiled norm order
// (row_norm.h) and to_s8 - results are bit-identical to the unfused path.
#include "device_utils.h"
#include "kernels.h"
#include "row_norm.h"
#include "runtime.h"

namespace ct2x {

namespace {

struct RowGemvLnArgs {
  const int8_t* qa;       // [rows, K] quantised input rows
  const float* a_scale;   // [rows]
  const int8_t* w;        // [N, K]
  const float* w_scale;   // [N]
  const float* bias;      // [N] or nullptr
  const float* residual;  // [rows, N] or nullptr
  float* y;               // [rows, N] (may be the residual buffer's partner, never qa)
  int N, K;
  // the norm that consumes y next (beta == nullptr: RMSNorm) and its Quantize
  const float* gamma;
  const float* beta;
  float eps;
  int rms_use_residual;
  int quant_mode;
  int8_t* q_out;          // [rows, N]
  float* q_scale_out;     // [rows]
};

// TN: 16-column tiles per wave; the waves of the workgroup split N, every wave walks all of K in
// rounds of U 64-byte steps with all loads of a round in flight before the first MFMA.
template <int TN, int U>
__global__ __launch_bounds__(512) void rowgemv_ln_kernel(RowGemvLnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds_f[];
  __shared__ float scratch[16];
  const int N = a.N, K = a.K;
  float* row_buf = lds_f;                                         // [N] fp32 output row
  int8_t* a_row = reinterpret_cast<int8_t*>(lds_f + N);           // [K] int8 input row
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int64_t row = blockIdx.x;
  const int col0 = wave * (TN * 16);

  // Epilogue operands first: their latency hides under the weight stream.
  const float r_a = 1.f / a.a_scale[row];
  float e_ws[TN], e_bias[TN], e_res[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = col0 + j * 16 + r;
    e_ws[j] = a.w_scale[col];
    e_bias[j] = a.bias ? a.bias[col] : 0.f;
    e_res[j] = a.residual ? a.residual[row * N + col] : 0.f;
  }
  for (int i = tid * 16; i < K; i += blockDim.x * 16)
    *reinterpret_cast<v4i*>(a_row + i) = *reinterpret_cast<const v4i*>(a.qa + row * K + i);
  // this thread's share of the norm parameters (consumed after the GEMV)
  float e_gamma[2], e_beta[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int c = tid + u * blockDim.x;
    e_gamma[u] = (c < N) ? a.gamma[c] : 0.f;
    e_beta[u] = (c < N && a.beta) ? a.beta[c] : 0.f;
  }
  __syncthreads();

  v4i acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) acc[j] = v4i{0, 0, 0, 0};
  const int ksteps = K / 64;
  for (int ks0 = 0; ks0 < ksteps; ks0 += U) {
    v4i af[U], bf[U][TN];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ks = ks0 + u;
      const int kk = ks * 64 + g * 16;
      const bool ok = ks < ksteps;
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bf[u][j] = ok ? *reinterpret_cast<const v4i*>(a.w + static_cast<int64_t>(col0 + j * 16 + r) * K + kk)
                      : v4i{0, 0, 0, 0};
      // one live row: only the lanes of MFMA row 0 carry activation bytes
      af[u] = (ok && r == 0) ? *reinterpret_cast<const v4i*>(a_row + kk) : v4i{0, 0, 0, 0};
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[u], bf[u][j], acc[j], 0, 0, 0);
  }

  // C layout: row = g * 4 + reg, col = r: output row 0 lives in reg 0 of the lanes with g == 0.
  if (g == 0) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = col0 + j * 16 + r;
      // CPU formula order (src/cpu/kernels.cc:666-673): c * (1/a_scale) / b_scale + bias
      float v = static_cast<float>(acc[j][0]) * r_a;
      v = v / e_ws[j];
      if (a.bias) v = v + e_bias[j];
      if (a.residual) v = e_res[j] + v;  // ops::Add()(residual, output, output)
      a.y[row * N + col] = v;
      row_buf[col] = v;
    }
  }
  __syncthreads();

  // ---- the next sub-layer's norm + Quantize on the row (norm_vec_kernel, rowops.hip) ----------
  float sum, sumsq;
  ordered_row_stats(row_buf, N, sum, sumsq);  // every wave, redundantly
  const bool ln = a.beta != nullptr;
  const NormStats st = norm_stats(sum, sumsq, N, a.eps, ln);
  float v[2];
  float amax = 0.f;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int c = tid + u * blockDim.x;
    v[u] = 0.f;
    if (c < N) {
      v[u] = norm_apply(row_buf[c], e_gamma[u], e_beta[u], st, ln, a.rms_use_residual != 0, c);
      amax = fmaxf(amax, fabsf(v[u]));
    }
  }
  amax = block_max(amax, scratch);
  const float scale = (amax != 0.f) ? 127.f / amax : 1.f;
  if (tid == 0) a.q_scale_out[row] = scale;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int c = tid + u * blockDim.x;
    if (c < N) a.q_out[row * N + c] = to_s8(v[u], scale, a.quant_mode);
  }
}

template <int TN, int U>
void launch_rowgemv(const RowGemvLnArgs& a, int64_t rows, int waves, hipStream_t stream) {
  const size_t lds = static_cast<size_t>(a.N) * sizeof(float) + static_cast<size_t>(a.K);
  hipLaunchKernelGGL((rowgemv_ln_kernel<TN, U>), dim3(static_cast<unsigned>(rows)), dim3(waves * 64),
                     lds, stream, a);
}

// MEASURED AND REJECTED as the default (profiles/r02_rowgemv_fusion.txt): the launch it removes
// costs ~4.4 us, but a row-owning workgroup has to pull the whole 256 KB weight through ONE CU's
// L2 port (~30 B/clk): 8.6 us in the kernel against 3.2 + 4.4 us for the two launches it replaces
// (Transformer-base: 80.0k -> 68.2k tokens/s). Kept behind CT2X_ROW_FUSION=1 (bit-identical).
bool rowblock_disabled() {
  static const bool on = [] {
    const char* e = tuning_env("CT2X_ROW_FUSION");
    return e != nullptr && e[0] == '1';
  }();
  return !on;
}

}  // namespace

// Shapes the row-owning Dense + norm + Quantize serves: decode rows, a square-ish projection whose
// weight is small enough to be re-read from L2 by every row (<= 1 MiB), N split over <= 8 waves in
// whole 16-column tiles, at most two output columns per thread.
bool rowgemv_ln_applicable(int64_t rows, int64_t n, int64_t k) {
  if (rowblock_disabled()) return false;
  if (rows < 1 || rows > 64) return false;
  if (n % 16 != 0 || k % 64 != 0 || n * k > (1 << 20)) return false;
  int waves = static_cast<int>(n / 16);
  if (waves > 8) waves = 8;
  if (n % (waves * 16) != 0) return false;
  const int tn = static_cast<int>(n / (waves * 16));
  if (tn != 1 && tn != 2 && tn != 4) return false;
  return n <= 2 * waves * 64;
}

void launch_rowgemv_ln(const int8_t* qa, const float* a_scale, const int8_t* w, const float* w_scale,
                       const float* bias, const float* residual, float* y, int64_t rows, int64_t n,
                       int64_t k, const float* gamma, const float* beta, float eps,
                       bool rms_use_residual, int quant_mode, int8_t* q_out, float* q_scale_out,
                       hipStream_t stream) {
  RowGemvLnArgs a;
  a.qa = qa;
  a.a_scale = a_scale;
  a.w = w;
  a.w_scale = w_scale;
  a.bias = bias;
  a.residual = residual;
  a.y = y;
  a.N = static_cast<int>(n);
  a.K = static_cast<int>(k);
  a.gamma = gamma;
  a.beta = beta;
  a.eps = eps;
  a.rms_use_residual = rms_use_residual ? 1 : 0;
  a.quant_mode = quant_mode;
  a.q_out = q_out;
  a.q_scale_out = q_scale_out;
  int waves = static_cast<int>(n / 16);
  if (waves > 8) waves = 8;
  const int tn = static_cast<int>(n / (waves * 16));
  // U * TN fragment loads of 16 bytes per lane are in flight per round (32 = 128 VGPRs)
  if (tn == 4) launch_rowgemv<4, 8>(a, rows, waves, stream);
  else if (tn == 2) launch_rowgemv<2, 8>(a, rows, waves, stream);
  else launch_rowgemv<1, 8>(a, rows, waves, stream);
}

}  // namespace ct2x
