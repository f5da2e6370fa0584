// Row-wise HBM-bound kernels of the decoder hot path for gfx950 (wave64):
// Quantize, Dequantize, LayerNorm/RMSNorm (+fused Quantize), (Log)SoftMax, Rotary, TopK,
// Gather, embedding lookup. One workgroup per row, 16-byte loads where the row allows,
// wave64 shuffle reductions + one LDS hop per block reduction.
//
// Float formulas follow the reference CPU path (the parity oracle), cited per kernel.
// Compiled with -ffp-contract=off so every expression rounds like the plain C oracle.

#include "device_utils.h"
#include "row_norm.h"
#include "kernels.h"
#include "runtime.h"

#include <float.h>
#include <limits.h>

namespace ct2x {

static inline int pick_block(int64_t depth) {
  // One 16-byte chunk per thread is the sweet spot; clamp to [64, 1024] lanes, wave multiple.
  int64_t t = (depth + 3) / 4;
  int b = 64;
  while (b < t && b < 1024) b <<= 1;
  return b;
}

// ---------------------------------------------------------------------------------------
// ops::Quantize (int8, per row). Reference: src/cpu/kernels.cc:577-651 (quantize_s8_row),
// GPU counterpart src/ops/quantize_gpu.cu:58-103.
// ---------------------------------------------------------------------------------------
__global__ void quantize_s8_kernel(const float* __restrict__ x, int64_t depth,
                                   int quant_mode, int8_t* __restrict__ q,
                                   float* __restrict__ scales) {
  __shared__ float scratch[16];
  const int64_t row = blockIdx.x;
  const float* xr = x + row * depth;
  int8_t* qr = q + row * depth;
  const bool vec = (depth % 4 == 0) && ((reinterpret_cast<uintptr_t>(xr) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(qr) & 3) == 0);
  float amax = 0.f;
  if (vec) {
    const float4* xv = reinterpret_cast<const float4*>(xr);
    for (int64_t i = threadIdx.x; i < depth / 4; i += blockDim.x) {
      float4 v = xv[i];
      amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
  } else {
    for (int64_t i = threadIdx.x; i < depth; i += blockDim.x) amax = fmaxf(amax, fabsf(xr[i]));
  }
  amax = block_max(amax, scratch);
  const float scale = (amax != 0.f) ? 127.f / amax : 1.f;
  if (threadIdx.x == 0) scales[row] = scale;
  if (vec) {
    const float4* xv = reinterpret_cast<const float4*>(xr);
    char4* qv = reinterpret_cast<char4*>(qr);
    for (int64_t i = threadIdx.x; i < depth / 4; i += blockDim.x) {
      float4 v = xv[i];
      char4 o;
      o.x = to_s8(v.x, scale, quant_mode);
      o.y = to_s8(v.y, scale, quant_mode);
      o.z = to_s8(v.z, scale, quant_mode);
      o.w = to_s8(v.w, scale, quant_mode);
      qv[i] = o;
    }
  } else {
    for (int64_t i = threadIdx.x; i < depth; i += blockDim.x)
      qr[i] = to_s8(xr[i], scale, quant_mode);
  }
}

// Two-step form of the same Quantize for tensor-parallel row-parallel layers: the row absmax is
// taken over the rank's K slice, max-reduced across ranks, and the slice is then quantised with
// the GLOBAL scale - bit-identical to quantising the gathered row (what the reference does by
// all-gathering the activations, src/layers/common.cc:360-387), for m floats of traffic instead.
__global__ void row_amax_kernel(const float* __restrict__ x, int64_t depth,
                                float* __restrict__ amax_out) {
  __shared__ float scratch[16];
  const int64_t row = blockIdx.x;
  const float* xr = x + row * depth;
  float amax = 0.f;
  for (int64_t i = threadIdx.x; i < depth; i += blockDim.x) amax = fmaxf(amax, fabsf(xr[i]));
  amax = block_max(amax, scratch);
  if (threadIdx.x == 0) amax_out[row] = amax;
}

__global__ void quantize_with_amax_kernel(const float* __restrict__ x,
                                          const float* __restrict__ amax, int64_t depth,
                                          int quant_mode, int8_t* __restrict__ q,
                                          float* __restrict__ scales) {
  const int64_t row = blockIdx.x;
  const float a = amax[row];
  const float scale = (a != 0.f) ? 127.f / a : 1.f;
  if (threadIdx.x == 0) scales[row] = scale;
  const float* xr = x + row * depth;
  int8_t* qr = q + row * depth;
  for (int64_t i = threadIdx.x; i < depth; i += blockDim.x) qr[i] = to_s8(xr[i], scale, quant_mode);
}

__global__ void quantize_cols_kernel(const float* __restrict__ x, const float* __restrict__ amax,
                                     int64_t depth, int quant_mode, int8_t* __restrict__ q,
                                     float* __restrict__ scales, int rows, PrefetchArgs pf,
                                     int amax_parts) {
  if (static_cast<int>(blockIdx.y) >= rows) {  // spare workgroups: warm the next GEMM's weight
    weight_prefetch_wg(pf, (static_cast<int>(blockIdx.y) - rows) * gridDim.x + blockIdx.x, pf.wgs);
    return;
  }
  const int64_t row = blockIdx.y;
  // amax [rows, amax_parts]: partial maxima left by the producer's workgroups (plain stores)
  float a = amax[row * amax_parts];
  for (int i = 1; i < amax_parts; ++i) a = fmaxf(a, amax[row * amax_parts + i]);
  const float scale = (a != 0.f) ? 127.f / a : 1.f;
  if (blockIdx.x == 0 && threadIdx.x == 0) scales[row] = scale;
  const int64_t i = (blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x) * 4;
  if (i >= depth) return;
  const float4 v = *reinterpret_cast<const float4*>(x + row * depth + i);
  char4 o;
  o.x = to_s8(v.x, scale, quant_mode);
  o.y = to_s8(v.y, scale, quant_mode);
  o.z = to_s8(v.z, scale, quant_mode);
  o.w = to_s8(v.w, scale, quant_mode);
  *reinterpret_cast<char4*>(q + row * depth + i) = o;
}

void launch_quantize_cols(const float* x, const float* amax, int64_t rows, int64_t depth,
                          int quant_mode, int8_t* q, float* scales, hipStream_t stream,
                          const PrefetchArgs* pf_in, int amax_parts) {
  if (rows == 0 || depth == 0) return;
  if (depth % 4 != 0 || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(q) & 3)) {
    if (amax_parts != 1) throw Error("quantize_cols: partial maxima need the aligned form");
    launch_quantize_with_amax(x, amax, rows, depth, quant_mode, q, scales, stream);
    return;
  }
  const int threads = 256;
  dim3 grid(static_cast<unsigned>((depth / 4 + threads - 1) / threads), static_cast<unsigned>(rows));
  PrefetchArgs pf;
  if (pf_in != nullptr && pf_in->wgs > 0) {
    pf = *pf_in;
    const unsigned extra_y = (static_cast<unsigned>(pf.wgs) + grid.x - 1) / grid.x;
    pf.wgs = static_cast<int>(extra_y * grid.x);  // whole grid rows of spare workgroups
    grid.y += extra_y;
  }
  hipLaunchKernelGGL(quantize_cols_kernel, grid, dim3(threads), 0, stream, x, amax, depth, quant_mode,
                     q, scales, static_cast<int>(rows), pf, amax_parts);
}

void launch_row_amax(const float* x, int64_t rows, int64_t depth, float* amax,
                     hipStream_t stream) {
  if (rows == 0) return;
  hipLaunchKernelGGL(row_amax_kernel, dim3(rows), dim3(pick_block(depth)), 0, stream, x, depth,
                     amax);
}

void launch_quantize_with_amax(const float* x, const float* amax, int64_t rows, int64_t depth,
                               int quant_mode, int8_t* q, float* scales, hipStream_t stream) {
  if (rows == 0) return;
  hipLaunchKernelGGL(quantize_with_amax_kernel, dim3(rows), dim3(pick_block(depth)), 0, stream, x,
                     amax, depth, quant_mode, q, scales);
}

void launch_quantize_s8(const float* x, int64_t rows, int64_t depth, int quant_mode,
                        int8_t* q, float* scales, hipStream_t stream) {
  if (rows == 0 || depth == 0) return;
  hipLaunchKernelGGL(quantize_s8_kernel, dim3(rows), dim3(pick_block(depth)), 0, stream, x,
                     depth, quant_mode, q, scales);
}

// ---------------------------------------------------------------------------------------
// ops::Dequantize rows: y = x * (1 / scale[row])   (src/ops/dequantize_cpu.cc:10-19,
// dequantize_kernel: r_scale = 1.f / scale, then v * r_scale; the reference's GPU kernel divides,
// src/ops/dequantize_gpu.cu:17-28 - the CPU form is the parity target).
// ---------------------------------------------------------------------------------------
__global__ void dequantize_rows_kernel(const int8_t* __restrict__ x,
                                       const float* __restrict__ scale, int64_t depth,
                                       float* __restrict__ y) {
  const int64_t row = blockIdx.x;
  const float r_s = 1.f / scale[row];
  const int8_t* xr = x + row * depth;
  float* yr = y + row * depth;
  for (int64_t i = threadIdx.x; i < depth; i += blockDim.x)
    yr[i] = static_cast<float>(xr[i]) * r_s;
}

void launch_dequantize_rows(const int8_t* x, const float* scale, int64_t rows, int64_t depth,
                            float* y, hipStream_t stream) {
  if (rows == 0 || depth == 0) return;
  hipLaunchKernelGGL(dequantize_rows_kernel, dim3(rows), dim3(256), 0, stream, x, scale, depth,
                     y);
}

// ---------------------------------------------------------------------------------------
// ops::Dequantize GEMM-output form, standalone (the fused form lives in gemm_s8.hip).
// CPU formula order (src/cpu/kernels.cc:653-692): c * (1/a_scale) / b_scale + bias -> act.
// ---------------------------------------------------------------------------------------
__global__ void dequantize_gemm_output_kernel(const int32_t* __restrict__ c,
                                              const float* __restrict__ a_scale,
                                              const float* __restrict__ b_scale,
                                              const float* __restrict__ bias,
                                              const float* residual, int64_t n, int act,
                                              float* y) {
  const int64_t row = blockIdx.x;
  const float r_a = 1.f / a_scale[row];
  const int32_t* cr = c + row * n;
  float* yr = y + row * n;
  for (int64_t j = threadIdx.x; j < n; j += blockDim.x) {
    float v = static_cast<float>(cr[j]) * r_a;
    v = v / b_scale[j];
    if (bias) v = v + bias[j];
    v = apply_activation(v, act);
    if (residual) v = residual[row * n + j] + v;  // ops::Add()(residual, output, output)
    yr[j] = v;
  }
}

void launch_dequantize_gemm_output(const int32_t* c, const float* a_scale, const float* b_scale,
                                   const float* bias, int64_t m, int64_t n, int act, float* y,
                                   hipStream_t stream, const float* residual) {
  if (m == 0 || n == 0) return;
  hipLaunchKernelGGL(dequantize_gemm_output_kernel, dim3(m), dim3(256), 0, stream, c, a_scale,
                     b_scale, bias, residual, n, act, y);
}

// y[i, j] = residual[i, j] + (x[i, j] + bias[j]): the epilogue of a float row-parallel Dense after
// its partial products have been summed over the ranks.
__global__ void bias_residual_kernel(const float* x, const float* __restrict__ bias,
                                     const float* residual, int64_t n, int64_t total, float* y) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= total) return;
  float v = x[i];
  if (bias) v = v + bias[i % n];
  if (residual) v = residual[i] + v;
  y[i] = v;
}

void launch_bias_residual(const float* x, const float* bias, const float* residual, int64_t m,
                          int64_t n, float* y, hipStream_t stream) {
  const int64_t total = m * n;
  if (total == 0) return;
  hipLaunchKernelGGL(bias_residual_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, x, bias,
                     residual, n, total, y);
}

// Element-wise combine of up to 8 equally sized buffers in rank order (the in-process
// "loopback" communicator of tp_comm.cc; RCCL does this on real multi-GPU runs).
struct ReducePtrs { const void* p[8]; };
template <typename T, bool kMax>
__global__ void reduce_ranks_kernel(ReducePtrs src, int world, int64_t n, T* out) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  T v = static_cast<const T*>(src.p[0])[i];
  for (int r = 1; r < world; ++r) {
    const T o = static_cast<const T*>(src.p[r])[i];
    v = kMax ? (o > v ? o : v) : v + o;
  }
  out[i] = v;
}

void launch_reduce_ranks(const void* const* src, int world, int64_t n, bool is_int32, bool is_max,
                         void* out, hipStream_t stream) {
  if (n == 0) return;
  ReducePtrs ptrs;
  for (int r = 0; r < 8; ++r) ptrs.p[r] = r < world ? src[r] : nullptr;
  const dim3 grid((n + 255) / 256), block(256);
  if (is_int32)
    hipLaunchKernelGGL((reduce_ranks_kernel<int32_t, false>), grid, block, 0, stream, ptrs, world,
                       n, static_cast<int32_t*>(out));
  else if (is_max)
    hipLaunchKernelGGL((reduce_ranks_kernel<float, true>), grid, block, 0, stream, ptrs, world, n,
                       static_cast<float*>(out));
  else
    hipLaunchKernelGGL((reduce_ranks_kernel<float, false>), grid, block, 0, stream, ptrs, world, n,
                       static_cast<float*>(out));
}

// ---------------------------------------------------------------------------------------
// LayerNorm / RMSNorm (+ optional fused int8 Quantize of the normalised row), BIT-EXACT with the
// reference CPU kernels as compiled (src/cpu/kernels.cc:461-496 and :543-573: a CT2_FFAST_MATH
// region that GCC -O3 - the reference's Release build, CMakeLists.txt:59-61 - vectorises for
// AVX2; oracle/ct2_oracle.c row_stats / orc_layer_norm / orc_rms_norm give the decoded
// arithmetic and are pinned against the reference library):
//   statistics: eight lane accumulators over the leading depth & ~7 elements (sum += x,
//     sumsq = fma(x, x, sumsq)), reduced lane l + l^4, then ^2, then ^1; a remainder of >= 4
//     elements goes four at once (rounded squares, (q1+q3) + (q0+q2)), then <= 3 scalar steps;
//   LayerNorm: mean = sum*inv_n; var = max(fma(sumsq, inv_n, -(mean*mean)), 0);
//              rstd = 1/sqrt(var+eps); y = fma(x - mean, rstd*gamma, beta)
//   RMSNorm:   inv = 1/sqrt(fma(sumsq, inv_n, eps)); x*inv*g re-associated per loop version:
//              (x*g)*inv [gamma] / (inv*x)*g [gamma + 1] in the 8-wide body and the scalar tail,
//              x*(inv*g) / (x*g)*inv in the 4-wide remainder.
// The int8 decoder re-quantises this output: a last-ulp difference flips int8 activations, so the
// order is reproduced as it is. The row is staged in LDS; lanes 0..7 of every wave walk their
// strided chains (depth / 8 dependent steps: ~64 x 13 cycles at depth 512 - a dependent fp32 add
// costs ~10 cycles on this chip, which is why the eight-lane order is 8x cheaper here than a
// scalar one), then each lane normalises the elements it holds in registers.
// ---------------------------------------------------------------------------------------
__global__ void norm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                            const float* __restrict__ beta, int64_t depth, float eps,
                            bool rms_use_residual, int quant_mode,
                            float* __restrict__ y, int8_t* __restrict__ q,
                            float* __restrict__ scales) {
  extern __shared__ __attribute__((aligned(16))) float row_buf[];
  __shared__ float scratch[16];
  const int64_t row = blockIdx.x;
  const float* xr = x + row * depth;
  for (int64_t i = threadIdx.x; i < depth; i += blockDim.x) row_buf[i] = xr[i];
  __syncthreads();
  float sum, sumsq;
  ordered_row_stats(row_buf, depth, sum, sumsq);  // every wave, redundantly
  const bool ln = beta != nullptr;
  const NormStats st = norm_stats(sum, sumsq, depth, eps, ln);
  __syncthreads();  // everyone has read the row before it is overwritten
  float amax = 0.f;
  for (int64_t i = threadIdx.x; i < depth; i += blockDim.x) {
    const float v = norm_apply(row_buf[i], gamma[i], ln ? beta[i] : 0.f, st, ln, rms_use_residual,
                               static_cast<int>(i));
    row_buf[i] = v;
    amax = fmaxf(amax, fabsf(v));
  }
  // Each thread re-reads only the elements it wrote itself: no barrier needed for row_buf.
  if (y != nullptr) {
    float* yr = y + row * depth;
    for (int64_t i = threadIdx.x; i < depth; i += blockDim.x) yr[i] = row_buf[i];
  }
  if (q != nullptr) {
    amax = block_max(amax, scratch);
    const float scale = (amax != 0.f) ? 127.f / amax : 1.f;
    if (threadIdx.x == 0) scales[row] = scale;
    int8_t* qr = q + row * depth;
    for (int64_t i = threadIdx.x; i < depth; i += blockDim.x)
      qr[i] = to_s8(row_buf[i], scale, quant_mode);
  }
}


// Register-resident variant for the decoder shapes (depth % 4 == 0, depth <= 4096): every lane
// keeps its CH float4 chunks in VGPRs and drops a TRANSPOSED copy into LDS for the ordered
// statistics (row_norm.h ordered_row_stats_t); all loads are issued before the first use. Rows of
// <= 1024 elements run as ONE wave (no block barrier on the way to the absmax), wider rows as four.
// Phase costs measured with tools/norm_probe.cpp ([32, 512], cycles): loads 850, strided chains
// 2500 -> contiguous chains, ds_bpermute block max 1130 -> DPP, stats + apply 670, quantize 630.
template <int CH>
__global__ __launch_bounds__(256) void norm_vec_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                const float* __restrict__ beta, int64_t depth, float eps,
                                bool rms_use_residual, int quant_mode, int pad,
                                float* __restrict__ y, int8_t* __restrict__ q,
                                float* __restrict__ scales, int rows, PrefetchArgs pf,
                                bool fast_stats) {
  extern __shared__ __attribute__((aligned(16))) float chain_buf[];
  __shared__ float scratch[32];
  if (static_cast<int>(blockIdx.x) >= rows) {  // spare workgroups: warm the next GEMM's weight
    weight_prefetch_wg(pf, static_cast<int>(blockIdx.x) - rows, pf.wgs);
    return;
  }
  const int64_t row = blockIdx.x;
  const int n4 = static_cast<int>(depth / 4);
  const int vsize = static_cast<int>(depth - depth % 8);
  const float4* xr = reinterpret_cast<const float4*>(x + row * depth);
  const float4* g4 = reinterpret_cast<const float4*>(gamma);
  const float4* b4 = reinterpret_cast<const float4*>(beta);
  float4 v[CH], gv[CH], bv[CH];
#pragma unroll
  for (int u = 0; u < CH; ++u) {
    const int i = threadIdx.x + u * blockDim.x;
    v[u] = gv[u] = bv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4) {
      v[u] = xr[i];
      gv[u] = g4[i];
      if (beta != nullptr) bv[u] = b4[i];
    }
  }
  float sum, sumsq;
  const bool ln = beta != nullptr;
  if (fast_stats) {
    // 16-bit compute types: their GEMMs round to the 16-bit type anyway, parity is a tolerance, and
    // the reference's own GPU kernel reduces in parallel (src/ops/layer_norm_gpu.cu:174-213): the
    // statistics are a tree over the lanes (DPP) and waves instead of the CPU path's 8 x depth / 8
    // dependent chain (at depth 2048: 256 dependent steps, most of this launch's 7 us)
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int u = 0; u < CH; ++u) {  // (absent chunks hold zeros)
      s += (v[u].x + v[u].y) + (v[u].z + v[u].w);
      ss += (v[u].x * v[u].x + v[u].y * v[u].y) + (v[u].z * v[u].z + v[u].w * v[u].w);
    }
    s = wave_sum_dpp(s);
    ss = wave_sum_dpp(ss);
    if (blockDim.x > 64) {
      const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
      if ((threadIdx.x & 63) == 0) {
        scratch[wave] = s;
        scratch[8 + wave] = ss;
      }
      __syncthreads();
      s = ss = 0.f;
      for (int w = 0; w < nw; ++w) {
        s += scratch[w];
        ss += scratch[8 + w];
      }
      __syncthreads();  // scratch is reused by block_max below
    }
    sum = s;
    sumsq = ss;
  } else {
#pragma unroll
  for (int u = 0; u < CH; ++u) {
    const int i = threadIdx.x + u * blockDim.x;
    if (i < n4) {
      const int c = 4 * i;
      chain_buf[norm_chain_slot(c, vsize, pad)] = v[u].x;
      chain_buf[norm_chain_slot(c + 1, vsize, pad)] = v[u].y;
      chain_buf[norm_chain_slot(c + 2, vsize, pad)] = v[u].z;
      chain_buf[norm_chain_slot(c + 3, vsize, pad)] = v[u].w;
    }
  }
  __syncthreads();
  // every wave, redundantly (RMSNorm: the sum is not needed - one dependent chain instead of two)
  if (ln) ordered_row_stats_t<true>(chain_buf, pad, depth, sum, sumsq);
  else ordered_row_stats_t<false>(chain_buf, pad, depth, sum, sumsq);
  }
  const NormStats st = norm_stats(sum, sumsq, depth, eps, ln);
  float amax = 0.f;
#pragma unroll
  for (int u = 0; u < CH; ++u) {
    const int c = 4 * (threadIdx.x + u * blockDim.x);
    v[u].x = norm_apply(v[u].x, gv[u].x, bv[u].x, st, ln, rms_use_residual, c);
    v[u].y = norm_apply(v[u].y, gv[u].y, bv[u].y, st, ln, rms_use_residual, c + 1);
    v[u].z = norm_apply(v[u].z, gv[u].z, bv[u].z, st, ln, rms_use_residual, c + 2);
    v[u].w = norm_apply(v[u].w, gv[u].w, bv[u].w, st, ln, rms_use_residual, c + 3);
  }
#pragma unroll
  for (int u = 0; u < CH; ++u) {
    const int i = threadIdx.x + u * blockDim.x;
    if (i < n4) {
      amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[u].x), fabsf(v[u].y)), fmaxf(fabsf(v[u].z), fabsf(v[u].w))));
      if (y != nullptr) reinterpret_cast<float4*>(y + row * depth)[i] = v[u];
    }
  }
  if (q != nullptr) {
    amax = (blockDim.x == 64) ? wave_max(amax) : block_max(amax, scratch);
    const float scale = (amax != 0.f) ? 127.f / amax : 1.f;
    if (threadIdx.x == 0) scales[row] = scale;
    char4* qr = reinterpret_cast<char4*>(q + row * depth);
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int i = threadIdx.x + u * blockDim.x;
      if (i < n4) {
        char4 o;
        o.x = to_s8(v[u].x, scale, quant_mode);
        o.y = to_s8(v[u].y, scale, quant_mode);
        o.z = to_s8(v[u].z, scale, quant_mode);
        o.w = to_s8(v[u].w, scale, quant_mode);
        qr[i] = o;
      }
    }
  }
}

void launch_norm(const float* x, const float* gamma, const float* beta, int64_t rows,
                 int64_t depth, float eps, bool rms_use_residual, int quant_mode,
                 float* y, int8_t* q, float* scales, hipStream_t stream, const PrefetchArgs* pf_in,
                 bool fast_stats) {
  if (rows == 0 || depth == 0) return;
  const bool aligned = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gamma) |
                         reinterpret_cast<uintptr_t>(beta) | reinterpret_cast<uintptr_t>(y) |
                         reinterpret_cast<uintptr_t>(q)) & 15) == 0;
  if (depth % 4 == 0 && depth <= 4096 && aligned) {
    const int n4 = static_cast<int>(depth / 4);
    const int threads = depth <= 1024 ? 64 : 256;
    const int ch = (n4 + threads - 1) / threads;  // 1..4
    PrefetchArgs pf;
    if (pf_in != nullptr && threads == 256) pf = *pf_in;  // spare workgroups are 256 threads wide
    const int pad = norm_chain_pad(depth);
    const size_t lds = (static_cast<size_t>(8) * pad + 8) * sizeof(float);
#define CT2X_NORMV(CH)                                                                          \
  hipLaunchKernelGGL(norm_vec_kernel<CH>, dim3(rows + pf.wgs), dim3(threads), lds, stream, x, gamma,  \
                     beta, depth, eps, rms_use_residual, quant_mode, pad, y, q, scales,               \
                     static_cast<int>(rows), pf, fast_stats && depth % 8 == 0)
    if (ch <= 1) CT2X_NORMV(1);
    else if (ch == 2) CT2X_NORMV(2);
    else CT2X_NORMV(4);
#undef CT2X_NORMV
    return;
  }
  int block = 64;
  while (block < depth && block < 512) block <<= 1;
  hipLaunchKernelGGL(norm_kernel, dim3(rows), dim3(block), depth * sizeof(float), stream, x,
                     gamma, beta, depth, eps, rms_use_residual, quant_mode, y, q, scales);
}

// ---------------------------------------------------------------------------------------
// SoftMax / LogSoftMax with optional per-row valid length, BIT-EXACT with cpu::softmax<AVX2>
// (src/cpu/kernels.cc:402-459; oracle/ct2_oracle.c orc_softmax): masked tail = 0; exp256_ps with
// GCC's FMA contraction (exp_avx2, attention_exact.hip); the sum runs as eight strided partial
// sums + reduce_m256 + scalar tail; softmax = e * (1/sum); log-softmax = x + (-max - log(sum)).
// The exps are parked in the output row (global memory, re-read by wave 0 for the ordered sum).
// The final log is glibc's logf restated (logf_glibc, device_utils.h). One documented residual of
// the log form: the scalar tail (depth % 8 elements, or a whole row of <= 8) uses libm expf on
// the CPU = glibc's expf, restated exactly as expf_glibc (activation_exact.h; exhaustively equal to
// libm on the host).
// ---------------------------------------------------------------------------------------
__global__ void softmax_kernel(const float* x, const int32_t* __restrict__ lengths,
                               const float* __restrict__ row_bias, int64_t depth, bool log,
                               int chain_pad, float* y) {  // x == y is allowed (ops::SoftMax runs in place)
  extern __shared__ __attribute__((aligned(16))) float exp_lds[];
  __shared__ float scratch[16];
  const int64_t row = blockIdx.x;
  const float* xr = x + row * depth;
  float* yr = y + row * depth;
  int64_t size = depth;
  if (lengths != nullptr) {
    size = lengths[row];
    if (size > depth) size = depth;
    if (size < 0) size = 0;
    for (int64_t j = size + threadIdx.x; j < depth; j += blockDim.x) yr[j] = 0.f;
    if (size == 0) return;
  }
  float m = -FLT_MAX;
  for (int64_t j = threadIdx.x; j < size; j += blockDim.x) m = fmaxf(m, xr[j]);
  m = block_max(m, scratch);
  // body / tail split of the reference: the log form sends the last size % 8 elements (a whole
  // row of <= 8) through libm expf, the plain form sends everything through the polynomial
  const int64_t poly_end = log ? (size > 8 ? size - size % 8 : 0) : size;
  const int64_t vsize = size > 8 ? size - size % 8 : 0;  // part summed by the eight chains
  float s;
  if (chain_pad > 0) {
    // exps parked transposed in LDS for the ordered sum (and written to y for the plain form)
    float* tail = exp_lds + 8 * chain_pad;
    for (int64_t j = threadIdx.x; j < size; j += blockDim.x) {
      const float e = (j < poly_end) ? exp_avx2(xr[j] - m) : expf_glibc(xr[j] - m);
      if (j < vsize)
        exp_lds[chain_slot(j, chain_pad)] = e;
      else
        tail[j - vsize] = e;
      if (!log) yr[j] = e;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
      float t;
      if (size <= 8) {  // front to back
        t = 0.f;
        for (int64_t j = 0; j < size; ++j) t = t + tail[j];
      } else {
        t = ordered_chain_sum_lds(exp_lds, chain_pad, vsize, size);
      }
      if (threadIdx.x == 0) scratch[0] = t;
    }
    __syncthreads();
    s = scratch[0];
  } else if (log && x == y) {
    // in place and too long for LDS: the inputs are still needed for the result, so the eight
    // chain lanes of wave 0 evaluate their own exps (slow; the engine never takes this arm)
    if (threadIdx.x < 64) {
      const int l = threadIdx.x & 7;
      float acc = 0.f;
      if (size <= 8) {
        for (int64_t j = 0; j < size; ++j) acc += expf_glibc(xr[j] - m);
      } else {
        for (int64_t j = l; j < vsize; j += 8) acc += exp_avx2(xr[j] - m);
        acc = acc + __shfl_xor(acc, 4, 64);
        acc = acc + __shfl_xor(acc, 2, 64);
        acc = acc + __shfl_xor(acc, 1, 64);
        for (int64_t j = vsize; j < size; ++j) acc += expf_glibc(xr[j] - m);
      }
      if (threadIdx.x == 0) scratch[0] = acc;
    }
    __syncthreads();
    s = scratch[0];
  } else {
    for (int64_t j = threadIdx.x; j < size; j += blockDim.x)
      yr[j] = (j < poly_end) ? exp_avx2(xr[j] - m) : expf_glibc(xr[j] - m);
    __syncthreads();  // the row of exps is visible to wave 0 (same CU)
    if (threadIdx.x < 64) {
      const float t = ordered_sum8(yr, static_cast<int>(size));
      if (threadIdx.x == 0) scratch[0] = t;
    }
    __syncthreads();
    s = scratch[0];
  }
  if (log) {
    const float c = -m - logf_glibc(s);
    for (int64_t j = threadIdx.x; j < size; j += blockDim.x) {
      float v = xr[j] + c;
      if (row_bias) v = v + row_bias[row];
      yr[j] = v;
    }
  } else {
    const float r = 1.f / s;
    for (int64_t j = threadIdx.x; j < size; j += blockDim.x) yr[j] = yr[j] * r;
  }
}

static void launch_softmax_impl(const float* x, const int32_t* lengths, const float* row_bias,
                                int64_t rows, int64_t depth, bool log, float* y, hipStream_t stream) {
  const ExpSumLayout el = exp_sum_layout(depth);
  const size_t lds = static_cast<size_t>(el.lds_floats) * sizeof(float);
  allow_big_lds(reinterpret_cast<const void*>(softmax_kernel), lds);
  hipLaunchKernelGGL(softmax_kernel, dim3(rows), dim3(pick_block(depth)), lds, stream, x, lengths,
                     row_bias, depth, log, el.chain_pad, y);
}

void launch_softmax(const float* x, const int32_t* lengths, int64_t rows, int64_t depth,
                    bool log, float* y, hipStream_t stream) {
  if (rows == 0 || depth == 0) return;
  launch_softmax_impl(x, lengths, nullptr, rows, depth, log, y, stream);
}

void launch_log_softmax_bias(const float* x, const float* row_bias, int64_t rows, int64_t depth,
                             float* y, hipStream_t stream) {
  if (rows == 0 || depth == 0) return;
  launch_softmax_impl(x, nullptr, row_bias, rows, depth, true, y, stream);
}

// ---------------------------------------------------------------------------------------
// Rotary embeddings. Reference: src/ops/rotary_cpu.cc:9-44.
//   interleave:     y[i] = x[i]*c[i] + (i even ? -x[i+1] : x[i-1]) * s[i]
//   non-interleave: y[i] = x[i]*c[i] + (i < nd/2 ? -x[i+nd/2] : x[i-nd/2]) * s[i]
// x [batch, time, depth] (batch = B*H), sin/cos [time, ndims]; dims >= ndims are copied.
// ---------------------------------------------------------------------------------------
__global__ void rotary_kernel(const float* __restrict__ x, const float* __restrict__ sin,
                              const float* __restrict__ cos, int64_t time, int64_t depth,
                              int64_t ndims, bool interleave, float* __restrict__ y) {
  const int64_t r = blockIdx.x;  // over batch*time
  const int64_t t = r % time;
  const float* xr = x + r * depth;
  float* yr = y + r * depth;
  const float* s = sin + t * ndims;
  const float* c = cos + t * ndims;
  const int64_t middle = ndims / 2;
  for (int64_t i = threadIdx.x; i < depth; i += blockDim.x) {
    if (i < ndims) {
      float other;
      if (interleave)
        other = (i % 2 == 0) ? -xr[i + 1] : xr[i - 1];
      else
        other = (i < middle) ? -xr[i + middle] : xr[i - middle];
      yr[i] = xr[i] * c[i] + other * s[i];
    } else {
      yr[i] = xr[i];
    }
  }
}

void launch_rotary(const float* x, const float* sin, const float* cos, int64_t batch,
                   int64_t time, int64_t depth, int64_t ndims, bool interleave, float* y,
                   hipStream_t stream) {
  if (batch * time == 0) return;
  if (ndims == 0) ndims = depth;
  int block = 64;
  while (block < depth && block < 256) block <<= 1;
  hipLaunchKernelGGL(rotary_kernel, dim3(batch * time), dim3(block), 0, stream, x, sin, cos,
                     time, depth, ndims, interleave, y);
}

// ---------------------------------------------------------------------------------------
// TopK. Reference: src/ops/topk_cpu.cc:13-55 (k==1: std::max_element -> first maximum;
// k>1: partial_sort by value). Selection here is k rounds of block arg-max over the
// lexicographic order (value desc, index asc); round j only considers candidates strictly
// after the (value,index) picked in round j-1, so the input is never mutated (the reference
// GPU kernel mutates and restores, src/ops/topk_gpu.cu:182-335).
// ---------------------------------------------------------------------------------------
__global__ void topk_kernel(const float* __restrict__ x, int64_t depth, int k,
                            float* __restrict__ values, int32_t* __restrict__ indices) {
  __shared__ float s_v[16];
  __shared__ int s_i[16];
  __shared__ float prev_v;
  __shared__ int prev_i;
  const int64_t row = blockIdx.x;
  const float* xr = x + row * depth;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nwaves = (blockDim.x + 63) >> 6;
  float pv = 0.f;
  int pi = -1;  // -1: nothing selected yet
  for (int j = 0; j < k; ++j) {
    ValIdx best;
    best.v = -FLT_MAX;
    best.i = INT_MAX;
    for (int64_t i = threadIdx.x; i < depth; i += blockDim.x) {
      const float v = xr[i];
      const bool eligible = (pi < 0) || (v < pv) || (v == pv && static_cast<int>(i) > pi);
      if (eligible) {
        ValIdx c;
        c.v = v;
        c.i = static_cast<int>(i);
        // -FLT_MAX sentinel: an eligible element always beats the empty slot.
        best = better(best, c);
      }
    }
    // wave reduce (empty slots have i == INT_MAX and v == -FLT_MAX; `better` prefers real
    // entries with the same value because their index is lower).
    best = wave_argmax(best);
    __syncthreads();
    if (lane == 0) {
      s_v[wave] = best.v;
      s_i[wave] = best.i;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      ValIdx r;
      r.v = s_v[0];
      r.i = s_i[0];
      for (int w = 1; w < nwaves; ++w) {
        ValIdx c;
        c.v = s_v[w];
        c.i = s_i[w];
        r = better(r, c);
      }
      prev_v = r.v;
      prev_i = r.i;
      if (r.i == INT_MAX) {  // fewer than k finite candidates
        values[row * k + j] = -FLT_MAX;
        indices[row * k + j] = 0;
      } else {
        values[row * k + j] = r.v;
        indices[row * k + j] = r.i;
      }
    }
    __syncthreads();
    pv = prev_v;
    pi = prev_i;
    if (pi == INT_MAX) pi = INT_MAX - 1;  // exhausted: nothing is eligible any more
  }
}

// Wide rows (the vocabulary, or beam * vocabulary in beam search) in two launches: the one-
// workgroup-per-row kernel above walks a 128000-column row k times with one dependent load per
// iteration (296 us for top-8 over [32, 128000]) and uses 32 of the 256 CUs.
//  1. topk_partial_kernel: grid (chunks, rows), chunk = 4096 columns = 16 per thread, loaded ONCE
//     into registers (all loads in flight); k rounds of block arg-max over the registers (DPP
//     wave reductions), the winner's owner retires its element. Output: the chunk's k best
//     (value, index), sorted.
//  2. topk_merge_kernel: per row, k rounds of block arg-max over the chunks * k partial results
//     (registers again).
// Exact for the same total order (value descending, index ascending): every element of the
// global top-k is in the top-k of its chunk.
constexpr int kTopkChunk = 4096;
constexpr int kTopkThreads = 256;
constexpr int kTopkEPT = kTopkChunk / kTopkThreads;  // 16 columns per thread (4 x float4)
constexpr int kTopkMergeMax = 8;  // partial candidates per thread of the merge (256 threads)

// Block arg-max over (value desc, index asc); empty slots are (-inf, INT_MAX).
__device__ __forceinline__ ValIdx block_argmax(ValIdx best, float* s_v, int* s_i) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nwaves = (blockDim.x + 63) >> 6;
  best = wave_argmax_dpp(best);
  __syncthreads();  // previous round's readers are done with s_v / s_i
  if (lane == 0) {
    s_v[wave] = best.v;
    s_i[wave] = best.i;
  }
  __syncthreads();
  ValIdx r;
  r.v = s_v[0];
  r.i = s_i[0];
  for (int w = 1; w < nwaves; ++w) {
    const float v = s_v[w];
    const int i = s_i[w];
    if (v > r.v || (v == r.v && i < r.i)) {
      r.v = v;
      r.i = i;
    }
  }
  return r;
}

__device__ __forceinline__ ValIdx local_argmax(const float* v, const int* idx, int n) {
  ValIdx best;
  best.v = -__builtin_huge_valf();
  best.i = INT_MAX;
#pragma unroll
  for (int e = 0; e < n; ++e)
    if (v[e] > best.v || (v[e] == best.v && idx[e] < best.i)) {
      best.v = v[e];
      best.i = idx[e];
    }
  return best;
}

__global__ __launch_bounds__(kTopkThreads) void topk_partial_kernel(const float* __restrict__ x,
                                                                    int64_t depth, int k,
                                                                    float* __restrict__ part_v,
                                                                    int32_t* __restrict__ part_i) {
  __shared__ float s_v[16];
  __shared__ int s_i[16];
  const int64_t row = blockIdx.y;
  const int chunk = blockIdx.x, chunks = gridDim.x;
  const float* xr = x + row * depth;
  const float ninf = -__builtin_huge_valf();
  float v[kTopkEPT];
  int idx[kTopkEPT];
  // thread t owns float4 t, t + 256, ... of the chunk: coalesced, every load issued up front
#pragma unroll
  for (int q = 0; q < kTopkEPT / 4; ++q) {
    const int64_t base = static_cast<int64_t>(chunk) * kTopkChunk + (threadIdx.x + q * kTopkThreads) * 4;
    if (base + 3 < depth && ((reinterpret_cast<uintptr_t>(xr + base) & 15) == 0)) {
      const float4 f = *reinterpret_cast<const float4*>(xr + base);
      v[4 * q] = f.x; v[4 * q + 1] = f.y; v[4 * q + 2] = f.z; v[4 * q + 3] = f.w;
#pragma unroll
      for (int e = 0; e < 4; ++e) idx[4 * q + e] = static_cast<int>(base + e);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool ok = base + e < depth;
        v[4 * q + e] = ok ? xr[base + e] : ninf;
        idx[4 * q + e] = ok ? static_cast<int>(base + e) : INT_MAX;  // INT_MAX: empty slot
      }
    }
  }
  float* ov = part_v + (row * chunks + chunk) * k;
  int32_t* oi = part_i + (row * chunks + chunk) * k;
  for (int j = 0; j < k; ++j) {
    const ValIdx r = block_argmax(local_argmax(v, idx, kTopkEPT), s_v, s_i);
#pragma unroll
    for (int e = 0; e < kTopkEPT; ++e)
      if (idx[e] == r.i) {  // retired (indices are unique; INT_MAX stays empty)
        idx[e] = INT_MAX;
        v[e] = ninf;
      }
    if (threadIdx.x == 0) {
      ov[j] = r.v;
      oi[j] = r.i;
    }
  }
}

__global__ __launch_bounds__(256) void topk_merge_kernel(const float* __restrict__ part_v,
                                                         const int32_t* __restrict__ part_i,
                                                         int n_cand, int k,
                                                         float* __restrict__ values,
                                                         int32_t* __restrict__ indices) {
  __shared__ float s_v[16];
  __shared__ int s_i[16];
  const int64_t row = blockIdx.x;
  const float* pv = part_v + row * n_cand;
  const int32_t* pi = part_i + row * n_cand;
  const float ninf = -__builtin_huge_valf();
  float v[kTopkMergeMax];
  int idx[kTopkMergeMax];
#pragma unroll
  for (int e = 0; e < kTopkMergeMax; ++e) {
    const int c = threadIdx.x + e * 256;
    idx[e] = c < n_cand ? pi[c] : INT_MAX;
    v[e] = (c < n_cand && idx[e] != INT_MAX) ? pv[c] : ninf;
  }
  for (int j = 0; j < k; ++j) {
    const ValIdx r = block_argmax(local_argmax(v, idx, kTopkMergeMax), s_v, s_i);
#pragma unroll
    for (int e = 0; e < kTopkMergeMax; ++e)
      if (idx[e] == r.i) {
        idx[e] = INT_MAX;
        v[e] = ninf;
      }
    if (threadIdx.x == 0) {
      if (r.i == INT_MAX) {  // fewer than k candidates
        values[row * k + j] = -FLT_MAX;
        indices[row * k + j] = 0;
      } else {
        values[row * k + j] = r.v;
        indices[row * k + j] = r.i;
      }
    }
  }
}

size_t topk_workspace_bytes(int64_t rows, int64_t depth, int64_t k) {
  const int64_t chunks = (depth + kTopkChunk - 1) / kTopkChunk;
  if (depth < 2 * kTopkChunk || chunks * k > 256 * kTopkMergeMax) return 0;  // single-launch path
  return static_cast<size_t>(rows * chunks * k) * (sizeof(float) + sizeof(int32_t));
}

void launch_topk(const float* x, int64_t rows, int64_t depth, int64_t k, float* values,
                 int32_t* indices, hipStream_t stream, void* workspace, size_t workspace_bytes) {
  if (rows == 0 || depth == 0 || k == 0) return;
  const size_t need = topk_workspace_bytes(rows, depth, k);
  if (need != 0 && workspace != nullptr && workspace_bytes >= need) {
    const int64_t chunks = (depth + kTopkChunk - 1) / kTopkChunk;
    float* part_v = static_cast<float*>(workspace);
    int32_t* part_i = reinterpret_cast<int32_t*>(part_v + rows * chunks * k);
    hipLaunchKernelGGL(topk_partial_kernel, dim3(chunks, rows), dim3(kTopkThreads), 0, stream, x, depth,
                       static_cast<int>(k), part_v, part_i);
    hipLaunchKernelGGL(topk_merge_kernel, dim3(rows), dim3(256), 0, stream, part_v, part_i,
                       static_cast<int>(chunks * k), static_cast<int>(k), values, indices);
    return;
  }
  int block = 64;
  while (block < depth / 4 && block < 1024) block <<= 1;
  hipLaunchKernelGGL(topk_kernel, dim3(rows), dim3(block), 0, stream, x, depth,
                     static_cast<int>(k), values, indices);
}

// ---------------------------------------------------------------------------------------
// Gather rows (ops::Gather axis 0, src/ops/gather_gpu.cu:53-91): out[i] = data[idx[i]].
// 16-byte lanes when the row size allows (it always does for fp32 rows with depth%4==0).
// ---------------------------------------------------------------------------------------
__global__ void gather_rows_kernel(const char* __restrict__ data, const int32_t* __restrict__ idx,
                                   int64_t row_bytes, char* __restrict__ out) {
  const int64_t i = blockIdx.x;
  const char* src = data + static_cast<int64_t>(idx[i]) * row_bytes;
  char* dst = out + i * row_bytes;
  const bool vec = (row_bytes % 16 == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(dst) & 15) == 0);
  if (vec) {
    const uint4* s = reinterpret_cast<const uint4*>(src);
    uint4* d = reinterpret_cast<uint4*>(dst);
    for (int64_t j = threadIdx.x; j < row_bytes / 16; j += blockDim.x) d[j] = s[j];
  } else {
    for (int64_t j = threadIdx.x; j < row_bytes; j += blockDim.x) dst[j] = src[j];
  }
}

// out[t, i] = in[t, idx[i]] for t < steps (4-byte elements): the [step][row] histories of a decoding
// session whose finished rows are being dropped (Engine::compact_rows).
__global__ void gather_cols_kernel(const int32_t* __restrict__ in, const int32_t* __restrict__ idx,
                                   int64_t n_old, int64_t n_new, int64_t total,
                                   int32_t* __restrict__ out) {
  const int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (e >= total) return;
  const int64_t t = e / n_new, i = e - t * n_new;
  out[e] = in[t * n_old + idx[i]];
}

void launch_gather_cols(const void* in, const int32_t* idx, int64_t steps, int64_t n_old, int64_t n_new,
                        void* out, hipStream_t stream) {
  const int64_t total = steps * n_new;
  if (total == 0) return;
  hipLaunchKernelGGL(gather_cols_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0,
                     stream, static_cast<const int32_t*>(in), idx, n_old, n_new, total,
                     static_cast<int32_t*>(out));
}

void launch_gather_rows(const void* data, const int32_t* idx, int64_t n_idx, int64_t row_bytes,
                        void* out, hipStream_t stream) {
  if (n_idx == 0 || row_bytes == 0) return;
  int block = 64;
  while (block < row_bytes / 16 && block < 256) block <<= 1;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(n_idx), dim3(block), 0, stream,
                     static_cast<const char*>(data), idx, row_bytes, static_cast<char*>(out));
}

// ---------------------------------------------------------------------------------------
// Embedding lookup fused with int8 dequantisation, sqrt(d) scaling and position encoding.
// Reference op order: Gather (+Dequantize x/scale) [src/layers/common.cc:64-81] -> Mul by the
// embedding scale [src/layers/transformer.cc:640] -> add position encoding row
// [src/layers/common.cc:151-174]. Row r = (b, t): position = pos_index + t.
// ---------------------------------------------------------------------------------------
__global__ void embed_kernel(const void* __restrict__ table, bool table_is_int8,
                             const float* __restrict__ table_scale,
                             const int32_t* __restrict__ ids, int64_t depth, float emb_scale,
                             bool apply_scale, const float* __restrict__ pos_table, int64_t time,
                             int64_t pos_index, const int32_t* __restrict__ step_ptr,
                             float* __restrict__ out, float* __restrict__ zero_buf,
                             int64_t zero_count) {
  const int64_t r = blockIdx.x;
  // first launch of a decode step: clear the producer-side row-absmax slots the step max-combines
  // into (GemmS8Epilogue::row_amax_out) - saves the memset node of the step's graph
  if (r == 0)
    for (int64_t i = threadIdx.x; i < zero_count; i += blockDim.x) zero_buf[i] = 0.f;
  const int64_t id = ids[r];
  if (step_ptr != nullptr) pos_index = *step_ptr;
  const int64_t pos = pos_index + (r % time);
  float* o = out + r * depth;
  const float* p = pos_table ? pos_table + pos * depth : nullptr;
  if (table_is_int8) {
    const int8_t* e = static_cast<const int8_t*>(table) + id * depth;
    // ops::Dequantize on the CPU multiplies by the reciprocal (dequantize_kernel,
    // src/ops/dequantize_cpu.cc:10-19: r_scale = 1.f / scale; v * r_scale), which is not x / scale
    const float r_s = 1.f / table_scale[id];
    for (int64_t i = threadIdx.x; i < depth; i += blockDim.x) {
      float v = static_cast<float>(e[i]) * r_s;
      if (apply_scale) v = v * emb_scale;
      if (p) v = v + p[i];
      o[i] = v;
    }
  } else {
    const float* e = static_cast<const float*>(table) + id * depth;
    for (int64_t i = threadIdx.x; i < depth; i += blockDim.x) {
      float v = e[i];
      if (apply_scale) v = v * emb_scale;
      if (p) v = v + p[i];
      o[i] = v;
    }
  }
}

void launch_embed(const void* table, bool table_is_int8, const float* table_scale,
                  const int32_t* ids, int64_t rows, int64_t depth, float emb_scale,
                  bool apply_scale, const float* pos_table, int64_t time, int64_t pos_index,
                  const int32_t* step_ptr, float* out, hipStream_t stream, float* zero_buf,
                  int64_t zero_count) {
  if (rows == 0) return;
  int block = 64;
  while (block < depth && block < 256) block <<= 1;
  hipLaunchKernelGGL(embed_kernel, dim3(rows), dim3(block), 0, stream, table, table_is_int8,
                     table_scale, ids, depth, emb_scale, apply_scale, pos_table, time, pos_index,
                     step_ptr, out, zero_buf, zero_count);
}

// ---------------------------------------------------------------------------------------
// Element-wise add / mul (grid-stride, 16-byte lanes when aligned).
// ---------------------------------------------------------------------------------------
template <bool kMul>
__global__ void binary_kernel(const float* __restrict__ a, const float* __restrict__ b,
                              int64_t n, float* __restrict__ y) {
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
    y[i] = kMul ? a[i] * b[i] : a[i] + b[i];
}

static int grid_for(int64_t n) {
  int64_t g = (n + 255) / 256;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return static_cast<int>(g);
}

void launch_add(const float* a, const float* b, int64_t n, float* y, hipStream_t stream) {
  if (n == 0) return;
  hipLaunchKernelGGL(binary_kernel<false>, dim3(grid_for(n)), dim3(256), 0, stream, a, b, n, y);
}

void launch_mul(const float* a, const float* b, int64_t n, float* y, hipStream_t stream) {
  if (n == 0) return;
  hipLaunchKernelGGL(binary_kernel<true>, dim3(grid_for(n)), dim3(256), 0, stream, a, b, n, y);
}

}  // namespace ct2x
