// Row statistics and the normalisation step of LayerNorm / RMSNorm in the reference CPU path's
// compiled order (src/cpu/kernels.cc:461-573 as GCC -O3 vectorises it for AVX2; oracle
// ct2_oracle.c row_stats / orc_layer_norm / orc_rms_norm): shared by the norm kernels
// (rowops.hip) and the row-owning fused blocks (rowblock.hip).
#pragma once

#include "device_utils.h"

namespace ct2x {

// `row`: the depth floats of the row in LDS (linear). Every lane of the calling wave returns the
// statistics.
__device__ __forceinline__ void ordered_row_stats(const float* row, int64_t depth, float& sum,
                                                  float& sumsq) {
  const int l = threadIdx.x & 7;
  const int vsize = static_cast<int>(depth >= 8 ? depth - depth % 8 : 0);
  float s = 0.f, ss = 0.f;
  if (vsize > 0) {
#pragma unroll 8
    for (int j = l; j < vsize; j += 8) {
      const float v = row[j];
      ss = fmaf(v, v, ss);
      s = s + v;
    }
    ss = __shfl_xor(ss, 4, 64) + ss;
    s = __shfl_xor(s, 4, 64) + s;
    ss = __shfl_xor(ss, 2, 64) + ss;
    s = __shfl_xor(s, 2, 64) + s;
    ss = __shfl_xor(ss, 1, 64) + ss;
    s = __shfl_xor(s, 1, 64) + s;
  }
  int j = vsize;
  if (depth - j >= 4) {
    const float x0 = row[j], x1 = row[j + 1], x2 = row[j + 2], x3 = row[j + 3];
    const float q0 = x0 * x0, q1 = x1 * x1, q2 = x2 * x2, q3 = x3 * x3;
    ss = ss + ((q3 + q1) + (q2 + q0));
    s = s + ((x3 + x1) + (x2 + x0));
    j += 4;
  }
  for (; j < depth; ++j) {
    const float v = row[j];
    ss = fmaf(v, v, ss);
    s = s + v;
  }
  sum = s;
  sumsq = ss;
}

struct NormStats {
  float mean, mul;  // LayerNorm: y = fma(x - mean, mul*gamma, beta); RMSNorm: see norm_apply
  int four_begin, four_end;  // columns served by the 4-wide remainder version of the loop
};

__device__ __forceinline__ NormStats norm_stats(float sum, float sumsq, int64_t depth, float eps,
                                                bool layer_norm) {
  const float inv_n = 1.f / static_cast<float>(depth);
  NormStats st;
  if (layer_norm) {
    st.mean = sum * inv_n;
    const float mm = st.mean * st.mean;
    float variance = fmaf(sumsq, inv_n, -mm);
    variance = (variance > 0.f) ? variance : 0.f;
    st.mul = 1.f / sqrtf(variance + eps);
  } else {
    st.mean = 0.f;
    st.mul = 1.f / sqrtf(fmaf(sumsq, inv_n, eps));
  }
  const int vsize = static_cast<int>(depth >= 8 ? depth - depth % 8 : 0);
  st.four_begin = vsize;
  st.four_end = (depth - vsize >= 4) ? vsize + 4 : vsize;
  return st;
}

__device__ __forceinline__ float norm_apply(float x, float g, float b, const NormStats& st,
                                            bool layer_norm, bool rms_use_residual, int col) {
  if (layer_norm) return fmaf(x - st.mean, st.mul * g, b);
  const bool four = col >= st.four_begin && col < st.four_end;
  if (rms_use_residual) {
    const float g1 = g + 1.f;
    return four ? (x * g1) * st.mul : (st.mul * x) * g1;
  }
  return four ? x * (st.mul * g) : (x * g) * st.mul;
}


}  // namespace ct2x
