// Row statistics and the normalisation step of LayerNorm / RMSNorm in the reference CPU path's
// compiled order (src/cpu/kernels.cc:461-573 as GCC -O3 vectorises it for AVX2; oracle
// ct2_oracle.c row_stats / orc_layer_norm / orc_rms_norm): shared by the norm kernels
// (rowops.hip) and the row-owning fused blocks (rowblock.hip).
#pragma once

#include "device_utils.h"

namespace ct2x {

__device__ __forceinline__ float dpp_xor4(float v) {
  const float m = dpp_move<0x141, 0xf>(0.f, v);  // row_half_mirror: lane i <- lane 7 - i (per 8)
  return dpp_move<0x1B, 0xf>(0.f, m);             // quad_perm [3,2,1,0]: lane i <- lane i ^ 3
}

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
    // lane ^ 4, ^ 2, ^ 1 on the DPP network (no LDS round trips): ^4 = half-row mirror (i -> 7 - i)
    // followed by a reversal inside the quad ((7 - i) ^ 3 = i ^ 4); ^2 and ^1 are quad permutes
    ss = dpp_xor4(ss) + ss;
    s = dpp_xor4(s) + s;
    ss = dpp_move<0x4E, 0xf>(0.f, ss) + ss;  // quad_perm [2,3,0,1]
    s = dpp_move<0x4E, 0xf>(0.f, s) + s;
    ss = dpp_move<0xB1, 0xf>(0.f, ss) + ss;  // quad_perm [1,0,3,2]
    s = dpp_move<0xB1, 0xf>(0.f, s) + s;
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
