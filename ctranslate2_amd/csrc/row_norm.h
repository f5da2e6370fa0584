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

// The same statistics from a TRANSPOSED staging of the row: element j < depth & ~7 sits at
// (j & 7) * pad + (j >> 3), so lane l's chain is contiguous (four elements per ds_read_b128, which
// the compiler keeps in flight ahead of the dependent adds - measured 13 cycles per step against
// ~39 with the strided reads of the linear layout); the depth % 8 tail sits at 8 * pad + t.
// pad = norm_chain_pad(depth): >= depth / 8, a multiple of 4, and 4 mod 32 so that the eight
// lanes' 16-byte reads hit eight different bank quads.
__host__ __device__ __forceinline__ int norm_chain_pad(int64_t depth) {
  const int n = static_cast<int>(depth / 8);
  int pad = (n + 3) & ~3;
  while (pad % 32 != 4) pad += 4;
  return pad;
}
__device__ __forceinline__ int norm_chain_slot(int col, int vsize, int pad) {
  return col < vsize ? (col & 7) * pad + (col >> 3) : 8 * pad + (col - vsize);
}
// WANT_SUM = false (RMSNorm: only the sum of squares enters the result) drops the second dependent
// chain of the loop - at depth 4096 the statistics are 512 dependent steps per row.
template <bool WANT_SUM = true, int kB = 8>
__device__ __forceinline__ void ordered_row_stats_t(const float* buf, int pad, int64_t depth,
                                                    float& sum, float& sumsq) {
  const int l = threadIdx.x & 7;
  const int n = static_cast<int>(depth / 8);
  float s = 0.f, ss = 0.f;
  if (n > 0) {
    const float* chain = buf + l * pad;
    const v4f* c4 = reinterpret_cast<const v4f*>(chain);
    const int n4 = n / 4;
    auto step4 = [&](const v4f v) {
      ss = fmaf(v.x, v.x, ss); if (WANT_SUM) s = s + v.x;
      ss = fmaf(v.y, v.y, ss); if (WANT_SUM) s = s + v.y;
      ss = fmaf(v.z, v.z, ss); if (WANT_SUM) s = s + v.z;
      ss = fmaf(v.w, v.w, ss); if (WANT_SUM) s = s + v.w;
    };
    // Software pipeline (round 6): the next eight 16-byte reads are requested BEFORE the 32 dependent
    // steps on the eight already here (the compiler's own unrolled loop waited for each batch right
    // after asking for it: 13 clocks per step measured against the ~5 of the dependent FMA itself).
    int i = 0;  // (kB: reads per batch - 8 * kB registers for the two batches in flight)
    if constexpr (kB == 0) {  // the compiler's own unrolling (decoder_cluster.hip: tuned to zero spills as is)
#pragma unroll 8
      for (; i < n4; ++i) step4(c4[i]);
    } else if (n4 >= kB) {
      v4f cur[kB > 0 ? kB : 1], nxt[kB > 0 ? kB : 1];
#pragma unroll
      for (int u = 0; u < kB; ++u) cur[u] = c4[u];
      i = kB;
      while (i + 2 * kB <= n4) {  // (two batches per trip: the buffers swap roles, no moves)
#pragma unroll
        for (int u = 0; u < kB; ++u) nxt[u] = c4[i + u];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < kB; ++u) step4(cur[u]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < kB; ++u) cur[u] = c4[i + kB + u];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < kB; ++u) step4(nxt[u]);
        __builtin_amdgcn_sched_barrier(0);
        i += 2 * kB;
      }
      if (i + kB <= n4) {
#pragma unroll
        for (int u = 0; u < kB; ++u) nxt[u] = c4[i + u];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < kB; ++u) step4(cur[u]);
#pragma unroll
        for (int u = 0; u < kB; ++u) step4(nxt[u]);
        i += kB;
      } else {
#pragma unroll
        for (int u = 0; u < kB; ++u) step4(cur[u]);
      }
    }
    for (; i < n4; ++i) step4(c4[i]);
    for (int i = n4 * 4; i < n; ++i) {
      const float v = chain[i];
      ss = fmaf(v, v, ss);
      s = s + v;
    }
    ss = dpp_xor4(ss) + ss;
    s = dpp_xor4(s) + s;
    ss = dpp_move<0x4E, 0xf>(0.f, ss) + ss;
    s = dpp_move<0x4E, 0xf>(0.f, s) + s;
    ss = dpp_move<0xB1, 0xf>(0.f, ss) + ss;
    s = dpp_move<0xB1, 0xf>(0.f, s) + s;
  }
  const float* tail = buf + 8 * pad;
  int t = 0;
  const int rest = static_cast<int>(depth - 8 * static_cast<int64_t>(n));
  if (rest >= 4) {
    const float x0 = tail[0], x1 = tail[1], x2 = tail[2], x3 = tail[3];
    const float q0 = x0 * x0, q1 = x1 * x1, q2 = x2 * x2, q3 = x3 * x3;
    ss = ss + ((q3 + q1) + (q2 + q0));
    s = s + ((x3 + x1) + (x2 + x0));
    t = 4;
  }
  for (; t < rest; ++t) {
    const float v = tail[t];
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
