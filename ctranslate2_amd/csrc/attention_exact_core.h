// The arithmetic core of the exact decode attention (attention_exact.hip), shared with the
// cluster-resident decoder stack (decoder_cluster.hip): MKL's one-query-row dot-product forms and
// ONE wave's pass over a (row, head) - QK^T, SoftMax, P.V - in the reference CPU path's order.
// See the header comment of attention_exact.hip for where each order comes from; everything here
// is checked bit for bit against oracle/ct2_oracle.c (orc_attention_decode).
#pragma once

#include <float.h>

#include "device_utils.h"

namespace ct2x {

__device__ __forceinline__ float tree16(float (&a)[16]) {
#pragma unroll
  for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
    for (int l = 0; l < w; ++l) a[l] = a[l] + a[l + w];
  return a[0];
}

// MKL's m = 1 dot product over k = K (K <= 64; longer rows use another tail form that was not
// recovered), "main" form (keys in the groups of four): lane accumulators over k = 1 .. body,
// lane (k - 1) % 16, accumulator 0 seeded with q0*k0; the remainder k = body + 1 .. K - 1 goes
// through a second 16-lane tree whose lane 0 fuses its product onto the first tree's sum.
template <int K>
__device__ __forceinline__ float dot_main(const float (&q)[K], const float (&k)[K]) {
  constexpr int body = ((K - 1) / 16) * 16;
  float acc[16], r[16];
  acc[0] = q[0] * k[0];
#pragma unroll
  for (int l = 1; l < 16; ++l) acc[l] = 0.f;
#pragma unroll
  for (int t = 1; t <= body; ++t) acc[(t - 1) & 15] = fmaf(q[t], k[t], acc[(t - 1) & 15]);
  const float s = tree16(acc);
#pragma unroll
  for (int l = 0; l < 16; ++l) r[l] = 0.f;
  r[0] = fmaf(q[body + 1], k[body + 1], s);
#pragma unroll
  for (int t = body + 2; t < K; ++t) r[t - body - 1] = q[t] * k[t];
  return tree16(r);
}

// ... and the "tail" forms (the n % 4 last keys): element 0 is added last as a rounded product;
// type A (first and third tail key) adds the second 16-channel chunk as a rounded product when
// the row is longer than two chunks, type B (second tail key) fuses every chunk.
template <int K>
__device__ __forceinline__ float dot_tail(const float (&q)[K], const float (&k)[K], bool type_a) {
  float acc[16];
#pragma unroll
  for (int l = 0; l < 16; ++l) acc[l] = 0.f;
#pragma unroll
  for (int t = 1; t < K; ++t) {
    const int l = (t - 1) & 15, c = (t - 1) >> 4;
    if (c == 0) {
      acc[l] = q[t] * k[t];
    } else if (c == 1 && K > 33) {
      const float prod = q[t] * k[t];
      const float fused = fmaf(q[t], k[t], acc[l]);
      acc[l] = type_a ? acc[l] + prod : fused;
    } else {
      acc[l] = fmaf(q[t], k[t], acc[l]);
    }
  }
  const float s = tree16(acc);
  const float p0 = q[0] * k[0];
  return s + p0;
}


template <int K>
__device__ __forceinline__ float dot_seq(const float (&q)[K], const float (&k)[K]) {
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < K; ++t) s = fmaf(q[t], k[t], s);
  return s;
}

// One wave, one (row, head), d_head 64, no RoPE: the arithmetic of attn_x_decode_kernel<64> for
// callers whose workgroup has already STAGED
// the first 64 cache rows in LDS (the cache of past steps does not depend on the step's
// activations, so its loads ride under the LayerNorm / projection that precede the attention):
//   qd, knew, vnew  this lane's channel of q, of the step's new key and value (has_new)
//   ks [64][68]     keys 0 .. 63 (row min(t, t_last) for t < n_keys); position t_cached is patched
//                   with knew here. vs [64][64]: values likewise (vnew taken from the register)
//   kc, vc          the cache rows in memory for the keys >= 64: key t at kc + t * kvs
//   t_cached        keys in the cache before this step (self-attention) / valid memory length
//   n_keys          n of the reference's MatMul: t_cached + 1 (self) or the PADDED memory length
//   n_valid         size of the SoftMax
//   tq [64], tp [t_pad]: this wave's LDS scratch
//   m_rows          query rows that share these keys in the reference's MatMul: 1 = the one-row
//                   forms; > 1 (beam search over the encoder memory, except at its first step) =
//                   MKL's GEMM forms (k-ordered chains, alpha on the packed keys)
//   anc             beam search: position t of this row's history lives in cache row
//                   anc[t * anc_stride] (kc / vc point at THIS row's own cache row `own_row`;
//                   rows are kv_row_stride floats apart); nullptr = the row's own
// Returns this lane's channel of the context.
__device__ __forceinline__ float attn_exact_wave64(float qd, bool has_new, float knew, float vnew,
                                                   const float* __restrict__ kc,
                                                   const float* __restrict__ vc, int kvs, int t_cached,
                                                   int n_keys, int n_valid, float scale, float* tq,
                                                   float* tp, int t_pad, float* ks, const float* vs,
                                                   int m_rows = 1, const int32_t* anc = nullptr,
                                                   int64_t anc_stride = 0, int64_t kv_row_stride = 0,
                                                   int own_row = 0, unsigned long long* phase_stamps = nullptr) {
  constexpr int DH = 64, XKS = DH + 4, R4 = DH / 4, KPI = 64 / R4, XVB = 16;
  const bool seq = m_rows > 1;
  const bool pack_b = seq && scale != 1.f && (n_keys >= 192 || (n_keys > m_rows && n_keys >= DH));
#define CT2X_CW_HOP(T) \
  (anc ? (static_cast<int64_t>(anc[static_cast<int64_t>(T) * anc_stride]) - own_row) * kv_row_stride : int64_t(0))
  const int lane = threadIdx.x & 63;
  const int cl = lane;
  const int pos = t_cached;
  const int t_last = max(t_cached - 1, 0);  // loads are clamped to rows that were written
  tq[cl] = qd;
  if (has_new && pos < 64) {
    ks[pos * XKS + cl] = knew;
    const_cast<float*>(vs)[pos * DH + cl] = vnew;  // (this lane's channel: read back by this lane only)
  }
  __builtin_amdgcn_wave_barrier();

  // ---- QK^T: lane = key
  const int main_n = (n_keys / 4) * 4;
  float s_reg = 0.f;
  float m_run = -FLT_MAX;
  const bool single = n_keys <= 64;
  for (int t0 = 0; t0 < n_keys; t0 += 64) {
    const int n_tile = min(64, n_keys - t0);
    if (t0 != 0) {
      v4f st[R4];
#pragma unroll
      for (int it = 0; it < R4; ++it) {
        st[it] = v4f{0.f, 0.f, 0.f, 0.f};
        if (it * KPI < n_tile) {
          const int i_ = it * 64 + lane;
          const int tl = min(t0 + i_ / R4, t_last);
          st[it] = *reinterpret_cast<const v4f*>(kc + CT2X_CW_HOP(tl) + tl * kvs + (i_ % R4) * 4);
        }
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < R4; ++it) {
        if (it * KPI < n_tile) {
          const int i_ = it * 64 + lane;
          *reinterpret_cast<v4f*>(ks + (i_ / R4) * XKS + (i_ % R4) * 4) = st[it];
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (has_new && pos >= t0 && pos < t0 + 64) ks[(pos - t0) * XKS + cl] = knew;
      __builtin_amdgcn_wave_barrier();
    }
    const int t = t0 + lane;
    float s = 0.f;
    // the query is the same for every lane (key): it lives in SCALAR registers - 64 vector
    // registers less in the one phase that needs a whole key row per lane as well
    float qv[DH];
    {
      const v4f* qrow = reinterpret_cast<const v4f*>(tq);
#pragma unroll
      for (int c = 0; c < R4; ++c) {
        const v4f q4 = qrow[c];
#pragma unroll
        for (int e = 0; e < 4; ++e)
          qv[4 * c + e] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(q4[e])));
      }
    }
    if (lane < n_tile) {
      float kv[DH];
      const v4f* krow = reinterpret_cast<const v4f*>(ks + lane * XKS);
#pragma unroll
      for (int c = 0; c < R4; ++c) {
        const v4f k4 = krow[c];
        kv[4 * c + 0] = k4.x; kv[4 * c + 1] = k4.y; kv[4 * c + 2] = k4.z; kv[4 * c + 3] = k4.w;
      }
      if (seq) {
        if (pack_b) {
#pragma unroll
          for (int c = 0; c < DH; ++c) kv[c] = kv[c] * scale;
        }
        s = dot_seq<DH>(qv, kv);
      } else if (t < main_n) {
        s = dot_main<DH>(qv, kv);
      } else {
        s = dot_tail<DH>(qv, kv, (t - main_n) != 1);
      }
      if (!pack_b) s = s * scale;
      if (!single) tp[t] = s;
      if (t < n_valid) m_run = fmaxf(m_run, s);
    }
    s_reg = s;
    __builtin_amdgcn_wave_barrier();
  }

  if (phase_stamps != nullptr && lane == 0) phase_stamps[0] = wall_clock64();
  // ---- SoftMax over the n_valid leading keys; masked keys get probability 0
  const float m = wave_max_dpp(m_run);
  float e_reg = 0.f;
  if (single) {
    e_reg = (lane < n_valid) ? exp_avx2(s_reg - m) : 0.f;
    for (int t = lane; t < t_pad; t += 64) tp[t] = (t == lane) ? e_reg : 0.f;
  } else {
    for (int t = lane; t < t_pad; t += 64) {
      const float sc = tp[t];
      tp[t] = (t < n_valid) ? exp_avx2(sc - m) : 0.f;
    }
  }
  __builtin_amdgcn_wave_barrier();
  const float esum = ordered_sum8_short(tp, n_valid);
  const float rcp = 1.f / esum;
  __builtin_amdgcn_wave_barrier();
  if (single) {
    if (lane < n_valid) tp[lane] = e_reg * rcp;
  } else {
    for (int t = lane; t < n_valid; t += 64) tp[t] = tp[t] * rcp;
  }
  __builtin_amdgcn_wave_barrier();

  if (phase_stamps != nullptr && lane == 0) phase_stamps[1] = wall_clock64();
  // ---- P.V: lane = channel
  float y = 0.f;
  bool has = false;
  const int n_blocks = seq ? 0 : n_keys / 8;  // (the GEMM forms add one key at a time)
#define CT2X_CW_BLOCK(VR, J0, T0) CT2X_CW_BLOCK_P(VR, J0, T0, *reinterpret_cast<const v4f*>(tp + (T0)), \
                                                  *reinterpret_cast<const v4f*>(tp + (T0) + 4))
#define CT2X_CW_BLOCK_P(VR, J0, T0, PA, PB)                                                  \
  {                                                                                          \
    const v4f pa_ = PA;                                                                      \
    const v4f pb_ = PB;                                                                      \
    if ((T0) + 8 <= n_blocks * 8) {                                                          \
      const float A_ = fmaf(pa_.x, VR[J0 + 0], pa_.z * VR[J0 + 2]);                          \
      const float B_ = fmaf(pa_.y, VR[J0 + 1], pa_.w * VR[J0 + 3]);                          \
      const float c6_ = has ? fmaf(pb_.z, VR[J0 + 6], y) : pb_.z * VR[J0 + 6];               \
      const float C_ = fmaf(pb_.x, VR[J0 + 4], c6_);                                         \
      const float D_ = fmaf(pb_.y, VR[J0 + 5], pb_.w * VR[J0 + 7]);                          \
      y = (C_ + D_) + (A_ + B_);                                                             \
      has = true;                                                                            \
    } else {                                                                                 \
      const float pj_[8] = {pa_.x, pa_.y, pa_.z, pa_.w, pb_.x, pb_.y, pb_.z, pb_.w};         \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                        \
        if ((T0) + j < n_keys) {                                                             \
          y = has ? fmaf(pj_[j], VR[J0 + j], y) : pj_[j] * VR[J0 + j];                       \
          has = true;                                                                        \
        }                                                                                    \
      }                                                                                      \
    }                                                                                        \
  }
  // keys 0 .. 63: values from the staged tile, eight at a time. ALL 64 reads leave before the first
  // block (rows past the keys are clamped to a staged row, so every read is in bounds): the chain
  // through y is 8 blocks x ~6 dependent operations, and a loop that waited for its eight LDS reads
  // at the top of every block made this phase 2.4 us of the 3.9 us attention core (round 6 stamps)
  {
    // (row t of the staged tile IS row min(t, t_last) of the cache - AttnStage64 - and the step's
    // own value was patched in next to its key above: 64 reads at constant offsets, no clamp, no
    // select - the address arithmetic and selects of the first form were most of this phase)
    float vall[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) vall[j] = vs[j * DH + cl];
    // ... and so do the probabilities (the same address for every lane: 16 broadcast reads; the
    // scratch row holds at least 32 entries, reads past it are clamped)
    v4f pall[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) pall[c] = *reinterpret_cast<const v4f*>(tp + min(c * 4, t_pad - 4));
    const int n_staged = min(n_keys, 64);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      if (b * 8 < n_staged) CT2X_CW_BLOCK_P(vall, b * 8, b * 8, pall[2 * b], pall[2 * b + 1])
    }
  }
#define CT2X_CW_LOAD_V(VR, T0)                                            \
  _Pragma("unroll") for (int j = 0; j < XVB; ++j) {                       \
    const int t_ = (T0) + j;                                              \
    const int tc_ = min(t_, t_last);                                      \
    const float g_ = vc[CT2X_CW_HOP(tc_) + tc_ * kvs + cl];               \
    VR[j] = (has_new && t_ == pos) ? vnew : g_;                           \
  }
  if (n_keys > 4 * XVB) {  // longer rows from memory: two batches in flight, alternating
    float vr[2][XVB];
    CT2X_CW_LOAD_V(vr[0], 4 * XVB)
    for (int t0 = 4 * XVB; t0 < n_keys; t0 += 2 * XVB) {
      CT2X_CW_LOAD_V(vr[1], t0 + XVB)
      CT2X_CW_BLOCK(vr[0], 0, t0)
      if (t0 + 8 < n_keys) CT2X_CW_BLOCK(vr[0], 8, t0 + 8)
      if (!(t0 + XVB < n_keys)) break;
      CT2X_CW_LOAD_V(vr[0], t0 + 2 * XVB)
      CT2X_CW_BLOCK(vr[1], 0, t0 + XVB)
      if (t0 + XVB + 8 < n_keys) CT2X_CW_BLOCK(vr[1], 8, t0 + XVB + 8)
    }
  }
#undef CT2X_CW_LOAD_V
#undef CT2X_CW_BLOCK
#undef CT2X_CW_BLOCK_P
#undef CT2X_CW_HOP
  asm volatile("" ::"v"(y));
  if (phase_stamps != nullptr && lane == 0) phase_stamps[2] = wall_clock64();
  return y;
}

// The staging that goes with it, by ALL threads of a 512-thread workgroup: rows t < min(n_keys, 64)
// of the (row, head)'s keys and values - row min(t, t_last), as the one-wave kernel clamps its
// loads - as two 16-byte pieces per thread and tensor. request() before the work that hides the
// latency, store() after it (then a barrier before the attention wave reads the tiles).
struct AttnStage64 {
  v4f k[2], v[2];
  __device__ __forceinline__ void request(const float* __restrict__ kc, const float* __restrict__ vc, int kvs,
                                          int t_cached, int n_keys, const int32_t* anc = nullptr,
                                          int64_t anc_stride = 0, int64_t kv_row_stride = 0, int own_row = 0) {
    const int t_last = max(t_cached - 1, 0);
    // (the lane's piece index behind an opaque move: the 64-bit addresses built from it are not
    // hoisted out of the caller's layer loop - hoisted addresses get spilled)
    int tid = static_cast<int>(threadIdx.x);
    asm volatile("" : "+v"(tid));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int piece = i * 512 + tid, r = piece >> 4, c4 = piece & 15;
      k[i] = v4f{0.f, 0.f, 0.f, 0.f};
      v[i] = v4f{0.f, 0.f, 0.f, 0.f};
      if (r < n_keys) {
        const int rc = min(r, t_last);
        const int64_t hop =
            anc ? (static_cast<int64_t>(anc[static_cast<int64_t>(rc) * anc_stride]) - own_row) * kv_row_stride : 0;
        k[i] = *reinterpret_cast<const v4f*>(kc + hop + static_cast<int64_t>(rc) * kvs + c4 * 4);
        v[i] = *reinterpret_cast<const v4f*>(vc + hop + static_cast<int64_t>(rc) * kvs + c4 * 4);
      }
    }
  }
  __device__ __forceinline__ void store(float* ks, float* vs) const {
    int tid = static_cast<int>(threadIdx.x);
    asm volatile("" : "+v"(tid));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int piece = i * 512 + tid, r = piece >> 4, c4 = piece & 15;
      *reinterpret_cast<v4f*>(ks + r * 68 + c4 * 4) = k[i];
      *reinterpret_cast<v4f*>(vs + r * 64 + c4 * 4) = v[i];
    }
  }
};

}  // namespace ct2x
