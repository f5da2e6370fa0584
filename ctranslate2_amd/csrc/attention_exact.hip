// Attention that reproduces the reference CPU path BIT FOR BIT (d_head = 64), for gfx950.
//
// The int8 decoder re-quantises every Dense input, so a last-ulp difference in an fp32 op that
// lands on a rounding boundary flips an int8 activation and the layers above amplify it to the
// quantisation-noise floor: "token-identical to the CPU reference" needs the fp32 ops of
// MultiHeadAttention (src/layers/attention.cc:178-287: MatMul(trans_b, alpha) -> SoftMax(lengths)
// -> MatMul) to round exactly like the reference's. On the CPU these are
//   * MatMul = MKL cblas_sgemm_batch_strided (src/cpu/primitives.cc:1085-1101). MKL's summation
//     trees were recovered on the authoring host (see oracle/ct2_oracle.c, orc_matmul):
//       - one query per (row, head) (m = 1): the QK^T dot products use 16 lane accumulators over
//         k = 1..63 with element 0 apart, in a "main" form for the key groups of four and two
//         "tail" forms for the n % 4 last keys; P.V consumes the keys in blocks of eight through
//         four 2-element FMA chains, then one FMA per remaining key (d_head >= 16; narrower
//         outputs are plain k-ordered chains);
//       - several queries (sequence mode, beam >= 3 over the encoder memory): k-ordered FMA chains;
//   * SoftMax = cpu::softmax<AVX2> (src/cpu/kernels.cc:402-459): exp256_ps with GCC's FMA
//     contraction, eight strided partial sums + reduce_m256 + scalar tail, multiply by 1 / sum.
// The lane roles follow attention_decode2.hip (QK^T: lane = key, P.V: lane = channel); what
// changes is the ORDER in which each lane adds. One wave per (row, query head); the heads of a
// row share a workgroup so that ops::Quantize of the context can stay fused.
//
// Checked bit for bit against oracle/ct2_oracle.c (orc_attention_decode), which is itself pinned
// bit for bit against the reference library (tests/test_oracle.py::test_exact_*).

#include "attention_exact_core.h"
#include "device_utils.h"
#include "kernels.h"
#include "runtime.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <float.h>

namespace ct2x {

namespace {

struct AttnXArgs {
  AttnDecodeParams p;
  int8_t* q_out;
  float* q_scales;
  int quant_mode;
  int teams;       // waves (query heads) per workgroup
  int wg_per_row;
  int group;       // query heads per KV head
  int t_pad;       // score slots per wave (multiple of 16, >= keys + 16)
  int wave_floats; // LDS floats per wave
  int seq;         // several query rows share a cache row: the MatMul has m > 1 (k-ordered forms)
};

constexpr int XVB = 16;      // V rows per batch (two blocks of eight)

// tools/attn_probe.cpp defines this to stamp the phases with s_memtime
#ifndef CT2X_ATTN_STAMP
#define CT2X_ATTN_STAMP(i)
#endif

// The sixteen leading scalar arguments are the ones the hardware preloads into SGPRs at wave launch
// (Makefile: -amdgpu-kernarg-preload-count=16); everything inside AttnXArgs costs an s_load round
// trip to kernarg memory first. They duplicate exactly what the launch's FIRST memory requests
// need - the cache length (a dependent load itself), the first K tile, q and the new key - so those
// requests leave before the argument block has arrived: three dependent round trips (arguments ->
// length -> K tile) become one.
// (14 dwords fit next to the kernarg segment pointer in the 16 user SGPRs.)
//   geom  = wg_per_row | teams << 8 | flags << 12 | group << 16 | kv_group << 24
//   shape = heads | len_imm << 8 | t_max << 9   (len_imm: the len_ptr slot carries the length itself)
//   flags: 1 = len_ptr is indexed by the cache row (cross-attention lengths), 2 = the first K tile
//   may be requested before the length is known (no ancestry table: rows past the length are
//   allocated but unwritten memory that is never used in arithmetic),
//   4 = only the first half of the tile (32 keys at d_head 64) is requested early, the rest once
//   the length is known (CT2X_ATTN_SPEC_K=half),
//   8 = q and the new key share row_stride_e (or there is no new key): they are requested early.
constexpr unsigned kXLenPerRow = 1u, kXSpecK = 2u, kXSpecHalf = 4u, kXEarlyQ = 8u;

template <int DH>
__global__ __launch_bounds__(512) void attn_x_decode_kernel(
    const float* __restrict__ k_cache_e, const int32_t* __restrict__ len_ptr_e,
    const float* __restrict__ q_e, const float* __restrict__ k_new_e, int kv_row_stride_e,
    int kv_head_stride_e, int kv_time_stride_e, int row_stride_e, unsigned geom_e,
    unsigned shape_e, AttnXArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int XKS = DH + 4;   // LDS row stride of the K staging tile (floats)
  constexpr int R4 = DH / 4;    // float4 per cache row
  constexpr int KPI = 64 / R4;  // keys per staging iteration
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const bool chan = lane < DH;       // lanes that own a channel in the prologue / P.V phases
  const int cl = chan ? lane : 0;
  // ---- requests that need nothing but the preloaded arguments --------------------------------
  const int wg_per_row_e = geom_e & 255u, teams_e = (geom_e >> 8) & 15u, group_e = (geom_e >> 16) & 255u;
  const unsigned flags_e = (geom_e >> 12) & 15u;
  const int kv_group_e = geom_e >> 24, heads_e = shape_e & 255u, t_max_e = shape_e >> 9;
  const bool len_imm_e = (shape_e >> 8) & 1u;
  const int row_e = blockIdx.x / wg_per_row_e;
  const int hq_e = (blockIdx.x % wg_per_row_e) * teams_e + wave;  // >= heads: an idle wave (clamped below)
  const int kvrow_e = row_e / kv_group_e;
  int len_e = static_cast<int>(reinterpret_cast<uintptr_t>(len_ptr_e));  // len_imm
  if (!len_imm_e) len_e = len_ptr_e[(flags_e & kXLenPerRow) ? kvrow_e : 0];
  v4f kpre[R4];
  float q_early = 0.f, knew_early = 0.f;
  {
    // (an idle wave of the last workgroup reads head 0's data and discards it, as before)
    const int hq_c = hq_e < heads_e ? hq_e : 0;
    const int hk_c = hq_c / group_e;
    // (branch-free on purpose: a load under its own branch is waited for before the next one is
    // issued; the row index is clamped to the allocation instead, t_max >= 1)
    const bool spec = (flags_e & kXSpecK) != 0;
    const float* kc_e = k_cache_e + static_cast<int64_t>(kvrow_e) * kv_row_stride_e +
                        static_cast<int64_t>(hk_c) * kv_head_stride_e;
    constexpr int HALF = R4 > 1 ? R4 / 2 : 1;
    if (spec) {
#pragma unroll
      for (int it = 0; it < HALF; ++it) {
        const int i_ = it * 64 + lane;
        const int tl = min(i_ / R4, t_max_e - 1);
        kpre[it] = *reinterpret_cast<const v4f*>(kc_e + static_cast<int64_t>(tl) * kv_time_stride_e + (i_ % R4) * 4);
      }
      if (!(flags_e & kXSpecHalf)) {
#pragma unroll
        for (int it = HALF; it < R4; ++it) {
          const int i_ = it * 64 + lane;
          const int tl = min(i_ / R4, t_max_e - 1);
          kpre[it] = *reinterpret_cast<const v4f*>(kc_e + static_cast<int64_t>(tl) * kv_time_stride_e + (i_ % R4) * 4);
        }
      }
    }
    if (flags_e & kXEarlyQ) {
      q_early = q_e[static_cast<int64_t>(row_e) * row_stride_e + hq_c * DH + cl];
      if (k_new_e != nullptr)
        knew_early = k_new_e[static_cast<int64_t>(row_e) * row_stride_e + hk_c * DH + cl];
    }
  }
  // nothing below may be scheduled above the early requests (the argument block's s_loads would
  // otherwise be hoisted to the top and waited for first)
  __builtin_amdgcn_sched_barrier(0);
  const AttnDecodeParams& p = a.p;
  const int row = blockIdx.x / a.wg_per_row;
  const int head = (blockIdx.x % a.wg_per_row) * a.teams + wave;
  const bool active = head < p.heads;
  const int hq = active ? head : 0;
  const int hk = hq / a.group;
  const bool appends = (hq % a.group) == 0;
  const int kvrow = row / static_cast<int>(p.kv_group);

  float* tq = smem + static_cast<size_t>(wave) * a.wave_floats;  // q [64]
  float* tp = tq + 64;                                           // scores / probabilities [t_pad]
  float* ks = tp + a.t_pad;                                      // K staging tile [64][XKS]
  float* team_amax = smem + static_cast<size_t>(a.teams) * a.wave_floats;

  float* kc = p.k_cache + static_cast<int64_t>(kvrow) * p.kv_row_stride +
              static_cast<int64_t>(hk) * p.kv_head_stride;
  float* vc = p.v_cache + static_cast<int64_t>(kvrow) * p.kv_row_stride +
              static_cast<int64_t>(hk) * p.kv_head_stride;
  const int kvs = static_cast<int>(p.kv_time_stride);
  // Beam search through the ancestry table: position t of this row's history is read from cache
  // row anc[t * stride + row] (element offset of that row relative to this row's own: `hop`).
  const int32_t* anc = p.ancestry ? p.ancestry + row : nullptr;
  const int64_t anc_stride = p.ancestry_stride;
#define CT2X_XHOP(T)                                                                               \
  (anc ? (static_cast<int64_t>(anc[static_cast<int64_t>(T) * anc_stride]) - kvrow) * p.kv_row_stride \
       : static_cast<int64_t>(0))

  int t_cached = len_e;  // p.lengths[kvrow], *p.step_ptr or p.length: requested / delivered at the top
  t_cached = max(0, min(t_cached, static_cast<int>(p.t_max)));
  const bool has_new = p.k_new != nullptr;
  const int pos = t_cached;
  // n of the reference's MatMul: the cache length after the Concat (self-attention), or the
  // PADDED memory length (cross-attention: masked keys still take part in the product);
  // size of its SoftMax: the valid keys.
  const int n_valid = t_cached + (has_new ? 1 : 0);
  const int n_keys = p.lengths != nullptr ? static_cast<int>(p.t_max) : n_valid;
  const int t_last = max(t_cached - 1, 0);  // loads are clamped to rows that were written
  // m of the reference's MatMul: the rows of a kv group (beam search on the encoder memory) -
  // except at the first step of a CPU beam search, which still runs one row per batch entry
  // (expand_after_first_step, src/decoding.cc:453-454, 700-712) and so takes the one-row forms.
  const bool kSeq = a.seq != 0 &&
                    !(p.one_row_step_ptr != nullptr && *p.one_row_step_ptr == p.one_row_step);

  // Where alpha enters MKL's several-row product: once B is packed (n >= 192, or n > m and
  // n >= max(k, 12)) the packed keys carry alpha (oracle/ct2_oracle.c orc_matmul, round 3).
  const bool pack_b = kSeq && p.scale != 1.f &&
                      (n_keys >= 192 || (n_keys > static_cast<int>(p.kv_group) && n_keys >= (DH > 12 ? DH : 12)));

  float acc_out = 0.f;
  CT2X_ATTN_STAMP(0)
  if (active) {
    // Every global load of the launch that does not depend on another is requested up front -
    // the first K tile, q, the new key / value, then (below) the first batch of V rows - so that
    // the prologue and both phases share ONE memory round trip.
    // (only the keys that exist are staged: an iteration covers 64 / R4 keys; the K tile's LDS
    // traffic - 32 KB per wave for 64 keys - is what bounds the QK^T phase)
    if (!(flags_e & kXSpecK) || (flags_e & kXSpecHalf)) {
      constexpr int HALF2 = R4 > 1 ? R4 / 2 : 1;
#pragma unroll
      for (int it = 0; it < R4; ++it) {
        if ((flags_e & kXSpecK) && it < HALF2) continue;  // requested at the top
        kpre[it] = v4f{0.f, 0.f, 0.f, 0.f};
        if (it * KPI < n_keys) {
          const int i_ = it * 64 + lane;  // R4 float4 per row
          const int tl = min(i_ / R4, t_last);
          kpre[it] = *reinterpret_cast<const v4f*>(kc + CT2X_XHOP(tl) + tl * kvs + (i_ % R4) * 4);
        }
      }
    }
    // ---- prologue: lane = channel ---------------------------------------------------------
    float knew = 0.f, vnew = 0.f, tq_pending = 0.f;
    {
      const float* qh = p.q + static_cast<int64_t>(row) * p.q_row_stride + hq * DH;
      float qd = (flags_e & kXEarlyQ) ? q_early : qh[cl];  // requested at the top when possible
      // RoPE at position `pos` on q and the new key (RotaryEmbeddings::apply before the cache
      // Concat, src/layers/attention.cc:521-533; ops::Rotary src/ops/rotary_cpu.cc:30-35:
      // y[i] = x[i] * cos[i] + rot(x)[i] * sin[i] with rot = -x[partner] on the first element of a
      // pair / the first half, two products and one sum - that translation unit is built without
      // FMA). The sign is folded into sin, which is exact.
      const bool rope = p.rope_sin != nullptr && cl < static_cast<int>(p.rot_dim);
      int partner = cl;
      float cs = 1.f, sn = 0.f, qo = 0.f;
      if (rope) {
        const int rd = static_cast<int>(p.rot_dim), middle = rd / 2;
        bool neg;
        if (p.rope_interleave) {
          partner = cl ^ 1;
          neg = (cl & 1) == 0;
        } else {
          partner = cl < middle ? cl + middle : cl - middle;
          neg = cl < middle;
        }
        cs = p.rope_cos[static_cast<int64_t>(pos) * rd + cl];
        const float sv = p.rope_sin[static_cast<int64_t>(pos) * rd + cl];
        sn = neg ? -sv : sv;
        qo = qh[partner];
        qd = qd * cs + qo * sn;
      }
      if (has_new) {
        const float* kn = p.k_new + static_cast<int64_t>(row) * p.new_row_stride + hk * DH;
        knew = (flags_e & kXEarlyQ) ? knew_early : kn[cl];
        if (rope) knew = knew * cs + kn[partner] * sn;
        vnew = p.v_new[static_cast<int64_t>(row) * p.new_row_stride + hk * DH + cl];
        if (appends && chan) {  // cache append in place of ops::Concat (attention.cc:541-546)
          kc[pos * kvs + cl] = knew;
          vc[pos * kvs + cl] = vnew;
        }
      }
      // (q reaches LDS only after the V request below has been issued)
      tq_pending = qd;
    }

    // The first batch of V rows is requested now, before the QK^T phase: one memory round trip
    // covers both phases (the rows are only consumed by P.V at the end).
#define CT2X_XLOAD_V(VR, T0)                                              \
  _Pragma("unroll") for (int j = 0; j < XVB; ++j) {                       \
    const int t_ = (T0) + j;                                              \
    const int tc_ = min(t_, t_last);                                      \
    const float g_ = vc[CT2X_XHOP(tc_) + tc_ * kvs + cl];                 \
    VR[j] = (has_new && t_ == pos) ? vnew : g_;                           \
  }
    // (four batches = 64 keys: a decode step of Transformer-base never waits for V twice)
    float vr[4][XVB];
#pragma unroll
    for (int b = 0; b < 4; ++b)
      if (b * XVB < n_keys) CT2X_XLOAD_V(vr[b], b * XVB)
    if (chan) tq[cl] = tq_pending;
    __builtin_amdgcn_wave_barrier();
    CT2X_ATTN_STAMP(1)

    // ---- QK^T: lane = key -------------------------------------------------------------------
    const int main_n = (n_keys / 4) * 4;
    float s_reg = 0.f;        // this lane's score of the LAST tile (the only one when n_keys <= 64)
    float m_run = -FLT_MAX;   // running maximum over the valid keys this lane has seen
    const bool single = n_keys <= 64;
    for (int t0 = 0; t0 < n_keys; t0 += 64) {
      const int n_tile = min(64, n_keys - t0);
      v4f st[R4];
      if (t0 == 0) {
#pragma unroll
        for (int it = 0; it < R4; ++it) st[it] = kpre[it];
      } else {
#pragma unroll
        for (int it = 0; it < R4; ++it) {
          st[it] = v4f{0.f, 0.f, 0.f, 0.f};
          if (it * KPI < n_tile) {
            const int i_ = it * 64 + lane;  // R4 float4 per row
            const int tl = min(t0 + i_ / R4, t_last);
            st[it] = *reinterpret_cast<const v4f*>(kc + CT2X_XHOP(tl) + tl * kvs + (i_ % R4) * 4);
          }
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
      if (has_new && chan && pos >= t0 && pos < t0 + 64) ks[(pos - t0) * XKS + cl] = knew;
      __builtin_amdgcn_wave_barrier();
      const int t = t0 + lane;
      float s = 0.f;
      if (lane < n_tile) {
        float qv[DH], kv[DH];
        const v4f* krow = reinterpret_cast<const v4f*>(ks + lane * XKS);
        const v4f* qrow = reinterpret_cast<const v4f*>(tq);
#pragma unroll
        for (int c = 0; c < R4; ++c) {
          const v4f k4 = krow[c], q4 = qrow[c];
          kv[4 * c + 0] = k4.x; kv[4 * c + 1] = k4.y; kv[4 * c + 2] = k4.z; kv[4 * c + 3] = k4.w;
          qv[4 * c + 0] = q4.x; qv[4 * c + 1] = q4.y; qv[4 * c + 2] = q4.z; qv[4 * c + 3] = q4.w;
        }
        if (kSeq) {
          if (pack_b) {
#pragma unroll
            for (int c = 0; c < DH; ++c) kv[c] = kv[c] * p.scale;
          }
          s = dot_seq<DH>(qv, kv);
        } else if (t < main_n) {
          s = dot_main<DH>(qv, kv);
        } else {
          s = dot_tail<DH>(qv, kv, (t - main_n) != 1);
        }
        if (!pack_b) s = s * p.scale;
        if (!single) tp[t] = s;
        if (t < n_valid) m_run = fmaxf(m_run, s);
      }
      s_reg = s;
      __builtin_amdgcn_wave_barrier();
    }

    CT2X_ATTN_STAMP(2)
    // ---- SoftMax over the n_valid leading keys; masked keys get probability 0 ---------------
    // (the scores of a single tile never visit LDS: maximum, exp and the scaling run on the
    // register copy; LDS sees the exps once for the ordered sum and the probabilities once for P.V)
    const float m = wave_max_dpp(m_run);
    float e_reg = 0.f;
    if (single) {
      e_reg = (lane < n_valid) ? exp_avx2(s_reg - m) : 0.f;
      for (int t = lane; t < a.t_pad; t += 64) tp[t] = (t == lane) ? e_reg : 0.f;
    } else {
      for (int t = lane; t < a.t_pad; t += 64) {
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

    CT2X_ATTN_STAMP(3)
    // ---- P.V: lane = channel --------------------------------------------------------------
    float y = 0.f;
    bool has = false;
    // narrow outputs (d_head 4 / 8) take MKL's plain k-ordered form even for one query row
    const int n_blocks = (kSeq || DH < 16) ? 0 : n_keys / 8;
#define CT2X_XBLOCK(VR, J0, T0)                                                              \
  {                                                                                          \
    const v4f pa_ = *reinterpret_cast<const v4f*>(tp + (T0));                                \
    const v4f pb_ = *reinterpret_cast<const v4f*>(tp + (T0) + 4);                            \
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
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if (b * XVB < n_keys) {
        CT2X_XBLOCK(vr[b], 0, b * XVB)
        if (b * XVB + 8 < n_keys) CT2X_XBLOCK(vr[b], 8, b * XVB + 8)
      }
    }
    if (n_keys > 4 * XVB) {  // longer rows: two batches in flight, alternating
      CT2X_XLOAD_V(vr[0], 4 * XVB)
      for (int t0 = 4 * XVB; t0 < n_keys; t0 += 2 * XVB) {
        CT2X_XLOAD_V(vr[1], t0 + XVB)
        CT2X_XBLOCK(vr[0], 0, t0)
        if (t0 + 8 < n_keys) CT2X_XBLOCK(vr[0], 8, t0 + 8)
        if (!(t0 + XVB < n_keys)) break;
        CT2X_XLOAD_V(vr[0], t0 + 2 * XVB)
        CT2X_XBLOCK(vr[1], 0, t0 + XVB)
        if (t0 + XVB + 8 < n_keys) CT2X_XBLOCK(vr[1], 8, t0 + XVB + 8)
      }
    }
#undef CT2X_XLOAD_V
#undef CT2X_XHOP
#undef CT2X_XBLOCK
    acc_out = y;
    asm volatile("" ::"v"(acc_out));
    CT2X_ATTN_STAMP(4)
  }

  if (a.q_out == nullptr) {
    if (active && chan) p.out[static_cast<int64_t>(row) * (p.heads * DH) + hq * DH + cl] = acc_out;
    return;
  }
  // ---- fused ops::Quantize of the row [heads * DH] (src/cpu/kernels.cc:577-620) -------------
  float amax = (active && chan) ? fabsf(acc_out) : 0.f;
  amax = wave_max_dpp(amax);
  if (lane == 0) team_amax[wave] = amax;
  __syncthreads();
  amax = team_amax[0];
  for (int w = 1; w < a.teams; ++w) amax = fmaxf(amax, team_amax[w]);
  const float qscale = (amax != 0.f) ? 127.f / amax : 1.f;
  if (threadIdx.x == 0) a.q_scales[row] = qscale;
  if (active && chan)
    a.q_out[static_cast<int64_t>(row) * (p.heads * DH) + hq * DH + cl] =
        to_s8(acc_out, qscale, a.quant_mode);
  CT2X_ATTN_STAMP(5)
}

// Sequence mode (encoder self-attention, prompt prefill), d_head = 64: the reference runs
// MatMul / SoftMax / MatMul over [B*H] matrices of Tq x Tk; MKL's GEMM kernels add in k order
// (fma chain), the SoftMax row is as above, and the masked / padded keys keep probability 0 in
// the second product. One workgroup per (batch, head): K and V of the head live in LDS, each
// wave takes every 4th query; lane = key for the scores, lane = channel for P.V.
struct AttnXFullArgs {
  AttnFullParams p;
  int t_pad;  // score slots per wave
};

template <int DH>
__global__ __launch_bounds__(256) void attn_x_full_kernel(AttnXFullArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const AttnFullParams& p = a.p;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y, h = blockIdx.x;
  const int hk = h / static_cast<int>(p.heads / p.heads_kv);
  const int tk = static_cast<int>(p.tk);
  const int length = p.lengths ? min(max(p.lengths[b], 0), tk) : tk;
  constexpr int KS = DH + 1;  // conflict-free row reads (lane = key)
  constexpr int R4 = DH / 4;
  const bool chan = lane < DH;
  const int cl = chan ? lane : 0;
  float* k_s = smem;                                                      // [tk][KS]
  float* v_s = k_s + ((static_cast<size_t>(tk) * KS + 3) & ~size_t(3));   // [tk][DH], 16-byte aligned
  float* wq = v_s + static_cast<size_t>(tk) * DH + static_cast<size_t>(wave) * (64 + a.t_pad);
  float* wp = wq + 64;                                                    // [t_pad]

  const float* qb = p.q + static_cast<int64_t>(b) * p.q_batch_stride + static_cast<int64_t>(h) * DH;
  const float* kb = p.k + static_cast<int64_t>(b) * p.k_batch_stride + static_cast<int64_t>(hk) * p.k_head_stride;
  const float* vb = p.v + static_cast<int64_t>(b) * p.v_batch_stride + static_cast<int64_t>(hk) * p.v_head_stride;
  for (int e = tid; e < tk * R4; e += 256) {
    const int t = e / R4, c4 = e % R4;
    const float4 k4 = *reinterpret_cast<const float4*>(kb + static_cast<int64_t>(t) * p.k_time_stride + c4 * 4);
    const float4 v4 = *reinterpret_cast<const float4*>(vb + static_cast<int64_t>(t) * p.v_time_stride + c4 * 4);
    float* kd = k_s + t * KS + c4 * 4;
    kd[0] = k4.x; kd[1] = k4.y; kd[2] = k4.z; kd[3] = k4.w;
    *reinterpret_cast<float4*>(v_s + t * DH + c4 * 4) = v4;
  }
  __syncthreads();

  const int64_t ots = p.out_time_stride ? p.out_time_stride : p.heads * DH;
  const int64_t obs = p.out_batch_stride ? p.out_batch_stride : p.tq * ots;
  for (int qi = wave; qi < p.tq; qi += 4) {
    if (chan) wq[cl] = qb[static_cast<int64_t>(qi) * p.q_time_stride + cl];
    __builtin_amdgcn_wave_barrier();
    // valid keys of this query (prepare_length_mask, src/layers/attention_layer.cc:152-174)
    int size = length;
    if (p.causal) size = min(size, qi + static_cast<int>(p.q_offset) + 1);
    float m = -FLT_MAX;
    for (int t0 = 0; t0 < tk; t0 += 64) {
      const int t = t0 + lane;
      if (t < tk) {
        const float* kr = k_s + t * KS;
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < DH; ++c) s = fmaf(wq[c], kr[c], s);
        s = s * p.scale;
        wp[t] = s;
        if (t < size) m = fmaxf(m, s);
      }
    }
    m = wave_max_dpp(m);
    __builtin_amdgcn_wave_barrier();
    for (int t = lane; t < a.t_pad; t += 64) {
      const float sc = wp[t];
      wp[t] = (t < size) ? exp_avx2(sc - m) : 0.f;
    }
    __builtin_amdgcn_wave_barrier();
    float y = 0.f;
    if (size > 0) {
      const float rcp = 1.f / ordered_sum8(wp, size);
      __builtin_amdgcn_wave_barrier();
      for (int t = lane; t < size; t += 64) wp[t] = wp[t] * rcp;
      __builtin_amdgcn_wave_barrier();
      for (int t = 0; t < tk; ++t) y = fmaf(wp[t], v_s[t * DH + cl], y);
    }
    if (chan) p.out[static_cast<int64_t>(b) * obs + static_cast<int64_t>(qi) * ots + h * DH + cl] = y;
    __builtin_amdgcn_wave_barrier();
  }
}

bool exact_disabled() {
  static const bool off = [] {
    const char* e = std::getenv("CT2X_FAST_ATTENTION");
    return e != nullptr && e[0] == '1';
  }();
  return off;
}

void ensure_lds(const void* fn, size_t lds) {
  if (lds <= 64 * 1024) return;
  // > 64 KiB of dynamic LDS has to be opted into (once per function; cheap to repeat)
  CT2X_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
}

int x_t_pad(const AttnDecodeParams& p) {
  return static_cast<int>(((p.t_max + 1 + 15) / 16) * 16 + 16);
}

}  // namespace

// Decode shapes the exact kernel covers: d_head 4 .. 64 (a multiple of 4; MKL's one-row tail
// form for longer rows was not recovered), RoPE on self-attention (new key present), aligned
// caches, and the per-wave LDS (q + scores + one K tile) of `teams` heads within 160 KiB.
static bool x_dhead_ok(int64_t dh) { return dh == 4 || dh == 8 || dh == 16 || dh == 32 || dh == 64; }

bool attention_exact_decode_applicable(const AttnDecodeParams& p, bool fused_quantize) {
  if (exact_disabled() || p.allow_inexact) return false;
  if (p.dhead == 128) return !fused_quantize && attention_exact128_decode_applicable(p);
  if (p.ancestry != nullptr && (p.kv_group != 1 || p.lengths != nullptr)) return false;
  if (!x_dhead_ok(p.dhead) || p.heads < 1 || p.heads_kv < 1 || p.heads % p.heads_kv != 0) return false;
  if (p.rope_sin != nullptr && (p.rope_cos == nullptr || p.rot_dim < 2 || p.rot_dim > p.dhead ||
                                p.rot_dim % 2 != 0 || p.k_new == nullptr))
    return false;
  if (p.kv_time_stride < p.dhead || p.kv_time_stride % 4 != 0 || p.t_max < 1) return false;
  if ((reinterpret_cast<uintptr_t>(p.k_cache) & 15) || (reinterpret_cast<uintptr_t>(p.v_cache) & 15) ||
      p.kv_head_stride % 4 != 0 || p.kv_row_stride % 4 != 0)
    return false;
  const size_t wave_floats = 64 + x_t_pad(p) + 64 * (p.dhead + 4);
  if (wave_floats * 4 + 64 > 160 * 1024) return false;
  if (fused_quantize)
    return p.heads <= 8 && (static_cast<size_t>(p.heads) * wave_floats + 16) * 4 <= 160 * 1024;
  return true;
}

template <int DH>
static void launch_x(const AttnXArgs& a, size_t lds, hipStream_t stream) {
  const unsigned grid = static_cast<unsigned>(a.p.rows * a.wg_per_row);
  ensure_lds(reinterpret_cast<const void*>(attn_x_decode_kernel<DH>), lds);
  const AttnDecodeParams& p = a.p;
  const int32_t* len_ptr = p.lengths != nullptr ? p.lengths : p.step_ptr;
  const int64_t big = int64_t(1) << 31;
  unsigned flags = 0;
  const bool fits = p.kv_row_stride < big && p.kv_head_stride < big && p.kv_time_stride < big &&
                    a.wg_per_row <= 255 && a.teams <= 15 && a.group <= 255 && p.kv_group <= 255 &&
                    p.heads <= 255 && p.t_max < (int64_t(1) << 23) && p.length >= 0 && p.length < big;
  if (!fits)
    throw Error("attention: geometry outside the exact decode kernel's packed arguments");
  const bool len_imm = len_ptr == nullptr;
  if (len_imm) len_ptr = reinterpret_cast<const int32_t*>(static_cast<uintptr_t>(p.length));
  if (p.lengths != nullptr) flags |= kXLenPerRow;
  // CT2X_ATTN_SPEC_K = 0 (no early K request) | half | 1 (whole first tile; default)
  static const int spec_mode = [] {
    const char* e = tuning_env("CT2X_ATTN_SPEC_K");
    if (e == nullptr) return 2;
    if (e[0] == '0') return 0;
    if (e[0] == 'h') return 1;
    return 2;
  }();
  if (p.ancestry == nullptr && spec_mode > 0) flags |= kXSpecK;
  if (spec_mode == 1) flags |= kXSpecHalf;
  if (p.q_row_stride < big && (p.k_new == nullptr || p.new_row_stride == p.q_row_stride))
    flags |= kXEarlyQ;
  const unsigned geom = static_cast<unsigned>(a.wg_per_row) | static_cast<unsigned>(a.teams) << 8 |
                        flags << 12 | static_cast<unsigned>(a.group) << 16 |
                        static_cast<unsigned>(p.kv_group) << 24;
  const unsigned shape = static_cast<unsigned>(p.heads) | (len_imm ? 256u : 0u) |
                         static_cast<unsigned>(p.t_max) << 9;
  hipLaunchKernelGGL((attn_x_decode_kernel<DH>), dim3(grid), dim3(a.teams * 64), lds, stream,
                     static_cast<const float*>(p.k_cache), len_ptr, p.q, p.k_new,
                     static_cast<int>(p.kv_row_stride), static_cast<int>(p.kv_head_stride),
                     static_cast<int>(p.kv_time_stride), static_cast<int>(p.q_row_stride), geom,
                     shape, a);
}

void launch_attention_exact_decode(const AttnDecodeParams& p, int quant_mode, int8_t* q_out,
                                   float* q_scales, hipStream_t stream) {
  if (p.dhead == 128) {
    CT2X_CHECK(q_out == nullptr, "attention: the d_head 128 exact kernel has no fused Quantize");
    launch_attention_exact128_decode(p, stream);
    return;
  }
  AttnXArgs a;
  a.p = p;
  a.q_out = q_out;
  a.q_scales = q_scales;
  a.quant_mode = quant_mode;
  a.group = static_cast<int>(p.heads / p.heads_kv);
  a.t_pad = x_t_pad(p);
  a.wave_floats = 64 + a.t_pad + 64 * (static_cast<int>(p.dhead) + 4);
  const int fit = static_cast<int>((160 * 1024 - 64) / (static_cast<size_t>(a.wave_floats) * 4));
  a.teams = static_cast<int>(p.heads);
  if (q_out == nullptr && a.teams > fit) a.teams = fit;
  if (a.teams > 8) a.teams = 8;  // 2 waves per SIMD: the dot-product forms keep ~170 VGPRs live
  a.wg_per_row = static_cast<int>((p.heads + a.teams - 1) / a.teams);
  const size_t lds = (static_cast<size_t>(a.teams) * a.wave_floats + 16) * sizeof(float);
  // Several query rows on one cache row = the m dimension of the reference's MatMul: MKL's GEMM
  // path (k-ordered) instead of its one-row path.
  a.seq = p.kv_group > 1 ? 1 : 0;
  switch (p.dhead) {
    case 4: launch_x<4>(a, lds, stream); break;
    case 8: launch_x<8>(a, lds, stream); break;
    case 16: launch_x<16>(a, lds, stream); break;
    case 32: launch_x<32>(a, lds, stream); break;
    default: launch_x<64>(a, lds, stream); break;
  }
}

// ---------------------------------------------------------------------------------------------
// ops::MatMul standalone (the unfused front-end of the Level-1 integration calls MatMul, SoftMax,
// MatMul as three ops): C[b] = alpha * A[b] . op(B[b]) with MKL's summation order per output
// element - the same forms as above, one thread per output (a parity path, not a fast one):
//   trans_b, m == 1, k in {4, 8, .. 64}:   main / tail dot forms by column
//   no transpose, m == 1, n >= 16:         blocks of eight keys, then one FMA per remaining key
//   everything else in the envelope:       k-ordered FMA chain
// ---------------------------------------------------------------------------------------------
namespace {

template <int K>
__device__ __forceinline__ float matmul_x_qk(const float* q, const float* kr, bool is_main, bool type_a) {
  float qv[K], kv[K];
#pragma unroll
  for (int c = 0; c < K; ++c) {
    qv[c] = q[c];
    kv[c] = kr[c];
  }
  return is_main ? dot_main<K>(qv, kv) : dot_tail<K>(qv, kv, type_a);
}

__device__ __forceinline__ float matmul_x_qk_any(int k, const float* q, const float* kr, bool is_main,
                                                 bool type_a) {
  switch (k) {
    case 4: return matmul_x_qk<4>(q, kr, is_main, type_a);
    case 8: return matmul_x_qk<8>(q, kr, is_main, type_a);
    case 12: return matmul_x_qk<12>(q, kr, is_main, type_a);
    case 16: return matmul_x_qk<16>(q, kr, is_main, type_a);
    case 20: return matmul_x_qk<20>(q, kr, is_main, type_a);
    case 24: return matmul_x_qk<24>(q, kr, is_main, type_a);
    case 28: return matmul_x_qk<28>(q, kr, is_main, type_a);
    case 32: return matmul_x_qk<32>(q, kr, is_main, type_a);
    case 36: return matmul_x_qk<36>(q, kr, is_main, type_a);
    case 40: return matmul_x_qk<40>(q, kr, is_main, type_a);
    case 44: return matmul_x_qk<44>(q, kr, is_main, type_a);
    case 48: return matmul_x_qk<48>(q, kr, is_main, type_a);
    case 52: return matmul_x_qk<52>(q, kr, is_main, type_a);
    case 56: return matmul_x_qk<56>(q, kr, is_main, type_a);
    case 60: return matmul_x_qk<60>(q, kr, is_main, type_a);
    default: return matmul_x_qk<64>(q, kr, is_main, type_a);
  }
}

__global__ void matmul_x_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                int64_t batch, int m, int n, int k, bool trans_b, float alpha,
                                float* __restrict__ c) {
  const int64_t total = batch * m * n;
  for (int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; e < total;
       e += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t bi = e / (static_cast<int64_t>(m) * n);
    const int i = static_cast<int>((e / n) % m), j = static_cast<int>(e % n);
    const float* A = a + bi * m * k + static_cast<int64_t>(i) * k;
    const float* B = b + bi * static_cast<int64_t>(n) * k;
    float s;
    if (trans_b && m == 1) {
      const int main_n = (n / 4) * 4;
      s = matmul_x_qk_any(k, A, B + static_cast<int64_t>(j) * k, j < main_n, (j - main_n) != 1);
    } else if (!trans_b && m == 1 && n >= 16) {
      float y = 0.f;
      bool has = false;
      const int nb = k / 8;
      for (int blk = 0; blk < nb; ++blk) {
        const float* pp = A + 8 * blk;
        const float* vv = B + static_cast<int64_t>(8 * blk) * n + j;
        const float A_ = fmaf(pp[0], vv[0], pp[2] * vv[2 * n]);
        const float B_ = fmaf(pp[1], vv[n], pp[3] * vv[3 * n]);
        const float c6 = has ? fmaf(pp[6], vv[6 * n], y) : pp[6] * vv[6 * n];
        const float C_ = fmaf(pp[4], vv[4 * n], c6);
        const float D_ = fmaf(pp[5], vv[5 * n], pp[7] * vv[7 * n]);
        y = (C_ + D_) + (A_ + B_);
        has = true;
      }
      for (int t = 8 * nb; t < k; ++t) {
        const float bv = B[static_cast<int64_t>(t) * n + j];
        y = has ? fmaf(A[t], bv, y) : A[t] * bv;
        has = true;
      }
      s = y;
    } else if (trans_b) {
      // several rows: k-ordered chain; alpha on the sum, or carried by MKL's packed copy of B
      const bool pack_b = alpha != 1.f && (n >= 192 || (n > m && n >= (k > 12 ? k : 12)));
      s = 0.f;
      const float* Bj = B + static_cast<int64_t>(j) * k;
      if (pack_b) {
        for (int t = 0; t < k; ++t) s = fmaf(A[t], Bj[t] * alpha, s);
        c[e] = s;
        continue;
      }
      for (int t = 0; t < k; ++t) s = fmaf(A[t], Bj[t], s);
    } else {
      // k-ordered chains over MKL's partition of a long k (384-blocks while > 768 remain, then
      // ceil / floor halves), blocks added in order
      int t0 = 0, rem = k;
      bool first = true;
      s = 0.f;
      while (rem > 0) {
        const int blk = rem > 768 ? 384 : (rem > 384 ? (rem + 1) / 2 : rem);
        float acc = 0.f;
        for (int t = t0; t < t0 + blk; ++t) acc = fmaf(A[t], B[static_cast<int64_t>(t) * n + j], acc);
        s = first ? acc : s + acc;
        first = false;
        t0 += blk;
        rem -= blk;
      }
    }
    c[e] = (alpha != 1.f) ? s * alpha : s;
  }
}

}  // namespace

// The envelope in which MKL's order is known (oracle/ct2_oracle.c orc_matmul).
bool matmul_exact_applicable(int64_t m, int64_t n, int64_t k, bool trans_b) {
  if (exact_disabled()) return false;
  if (m == 1 && trans_b) return k >= 4 && k <= 64 && k % 4 == 0;
  if (m == 1 && !trans_b) return true;
  if (trans_b) return m >= 3 && n >= 5;
  return true;
}

void launch_matmul_exact(const float* a, const float* b, int64_t batch, int64_t m, int64_t n,
                         int64_t k, bool trans_b, float alpha, float* c, hipStream_t stream) {
  const int64_t total = batch * m * n;
  if (total <= 0) return;
  const unsigned grid = static_cast<unsigned>(std::min<int64_t>((total + 127) / 128, 65535));
  hipLaunchKernelGGL(matmul_x_kernel, dim3(grid), dim3(128), 0, stream, a, b, batch,
                     static_cast<int>(m), static_cast<int>(n), static_cast<int>(k), trans_b, alpha, c);
}

bool attention_exact_full_applicable(const AttnFullParams& p) {
  if (exact_disabled()) return false;
  // (this kernel scales the finished sum and adds the keys in one chain: shapes where MKL packs
  // B or cuts k go to attention_seq_exact.hip)
  if (mkl_sequence_packs_b(p.tq, p.tk, p.dhead, p.scale) || p.tk > 384) return false;
  if (!x_dhead_ok(p.dhead) || p.heads_kv < 1 || p.heads % p.heads_kv != 0 || p.tk < 1) return false;
  auto aligned16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if (!(aligned16(p.k) && aligned16(p.v) && p.k_time_stride % 4 == 0 && p.v_time_stride % 4 == 0 &&
        p.k_head_stride % 4 == 0 && p.v_head_stride % 4 == 0 && p.k_batch_stride % 4 == 0 &&
        p.v_batch_stride % 4 == 0))
    return false;
  const size_t t_pad = ((p.tk + 63) / 64) * 64;
  const size_t floats = static_cast<size_t>(p.tk) * (2 * p.dhead + 1) + 4 + 4 * (64 + t_pad);
  return floats * 4 <= 160 * 1024;  // K and V of one head in LDS (tk <= ~300 at d_head 64)
}

template <int DH>
static void launch_xfull(const AttnXFullArgs& a, size_t lds, hipStream_t stream) {
  ensure_lds(reinterpret_cast<const void*>(attn_x_full_kernel<DH>), lds);
  hipLaunchKernelGGL(attn_x_full_kernel<DH>,
                     dim3(static_cast<unsigned>(a.p.heads), static_cast<unsigned>(a.p.batch)),
                     dim3(256), lds, stream, a);
}

void launch_attention_exact_full(const AttnFullParams& p, hipStream_t stream) {
  AttnXFullArgs a;
  a.p = p;
  a.t_pad = static_cast<int>(((p.tk + 63) / 64) * 64);
  const size_t lds =
      (static_cast<size_t>(p.tk) * (2 * p.dhead + 1) + 4 + 4 * (64 + a.t_pad)) * sizeof(float);
  switch (p.dhead) {
    case 4: launch_xfull<4>(a, lds, stream); break;
    case 8: launch_xfull<8>(a, lds, stream); break;
    case 16: launch_xfull<16>(a, lds, stream); break;
    case 32: launch_xfull<32>(a, lds, stream); break;
    default: launch_xfull<64>(a, lds, stream); break;
  }
}

}  // namespace ct2x
