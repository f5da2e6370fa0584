// Single-query attention over the pre-allocated KV cache, second generation (decode steps of
// MultiHeadAttention::operator(), src/layers/attention.cc:442-615, for dhead 64 / 128).
//
// attention.hip spreads one key over dhead/4 lanes and folds every dot product with cross-lane
// shuffles: at decode sizes that kernel is bound by shuffle latency (5 dependent ds_bpermute per
// key pair), not by memory. Here the lanes change role between the two GEMV-shaped phases so that
// neither needs a cross-lane reduction per key:
//   * QK^T: lane = key. A lane walks its own cache row with 16-byte loads and keeps the dot
//     products of ALL query heads that share this KV head (GQA group, no ops::Tile replication,
//     attention.cc:291-295) in registers; q comes from LDS as a broadcast ds_read_b128.
//   * softmax: two wave reductions per query head (max, sum), exp(x - max) * (1 / sum) as in
//     src/cpu/kernels.cc:452-455.
//   * PV: lane = channel. V rows are read as coalesced 256-byte segments, probabilities come from
//     LDS as broadcasts, one accumulator per (query head, channel).
// A "team" of TW waves owns one (row, KV head) and splits its keys in chunks of 64. Two launch
// shapes:
//   row mode   - one workgroup per row, one single-wave team per KV head: the whole context row is
//                in the workgroup, so ops::Quantize for the following int8 Dense is fused
//                (Transformer-base: 8 heads, <= 64 keys);
//   split mode - one workgroup per (row, KV head) with TW = 4 waves splitting the keys (long
//                caches, few rows: Llama-3-8B, 16 rows x 8 KV heads x 256 keys).
// The new key/value of the step are appended to the cache by the team and enter the softmax from
// registers (never read back from memory inside the launch).

#include "device_utils.h"
#include "kernels.h"
#include "runtime.h"

#include <atomic>

#include <cstdlib>
#include <float.h>

namespace ct2x {

long long* g_attn_trace = nullptr;

namespace {

// Native vector type for register arrays that live across program regions (arrays of HIP's
// float4 struct are not promoted out of scratch memory there).
typedef float v4f __attribute__((ext_vector_type(4)));

struct Attn2Args {
  AttnDecodeParams p;
  int8_t* q_out;
  float* q_scales;
  int quant_mode;
  int teams;        // teams per workgroup
  int wg_per_row;   // workgroups per row
  int per_q_head;   // 1: a team is ONE query head (GQ = 1) reading the cache of head / group
  int group;        // query heads per KV head
  int t_pad;        // score slots per query head (multiple of 64, > t_max)
  long long* trace; // debugging: per-phase s_memtime stamps of workgroup 0 (nullptr = off)
  int xcd_remap;    // 1: logical workgroup id = (id % 8) * (grid / 8) + id / 8 (grid % 8 == 0)
};

#define CT2X_STAMP(K)                                                                   \
  if (a.trace != nullptr && blockIdx.x == 0 && lane == 0) a.trace[wave * 16 + (K)] = clock64();

template <int DH, int GQ, int TW, bool ANC = false>
__global__ __launch_bounds__(TW == 1 ? 512 : TW * 64) void attn2_kernel(Attn2Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int E = DH / 64;  // channels per lane in the PV phase
  const AttnDecodeParams& p = a.p;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int team = wave / TW, tw = wave % TW;
  // XCD-aware placement: consecutive workgroup ids are dealt round-robin to the eight XCDs, each
  // with its own L2. In split mode the query heads of a GQA group (and the heads of a row) are
  // consecutive ids - i.e. on eight DIFFERENT L2s, every one of which fetches the group's K/V from
  // memory again. Remapped so that a run of gridDim / 8 logical ids shares an XCD, the group's
  // K/V is fetched once per L2 (Llama-3-8B shape: 4 query heads per KV head).
  unsigned bid = blockIdx.x;
  if (a.xcd_remap) {
    const unsigned per_xcd = gridDim.x >> 3;
    bid = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  }
  const int row = static_cast<int>(bid) / a.wg_per_row;
  // A team is either a KV head with its GQ query heads, or (per_q_head) one query head of a GQA
  // group: at decode sizes the chip is starved for waves, so re-reading K/V from L2 once per query
  // head beats sharing the loads inside one wave.
  const int unit = (static_cast<int>(bid) % a.wg_per_row) * a.teams + team;
  const int hk = a.per_q_head ? unit / a.group : unit;   // cache head
  const int hq0 = a.per_q_head ? unit : unit * GQ;       // first query head of the team
  const bool appends = a.per_q_head ? (unit % a.group == 0) : true;
  const int kvrow = row / static_cast<int>(p.kv_group);

  // LDS per team: q [GQ][DH] | probabilities [GQ][t_pad] | stats [TW][GQ][2] | partial out
  // [TW][GQ][DH] (TW > 1 only); then one float per team for the fused quantisation.
  const int team_floats =
      (GQ * DH + GQ * a.t_pad + TW * GQ * 2 + (TW > 1 ? TW * GQ * DH : 0) + 3) & ~3;
  float* tq = smem + static_cast<size_t>(team) * team_floats;
  float* tp = tq + GQ * DH;
  float* tstat = tp + GQ * a.t_pad;
  float* tpart = tstat + TW * GQ * 2;
  float* team_amax = smem + static_cast<size_t>(a.teams) * team_floats;
  // K staging tile of this wave: 64 keys x DH floats, rows padded by 4 floats (16-byte aligned,
  // and a 16-lane ds_read_b128 group then covers 16 distinct 4-bank slots).
  float* ks = smem + static_cast<size_t>(a.teams) * team_floats + ((a.teams + 3) & ~3) +
              static_cast<size_t>(wave) * (64 * 68);

  float* kc = p.k_cache + static_cast<int64_t>(kvrow) * p.kv_row_stride +
              static_cast<int64_t>(hk) * p.kv_head_stride;
  float* vc = p.v_cache + static_cast<int64_t>(kvrow) * p.kv_row_stride +
              static_cast<int64_t>(hk) * p.kv_head_stride;
  // Beam search without moving the caches (AttnDecodeParams::ancestry, 16-bit compute types):
  // position t of this row's history lives in cache row anc[t * anc_stride]; the step's new key /
  // value go to the row's own cache row. One dependent load in front of every batch of K / V
  // requests - against a cache_gather launch that rewrites every live position of every layer
  // (NLLB-3.3B shape, 256 rows: 1.03 ms of an 8.4 ms step).
  // (ANC is a template parameter: the instantiations without the table are the round-2 kernels,
  // instruction for instruction)
  const int32_t* anc = ANC ? p.ancestry + row : nullptr;
  const int64_t anc_stride = p.ancestry_stride;
  const float* kc0 = p.k_cache + static_cast<int64_t>(hk) * p.kv_head_stride;
  const float* vc0 = p.v_cache + static_cast<int64_t>(hk) * p.kv_head_stride;

  // As soon as the cache length is known, and before q and the new key are processed, the wave
  // requests the first 16 float4 per lane of its first K chunk and the first VB rows of its V
  // chunk: one memory round trip covers the prologue and the first batch of both phases.
  constexpr int VB = 16;       // V rows per batch
  constexpr int KS = 64 + 4;   // LDS row stride of the K staging tile: 64 channels + pad (floats)
  // (Macros, not lambdas: register arrays passed by reference through a closure end up in
  // scratch memory with a full vmcnt(0) wait per load.)
  // channel handled by this lane in the PV phase: one float (DH = 64) or an adjacent pair read as
  // one 8-byte load (DH = 128)
#define CT2X_PVD(e) (E == 1 ? lane : 2 * lane + (e))
  const int kvs = static_cast<int>(p.kv_time_stride);  // a head's cache spans < 2^31 floats
#define CT2X_STAGE_LOAD(ST, T0, COL0, TLIM)                                     \
  if constexpr (!ANC) {                                                         \
    _Pragma("unroll") for (int it = 0; it < 16; ++it) {                         \
      const int i_ = it * 64 + lane; /* 16 float4 per half row */               \
      const int tl_ = min((T0) + i_ / 16, (TLIM));                              \
      ST[it] = *reinterpret_cast<const v4f*>(kc + tl_ * kvs + (COL0) + (i_ % 16) * 4); \
    }                                                                           \
  } else {                                                                      \
    int ar_[16];                                                                \
    _Pragma("unroll") for (int it = 0; it < 16; ++it) {                         \
      const int tl_ = min((T0) + (it * 64 + lane) / 16, (TLIM));                \
      ar_[it] = anc[tl_ * anc_stride];                                          \
    }                                                                           \
    _Pragma("unroll") for (int it = 0; it < 16; ++it) {                         \
      const int i_ = it * 64 + lane;                                            \
      const int tl_ = min((T0) + i_ / 16, (TLIM));                              \
      ST[it] = *reinterpret_cast<const v4f*>(kc0 + static_cast<int64_t>(ar_[it]) * p.kv_row_stride + \
                                             tl_ * kvs + (COL0) + (i_ % 16) * 4); \
    }                                                                           \
  }
#define CT2X_STAGE_STORE(ST)                                                    \
  _Pragma("unroll") for (int it = 0; it < 16; ++it) {                           \
    const int i_ = it * 64 + lane;                                              \
    *reinterpret_cast<v4f*>(ks + (i_ / 16) * KS + (i_ % 16) * 4) = ST[it];      \
  }
  // wave-local key index u -> cache position (this wave owns chunks tw, tw + TW, ...)
#define CT2X_T_OF(U) (((((U) >> 6) * TW + tw) << 6) + ((U)&63))
#define CT2X_LOAD_V(VR, U0, TLIM)                                                              \
  if constexpr (!ANC) {                                                                        \
    _Pragma("unroll") for (int j = 0; j < VB; ++j) {                                           \
      const int tl_ = min(CT2X_T_OF((U0) + j), (TLIM));                                        \
      if (E == 1) {                                                                            \
        VR[j][0] = vc[tl_ * kvs + lane];                                                       \
      } else {                                                                                 \
        const float2 v2_ = *reinterpret_cast<const float2*>(vc + tl_ * kvs + 2 * lane);        \
        VR[j][0] = v2_.x;                                                                      \
        VR[j][E - 1] = v2_.y;                                                                  \
      }                                                                                        \
    }                                                                                          \
  } else {                                                                                     \
    int vr_[VB];                                                                               \
    _Pragma("unroll") for (int j = 0; j < VB; ++j)                                             \
      vr_[j] = anc[min(CT2X_T_OF((U0) + j), (TLIM)) * anc_stride];                             \
    _Pragma("unroll") for (int j = 0; j < VB; ++j) {                                           \
      const int tl_ = min(CT2X_T_OF((U0) + j), (TLIM));                                        \
      const float* vrow_ = vc0 + static_cast<int64_t>(vr_[j]) * p.kv_row_stride + tl_ * kvs;   \
      if (E == 1) {                                                                            \
        VR[j][0] = vrow_[lane];                                                                \
      } else {                                                                                 \
        const float2 v2_ = *reinterpret_cast<const float2*>(vrow_ + 2 * lane);                 \
        VR[j][0] = v2_.x;                                                                      \
        VR[j][E - 1] = v2_.y;                                                                  \
      }                                                                                        \
    }                                                                                          \
  }
  v4f kpre[16];
  float v_a[VB][E], v_b[VB][E];

  CT2X_STAMP(0)
  int t_cached;
  if (p.lengths != nullptr)
    t_cached = p.lengths[kvrow];
  else if (p.step_ptr != nullptr)
    t_cached = *p.step_ptr;
  else
    t_cached = static_cast<int>(p.length);
  t_cached = max(0, min(t_cached, static_cast<int>(p.t_max)));
  const int pos = t_cached;
  const bool has_new = p.k_new != nullptr;
  const int t_total = t_cached + (has_new ? 1 : 0);
  // Rows are clamped to the VALID range: lines past the length were never touched and would be
  // HBM misses that the whole wave waits for (measured: +2 us per launch).
  CT2X_STAGE_LOAD(kpre, tw * 64, 0, max(t_cached - 1, 0))
  CT2X_LOAD_V(v_a, 0, max(t_cached - 1, 0))

  const float* sin_t = p.rope_sin ? p.rope_sin + static_cast<int64_t>(pos) * p.rot_dim : nullptr;
  const float* cos_t = p.rope_cos ? p.rope_cos + static_cast<int64_t>(pos) * p.rot_dim : nullptr;
  const int rot_dim = static_cast<int>(p.rot_dim);

  CT2X_STAMP(1)
  // ---- prologue (wave 0 of the team): q -> LDS, new key/value -> cache, score of the new key --
  float vnew[E];
  float s_new[GQ];
#pragma unroll
  for (int e = 0; e < E; ++e) vnew[e] = 0.f;
#pragma unroll
  for (int g = 0; g < GQ; ++g) s_new[g] = 0.f;
  if (tw == 0) {
    // Every load of the prologue is issued before the first use (one memory round trip): the
    // element, its RoPE partner (reference formula src/ops/rotary_cpu.cc:31-36:
    // y[i] = x[i] * cos[i] + rot(x)[i] * sin[i], rot = -x[partner] for the first element of a
    // pair / the first half, +x[partner] otherwise; the sign is folded into sin, which is exact).
    const bool rope = sin_t != nullptr;
    const int middle = rot_dim / 2;
    int pd[E];
    bool in_rot[E];
    float cs[E], sn[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int d = lane + e * 64;
      in_rot[e] = rope && d < rot_dim;
      bool neg;
      if (p.rope_interleave) {
        pd[e] = d ^ 1;
        neg = (d & 1) == 0;
      } else {
        pd[e] = d < middle ? d + middle : d - middle;
        neg = d < middle;
      }
      if (!in_rot[e]) pd[e] = d;
      cs[e] = 1.f;
      sn[e] = 0.f;
      if (in_rot[e]) {
        cs[e] = cos_t[d];
        const float sv = sin_t[d];
        sn[e] = neg ? -sv : sv;
      }
    }
    float kx[E], ko[E], qx[GQ][E], qo[GQ][E];
#pragma unroll
    for (int e = 0; e < E; ++e) kx[e] = ko[e] = 0.f;
    if (has_new) {
      const float* kn = p.k_new + static_cast<int64_t>(row) * p.new_row_stride + hk * DH;
      const float* vn = p.v_new + static_cast<int64_t>(row) * p.new_row_stride + hk * DH;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        kx[e] = kn[lane + e * 64];
        ko[e] = kn[pd[e]];
        vnew[e] = vn[CT2X_PVD(e)];
      }
    }
#pragma unroll
    for (int g = 0; g < GQ; ++g) {
      const float* qh = p.q + static_cast<int64_t>(row) * p.q_row_stride + (hq0 + g) * DH;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        qx[g][e] = qh[lane + e * 64];
        qo[g][e] = qh[pd[e]];
      }
    }
    float knew[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      knew[e] = in_rot[e] ? kx[e] * cs[e] + ko[e] * sn[e] : kx[e];
      // Cache append (in place of ops::Concat, attention.cc:541-546).
      if (has_new && appends) {
        kc[pos * kvs + lane + e * 64] = knew[e];
        vc[pos * kvs + CT2X_PVD(e)] = vnew[e];
      }
    }
#pragma unroll
    for (int g = 0; g < GQ; ++g) {
      float part = 0.f;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const float qd = in_rot[e] ? qx[g][e] * cs[e] + qo[g][e] * sn[e] : qx[g][e];
        tq[g * DH + lane + e * 64] = qd;
        part += qd * knew[e];
      }
      s_new[g] = wave_sum_dpp(part) * p.scale;
    }
  }
  CT2X_STAMP(2)
  __syncthreads();
  CT2X_STAMP(3)

  // ---- QK^T: lane = key ---------------------------------------------------------------------
  float m_run[GQ];
#pragma unroll
  for (int g = 0; g < GQ; ++g) m_run[g] = -FLT_MAX;
  // The chunk's K rows go through LDS in halves of 64 channels (17 KiB per wave whatever the
  // head size: the workgroups of a long-context launch must all be resident at once): the
  // wave reads them as coalesced 16-byte-per-lane loads and re-reads them one row per lane
  // (conflict-free with the 4-float row pad).
  auto qk_half = [&](int col0, float (&s)[GQ]) __attribute__((always_inline)) {
    __builtin_amdgcn_wave_barrier();  // LDS operations of one wave execute in order
    const float4* krow = reinterpret_cast<const float4*>(ks + lane * KS);
    constexpr int KB = 8;
#pragma unroll 1
    for (int c0 = 0; c0 < 16; c0 += KB) {
      float4 kreg[KB];
#pragma unroll
      for (int c = 0; c < KB; ++c) kreg[c] = krow[c0 + c];
#pragma unroll
      for (int c = 0; c < KB; ++c) {
#pragma unroll
        for (int g = 0; g < GQ; ++g) {
          const float4 qq = *reinterpret_cast<const float4*>(tq + g * DH + col0 + (c0 + c) * 4);
          s[g] = fmaf(qq.x, kreg[c].x, s[g]);
          s[g] = fmaf(qq.y, kreg[c].y, s[g]);
          s[g] = fmaf(qq.z, kreg[c].z, s[g]);
          s[g] = fmaf(qq.w, kreg[c].w, s[g]);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();  // the tile is rewritten next
  };
  auto put_scores = [&](int t0, const float (&s)[GQ]) __attribute__((always_inline)) {
    const int t = t0 + lane;
#pragma unroll
    for (int g = 0; g < GQ; ++g) {
      const float sc = s[g] * p.scale;
      if (t < t_cached) {  // lanes past the length hold a duplicate of the last valid row
        tp[g * a.t_pad + t] = sc;
        m_run[g] = fmaxf(m_run[g], sc);
      }
    }
  };
  if (tw * 64 < t_cached) {  // first chunk: its first half is already in registers
    float s[GQ];
#pragma unroll
    for (int g = 0; g < GQ; ++g) s[g] = 0.f;
    v4f st[16];
    if (DH > 64) { CT2X_STAGE_LOAD(st, tw * 64, 64, t_cached - 1) }
    CT2X_STAGE_STORE(kpre)
    qk_half(0, s);
    if (DH > 64) {
      CT2X_STAGE_STORE(st)
      qk_half(64, s);
    }
    put_scores(tw * 64, s);
  }
  for (int t0 = (tw + TW) * 64; t0 < t_cached; t0 += TW * 64) {
    float s[GQ];
#pragma unroll
    for (int g = 0; g < GQ; ++g) s[g] = 0.f;
    v4f st[16], st2[16];
    CT2X_STAGE_LOAD(st, t0, 0, t_cached - 1)
    if (DH > 64) { CT2X_STAGE_LOAD(st2, t0, 64, t_cached - 1) }
    CT2X_STAGE_STORE(st)
    qk_half(0, s);
    if (DH > 64) {
      CT2X_STAGE_STORE(st2)
      qk_half(64, s);
    }
    put_scores(t0, s);
  }

  CT2X_STAMP(4)
  // ---- softmax over [0, t_total) ----------------------------------------------------------------
  float mx[GQ], rs[GQ];
#pragma unroll
  for (int g = 0; g < GQ; ++g) {
    float m = wave_max_dpp(m_run[g]);
    if (tw == 0 && has_new) m = fmaxf(m, s_new[g]);
    if (TW > 1) {
      if (lane == 0) tstat[(tw * GQ + g) * 2] = m;
    } else {
      mx[g] = m;
    }
  }
  if (TW > 1) {
    __syncthreads();
#pragma unroll
    for (int g = 0; g < GQ; ++g) {
      float m = tstat[g * 2];
#pragma unroll
      for (int w = 1; w < TW; ++w) m = fmaxf(m, tstat[(w * GQ + g) * 2]);
      mx[g] = m;
    }
  }
  float e_new[GQ];
#pragma unroll
  for (int g = 0; g < GQ; ++g) {
    float sum = 0.f;
    for (int t0 = tw * 64; t0 < t_cached; t0 += TW * 64) {
      const int t = t0 + lane;
      if (t < t_cached) {
        const float e = expf(tp[g * a.t_pad + t] - mx[g]);
        tp[g * a.t_pad + t] = e;
        sum += e;
      }
    }
    sum = wave_sum_dpp(sum);
    e_new[g] = 0.f;
    if (tw == 0 && has_new) {
      e_new[g] = expf(s_new[g] - mx[g]);
      sum += e_new[g];
    }
    if (TW > 1) {
      if (lane == 0) tstat[(tw * GQ + g) * 2 + 1] = sum;
    } else {
      rs[g] = (t_total > 0) ? 1.f / sum : 0.f;
    }
  }
  if (TW > 1) {
    __syncthreads();
#pragma unroll
    for (int g = 0; g < GQ; ++g) {
      float sum = tstat[g * 2 + 1];
#pragma unroll
      for (int w = 1; w < TW; ++w) sum += tstat[(w * GQ + g) * 2 + 1];
      rs[g] = (t_total > 0) ? 1.f / sum : 0.f;
    }
  }
#pragma unroll
  for (int g = 0; g < GQ; ++g) {
    for (int t0 = tw * 64; t0 < t_cached; t0 += TW * 64) {
      const int t = t0 + lane;
      if (t < t_cached) tp[g * a.t_pad + t] = tp[g * a.t_pad + t] * rs[g];
    }
  }
  // The probabilities of a wave's chunks are only read by that wave below: no barrier needed,
  // the LDS operations of one wave are ordered.

  CT2X_STAMP(5)
  // ---- PV: lane = channel ----------------------------------------------------------------------
  float acc[GQ][E];
#pragma unroll
  for (int g = 0; g < GQ; ++g)
#pragma unroll
    for (int e = 0; e < E; ++e) acc[g][e] = 0.f;
  // Batches of VB rows, double buffered: the next batch is in flight while this one is used
  // (batch 0 has been in flight since the top of the kernel). Rows past the length are
  // duplicates of the last valid row and get probability 0; with an empty cache (t_cached == 0)
  // the loops below do not run.
#define CT2X_PV_BATCH(VR, U0)                                                                    \
  {                                                                                              \
    const int tb_ = CT2X_T_OF(U0); /* batches never straddle a chunk */                          \
    _Pragma("unroll") for (int j4 = 0; j4 < VB; j4 += 4) {                                       \
      _Pragma("unroll") for (int g = 0; g < GQ; ++g) {                                           \
        const float4 pr_ = *reinterpret_cast<const float4*>(tp + g * a.t_pad + tb_ + j4);       \
        const float pj_[4] = {pr_.x, pr_.y, pr_.z, pr_.w};                                       \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                          \
          const float pt_ = (tb_ + j4 + j < t_cached) ? pj_[j] : 0.f;                            \
          _Pragma("unroll") for (int e = 0; e < E; ++e)                                          \
              acc[g][e] = fmaf(pt_, VR[j4 + j][e], acc[g][e]);                                   \
        }                                                                                        \
      }                                                                                          \
    }                                                                                            \
  }
  for (int u0 = 0; CT2X_T_OF(u0) < t_cached; u0 += 2 * VB) {
    CT2X_LOAD_V(v_b, u0 + VB, t_cached - 1)
    CT2X_PV_BATCH(v_a, u0)
    if (!(CT2X_T_OF(u0 + VB) < t_cached)) break;
    CT2X_LOAD_V(v_a, u0 + 2 * VB, t_cached - 1)
    CT2X_PV_BATCH(v_b, u0 + VB)
  }
  CT2X_STAMP(6)
  if (tw == 0 && has_new) {
#pragma unroll
    for (int g = 0; g < GQ; ++g) {
      const float pt = e_new[g] * rs[g];
#pragma unroll
      for (int e = 0; e < E; ++e) acc[g][e] = fmaf(pt, vnew[e], acc[g][e]);
    }
  }
  if (TW > 1) {
#pragma unroll
    for (int g = 0; g < GQ; ++g)
#pragma unroll
      for (int e = 0; e < E; ++e) tpart[(tw * GQ + g) * DH + CT2X_PVD(e)] = acc[g][e];
    __syncthreads();
    if (tw == 0) {
#pragma unroll
      for (int g = 0; g < GQ; ++g)
#pragma unroll
        for (int e = 0; e < E; ++e) {
          float v = tpart[g * DH + CT2X_PVD(e)];
#pragma unroll
          for (int w = 1; w < TW; ++w) v += tpart[(w * GQ + g) * DH + CT2X_PVD(e)];
          acc[g][e] = v;
        }
    }
  }

  CT2X_STAMP(7)
  if (a.q_out == nullptr) {
    if (tw == 0) {
      float amax = 0.f;
#pragma unroll
      for (int g = 0; g < GQ; ++g)
#pragma unroll
        for (int e = 0; e < E; ++e) {
          p.out[static_cast<int64_t>(row) * (p.heads * DH) + (hq0 + g) * DH + CT2X_PVD(e)] =
              acc[g][e];
          amax = fmaxf(amax, fabsf(acc[g][e]));
        }
      if (p.row_amax_out != nullptr) {
        // first half of the ops::Quantize that follows: max over the row's heads (bit patterns of
        // non-negative floats order like unsigned integers: exact, order independent)
        amax = wave_max_dpp(amax);
        if (lane == 0)
          atomicMax(reinterpret_cast<unsigned int*>(p.row_amax_out) + row, __float_as_uint(amax));
      }
    }
    return;
  }
  // ---- fused ops::Quantize of the row [heads * dhead] (src/cpu/kernels.cc:577-620); row mode:
  // every KV head of the row is a team of this workgroup.
  float amax = 0.f;
#pragma unroll
  for (int g = 0; g < GQ; ++g)
#pragma unroll
    for (int e = 0; e < E; ++e) amax = fmaxf(amax, fabsf(acc[g][e]));
  amax = wave_max_dpp(amax);
  if (tw == 0 && lane == 0) team_amax[team] = amax;
  __syncthreads();
  amax = team_amax[0];
  for (int w = 1; w < a.teams; ++w) amax = fmaxf(amax, team_amax[w]);
  const float qscale = (amax != 0.f) ? 127.f / amax : 1.f;
  if (threadIdx.x == 0) a.q_scales[row] = qscale;
  if (tw == 0) {
#pragma unroll
    for (int g = 0; g < GQ; ++g)
#pragma unroll
      for (int e = 0; e < E; ++e)
        a.q_out[static_cast<int64_t>(row) * (p.heads * DH) + (hq0 + g) * DH + CT2X_PVD(e)] =
            to_s8(acc[g][e], qscale, a.quant_mode);
  }
}

template <int DH, int GQ, int TW, bool ANC = false>
void launch(const Attn2Args& a, hipStream_t stream) {
  const int team_floats =
      (GQ * DH + GQ * a.t_pad + TW * GQ * 2 + (TW > 1 ? TW * GQ * DH : 0) + 3) & ~3;
  const size_t lds = (static_cast<size_t>(a.teams) * team_floats + ((a.teams + 3) & ~3) +
                      static_cast<size_t>(a.teams) * TW * 64 * 68) * sizeof(float);
  if (lds > 64 * 1024) {
    // > 64 KiB of dynamic LDS has to be opted into once per device.
    static std::atomic<bool> configured[64];
    int dev = 0;
    CT2X_HIP_CHECK(hipGetDevice(&dev));
    if (!configured[dev & 63].load(std::memory_order_acquire)) {
      CT2X_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(attn2_kernel<DH, GQ, TW, ANC>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      configured[dev & 63].store(true, std::memory_order_release);
    }
  }
  const unsigned grid = static_cast<unsigned>(a.p.rows * a.wg_per_row);
  hipLaunchKernelGGL((attn2_kernel<DH, GQ, TW, ANC>), dim3(grid), dim3(a.teams * TW * 64), lds,
                     stream, a);
}

template <int DH, int GQ>
void launch_tw(const Attn2Args& a, int tw, hipStream_t stream) {
  if (tw == 1) launch<DH, GQ, 1>(a, stream);
  else launch<DH, GQ, 4>(a, stream);
}

template <int DH>
void launch_gq(const Attn2Args& a, int gq, int tw, hipStream_t stream) {
  if (a.p.ancestry != nullptr) {
    // split mode, one query head per team (attention_decode2_applicable): beam search through the
    // ancestry table
    if (gq != 1) throw Error("attn2: the ancestry table needs one query head per team");
    if (tw == 1) launch<DH, 1, 1, true>(a, stream);
    else launch<DH, 1, 4, true>(a, stream);
    return;
  }
  switch (gq) {
    case 1: launch_tw<DH, 1>(a, tw, stream); break;
    case 2: launch_tw<DH, 2>(a, tw, stream); break;
    default: launch_tw<DH, 4>(a, tw, stream); break;
  }
}

#undef CT2X_STAMP
#undef CT2X_PVD
#undef CT2X_STAGE_LOAD
#undef CT2X_STAGE_STORE
#undef CT2X_T_OF
#undef CT2X_LOAD_V
#undef CT2X_PV_BATCH

}  // namespace

// Shapes the second-generation kernel covers; everything else stays on attention.hip.
bool attention_decode2_applicable(const AttnDecodeParams& p, bool fused_quantize) {
  static const bool disabled = tuning_env("CT2X_ATTN_V1") != nullptr;
  if (disabled) return false;
  if (p.dhead != 64 && p.dhead != 128) return false;
  if (p.heads_kv < 1 || p.heads % p.heads_kv != 0) return false;
  const int64_t gq = p.heads / p.heads_kv;
  if (gq != 1 && gq != 2 && gq != 4) return false;
  // cache rows may be strided (encoder memory K/V live interleaved in the projection output)
  if (p.kv_time_stride < p.dhead || p.kv_time_stride % 4 != 0 || p.t_max < 1) return false;
  if ((reinterpret_cast<uintptr_t>(p.k_cache) & 15) || (reinterpret_cast<uintptr_t>(p.v_cache) & 15) ||
      p.kv_head_stride % 4 != 0 || p.kv_row_stride % 4 != 0)
    return false;
  if (p.rot_dim > p.dhead) return false;
  // the ancestry table is followed in split mode only (self-attention of a beam search)
  if (p.ancestry != nullptr && (fused_quantize || p.kv_group != 1 || p.lengths != nullptr)) return false;
  const int64_t t_pad = ((p.t_max + 1 + 63) / 64) * 64;
  if (fused_quantize) {
    // row mode: all KV heads of a row in one workgroup of <= 16 waves, its LDS within 64 KiB
    const int64_t team_floats = gq * p.dhead + gq * t_pad + gq * 2 + 4;
    // ... and the row's K/V small enough that one CU streams it in a couple of microseconds;
    // otherwise the (row, KV head) split + a separate quantize launch is faster.
    return p.heads_kv <= 8 && (p.heads_kv * team_floats + p.heads_kv) * 4 <= 64 * 1024 &&
           p.heads_kv * p.t_max * p.dhead * 8 <= 512 * 1024;
  }
  const int64_t team_floats =
      p.dhead + t_pad + 4 * 2 + 4 * p.dhead + 4 * 64 * 68 + 8;  // split mode: one query head per team
  return team_floats * 4 <= 156 * 1024;
}

bool attention_decode_emits_row_amax(const AttnDecodeParams& p) {
  if (attention_exact_decode_applicable(p, false)) return p.dhead == 128;  // the exact kernel is preferred
  return attention_decode2_applicable(p, false);
}

void launch_attention_decode2(const AttnDecodeParams& p, int quant_mode, int8_t* q_out,
                              float* q_scales, hipStream_t stream) {
  Attn2Args a;
  a.p = p;
  a.q_out = q_out;
  a.q_scales = q_scales;
  a.quant_mode = quant_mode;
  a.t_pad = static_cast<int>(((p.t_max + 1 + 63) / 64) * 64);
  a.trace = g_attn_trace;
  int gq = static_cast<int>(p.heads / p.heads_kv);
  a.group = gq;
  int tw;
  if (q_out != nullptr) {
    a.teams = static_cast<int>(p.heads_kv);
    a.wg_per_row = 1;
    a.per_q_head = 0;
    tw = 1;
  } else {
    // Split mode. Default: one workgroup per QUERY head (more waves in flight, K/V re-read from
    // L2 once per head of a GQA group). CT2X_ATTN_SHARE_KV=1: one workgroup per KV head whose
    // waves carry the whole group in registers (K/V read once) - for comparison runs.
    static const bool share_kv = [] {
      const char* e = tuning_env("CT2X_ATTN_SHARE_KV");
      return e != nullptr && e[0] == '1';
    }();
    a.teams = 1;
    if (share_kv) {
      a.wg_per_row = static_cast<int>(p.heads_kv);
      a.per_q_head = 0;
    } else {
      a.wg_per_row = static_cast<int>(p.heads);
      a.per_q_head = 1;
      gq = 1;
    }
    tw = p.t_max > 64 ? 4 : 1;
  }
  static const bool no_remap = tuning_env("CT2X_ATTN_NO_XCD_REMAP") != nullptr;
  const int64_t grid = p.rows * a.wg_per_row;
  a.xcd_remap = (!no_remap && q_out == nullptr && a.per_q_head && a.group > 1 && grid % 8 == 0 &&
                 grid >= 64) ? 1 : 0;
  if (p.dhead == 64) launch_gq<64>(a, gq, tw, stream);
  else launch_gq<128>(a, gq, tw, stream);
}

}  // namespace ct2x

// Debugging hook (tools/attn_trace.py): device buffer of 16 x 16 int64 that receives s_memtime
// stamps of workgroup 0 at the phase boundaries of attn2_kernel; nullptr switches it off.
extern "C" void ct2x_debug_set_attn_trace(void* device_buffer) {
  ct2x::g_attn_trace = static_cast<long long*>(device_buffer);
}
