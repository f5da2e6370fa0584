// Sequence-mode attention (encoder self-attention, decoder-only prompt prefill) for gfx950:
// MultiHeadAttention::operator() with Tq > 1 (src/layers/attention.cc:442-615: MatMul QK^T *
// queries_scale -> prepare_length_mask / causal mask (src/layers/attention_layer.cc:152-174) ->
// SoftMax -> MatMul PV), d_head 64 / 128.
//
// attention.hip handles one query row per wave and re-reads the whole K/V of its batch for every
// row (8.6 GB of L2 traffic per Llama-3-8B layer at 16 x 128 prompt tokens). Here a workgroup of
// 4 waves owns a 64-query tile of one (batch, head); keys are visited in tiles of 32 that every
// query of the tile shares through LDS, and both products run on the matrix pipe in exact fp32
// (v_mfma_f32_16x16x4_f32: fp32 operands, fp32 accumulate - no precision is traded):
//   S  = Q K^T      per wave 16 queries x 16 keys per MFMA accumulator, K^T fragments from LDS;
//   softmax         online (running max m, running sum l, rescale of O) - the result equals
//                   exp(x - max) / sum of src/cpu/kernels.cc:452-455 up to fp32 rounding; row
//                   reductions over the 16 key columns are DPP butterflies inside one DPP row;
//   O += P V        P goes C-layout -> LDS -> A-layout (per-wave 16 x 32 tile), V fragments
//                   from LDS.
// GQA reads the KV head h / group (no ops::Tile replication, attention.cc:291-295).

#include "device_utils.h"
#include "kernels.h"
#include "runtime.h"

#include <cstdlib>
#include <type_traits>

namespace ct2x {

namespace {

constexpr int PF_BQ = 64;   // queries per workgroup (16 per wave)
constexpr int PF_BK = 32;   // keys per tile

// all-reduce over the 16 lanes of a DPP row: xor 1, xor 2 (quad_perm), then the two mirrors
template <bool kMax>
__device__ __forceinline__ float row16_allreduce(float v) {
  auto step = [&](float o) { return kMax ? fmaxf(v, o) : v + o; };
  v = step(__int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false)));   // quad_perm [1,0,3,2]
  v = step(__int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false)));   // quad_perm [2,3,0,1]
  v = step(__int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false)));  // row_half_mirror
  v = step(__int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, false)));  // row_mirror
  return v;
}

template <int DH>
__global__ __launch_bounds__(256) void attn_prefill_kernel(AttnFullParams p) {
  constexpr int KS = DH + 4;    // K tile row stride (floats): B-operand reads hit 64 distinct banks
  constexpr int VS = DH + 16;   // V tile row stride
  constexpr int PS = PF_BK + 4; // P tile row stride
  constexpr int DB = DH / 16;   // output column blocks per wave
  __shared__ __attribute__((aligned(16))) float k_tile[PF_BK * KS];
  __shared__ __attribute__((aligned(16))) float v_tile[PF_BK * VS];
  __shared__ __attribute__((aligned(16))) float p_tile[4][16 * PS];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;  // MFMA operand row / k group; C layout: col / row group
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * PF_BQ;
  const int hk = h / static_cast<int>(p.heads / p.heads_kv);
  const int length = p.lengths ? min(max(p.lengths[b], 0), static_cast<int>(p.tk)) : static_cast<int>(p.tk);

  const float* qb = p.q + static_cast<int64_t>(b) * p.q_batch_stride + static_cast<int64_t>(h) * DH;
  const float* kb = p.k + static_cast<int64_t>(b) * p.k_batch_stride + static_cast<int64_t>(hk) * p.k_head_stride;
  const float* vb = p.v + static_cast<int64_t>(b) * p.v_batch_stride + static_cast<int64_t>(hk) * p.v_head_stride;

  // Q fragments (A operand of v_mfma_f32_16x16x4_f32: lane = (row lr, k lg)): DH / 4 floats.
  const int qrow_a = q0 + wave * 16 + lr;
  float qf[DH / 4];
#pragma unroll
  for (int s = 0; s < DH / 4; ++s)
    qf[s] = (qrow_a < p.tq) ? qb[static_cast<int64_t>(qrow_a) * p.q_time_stride + 4 * s + lg] : 0.f;

  // C-layout rows of this lane: wave*16 + lg*4 + r
  int qpos[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) qpos[r] = q0 + wave * 16 + lg * 4 + r;

  v4f o[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d) o[d] = v4f{0.f, 0.f, 0.f, 0.f};
  float m_run[4], l_run[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    m_run[r] = -FLT_MAX;
    l_run[r] = 0.f;
  }

  // keys this tile can see: all valid keys, or up to the last query of the tile when causal
  int k_end = length;
  if (p.causal) k_end = min(k_end, q0 + PF_BQ - 1 + static_cast<int>(p.q_offset) + 1);

  for (int k0 = 0; k0 < k_end; k0 += PF_BK) {
    __syncthreads();  // previous tile fully consumed
    // stage K and V tiles (rows past the valid range are zero-filled and masked below)
    for (int e = tid; e < PF_BK * (DH / 4); e += 256) {
      const int row = e / (DH / 4), c4 = e % (DH / 4);
      const int t = k0 + row;
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      if (t < length) {
        kv = *reinterpret_cast<const float4*>(kb + static_cast<int64_t>(t) * p.k_time_stride + c4 * 4);
        vv = *reinterpret_cast<const float4*>(vb + static_cast<int64_t>(t) * p.v_time_stride + c4 * 4);
      }
      *reinterpret_cast<float4*>(k_tile + row * KS + c4 * 4) = kv;
      *reinterpret_cast<float4*>(v_tile + row * VS + c4 * 4) = vv;
    }
    __syncthreads();

    // S = Q K^T for the two 16-key blocks of the tile
    v4f sacc[2];
#pragma unroll
    for (int kb16 = 0; kb16 < 2; ++kb16) {
      sacc[kb16] = v4f{0.f, 0.f, 0.f, 0.f};
      const float* kr = k_tile + (kb16 * 16 + lr) * KS + lg;  // B operand: (k = lg, n = key lr)
#pragma unroll
      for (int s = 0; s < DH / 4; ++s)
        sacc[kb16] = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[s], kr[4 * s], sacc[kb16], 0, 0, 0);
    }
    // mask + online softmax; C layout: row = lg*4 + r, col (key) = kb16*16 + lr
    float pr[2][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float mx = -FLT_MAX;
#pragma unroll
      for (int kb16 = 0; kb16 < 2; ++kb16) {
        const int t = k0 + kb16 * 16 + lr;
        bool ok = t < length;
        if (p.causal) ok = ok && (t <= qpos[r] + static_cast<int>(p.q_offset));
        // scores = alpha * (q . k) with alpha = queries_scale (ops::MatMul, attention.cc:216-217)
        const float sv = ok ? sacc[kb16][r] * p.scale : -FLT_MAX;
        pr[kb16][r] = sv;
        mx = fmaxf(mx, sv);
      }
      mx = row16_allreduce<true>(mx);
      const float m_new = fmaxf(m_run[r], mx);
      // alpha = exp(m_old - m_new); nothing accumulated yet (m_old = -FLT_MAX) -> 0
      const float alpha = (m_run[r] == -FLT_MAX) ? 0.f : expf(m_run[r] - m_new);
      float sum = 0.f;
#pragma unroll
      for (int kb16 = 0; kb16 < 2; ++kb16) {
        const float e = (pr[kb16][r] == -FLT_MAX) ? 0.f : expf(pr[kb16][r] - m_new);
        pr[kb16][r] = e;
        sum += e;
      }
      sum = row16_allreduce<false>(sum);
      l_run[r] = l_run[r] * alpha + sum;
      m_run[r] = m_new;
#pragma unroll
      for (int d = 0; d < DB; ++d) o[d][r] = o[d][r] * alpha;
    }
    // P: C layout -> LDS -> A layout (this wave's private 16 x 32 tile)
    float* pt = p_tile[wave];
#pragma unroll
    for (int kb16 = 0; kb16 < 2; ++kb16)
#pragma unroll
      for (int r = 0; r < 4; ++r) pt[(lg * 4 + r) * PS + kb16 * 16 + lr] = pr[kb16][r];
    __builtin_amdgcn_wave_barrier();
    // O += P V: A operand (row lr, k = key 4s + lg), B operand (k = key 4s + lg, n = d lr)
#pragma unroll
    for (int s = 0; s < PF_BK / 4; ++s) {
      const float pa = pt[lr * PS + 4 * s + lg];
      const float* vr = v_tile + (4 * s + lg) * VS + lr;
#pragma unroll
      for (int d = 0; d < DB; ++d)
        o[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa, vr[d * 16], o[d], 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
  }

  // out = O / l  (softmax = exp(x - max) * (1 / sum)); fully masked rows stay zero
  float* ob = p.out + static_cast<int64_t>(b) * p.out_batch_stride + static_cast<int64_t>(h) * DH;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (qpos[r] >= p.tq) continue;
    const float inv = l_run[r] > 0.f ? 1.f / l_run[r] : 0.f;
#pragma unroll
    for (int d = 0; d < DB; ++d)
      ob[static_cast<int64_t>(qpos[r]) * p.out_time_stride + d * 16 + lr] = o[d][r] * inv;
  }
}

// ---- 16-bit compute types: the same tiling on v_mfma_f32_16x16x32_{f16,bf16} ---------------------
// The engine keeps activations in fp32 under float16 / bfloat16 compute (16-bit weights, fp32
// accumulate: gemm_h.hip), so Q, K, V and the probabilities arrive as fp32. Each enters the 16-bit
// matrix pipe as TWO 16-bit values, x = hi + lo (hi = RN16(x), lo = RN16(x - hi): 22 significant
// bits as float16 - an absolute 2^-25 below |x| = 1/8, where lo is a float16 subnormal -, 16 as
// bfloat16; tests/test_split16_cpu.py), and a product a.b is hi.hi + hi.lo + lo.hi accumulated in fp32:
// three MFMAs of 17 cycles for a 16x16x32 block where the fp32 pipe above spends eight of 32 - the
// tile's matrix time drops 5x and the result stays a float32 evaluation up to the dropped lo.lo
// term (2^-22 / 2^-16 relative per product), well inside what the 16-bit types themselves resolve.
//
// Everything is held TRANSPOSED so that no operand is re-laid-out between the two products:
//   S^T = K Q^T    A = K tile rows (LDS, d contiguous), B = Q^T (registers, loaded once);
//                  C layout: lane (query lr) holds keys kb16*16 + lg*4 + r
//   softmax        per lane ONE query: in-lane max over its 8 keys + two cross-row shuffles; the
//                  row sum stays a per-lane partial until the end (alpha is common to the 4 lanes)
//   O^T = V^T P^T  B = P^T straight from the softmax registers: the MFMA's k index is only a
//                  summation order, so k-slots 8 lg .. 8 lg + 7 are DEFINED as the keys this lane
//                  holds (lg*4 .. +3 and 16 + lg*4 .. +3) and the A operand (V^T tile, keys
//                  contiguous per d row in LDS) is read with the same assignment; P never visits LDS
//   output         lane holds 4 consecutive d of one query: one 16-byte store per 16-wide d block.
// The next tile's K / V rows are in flight (registers) while the current tile is computed.
// Workgroups of one (batch, KV head) are placed on ONE XCD (blockIdx % 8), so its K / V are pulled
// into one L2, not eight.

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef __bf16 v8b __attribute__((ext_vector_type(8)));

template <bool kBF16>
struct Split16;
template <>
struct Split16<false> {
  using vec = v8h;
  using elem = _Float16;
  __device__ static __forceinline__ v4f mfma(vec a, vec b, v4f c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};
template <>
struct Split16<true> {
  using vec = v8b;
  using elem = __bf16;
  __device__ static __forceinline__ v4f mfma(vec a, vec b, v4f c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};

template <typename E>
__device__ __forceinline__ void split16(float x, E& hi, E& lo) {
  hi = static_cast<E>(x);
  lo = static_cast<E>(x - static_cast<float>(hi));
}

template <typename E>
__device__ __forceinline__ uint16_t bits16(E v) {
  return __builtin_bit_cast(uint16_t, v);
}

template <int DH, bool kBF16, int NW>
__global__ __launch_bounds__(NW * 64) void attn_prefill_h_kernel(AttnFullParams p, int q_tiles) {
  using S = Split16<kBF16>;
  using vec = typename S::vec;
  using E = typename S::elem;
  constexpr int KSH = DH + 8;      // K tile row stride (16-bit elements): 16-byte reads, 16 lanes on distinct banks
  constexpr int VSH = PF_BK + 8;   // V^T tile row stride
  constexpr int KST = DH / 32;     // 32-wide k-steps of S
  constexpr int DB = DH / 16;      // 16-wide d blocks of O
  constexpr int NT = NW * 64;      // threads; NW waves x 16 queries per workgroup
  constexpr int BQ = NW * 16;
  constexpr int ITEMS = 4 * DH;    // 8-float pieces of a K tile = (key pair, 4 d) pieces of a V tile
  // staging roles: with twice as many threads as pieces, the lower waves stage K and the upper V
  constexpr bool kSplitRoles = NT >= 2 * ITEMS;
  constexpr int KIT = kSplitRoles ? 1 : ITEMS / NT;  // pieces of each kind per thread per tile
  __shared__ __attribute__((aligned(16))) uint16_t k_hi[PF_BK * KSH];
  __shared__ __attribute__((aligned(16))) uint16_t k_lo[PF_BK * KSH];
  __shared__ __attribute__((aligned(16))) uint16_t v_hi[DH * VSH];
  __shared__ __attribute__((aligned(16))) uint16_t v_lo[DH * VSH];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;

  // blockIdx -> (batch, head, query tile): all workgroups of a (batch, KV head) unit on one XCD
  const int group = static_cast<int>(p.heads / p.heads_kv);
  const int per_unit = group * q_tiles;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int unit = (slot / per_unit) * 8 + xcd;
  if (unit >= static_cast<int>(p.batch * p.heads_kv)) return;
  const int within = slot % per_unit;
  const int b = unit / static_cast<int>(p.heads_kv), hk = unit % static_cast<int>(p.heads_kv);
  const int h = hk * group + within / q_tiles;
  const int q0 = (within % q_tiles) * BQ;
  const int length = p.lengths ? min(max(p.lengths[b], 0), static_cast<int>(p.tk)) : static_cast<int>(p.tk);

  const float* qb = p.q + static_cast<int64_t>(b) * p.q_batch_stride + static_cast<int64_t>(h) * DH;
  const float* kb = p.k + static_cast<int64_t>(b) * p.k_batch_stride + static_cast<int64_t>(hk) * p.k_head_stride;
  const float* vb = p.v + static_cast<int64_t>(b) * p.v_batch_stride + static_cast<int64_t>(hk) * p.v_head_stride;

  // Q^T fragments (B operand: lane = (query lr, d 32 ks + 8 lg .. + 7)), split once
  const int qpos = q0 + wave * 16 + lr;
  vec qh[KST], ql[KST];
#pragma unroll
  for (int ks = 0; ks < KST; ++ks) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
    if (qpos < p.tq) {
      const float* src = qb + static_cast<int64_t>(qpos) * p.q_time_stride + 32 * ks + 8 * lg;
      a = *reinterpret_cast<const float4*>(src);
      c = *reinterpret_cast<const float4*>(src + 4);
    }
    const float f[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      E hi, lo;
      split16(f[e], hi, lo);
      qh[ks][e] = hi;
      ql[ks][e] = lo;
    }
  }

  v4f o[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d) o[d] = v4f{0.f, 0.f, 0.f, 0.f};
  float m_run = -FLT_MAX, l_part = 0.f;

  int k_end = length;
  if (p.causal) k_end = min(k_end, q0 + BQ - 1 + static_cast<int>(p.q_offset) + 1);

  // staging assignment. K: thread -> (key row, 8 consecutive d); V: thread -> (key pair, 4 d),
  // key pair = index & 15 so that the transposed 4-byte stores of a wave fall on distinct banks.
  // One register set serves both kinds when the roles are split by wave.
  const bool stages_k = !kSplitRoles || tid < ITEMS;
  const bool stages_v = !kSplitRoles || tid >= ITEMS;
  const int vtid = kSplitRoles ? tid - ITEMS : tid;
  float4 kreg[KIT][2];
  float4 vreg_own[kSplitRoles ? 1 : KIT][2];
  float4 (&vreg)[KIT][2] = *reinterpret_cast<float4 (*)[KIT][2]>(kSplitRoles ? &kreg[0] : &vreg_own[0]);
  auto fetch = [&](int k0) {
    if (stages_k) {
#pragma unroll
      for (int it = 0; it < KIT; ++it) {
        const int e = tid + it * NT;
        const int row = e / (DH / 8), c8 = e % (DH / 8);
        const int t = k0 + row;
        kreg[it][0] = kreg[it][1] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < length) {
          const float* src = kb + static_cast<int64_t>(t) * p.k_time_stride + c8 * 8;
          kreg[it][0] = *reinterpret_cast<const float4*>(src);
          kreg[it][1] = *reinterpret_cast<const float4*>(src + 4);
        }
      }
    }
    if (stages_v) {
#pragma unroll
      for (int it = 0; it < KIT; ++it) {
        const int e = vtid + it * NT;
        const int kp = e & 15, c4 = e >> 4;
        const int t = k0 + 2 * kp;
        vreg[it][0] = vreg[it][1] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < length) vreg[it][0] = *reinterpret_cast<const float4*>(vb + static_cast<int64_t>(t) * p.v_time_stride + c4 * 4);
        if (t + 1 < length)
          vreg[it][1] = *reinterpret_cast<const float4*>(vb + static_cast<int64_t>(t + 1) * p.v_time_stride + c4 * 4);
      }
    }
  };
  auto stage = [&]() {
    if (stages_k) {
#pragma unroll
      for (int it = 0; it < KIT; ++it) {
        const int e = tid + it * NT;
        const int row = e / (DH / 8), c8 = e % (DH / 8);
        const float f[8] = {kreg[it][0].x, kreg[it][0].y, kreg[it][0].z, kreg[it][0].w,
                            kreg[it][1].x, kreg[it][1].y, kreg[it][1].z, kreg[it][1].w};
        vec hi, lo;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          E a, c;
          split16(f[i], a, c);
          hi[i] = a;
          lo[i] = c;
        }
        *reinterpret_cast<vec*>(k_hi + row * KSH + c8 * 8) = hi;
        *reinterpret_cast<vec*>(k_lo + row * KSH + c8 * 8) = lo;
      }
    }
    if (stages_v) {
#pragma unroll
      for (int it = 0; it < KIT; ++it) {
        const int e = vtid + it * NT;
        const int kp = e & 15, c4 = e >> 4;
        const float f0[4] = {vreg[it][0].x, vreg[it][0].y, vreg[it][0].z, vreg[it][0].w};
        const float f1[4] = {vreg[it][1].x, vreg[it][1].y, vreg[it][1].z, vreg[it][1].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          E h0, l0, h1, l1;
          split16(f0[i], h0, l0);
          split16(f1[i], h1, l1);
          const int at = (c4 * 4 + i) * VSH + 2 * kp;  // V^T[d][key pair]
          *reinterpret_cast<uint32_t*>(v_hi + at) = bits16(h0) | (static_cast<uint32_t>(bits16(h1)) << 16);
          *reinterpret_cast<uint32_t*>(v_lo + at) = bits16(l0) | (static_cast<uint32_t>(bits16(l1)) << 16);
        }
      }
    }
  };

  // scores are kept in the log2 domain: s2 = (q.k) * scale * log2(e), p = 2^(s2 - m2) - one
  // v_exp_f32 per probability, no range reduction (the argument is <= 0; below -126 it is 0)
  const float scale2 = p.scale * 1.44269504088896340736f;
  auto tile = [&](int k0, auto masked_tag) {
    constexpr bool kMasked = decltype(masked_tag)::value;
    // S^T = K Q^T for the two 16-key blocks
    v4f sacc[2];
#pragma unroll
    for (int kb16 = 0; kb16 < 2; ++kb16) {
      v4f acc = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KST; ++ks) {
        const int at = (kb16 * 16 + lr) * KSH + 32 * ks + 8 * lg;
        const vec kh = *reinterpret_cast<const vec*>(k_hi + at);
        const vec kl = *reinterpret_cast<const vec*>(k_lo + at);
        acc = S::mfma(kl, qh[ks], acc);
        acc = S::mfma(kh, ql[ks], acc);
        acc = S::mfma(kh, qh[ks], acc);
      }
      sacc[kb16] = acc;
    }
    // mask + online softmax for query lr over keys k0 + kb16*16 + lg*4 + r
    float pr[8];
    float mx = -FLT_MAX;
#pragma unroll
    for (int kb16 = 0; kb16 < 2; ++kb16)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float sv = sacc[kb16][r] * scale2;
        if (kMasked) {
          const int t = k0 + kb16 * 16 + lg * 4 + r;
          bool ok = t < length;
          if (p.causal) ok = ok && (t <= qpos + static_cast<int>(p.q_offset));
          sv = ok ? sv : -FLT_MAX;
        }
        pr[kb16 * 4 + r] = sv;
        mx = fmaxf(mx, sv);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    // nothing visible so far (m_new = -FLT_MAX): subtract 0, so that masked scores stay at
    // -FLT_MAX and give 2^-inf = 0; alpha = 2^(m_old - m_new) is 0 the same way when m_old = -FLT_MAX
    const float m_sub = kMasked ? ((m_new == -FLT_MAX) ? 0.f : m_new) : m_new;
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_sub);
    float sum = 0.f;
    vec ph, pl;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float e = __builtin_amdgcn_exp2f(pr[i] - m_sub);
      sum += e;
      E a, c;
      split16(e, a, c);
      ph[i] = a;
      pl[i] = c;
    }
    l_part = l_part * alpha + sum;
    m_run = m_new;
    // O^T = alpha O^T + V^T P^T: A operand lane = (d lr, k-slots 8 lg .. + 7 = keys lg*4 .. + 3, 16 + lg*4 .. + 3)
#pragma unroll
    for (int d = 0; d < DB; ++d) {
      const int at = (d * 16 + lr) * VSH + lg * 4;
      typedef uint16_t u16x4 __attribute__((ext_vector_type(4)));
      const u16x4 h0 = *reinterpret_cast<const u16x4*>(v_hi + at), h1 = *reinterpret_cast<const u16x4*>(v_hi + at + 16);
      const u16x4 l0 = *reinterpret_cast<const u16x4*>(v_lo + at), l1 = *reinterpret_cast<const u16x4*>(v_lo + at + 16);
      const vec vh = __builtin_bit_cast(vec, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
      const vec vl = __builtin_bit_cast(vec, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
      v4f acc = o[d] * alpha;
      acc = S::mfma(vl, ph, acc);
      acc = S::mfma(vh, pl, acc);
      acc = S::mfma(vh, ph, acc);
      o[d] = acc;
    }
  };

  if (k_end > 0) fetch(0);
  for (int k0 = 0; k0 < k_end; k0 += PF_BK) {
    __syncthreads();  // previous tile fully consumed
    stage();
    __syncthreads();
    if (k0 + PF_BK < k_end) fetch(k0 + PF_BK);  // in flight while this tile is computed

    // interior tile: every key visible to every query of this wave (no length edge, below the
    // causal diagonal of the wave's first query) - a wave-uniform test, the mask arithmetic is skipped
    const bool interior = (k0 + PF_BK <= length) &&
                          (!p.causal || k0 + PF_BK - 1 <= q0 + wave * 16 + static_cast<int>(p.q_offset));
    if (interior)
      tile(k0, std::false_type{});
    else
      tile(k0, std::true_type{});
  }

  // out = O / l; the row sum was kept as four per-lane partials (same alphas on all four)
  float l = l_part;
  l += __shfl_xor(l, 16);
  l += __shfl_xor(l, 32);
  if (qpos < p.tq) {
    const float inv = l > 0.f ? 1.f / l : 0.f;
    float* ob = p.out + static_cast<int64_t>(b) * p.out_batch_stride + static_cast<int64_t>(h) * DH +
                static_cast<int64_t>(qpos) * p.out_time_stride;
#pragma unroll
    for (int d = 0; d < DB; ++d)
      *reinterpret_cast<float4*>(ob + d * 16 + lg * 4) =
          make_float4(o[d][0] * inv, o[d][1] * inv, o[d][2] * inv, o[d][3] * inv);
  }
}

}  // namespace

bool attention_prefill_applicable(const AttnFullParams& p) {
  static const bool disabled = tuning_env("CT2X_ATTN_PREFILL_V1") != nullptr;
  if (disabled) return false;
  if (p.dhead != 64 && p.dhead != 128) return false;
  if (p.heads_kv < 1 || p.heads % p.heads_kv != 0) return false;
  if (p.tq < 16) return false;  // a handful of rows: the per-row kernel is fine
  auto aligned16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  return aligned16(p.k) && aligned16(p.v) && p.k_time_stride % 4 == 0 && p.v_time_stride % 4 == 0 &&
         p.k_head_stride % 4 == 0 && p.v_head_stride % 4 == 0 && p.k_batch_stride % 4 == 0 &&
         p.v_batch_stride % 4 == 0;
}

// The 16-bit form also needs 16-byte alignment of q and out (vector loads of the Q rows, 16-byte stores).
static bool prefill_h_applicable(const AttnFullParams& p) {
  static const bool disabled = tuning_env("CT2X_ATTN_PREFILL_F32") != nullptr;
  if (disabled || p.half_kind == 0) return false;
  auto aligned16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  return aligned16(p.q) && aligned16(p.out) && p.q_time_stride % 4 == 0 && p.q_batch_stride % 4 == 0 &&
         p.out_time_stride % 4 == 0 && p.out_batch_stride % 4 == 0 &&
         (p.batch + 7) * p.heads * ((p.tq + PF_BQ - 1) / PF_BQ) < (int64_t{1} << 28);
}

void launch_attention_prefill(const AttnFullParams& p_in, hipStream_t stream) {
  AttnFullParams p = p_in;
  // the output is [B, Tq, H * dh] contiguous unless the caller gives strides
  if (p.out_time_stride == 0) p.out_time_stride = p.heads * p.dhead;
  if (p.out_batch_stride == 0) p.out_batch_stride = p.tq * p.out_time_stride;
  if (prefill_h_applicable(p)) {
    // 128 queries per workgroup (8 waves) when there are that many: a K / V tile is split and
    // staged once per 128 queries instead of per 64
    const bool wide = p.tq > PF_BQ;
    const int bq = wide ? 2 * PF_BQ : PF_BQ;
    const int q_tiles = static_cast<int>((p.tq + bq - 1) / bq);
    const int64_t units = p.batch * p.heads_kv;
    const int64_t per_unit = (p.heads / p.heads_kv) * q_tiles;
    const dim3 grid(static_cast<unsigned>((units + 7) / 8 * 8 * per_unit));
    const bool bf = p.half_kind == 2;
#define CT2X_LAUNCH_PREFILL_H(DH, BF, NW) \
  hipLaunchKernelGGL((attn_prefill_h_kernel<DH, BF, NW>), grid, dim3(NW * 64), 0, stream, p, q_tiles)
    if (p.dhead == 64) {
      if (bf) { if (wide) CT2X_LAUNCH_PREFILL_H(64, true, 8); else CT2X_LAUNCH_PREFILL_H(64, true, 4); }
      else { if (wide) CT2X_LAUNCH_PREFILL_H(64, false, 8); else CT2X_LAUNCH_PREFILL_H(64, false, 4); }
    } else {
      if (bf) { if (wide) CT2X_LAUNCH_PREFILL_H(128, true, 8); else CT2X_LAUNCH_PREFILL_H(128, true, 4); }
      else { if (wide) CT2X_LAUNCH_PREFILL_H(128, false, 8); else CT2X_LAUNCH_PREFILL_H(128, false, 4); }
    }
#undef CT2X_LAUNCH_PREFILL_H
    return;
  }
  const dim3 grid(static_cast<unsigned>((p.tq + PF_BQ - 1) / PF_BQ), static_cast<unsigned>(p.heads),
                  static_cast<unsigned>(p.batch));
  if (p.dhead == 64)
    hipLaunchKernelGGL(attn_prefill_kernel<64>, grid, dim3(256), 0, stream, p);
  else
    hipLaunchKernelGGL(attn_prefill_kernel<128>, grid, dim3(256), 0, stream, p);
}

}  // namespace ct2x
