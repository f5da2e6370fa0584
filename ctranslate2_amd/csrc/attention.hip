// Fused single-query-row attention for gfx950: QK^T * scale -> length mask -> softmax -> PV in
// one launch over a PRE-ALLOCATED KV cache, replacing the reference's per-step
// Concat(cache) + MatMul + SoftMax + MatMul chain (src/layers/attention.cc:541-546, 216-277)
// and ops::Tile head replication for GQA (attention.cc:291-295).
//
// One wave64 per (row, head). A key row of `dhead` floats is covered by G = dhead/4 lanes
// reading 16 bytes each (coalesced 1 KiB per wave-instruction for dhead = 64); 64/G keys are
// processed per iteration. Scores live in LDS; softmax is a wave64 shuffle reduction;
// PV accumulates float4 per lane and is folded across the key groups with xor-shuffles.
// HBM-bound: K and V are each read exactly once.
//
// Float formulas follow the reference CPU path: scores = alpha * (q . k)
// (ops::MatMul alpha = queries_scale, attention.cc:216-217), softmax = exp(x - max) * (1/sum)
// (src/cpu/kernels.cc:452-455), masked positions contribute exact 0.

#include "device_utils.h"
#include "kernels.h"

#include <float.h>

namespace ct2x {

namespace {

constexpr int kUnroll = 8;  // keys per lane group and block; a block is one memory round trip

// Loads 4 consecutive elements [i0, i0+4) of a head vector, applying RoPE at `pos` when
// requested (reference formula src/ops/rotary_cpu.cc:31-36).
__device__ __forceinline__ float4 load4_rope(const float* __restrict__ base, int i0,
                                             const float* __restrict__ sin_t,
                                             const float* __restrict__ cos_t, int rot_dim,
                                             bool interleave) {
  float4 v = *reinterpret_cast<const float4*>(base + i0);
  if (sin_t == nullptr || i0 >= rot_dim) return v;
  float x[4] = {v.x, v.y, v.z, v.w};
  float y[4];
  const int middle = rot_dim / 2;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int i = i0 + e;
    if (i < rot_dim) {
      float other;
      if (interleave)
        other = (i % 2 == 0) ? -base[i + 1] : base[i - 1];
      else
        other = (i < middle) ? -base[i + middle] : base[i - middle];
      y[e] = x[e] * cos_t[i] + other * sin_t[i];
    } else {
      y[e] = x[e];
    }
  }
  return make_float4(y[0], y[1], y[2], y[3]);
}

__device__ __forceinline__ float dot4(float4 a, float4 b) {
  return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
}

struct AttnKernelArgs {
  AttnDecodeParams p;
  int causal_tq;     // >0: rows are (batch, query i = row % causal_tq); keys limited to i+q_offset+1
  int64_t q_offset;
  int mask_tq;       // >0: rows are (batch, query) pairs without causality (encoder)
  // kFusedQ outputs: the quantised context row and its scale
  int8_t* q_out;
  float* q_scales;
  int quant_mode;
};

// kFusedQ = false: one wave64 workgroup per (row, head), fp32 context out.
// kFusedQ = true : one workgroup per ROW with one wave per head; after PV the waves exchange their
//                  |max| through LDS and emit the row already quantised for the following int8
//                  Dense (ops::Quantize of the combined heads, src/layers/common.cc:389), so the
//                  standalone quantize launch and the fp32 context round trip disappear.
// kMaxChunks = float4 chunks of a head vector per lane (dhead <= 4 * G * kMaxChunks).
template <int G, bool kFusedQ, int kMaxChunks = 1>
__global__ __launch_bounds__(kFusedQ ? 512 : 64) void attn_row_kernel(AttnKernelArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const AttnDecodeParams& p = a.p;
  constexpr int KPI = 64 / G;  // keys per iteration
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int kg = lane / G, c = lane % G;
  const int64_t row = kFusedQ ? static_cast<int64_t>(blockIdx.x) : blockIdx.x / p.heads;
  const int h = kFusedQ ? wave : static_cast<int>(blockIdx.x % p.heads);
  float* scores = smem + (kFusedQ ? static_cast<int64_t>(wave) * (p.t_max + 1) : 0);
  const int group = static_cast<int>(p.heads / p.heads_kv);
  const int hk = h / group;
  const int64_t kvrow = row / p.kv_group;
  const int nc = static_cast<int>(p.dhead / 4);

  int64_t t_cached;
  if (p.lengths != nullptr)
    t_cached = p.lengths[kvrow];
  else if (p.step_ptr != nullptr)
    t_cached = *p.step_ptr;
  else
    t_cached = p.length;
  if (a.causal_tq > 0) {
    const int64_t i = row % a.causal_tq;
    const int64_t lim = i + a.q_offset + 1;
    if (lim < t_cached) t_cached = lim;
  }
  if (t_cached > p.t_max) t_cached = p.t_max;
  if (t_cached < 0) t_cached = 0;
  const int64_t pos = t_cached;  // position of the token being decoded (self-attention)
  const bool has_new = (p.k_new != nullptr);
  const int64_t t_total = t_cached + (has_new ? 1 : 0);

  const float* sin_t = p.rope_sin ? p.rope_sin + pos * p.rot_dim : nullptr;
  const float* cos_t = p.rope_cos ? p.rope_cos + pos * p.rot_dim : nullptr;
  const int rot_dim = static_cast<int>(p.rot_dim);

  // ---- q (and the new k / v) into registers -------------------------------------------
  const float* qh = p.q + row * p.q_row_stride + static_cast<int64_t>(h) * p.dhead;
  float4 qv[kMaxChunks], knew[kMaxChunks], vnew[kMaxChunks];
#pragma unroll
  for (int u = 0; u < kMaxChunks; ++u) {
    const int cc = c + u * G;
    qv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    knew[u] = qv[u];
    vnew[u] = qv[u];
    if (cc < nc) {
      qv[u] = load4_rope(qh, cc * 4, sin_t, cos_t, rot_dim, p.rope_interleave);
      if (has_new) {
        const float* kn = p.k_new + row * p.new_row_stride + static_cast<int64_t>(hk) * p.dhead;
        const float* vn = p.v_new + row * p.new_row_stride + static_cast<int64_t>(hk) * p.dhead;
        knew[u] = load4_rope(kn, cc * 4, sin_t, cos_t, rot_dim, p.rope_interleave);
        vnew[u] = *reinterpret_cast<const float4*>(vn + cc * 4);
      }
    }
  }

  float* kc = p.k_cache + kvrow * p.kv_row_stride + static_cast<int64_t>(hk) * p.kv_head_stride;
  float* vc = p.v_cache + kvrow * p.kv_row_stride + static_cast<int64_t>(hk) * p.kv_head_stride;

  // Cache append (in place of ops::Concat): the first query head of each KV head owns it.
  if (has_new && (h % group) == 0 && kg == 0) {
#pragma unroll
    for (int u = 0; u < kMaxChunks; ++u) {
      const int cc = c + u * G;
      if (cc < nc) {
        *reinterpret_cast<float4*>(kc + pos * p.kv_time_stride + cc * 4) = knew[u];
        *reinterpret_cast<float4*>(vc + pos * p.kv_time_stride + cc * 4) = vnew[u];
      }
    }
  }

  // Latency-bound at decode sizes, so everything that does not depend on the scores is requested
  // up front: a block is KPI * kUnroll keys, two K blocks AND two V blocks are in flight before
  // the first score is computed (decode lengths <= 2 blocks cost ONE memory round trip for the
  // whole cache instead of one per phase and block); longer rows refill a block as soon as it
  // has been consumed.
  constexpr int BLK = KPI * kUnroll;
  auto load_blk = [&](const float* base, int64_t t0, float4 (&reg)[kUnroll][kMaxChunks]) {
#pragma unroll
    for (int j = 0; j < kUnroll; ++j) {
      const int64_t t = t0 + j * KPI + kg;
#pragma unroll
      for (int u = 0; u < kMaxChunks; ++u) {
        const int cc = c + u * G;
        reg[j][u] = (t < t_cached && cc < nc)
                        ? *reinterpret_cast<const float4*>(base + t * p.kv_time_stride + cc * 4)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  float4 k0[kUnroll][kMaxChunks], k1[kUnroll][kMaxChunks];
  float4 v0[kUnroll][kMaxChunks], v1[kUnroll][kMaxChunks];
  load_blk(kc, 0, k0);
  load_blk(kc, BLK, k1);
  load_blk(vc, 0, v0);
  load_blk(vc, BLK, v1);

  // ---- phase 1: scores = scale * (q . k_t) ----------------------------------------------
  auto score_blk = [&](int64_t t0, const float4 (&reg)[kUnroll][kMaxChunks]) {
#pragma unroll
    for (int j = 0; j < kUnroll; ++j) {
      const int64_t t = t0 + j * KPI + kg;
      float part = 0.f;
#pragma unroll
      for (int u = 0; u < kMaxChunks; ++u) part += dot4(qv[u], reg[j][u]);
#pragma unroll
      for (int o = G / 2; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
      if (c == 0 && t < t_cached) scores[t] = part * p.scale;
    }
  };
  for (int64_t t0 = 0; t0 < t_cached; t0 += 2 * BLK) {
    score_blk(t0, k0);
    if (t0 + 2 * BLK < t_cached) load_blk(kc, t0 + 2 * BLK, k0);
    if (t0 + BLK < t_cached) score_blk(t0 + BLK, k1);
    if (t0 + 3 * BLK < t_cached) load_blk(kc, t0 + 3 * BLK, k1);
  }
  if (has_new) {
    float part = 0.f;
#pragma unroll
    for (int u = 0; u < kMaxChunks; ++u) part += dot4(qv[u], knew[u]);
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
    if (lane == 0) scores[t_cached] = part * p.scale;
  }
  __syncthreads();

  // ---- phase 2: softmax over [0, t_total) -------------------------------------------------
  // A fully masked row (t_total == 0) has an all-zero softmax: the loops below do not execute and
  // the context stays zero (no early return: the barriers must be reached by every wave).
  float m = -FLT_MAX;
  for (int64_t t = lane; t < t_total; t += 64) m = fmaxf(m, scores[t]);
  m = wave_max(m);
  float s = 0.f;
  for (int64_t t = lane; t < t_total; t += 64) {
    const float e = expf(scores[t] - m);
    scores[t] = e;
    s += e;
  }
  s = wave_sum(s);
  const float r = (t_total > 0) ? 1.f / s : 0.f;
  for (int64_t t = lane; t < t_total; t += 64) scores[t] = scores[t] * r;
  __syncthreads();

  // ---- phase 3: out = sum_t p_t * v_t -------------------------------------------------------
  float4 acc[kMaxChunks];
#pragma unroll
  for (int u = 0; u < kMaxChunks; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  auto pv_blk = [&](int64_t t0, const float4 (&reg)[kUnroll][kMaxChunks]) {
#pragma unroll
    for (int j = 0; j < kUnroll; ++j) {
      const int64_t t = t0 + j * KPI + kg;
      const float pt = (t < t_cached) ? scores[t] : 0.f;
#pragma unroll
      for (int u = 0; u < kMaxChunks; ++u) {
        acc[u].x += pt * reg[j][u].x;
        acc[u].y += pt * reg[j][u].y;
        acc[u].z += pt * reg[j][u].z;
        acc[u].w += pt * reg[j][u].w;
      }
    }
  };
  for (int64_t t0 = 0; t0 < t_cached; t0 += 2 * BLK) {
    pv_blk(t0, v0);
    if (t0 + 2 * BLK < t_cached) load_blk(vc, t0 + 2 * BLK, v0);
    if (t0 + BLK < t_cached) pv_blk(t0 + BLK, v1);
    if (t0 + 3 * BLK < t_cached) load_blk(vc, t0 + 3 * BLK, v1);
  }
  if (has_new && kg == static_cast<int>(t_cached % KPI)) {
    const float pt = scores[t_cached];
#pragma unroll
    for (int u = 0; u < kMaxChunks; ++u) {
      acc[u].x += pt * vnew[u].x;
      acc[u].y += pt * vnew[u].y;
      acc[u].z += pt * vnew[u].z;
      acc[u].w += pt * vnew[u].w;
    }
  }
#pragma unroll
  for (int u = 0; u < kMaxChunks; ++u) {
#pragma unroll
    for (int o = G; o < 64; o <<= 1) {
      acc[u].x += __shfl_xor(acc[u].x, o, 64);
      acc[u].y += __shfl_xor(acc[u].y, o, 64);
      acc[u].z += __shfl_xor(acc[u].z, o, 64);
      acc[u].w += __shfl_xor(acc[u].w, o, 64);
    }
  }
  if (!kFusedQ) {
    float* out = p.out + row * (p.heads * p.dhead) + static_cast<int64_t>(h) * p.dhead;
    if (kg == 0) {
#pragma unroll
      for (int u = 0; u < kMaxChunks; ++u) {
        const int cc = c + u * G;
        if (cc < nc) *reinterpret_cast<float4*>(out + cc * 4) = acc[u];
      }
    }
    return;
  }
  // ---- fused ops::Quantize of the row [heads * dhead] (src/cpu/kernels.cc:577-620) ---------
  float amax = 0.f;
  if (kg == 0) {
#pragma unroll
    for (int u = 0; u < kMaxChunks; ++u) {
      const int cc = c + u * G;
      if (cc < nc)
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(acc[u].x), fabsf(acc[u].y)),
                                 fmaxf(fabsf(acc[u].z), fabsf(acc[u].w))));
    }
  }
  amax = wave_max(amax);
  float* wave_amax = smem + static_cast<int64_t>(p.heads) * (p.t_max + 1);
  __syncthreads();
  if (lane == 0) wave_amax[wave] = amax;
  __syncthreads();
  amax = wave_amax[0];
  for (int w = 1; w < static_cast<int>(p.heads); ++w) amax = fmaxf(amax, wave_amax[w]);
  const float qscale = (amax != 0.f) ? 127.f / amax : 1.f;
  if (threadIdx.x == 0) a.q_scales[row] = qscale;
  if (kg == 0) {
    int8_t* qo = a.q_out + row * (p.heads * p.dhead) + static_cast<int64_t>(h) * p.dhead;
#pragma unroll
    for (int u = 0; u < kMaxChunks; ++u) {
      const int cc = c + u * G;
      if (cc < nc) {
        char4 o;
        o.x = to_s8(acc[u].x, qscale, a.quant_mode);
        o.y = to_s8(acc[u].y, qscale, a.quant_mode);
        o.z = to_s8(acc[u].z, qscale, a.quant_mode);
        o.w = to_s8(acc[u].w, qscale, a.quant_mode);
        *reinterpret_cast<char4*>(qo + cc * 4) = o;
      }
    }
  }
}

template <int G, int CH = 1>
void launch_rows(const AttnKernelArgs& a, hipStream_t stream) {
  const AttnDecodeParams& p = a.p;
  if (a.q_out != nullptr) {
    const size_t lds = (static_cast<size_t>(p.heads) * (p.t_max + 1) + p.heads) * sizeof(float);
    hipLaunchKernelGGL((attn_row_kernel<G, true, CH>), dim3(static_cast<unsigned>(p.rows)),
                       dim3(static_cast<unsigned>(p.heads * 64)), lds, stream, a);
    return;
  }
  const size_t lds = static_cast<size_t>(p.t_max + 1) * sizeof(float);
  hipLaunchKernelGGL((attn_row_kernel<G, false, CH>), dim3(static_cast<unsigned>(p.rows * p.heads)),
                     dim3(64), lds, stream, a);
}

void dispatch(const AttnKernelArgs& a, hipStream_t stream) {
  const int nc = static_cast<int>(a.p.dhead / 4);
  if (nc <= 1) launch_rows<1>(a, stream);
  else if (nc <= 2) launch_rows<2>(a, stream);
  else if (nc <= 4) launch_rows<4>(a, stream);
  else if (nc <= 8) launch_rows<8>(a, stream);
  else if (nc <= 16) launch_rows<16>(a, stream);
  else if (nc <= 32) launch_rows<32>(a, stream);
  else if (nc <= 64) launch_rows<64>(a, stream);
  else launch_rows<64, 2>(a, stream);
}

}  // namespace

void launch_attention_decode(const AttnDecodeParams& p, hipStream_t stream) {
  if (p.rows == 0) return;
  // (the d_head 128 form emits the producer-side row absmax itself)
  if ((p.row_amax_out == nullptr || p.dhead == 128) && attention_exact_decode_applicable(p, false)) {
    launch_attention_exact_decode(p, 0, nullptr, nullptr, stream);  // bit-exact vs the CPU reference
    return;
  }
  if (attention_decode2_applicable(p, false)) {
    launch_attention_decode2(p, 0, nullptr, nullptr, stream);
    return;
  }
  AttnKernelArgs a;
  a.p = p;
  a.causal_tq = 0;
  a.q_offset = 0;
  a.mask_tq = 0;
  a.q_out = nullptr;
  a.q_scales = nullptr;
  a.quant_mode = 0;
  dispatch(a, stream);
}

bool attention_decode_can_fuse_quantize(const AttnDecodeParams& p) {
  if (attention_exact_decode_applicable(p, true)) return true;
  // shapes the exact kernel serves unfused must not fall back to an inexact fused form
  if (attention_exact_decode_applicable(p, false)) return false;
  if (attention_decode2_applicable(p, true)) return true;
  // one wave per head in one workgroup, scores of every head in LDS
  return p.heads >= 1 && p.heads <= 8 &&
         (static_cast<size_t>(p.heads) * (p.t_max + 1) + p.heads) * sizeof(float) <= 60 * 1024;
}

void launch_attention_decode_quantized(const AttnDecodeParams& p, int quant_mode, int8_t* q_out,
                                       float* q_scales, hipStream_t stream) {
  if (p.rows == 0) return;
  if (attention_exact_decode_applicable(p, true)) {
    launch_attention_exact_decode(p, quant_mode, q_out, q_scales, stream);
    return;
  }
  if (attention_decode2_applicable(p, true)) {
    launch_attention_decode2(p, quant_mode, q_out, q_scales, stream);
    return;
  }
  AttnKernelArgs a;
  a.p = p;
  a.causal_tq = 0;
  a.q_offset = 0;
  a.mask_tq = 0;
  a.q_out = q_out;
  a.q_scales = q_scales;
  a.quant_mode = quant_mode;
  dispatch(a, stream);
}

// Prefill / encoder attention expressed on the same kernel: every (batch, query) pair is a
// "row" that shares its batch's K/V (kv_group = tq). Reference masks:
// MultiHeadAttention::prepare_length_mask (src/layers/attention_layer.cc:152-174):
// valid = min(length[b], q + 1) when causal.
void launch_attention_full(const AttnFullParams& f, hipStream_t stream) {
  if (f.batch * f.tq == 0) return;
  if (f.allow_inexact && attention_prefill_applicable(f)) {
    launch_attention_prefill(f, stream);
    return;
  }
  if (attention_exact_full_applicable(f)) {
    launch_attention_exact_full(f, stream);  // bit-exact vs the CPU reference (K / V of a head in LDS)
    return;
  }
  if (attention_seq_exact_applicable(f)) {
    launch_attention_seq_exact(f, stream);  // bit-exact at any length / head size (fp32 MFMA chains)
    return;
  }
  if (attention_prefill_applicable(f)) {
    launch_attention_prefill(f, stream);
    return;
  }
  AttnKernelArgs a;
  AttnDecodeParams& p = a.p;
  p.q = f.q;
  p.q_row_stride = f.q_time_stride;  // rows are contiguous (b, t) pairs: requires
                                     // q_batch_stride == tq * q_time_stride
  p.k_cache = const_cast<float*>(f.k);
  p.v_cache = const_cast<float*>(f.v);
  p.kv_row_stride = f.k_batch_stride;
  p.kv_head_stride = f.k_head_stride;
  p.kv_time_stride = f.k_time_stride;
  p.lengths = f.lengths;
  p.length = f.tk;
  p.rows = f.batch * f.tq;
  p.heads = f.heads;
  p.heads_kv = f.heads_kv;
  p.dhead = f.dhead;
  p.kv_group = f.tq;
  p.t_max = f.tk;
  p.scale = f.scale;
  p.out = f.out;
  a.causal_tq = f.causal ? static_cast<int>(f.tq) : 0;
  a.q_offset = f.q_offset;
  a.mask_tq = static_cast<int>(f.tq);
  a.q_out = nullptr;
  a.q_scales = nullptr;
  a.quant_mode = 0;
  dispatch(a, stream);
}

// ---------------------------------------------------------------------------------------
// Prefill: scatter a fused QKV projection into the rotated query buffer and the KV caches.
// ---------------------------------------------------------------------------------------
__global__ void qkv_to_cache_kernel(const float* __restrict__ qkv, int64_t time, int64_t heads,
                                    int64_t heads_kv, int64_t dhead, int64_t t_max, int64_t pos0,
                                    const float* __restrict__ rope_sin,
                                    const float* __restrict__ rope_cos, int64_t rot_dim,
                                    bool interleave, float* __restrict__ q_out,
                                    float* __restrict__ cache_k, float* __restrict__ cache_v) {
  const int64_t row = blockIdx.x;  // (b, t)
  const int64_t b = row / time, t = row % time;
  const int64_t pos = pos0 + t;
  const int64_t qdim = heads * dhead, kvdim = heads_kv * dhead;
  const float* src = qkv + row * (qdim + 2 * kvdim);
  const float* s = rope_sin ? rope_sin + pos * rot_dim : nullptr;
  const float* c = rope_cos ? rope_cos + pos * rot_dim : nullptr;
  const int64_t middle = rot_dim / 2;
  // q heads then k heads: same rotation (reference formula src/ops/rotary_cpu.cc:31-36)
  for (int64_t e = threadIdx.x; e < qdim + kvdim; e += blockDim.x) {
    const bool is_q = e < qdim;
    const int64_t local = is_q ? e : e - qdim;
    const int64_t h = local / dhead, i = local % dhead;
    const float* x = src + (is_q ? 0 : qdim) + h * dhead;
    float v = x[i];
    if (s != nullptr && i < rot_dim) {
      float other;
      if (interleave)
        other = (i % 2 == 0) ? -x[i + 1] : x[i - 1];
      else
        other = (i < middle) ? -x[i + middle] : x[i - middle];
      v = x[i] * c[i] + other * s[i];
    }
    if (is_q)
      q_out[row * qdim + e] = v;
    else
      cache_k[((b * heads_kv + h) * t_max + pos) * dhead + i] = v;
  }
  for (int64_t e = threadIdx.x; e < kvdim; e += blockDim.x) {
    const int64_t h = e / dhead, i = e % dhead;
    cache_v[((b * heads_kv + h) * t_max + pos) * dhead + i] = src[qdim + kvdim + e];
  }
}

void launch_qkv_to_cache(const float* qkv, int64_t batch, int64_t time, int64_t heads,
                         int64_t heads_kv, int64_t dhead, int64_t t_max, int64_t pos0,
                         const float* rope_sin, const float* rope_cos, int64_t rot_dim,
                         bool interleave, float* q_out, float* cache_k, float* cache_v,
                         hipStream_t stream) {
  if (batch * time == 0) return;
  hipLaunchKernelGGL(qkv_to_cache_kernel, dim3(static_cast<unsigned>(batch * time)), dim3(256), 0,
                     stream, qkv, time, heads, heads_kv, dhead, t_max, pos0, rope_sin, rope_cos,
                     rot_dim, interleave, q_out, cache_k, cache_v);
}

// ---------------------------------------------------------------------------------------
// Beam reorder of the pre-allocated self-attention caches (Decoder::update_state ->
// ops::Gather on every cache tensor, src/layers/decoder.cc:33-56): only the `len` live
// positions of each (row, head) are moved, src -> dst (ping-pong buffers).
// ---------------------------------------------------------------------------------------
__global__ void cache_gather_kernel(const float* const* __restrict__ src,
                                    float* const* __restrict__ dst,
                                    const int32_t* __restrict__ parent, int64_t heads,
                                    int64_t t_max, int64_t dhead,
                                    const int32_t* __restrict__ step_ptr, int64_t length) {
  const int64_t r = blockIdx.x / heads;
  const int64_t h = blockIdx.x % heads;
  const int t = blockIdx.y;
  const int64_t len = step_ptr ? *step_ptr : length;
  const int64_t n4 = len * dhead / 4;
  const float4* s = reinterpret_cast<const float4*>(
      src[t] + (static_cast<int64_t>(parent[r]) * heads + h) * t_max * dhead);
  float4* d = reinterpret_cast<float4*>(dst[t] + (r * heads + h) * t_max * dhead);
  for (int64_t i = threadIdx.x; i < n4; i += blockDim.x) d[i] = s[i];
}

void launch_cache_gather(const float* const* tensors_src, float* const* tensors_dst,
                         int n_tensors, const int32_t* parent, int64_t rows, int64_t heads,
                         int64_t t_max, int64_t dhead, const int32_t* step_ptr, int64_t length,
                         hipStream_t stream) {
  if (rows == 0 || n_tensors == 0) return;
  hipLaunchKernelGGL(cache_gather_kernel, dim3(static_cast<unsigned>(rows * heads), n_tensors),
                     dim3(128), 0, stream, tensors_src, tensors_dst, parent, heads, t_max, dhead,
                     step_ptr, length);
}

}  // namespace ct2x
