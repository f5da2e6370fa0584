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

void launch_attention_prefill(const AttnFullParams& p_in, hipStream_t stream) {
  AttnFullParams p = p_in;
  // the output is [B, Tq, H * dh] contiguous unless the caller gives strides
  if (p.out_time_stride == 0) p.out_time_stride = p.heads * p.dhead;
  if (p.out_batch_stride == 0) p.out_batch_stride = p.tq * p.out_time_stride;
  const dim3 grid(static_cast<unsigned>((p.tq + PF_BQ - 1) / PF_BQ), static_cast<unsigned>(p.heads),
                  static_cast<unsigned>(p.batch));
  if (p.dhead == 64)
    hipLaunchKernelGGL(attn_prefill_kernel<64>, grid, dim3(256), 0, stream, p);
  else
    hipLaunchKernelGGL(attn_prefill_kernel<128>, grid, dim3(256), 0, stream, p);
}

}  // namespace ct2x
