// Sequence-mode attention (encoder self-attention, prompt prefill) that reproduces the reference
// CPU path BIT FOR BIT at any head size and length - on the matrix cores.
//
// The reference (src/layers/attention.cc:178-287 on Device::CPU) runs three ops per (batch, head):
// MatMul(trans_b, alpha = 1 / sqrt(d_head)) -> SoftMax(lengths) -> MatMul, the products through
// MKL's sgemm. With several query rows MKL adds in k order, one FMA per element (recovered and
// pinned bit for bit against the reference library: oracle/ct2_oracle.c orc_matmul,
// tests/test_oracle.py::test_exact_matmul_sequence*, _k_blocking), with two size dependent details:
//   * alpha: small problems scale the finished sum; once MKL packs B (n >= 192, or n > m and
//     n >= max(k, 12)) the packed copy carries alpha: chain(a, RN(alpha * b));
//   * a long k of the second product (keys > 384) is cut into blocks (384 while more than 768
//     remain, then ceil / floor halves), every block a chain from zero, blocks added in order.
// gfx950's fp32 MFMA (v_mfma_f32_16x16x4_f32) IS such a chain: its four k steps accumulate in order
// with one rounding per FMA - measured bit-identical to fmaf chains on 768 000 outputs incl.
// cancellation and denormal cases (tools/mfma_chain_probe.cpp, profiles/r03_mfma_chain_probe.txt).
// So the exact order costs nothing here: both products run at the fp32 matrix rate.
//
// Three launches per call (scores live in an L2-sized workspace between them):
//   xseq_scores_kernel   S = Q K^T (+ alpha, either form), valid-key count per query row
//   softmax_kernel       rowops.hip: cpu::softmax<AVX2> bit for bit, in place, masked keys -> 0
//   xseq_context_kernel  O = P V with MKL's k partition
// A workgroup owns 64 query rows of one (batch, head): wave w the rows 16 w .. 16 w + 15.

#include "device_utils.h"
#include "kernels.h"
#include "runtime.h"

#include <algorithm>
#include <cstdlib>
#include <float.h>

namespace ct2x {

namespace {

constexpr int XS_BQ = 64;  // query rows per workgroup
constexpr int XS_BK = 64;  // keys per staged K tile (scores)
constexpr int XS_BV = 32;  // keys per staged P / V tile (context)

struct XSeqArgs {
  AttnFullParams p;
  float* scores;        // [batch, heads, tq, tk_pad]
  int32_t* row_len;     // [batch, heads, tq]
  int64_t tk_pad;
  int pack_b;           // alpha travels with K (MKL packed B) instead of scaling the sum
  int s_stride;         // fused SoftMax: LDS row stride of the score tile (floats)
};

// FUSED: the 64 x tk score tile stays in LDS, the SoftMax of its rows runs in the same launch
// (cpu::softmax<AVX2> in the reference's order: exp256_ps, eight strided partial sums, e * (1 / sum)
// - the sequence of attn_x_full_kernel / softmax_kernel) and only the probabilities reach the
// workspace. Rows longer than the LDS budget take the three-launch form.
template <int DH, bool FUSED>
__global__ __launch_bounds__(256) void xseq_scores_kernel(XSeqArgs a) {
  constexpr int QS = DH + 2;  // LDS row strides (floats): lanes (r, g) -> bank 2 r + g
  constexpr int KS = DH + 2;
  constexpr int NK = DH / 4;  // MFMA k steps
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* qs = smem;               // [64][QS]
  float* ks = qs + XS_BQ * QS;    // [64][KS]
  float* ss = ks + XS_BK * KS;    // FUSED: [64][a.s_stride] scores, then probabilities
  const AttnFullParams& p = a.p;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int q0 = blockIdx.x * XS_BQ, h = blockIdx.y, b = blockIdx.z;
  const int hk = h / static_cast<int>(p.heads / p.heads_kv);
  const int tq = static_cast<int>(p.tq), tk = static_cast<int>(p.tk);

  const float* qb = p.q + static_cast<int64_t>(b) * p.q_batch_stride + static_cast<int64_t>(h) * DH;
  const float* kb = p.k + static_cast<int64_t>(b) * p.k_batch_stride + static_cast<int64_t>(hk) * p.k_head_stride;
  const int64_t bh = static_cast<int64_t>(b) * p.heads + h;

  // valid keys of every query row (prepare_length_mask, src/layers/attention_layer.cc:152-174)
  const int len_b = p.lengths ? min(max(p.lengths[b], 0), tk) : tk;
  if (!FUSED && tid < XS_BQ && q0 + tid < tq) {
    int size = len_b;
    if (p.causal) size = min(size, q0 + tid + static_cast<int>(p.q_offset) + 1);
    a.row_len[bh * tq + q0 + tid] = max(size, 0);
  }
  // keys no row of this tile can see: their scores are never read (SoftMax writes 0 behind a row's
  // length), so their tiles are not computed
  const int k_hi = p.causal ? max(0, min(len_b, q0 + XS_BQ + static_cast<int>(p.q_offset))) : len_b;

  // Q tile -> LDS (rows past tq are clamped: their scores are never stored)
  for (int e = tid; e < XS_BQ * (DH / 4); e += 256) {
    const int row = e / (DH / 4), c4 = e % (DH / 4);
    const int qi = min(q0 + row, tq - 1);
    const v4f v = *reinterpret_cast<const v4f*>(qb + static_cast<int64_t>(qi) * p.q_time_stride + c4 * 4);
    float* d = qs + row * QS + c4 * 4;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
  __syncthreads();
  float af[NK];
#pragma unroll
  for (int j = 0; j < NK; ++j) af[j] = qs[(16 * wave + r) * QS + 4 * j + g];

  float* srow = a.scores + bh * tq * a.tk_pad;
  for (int t0 = 0; t0 < k_hi; t0 += XS_BK) {
    __syncthreads();  // the previous tile has been consumed
    for (int e = tid; e < XS_BK * (DH / 4); e += 256) {
      const int row = e / (DH / 4), c4 = e % (DH / 4);
      const int t = min(t0 + row, tk - 1);
      v4f v = *reinterpret_cast<const v4f*>(kb + static_cast<int64_t>(t) * p.k_time_stride + c4 * 4);
      if (a.pack_b) {  // MKL's packed copy of B carries alpha: RN(alpha * k)
        v.x = v.x * p.scale; v.y = v.y * p.scale; v.z = v.z * p.scale; v.w = v.w * p.scale;
      }
      float* d = ks + row * KS + c4 * 4;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    v4f acc[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[s] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NK; ++j) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float bv = ks[(16 * s + r) * KS + 4 * j + g];  // B[k = 4 j + g][key 16 s + r]
        acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bv, acc[s], 0, 0, 0);
      }
    }
    // D[i = 4 g + ii][key = 16 s + r]
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int t = t0 + 16 * s + r;
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        const int qi = q0 + 16 * wave + 4 * g + ii;
        if (qi < tq && t < tk) {
          float v = acc[s][ii];
          if (!a.pack_b && p.scale != 1.f) v = v * p.scale;
          if (FUSED) ss[(16 * wave + 4 * g + ii) * a.s_stride + t] = v;
          else srow[static_cast<int64_t>(qi) * a.tk_pad + t] = v;
        }
      }
    }
  }
  if (!FUSED) return;
  __syncthreads();
  // SoftMax of this wave's 16 rows, one after the other (lane = key)
  for (int rr = 0; rr < 16; ++rr) {
    const int qi = q0 + 16 * wave + rr;
    if (qi >= tq) break;
    int size = len_b;
    if (p.causal) size = min(size, qi + static_cast<int>(p.q_offset) + 1);
    size = max(size, 0);
    float* e = ss + (16 * wave + rr) * a.s_stride;
    float* out = srow + static_cast<int64_t>(qi) * a.tk_pad;
    float m = -FLT_MAX;
    for (int t = lane; t < size; t += 64) m = fmaxf(m, e[t]);
    m = wave_max_dpp(m);
    for (int t = lane; t < size; t += 64) e[t] = exp_avx2(e[t] - m);
    __builtin_amdgcn_wave_barrier();
    float rcp = 0.f;
    if (size > 0) rcp = 1.f / ordered_sum8(e, size);
    __builtin_amdgcn_wave_barrier();
    for (int t = lane; t < tk; t += 64) out[t] = (t < size) ? e[t] * rcp : 0.f;
  }
}

// MKL's partition of a long k (oracle/ct2_oracle.c orc_matmul, no-transpose arm).
__device__ __forceinline__ int xseq_block(int rem) {
  if (rem > 768) return 384;
  if (rem > 384) return (rem + 1) / 2;
  return rem;
}

template <int DH>
__global__ __launch_bounds__(256) void xseq_context_kernel(XSeqArgs a) {
  constexpr int PS = XS_BV + 2;   // P tile row stride: lanes (r, g) -> bank 2 r + g
  constexpr int VS = DH + 16;     // V tile row stride: the two k rows of a 32-lane group 16 banks apart
  constexpr int NC = DH / 16;     // channel tiles
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* ps = smem;               // [64][PS]
  float* vs = ps + XS_BQ * PS;    // [32][VS]
  const AttnFullParams& p = a.p;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int q0 = blockIdx.x * XS_BQ, h = blockIdx.y, b = blockIdx.z;
  const int hk = h / static_cast<int>(p.heads / p.heads_kv);
  const int tq = static_cast<int>(p.tq), tk = static_cast<int>(p.tk);
  const float* vb = p.v + static_cast<int64_t>(b) * p.v_batch_stride + static_cast<int64_t>(hk) * p.v_head_stride;
  const int64_t bh = static_cast<int64_t>(b) * p.heads + h;
  const float* prow = a.scores + bh * tq * a.tk_pad;

  // keys no row of this tile can see carry probability 0: fma(0, v, acc) == acc, their tiles are
  // skipped (exact: acc is never -0)
  const int len_b = p.lengths ? min(max(p.lengths[b], 0), tk) : tk;
  const int k_hi = p.causal ? max(0, min(len_b, q0 + XS_BQ + static_cast<int>(p.q_offset))) : len_b;
  v4f tot[NC], acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) tot[c] = acc[c] = v4f{0.f, 0.f, 0.f, 0.f};
  bool first = true;
  int s0 = 0, rem = tk;
  while (rem > 0) {
    const int blk = xseq_block(rem);
    const int e0 = s0 + blk;
    for (int t0 = s0; t0 < min(e0, k_hi); t0 += XS_BV) {
      __syncthreads();
      // P tile: keys outside the block enter as 0 (fma(0, v, acc) == acc: acc is never -0)
      for (int e = tid; e < XS_BQ * XS_BV; e += 256) {
        const int row = e / XS_BV, j = e % XS_BV;
        const int qi = min(q0 + row, tq - 1), t = t0 + j;
        ps[row * PS + j] = (t < e0) ? prow[static_cast<int64_t>(qi) * a.tk_pad + t] : 0.f;
      }
      for (int e = tid; e < XS_BV * (DH / 4); e += 256) {
        const int j = e / (DH / 4), c4 = e % (DH / 4);
        const int t = min(t0 + j, tk - 1);
        *reinterpret_cast<v4f*>(vs + j * VS + c4 * 4) =
            *reinterpret_cast<const v4f*>(vb + static_cast<int64_t>(t) * p.v_time_stride + c4 * 4);
      }
      __syncthreads();
      const int steps = min(XS_BV, e0 - t0);
      for (int j4 = 0; j4 < steps; j4 += 4) {
        const float pa = ps[(16 * wave + r) * PS + j4 + g];   // A[query r][key j4 + g]
        const float* vr = vs + (j4 + g) * VS + r;             // B[key j4 + g][channel 16 c + r]
#pragma unroll
        for (int c = 0; c < NC; ++c)
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa, vr[16 * c], acc[c], 0, 0, 0);
      }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (first) {
        tot[c] = acc[c];
      } else {
        tot[c].x = tot[c].x + acc[c].x; tot[c].y = tot[c].y + acc[c].y;
        tot[c].z = tot[c].z + acc[c].z; tot[c].w = tot[c].w + acc[c].w;
      }
      acc[c] = v4f{0.f, 0.f, 0.f, 0.f};
    }
    first = false;
    s0 = e0;
    rem -= blk;
  }

  const int64_t ots = p.out_time_stride ? p.out_time_stride : p.heads * DH;
  const int64_t obs = p.out_batch_stride ? p.out_batch_stride : p.tq * ots;
  float* ob = p.out + static_cast<int64_t>(b) * obs + static_cast<int64_t>(h) * DH;
#pragma unroll
  for (int ii = 0; ii < 4; ++ii) {
    const int qi = q0 + 16 * wave + 4 * g + ii;
    if (qi >= tq) continue;
#pragma unroll
    for (int c = 0; c < NC; ++c) ob[static_cast<int64_t>(qi) * ots + 16 * c + r] = tot[c][ii];
  }
}

bool xseq_disabled() {
  static const bool off = [] {
    const char* e = std::getenv("CT2X_FAST_ATTENTION");
    return e != nullptr && e[0] == '1';
  }();
  return off;
}

template <int DH>
void launch_xseq(const XSeqArgs& a, hipStream_t stream) {
  const AttnFullParams& p = a.p;
  const dim3 grid(static_cast<unsigned>((p.tq + XS_BQ - 1) / XS_BQ), static_cast<unsigned>(p.heads),
                  static_cast<unsigned>(p.batch));
  const size_t lds1 = sizeof(float) * (XS_BQ * (DH + 2) + XS_BK * (DH + 2));
  const size_t lds3 = sizeof(float) * (XS_BQ * (XS_BV + 2) + XS_BV * (DH + 16));
  const size_t lds_fused = lds1 + sizeof(float) * XS_BQ * static_cast<size_t>(a.s_stride);
  if (lds_fused <= 160 * 1024) {
    allow_big_lds(reinterpret_cast<const void*>(xseq_scores_kernel<DH, true>), lds_fused);
    hipLaunchKernelGGL((xseq_scores_kernel<DH, true>), grid, dim3(256), lds_fused, stream, a);
  } else {
    allow_big_lds(reinterpret_cast<const void*>(xseq_scores_kernel<DH, false>), lds1);
    hipLaunchKernelGGL((xseq_scores_kernel<DH, false>), grid, dim3(256), lds1, stream, a);
    const int64_t rows = p.batch * p.heads * p.tq;
    // SoftMax over the padded row: columns tk .. tk_pad - 1 sit behind every row's length -> 0
    launch_softmax(a.scores, a.row_len, rows, a.tk_pad, false, a.scores, stream);
  }
  hipLaunchKernelGGL(xseq_context_kernel<DH>, grid, dim3(256), lds3, stream, a);
}

}  // namespace

// Where alpha enters MKL's sequence-mode product (oracle/ct2_oracle.c orc_matmul).
bool mkl_sequence_packs_b(int64_t m, int64_t n, int64_t k, float alpha) {
  return alpha != 1.f && (n >= 192 || (n > m && n >= std::max<int64_t>(k, 12)));
}

// The envelope of the recovered order: at least three query rows and five keys, dot products
// shorter than 24 rows' worth (k < 24 m: below that MKL takes a kernel that was not recovered),
// and outside the narrow-k corner (k < 12 with 8 .. 11 keys) - the oracle's envelope.
bool attention_seq_exact_applicable(const AttnFullParams& p) {
  if (xseq_disabled()) return false;
  if (p.dhead != 16 && p.dhead != 32 && p.dhead != 64 && p.dhead != 128) return false;
  if (p.heads_kv < 1 || p.heads % p.heads_kv != 0) return false;
  if (p.tq < 3 || p.tk < 5 || p.dhead >= 24 * p.tq) return false;
  auto aligned16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  return aligned16(p.q) && aligned16(p.k) && aligned16(p.v) && p.q_time_stride % 4 == 0 &&
         p.q_batch_stride % 4 == 0 && p.k_time_stride % 4 == 0 && p.v_time_stride % 4 == 0 &&
         p.k_head_stride % 4 == 0 && p.v_head_stride % 4 == 0 && p.k_batch_stride % 4 == 0 &&
         p.v_batch_stride % 4 == 0;
}

void launch_attention_seq_exact(const AttnFullParams& p_in, hipStream_t stream) {
  if (p_in.batch * p_in.tq == 0) return;
  // Scores workspace from the stream-ordered caching allocator (sequence mode runs eagerly, never
  // inside a captured step); batches are processed in chunks of at most ~256 MB of scores.
  const int64_t tk_pad = (p_in.tk + 7) / 8 * 8;
  const int64_t per_batch = p_in.heads * p_in.tq * tk_pad;
  int64_t chunk = std::max<int64_t>(1, (int64_t(64) << 20) / std::max<int64_t>(per_batch, 1));
  chunk = std::min(chunk, p_in.batch);
  float* scores = static_cast<float*>(device_malloc(static_cast<size_t>(chunk * per_batch) * sizeof(float)));
  int32_t* row_len = static_cast<int32_t*>(
      device_malloc(static_cast<size_t>(chunk * p_in.heads * p_in.tq) * sizeof(int32_t)));
  const int64_t ots = p_in.out_time_stride ? p_in.out_time_stride : p_in.heads * p_in.dhead;
  const int64_t obs = p_in.out_batch_stride ? p_in.out_batch_stride : p_in.tq * ots;
  for (int64_t b0 = 0; b0 < p_in.batch; b0 += chunk) {
    XSeqArgs a;
    a.p = p_in;
    a.p.batch = std::min(chunk, p_in.batch - b0);
    a.p.q = p_in.q + b0 * p_in.q_batch_stride;
    a.p.k = p_in.k + b0 * p_in.k_batch_stride;
    a.p.v = p_in.v + b0 * p_in.v_batch_stride;
    a.p.lengths = p_in.lengths ? p_in.lengths + b0 : nullptr;
    a.p.out = p_in.out + b0 * obs;
    a.p.out_time_stride = ots;
    a.p.out_batch_stride = obs;
    a.scores = scores;
    a.row_len = row_len;
    a.tk_pad = tk_pad;
    a.pack_b = mkl_sequence_packs_b(p_in.tq, p_in.tk, p_in.dhead, p_in.scale) ? 1 : 0;
    a.s_stride = static_cast<int>(tk_pad + 4);
    switch (p_in.dhead) {
      case 16: launch_xseq<16>(a, stream); break;
      case 32: launch_xseq<32>(a, stream); break;
      case 64: launch_xseq<64>(a, stream); break;
      default: launch_xseq<128>(a, stream); break;
    }
  }
  if (stream != active_stream()) CT2X_HIP_CHECK(hipStreamSynchronize(stream));
  device_free(row_len);
  device_free(scores);
}

}  // namespace ct2x
