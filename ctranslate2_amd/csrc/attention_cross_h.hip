// Encoder-decoder attention of a decode step for the 16-bit compute types, d_head 64 / 128 (round 4:
// 128 for the NLLB-3.3B geometry, whose per-(row, head) fp32 kernel read every entry's 32 memory
// positions once per beam: 27 us per layer), with the
// memory keys / values held in the compute type (float16 / bfloat16) - the first 16-bit cache of
// this backend. MultiHeadAttention::operator() with cached memory keys
// (src/layers/attention.cc:442-615; process_cross_attention :371-440): for every hypothesis row
//   softmax(q . K^T * queries_scale, memory_lengths) . V.
//
// At the Whisper-large-v3 shape (1500 memory positions, 20 heads, beam 5, 8 windows) the fp32 memory
// K/V is 123 MB per layer and 3.9 GB per decode step - more than the 1.7 GB of bfloat16 weights -
// and the per-(row, head) kernel reads it once per hypothesis row. Here ONE workgroup owns a
// (batch entry, head): the `beam` hypothesis rows of the entry share every key and value it reads
// (5x fewer L2 reads), and the bytes are halved.
//   scores   thread = key: the 128-byte key row is eight 16-byte loads (one cache line), dot
//            products against all hypothesis queries (LDS broadcasts), scores parked in LDS;
//   softmax  maxima / sums of all hypotheses in one LDS exchange each, exp(x - max) as in
//            src/cpu/kernels.cc:452-455, normalisation folded into the output;
//   context  a wave takes tiles of 8 keys, eight tiles in flight: 8 lanes per value row (16 bytes =
//            8 channels each), probabilities from LDS, partial sums over the lane's keys in
//            registers, then a 3-step butterfly over the 8 key lanes and an 8-wave sum through LDS.

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>

#include "device_utils.h"
#include "kernels.h"
#include "runtime.h"

namespace ct2x {

namespace {

constexpr int kXhMaxBeam = 8;

template <bool kBF16>
__device__ __forceinline__ void unpack8(const uint4& u, float* f) {
  const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if constexpr (kBF16) {
      f[2 * i] = __uint_as_float(w[i] << 16);
      f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    } else {
      const _Float16 lo = __builtin_bit_cast(_Float16, static_cast<uint16_t>(w[i] & 0xffffu));
      const _Float16 hi = __builtin_bit_cast(_Float16, static_cast<uint16_t>(w[i] >> 16));
      f[2 * i] = static_cast<float>(lo);
      f[2 * i + 1] = static_cast<float>(hi);
    }
  }
}

constexpr int kXhWaves = 8;          // 512 threads per workgroup
constexpr int kXhThreads = kXhWaves * 64;
constexpr int kXhUnroll = 8;         // value tiles in flight per wave

template <bool kBF16, int DH>
__global__ __launch_bounds__(kXhThreads) void xattn_h_kernel(CrossAttnHParams p) {
  constexpr int CG = DH / 8;      // 16-byte channel groups of a key / value row
  constexpr int KT = 64 / CG;     // keys per wave tile of the context phase
  extern __shared__ __attribute__((aligned(16))) float xh_lds[];
  __shared__ float red[kXhWaves][kXhMaxBeam];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / static_cast<int>(p.heads);
  const int h = blockIdx.x % static_cast<int>(p.heads);
  const int hk = h / static_cast<int>(p.heads / p.heads_kv);
  const int beam = static_cast<int>(p.beam);
  const int S = static_cast<int>(p.s_max);
  const int s_pad = (S + 7) & ~7;
  int length = p.lengths ? p.lengths[b] : S;
  length = length < 0 ? 0 : (length > S ? S : length);

  float* q_s = xh_lds;                               // [kXhMaxBeam][DH]
  float* probs = q_s + kXhMaxBeam * DH;              // [kXhMaxBeam][s_pad]
  float* part = probs + kXhMaxBeam * s_pad;          // [kXhWaves][kXhMaxBeam][DH]

  for (int e = tid; e < beam * DH; e += kXhThreads) {
    const int j = e / DH, c = e % DH;
    const int64_t row = static_cast<int64_t>(b) * beam + j;
    q_s[j * DH + c] = p.q[row * p.q_row_stride + static_cast<int64_t>(h) * DH + c] * p.scale;
  }
  __syncthreads();

  const int64_t kv_time = 2 * p.heads_kv * DH;       // elements per memory position
  const uint16_t* kbase = p.kv + static_cast<int64_t>(b) * S * kv_time + static_cast<int64_t>(hk) * DH;
  const uint16_t* vbase = kbase + p.heads_kv * DH;

  // ---- scores: thread = key (the 128-byte key row = one cache line, eight 16-byte loads in
  // flight; measured faster than 8 lanes per key with DPP folds: 1096 vs 1015 tokens/s on the
  // Whisper-large-v3 shape, profiles/r03_whisper_cross_kv16_ab.txt)
  float mx[kXhMaxBeam];
#pragma unroll
  for (int j = 0; j < kXhMaxBeam; ++j) mx[j] = -FLT_MAX;
  for (int s = tid; s < s_pad; s += kXhThreads) {
    float sc[kXhMaxBeam];
#pragma unroll
    for (int j = 0; j < kXhMaxBeam; ++j) sc[j] = 0.f;
    if (s < length) {
      const uint4* kr = reinterpret_cast<const uint4*>(kbase + static_cast<int64_t>(s) * kv_time);
      uint4 ku[CG];
#pragma unroll
      for (int c8 = 0; c8 < CG; ++c8) ku[c8] = kr[c8];
#pragma unroll
      for (int c8 = 0; c8 < CG; ++c8) {
        float kf[8];
        unpack8<kBF16>(ku[c8], kf);
#pragma unroll
        for (int j = 0; j < kXhMaxBeam; ++j) {
          if (j < beam) {
            const float4 qa = *reinterpret_cast<const float4*>(q_s + j * DH + c8 * 8);
            const float4 qb = *reinterpret_cast<const float4*>(q_s + j * DH + c8 * 8 + 4);
            sc[j] += kf[0] * qa.x + kf[1] * qa.y + kf[2] * qa.z + kf[3] * qa.w + kf[4] * qb.x +
                     kf[5] * qb.y + kf[6] * qb.z + kf[7] * qb.w;
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < kXhMaxBeam; ++j) {
      if (j < beam) {
        const float v = s < length ? sc[j] : -FLT_MAX;
        probs[j * s_pad + s] = v;
        mx[j] = fmaxf(mx[j], v);
      }
    }
  }
  // ---- softmax: one exchange for the maxima of all hypotheses, one for the sums ----------------
#pragma unroll
  for (int j = 0; j < kXhMaxBeam; ++j) {
    if (j < beam) {
      const float v = wave_max(mx[j]);
      if (lane == 0) red[wave][j] = v;
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kXhMaxBeam; ++j) {
    if (j < beam) {
      float m = red[0][j];
#pragma unroll
      for (int w = 1; w < kXhWaves; ++w) m = fmaxf(m, red[w][j]);
      mx[j] = m;
    }
  }
  __syncthreads();  // every thread has read the maxima: `red` is reused for the sums
  float sum[kXhMaxBeam];
#pragma unroll
  for (int j = 0; j < kXhMaxBeam; ++j) sum[j] = 0.f;
  for (int s = tid; s < s_pad; s += kXhThreads) {  // (scores are visible: barrier of the exchange)
#pragma unroll
    for (int j = 0; j < kXhMaxBeam; ++j) {
      if (j < beam) {
        const float e = s < length ? expf(probs[j * s_pad + s] - mx[j]) : 0.f;
        probs[j * s_pad + s] = e;
        sum[j] += e;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < kXhMaxBeam; ++j) {
    if (j < beam) {
      const float v = wave_sum(sum[j]);
      if (lane == 0) red[wave][j] = v;
    }
  }
  __syncthreads();  // sums exchanged, probabilities visible to every wave
  float inv_sum[kXhMaxBeam];
#pragma unroll
  for (int j = 0; j < kXhMaxBeam; ++j) {
    inv_sum[j] = 0.f;
    if (j < beam) {
      float t = red[0][j];
#pragma unroll
      for (int w = 1; w < kXhWaves; ++w) t += red[w][j];
      inv_sum[j] = t > 0.f ? 1.f / t : 0.f;
    }
  }

  // ---- context: a wave takes tiles of KT keys, kXhUnroll tiles in flight --------------------------
  const int kr = lane / CG, cg = lane % CG;  // key within the tile, channel group of 8
  const int tiles = s_pad / KT;
  float acc[kXhMaxBeam][8];
#pragma unroll
  for (int j = 0; j < kXhMaxBeam; ++j)
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[j][c] = 0.f;
  for (int t0 = wave; t0 < tiles; t0 += kXhWaves * kXhUnroll) {
    uint4 vu[kXhUnroll];
    int vs[kXhUnroll];
#pragma unroll
    for (int k = 0; k < kXhUnroll; ++k) {
      const int t = t0 + k * kXhWaves;
      const int s = t * KT + kr;
      vs[k] = (t < tiles && s < length) ? s : -1;
      vu[k] = make_uint4(0u, 0u, 0u, 0u);
      if (vs[k] >= 0)
        vu[k] = *reinterpret_cast<const uint4*>(vbase + static_cast<int64_t>(s) * kv_time + cg * 8);
    }
#pragma unroll
    for (int k = 0; k < kXhUnroll; ++k) {
      if (vs[k] >= 0) {
        float vf[8];
        unpack8<kBF16>(vu[k], vf);
#pragma unroll
        for (int j = 0; j < kXhMaxBeam; ++j) {
          if (j < beam) {
            const float pj = probs[j * s_pad + vs[k]];
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[j][c] += pj * vf[c];
          }
        }
      }
    }
  }
  // sum over the KT key lanes of a channel group (the lane bits above the channel group's)
#pragma unroll
  for (int j = 0; j < kXhMaxBeam; ++j) {
    if (j < beam) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        float v = acc[j][c];
        if (CG <= 8) v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        acc[j][c] = v;
      }
      if (kr == 0) {
#pragma unroll
        for (int c = 0; c < 8; ++c) part[(wave * kXhMaxBeam + j) * DH + cg * 8 + c] = acc[j][c];
      }
    }
  }
  __syncthreads();
  for (int e = tid; e < beam * DH; e += kXhThreads) {
    const int j = e / DH, c = e % DH;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < kXhWaves; ++w) v += part[(w * kXhMaxBeam + j) * DH + c];
    float inv = 0.f;
#pragma unroll
    for (int jj = 0; jj < kXhMaxBeam; ++jj)
      if (jj == j) inv = inv_sum[jj];
    const int64_t row = static_cast<int64_t>(b) * beam + j;
    p.out[row * p.out_row_stride + static_cast<int64_t>(h) * DH + c] = v * inv;
  }
}

size_t xattn_h_lds_bytes(int64_t s_max, int64_t dhead) {
  const int64_t s_pad = (s_max + 7) & ~static_cast<int64_t>(7);
  return static_cast<size_t>(kXhMaxBeam * dhead + kXhMaxBeam * s_pad + kXhWaves * kXhMaxBeam * dhead) *
         sizeof(float);
}

}  // namespace

bool cross_attention_h_applicable(const CrossAttnHParams& p) {
  return (p.dhead == 64 || p.dhead == 128) && p.beam >= 1 && p.beam <= kXhMaxBeam && p.heads_kv >= 1 &&
         p.heads % p.heads_kv == 0 && p.s_max >= 1 &&
         xattn_h_lds_bytes(p.s_max, p.dhead) <= 150 * 1024 &&
         (reinterpret_cast<uintptr_t>(p.kv) & 15) == 0;
}

void launch_cross_attention_h(const CrossAttnHParams& p, hipStream_t stream) {
  if (p.batch * p.heads == 0) return;
  if (!cross_attention_h_applicable(p)) throw Error("launch_cross_attention_h: shape not supported");
  const size_t lds = xattn_h_lds_bytes(p.s_max, p.dhead);
  const dim3 grid(static_cast<unsigned>(p.batch * p.heads));
#define CT2X_XH(BF, DH)                                                                          \
  do {                                                                                            \
    allow_big_lds(reinterpret_cast<const void*>(xattn_h_kernel<BF, DH>), lds);                    \
    hipLaunchKernelGGL((xattn_h_kernel<BF, DH>), grid, dim3(kXhThreads), lds, stream, p);         \
  } while (0)
  if (p.dhead == 64) {
    if (p.bf16) CT2X_XH(true, 64);
    else CT2X_XH(false, 64);
  } else {
    if (p.bf16) CT2X_XH(true, 128);
    else CT2X_XH(false, 128);
  }
#undef CT2X_XH
}

}  // namespace ct2x
