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
//   scores   (round 5) S^T = K Q^T on the 16-bit matrix cores per tile of 16 keys: the key rows as
//            they lie in memory are the A operand, Q^T (hypothesis rows as columns) split into
//            16-bit terms of 22 / 24 bits is the B operand; scores parked in LDS. (Rounds 3 - 4:
//            thread = key, dot products on the vector pipe against LDS broadcasts of the queries -
//            80 ds_read_b128 per key, 25k of the launch's 72k shader clocks.)
//   softmax  maxima / sums of all hypotheses in one LDS exchange each, exp(x - max) as in
//            src/cpu/kernels.cc:452-455, normalisation folded into the output; the statistics wait
//            in LDS while the context phase has the registers;
//   context  a wave takes tiles of 8 keys, eight tiles in flight - the first batch requested
//            before the softmax -: 8 ADJACENT lanes per value row (16 bytes = 8 channels each),
//            probabilities from LDS, partial sums over the lane's keys in registers, three DPP adds
//            over the 8 key lanes and an 8-wave sum through LDS.
//
// Round 5 - memory split over workgroups: 8 windows x 20 heads are 160 workgroups on 256 CUs, each
// walking 1500 positions (39 us per layer, 1.6 TB/s: the largest kernel of the Whisper decode step).
// With a scratch buffer the memory of a (batch entry, head) is cut into `splits` chunks of whole
// 8-key tiles, one workgroup each - as many chunks as keep every workgroup RESIDENT (two per CU at
// <= 128 VGPRs: register arrays sized by a beam bucket, `amdgpu_waves_per_eu`; at 160 VGPRs and one
// workgroup per CU five chunks ran in four rounds and took 68 us). A chunk publishes (max, sum,
// unnormalised context) per hypothesis row with write-through stores, takes a ticket, and the LAST
// arrival of the (entry, head) merges the chunks
//   out = sum_c exp(m_c - M) o_c / sum_c exp(m_c - M) l_c
// and puts the ticket back to zero (grid_sync.h: no workgroup waits, so no residency REQUIREMENT).
// 39.4 -> 24.9 us per layer (profiles/r05_xattn_probe.txt, r05_kernel_stats_bench_whisper_b.csv).

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cstdint>
#include <cstdlib>

#include "device_utils.h"
#include "grid_sync.h"
#include "kernels.h"
#include "runtime.h"

namespace ct2x {

namespace {

constexpr int kXhMaxBeam = 8;

// TUNING builds: shader-clock stamps of workgroup 0 at its phase boundaries (tools/xattn_probe.py)
#ifdef CT2X_TUNING
__device__ unsigned long long g_xh_stamps[16];
#define CT2X_XH_STAMP(i)                                                              \
  do {                                                                                \
    if (blockIdx.x == 0 && threadIdx.x == 0) g_xh_stamps[i] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#define CT2X_XH_STAMP_ANY(i)                                                          \
  do {                                                                                \
    if (blockIdx.x / p.splits == 0 && threadIdx.x == 0) g_xh_stamps[i] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define CT2X_XH_STAMP(i)
#define CT2X_XH_STAMP_ANY(i)
#endif

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

typedef _Float16 xh_v8h __attribute__((ext_vector_type(8)));
typedef __bf16 xh_v8b __attribute__((ext_vector_type(8)));
template <bool kBF16>
struct XhVec;
template <>
struct XhVec<false> {
  using vec = xh_v8h;
  using elem = _Float16;
  __device__ static __forceinline__ v4f mfma(vec a, vec b, v4f c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};
template <>
struct XhVec<true> {
  using vec = xh_v8b;
  using elem = __bf16;
  __device__ static __forceinline__ v4f mfma(vec a, vec b, v4f c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};

// sum over KT adjacent lanes (KT = 4 or 8) on the vector pipe: xor 1, xor 2 (quad_perm), then the
// mirror of each half row - every lane of the group ends with the total
template <int KT>
__device__ __forceinline__ float sum_adjacent(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));  // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));  // quad_perm [2,3,0,1]
  if (KT == 8) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false));  // row_half_mirror
  return v;
}

constexpr int kXhWaves = 8;          // 512 threads per workgroup
constexpr int kXhThreads = kXhWaves * 64;
constexpr int kXhUnroll = 8;         // value tiles in flight per wave

// MB: register arrays are sized for the hypothesis rows the launch can have (beam <= MB)
template <bool kBF16, int DH, int MB>
__global__ __launch_bounds__(kXhThreads) __attribute__((amdgpu_waves_per_eu(MB <= 5 ? 4 : 2))) void xattn_h_kernel(CrossAttnHParams p) {
  constexpr int CG = DH / 8;      // 16-byte channel groups of a key / value row
  constexpr int KT = 64 / CG;     // keys per wave tile of the context phase
  extern __shared__ __attribute__((aligned(16))) float xh_lds[];
  __shared__ float red[kXhWaves][kXhMaxBeam];
  __shared__ float fin_max[kXhMaxBeam];
  static_assert(kXhWaves == 8, "the maxima are folded over eight waves");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int splits = p.splits;
  const int bh = blockIdx.x / splits, chunk_id = blockIdx.x % splits;
  const int b = bh / static_cast<int>(p.heads);
  const int h = bh % static_cast<int>(p.heads);
  const int hk = h / static_cast<int>(p.heads / p.heads_kv);
  const int beam = static_cast<int>(p.beam);
  // this workgroup's chunk of the memory: positions [s_lo, s_lo + S), `length` of them valid
  const int s_lo = chunk_id * p.chunk;
  const int S = min(static_cast<int>(p.s_max) - s_lo, splits > 1 ? p.chunk : static_cast<int>(p.s_max));
  const int s_pad = (S + 7) & ~7;
  int length = p.lengths ? p.lengths[b] : static_cast<int>(p.s_max);
  length -= s_lo;
  length = length < 0 ? 0 : (length > S ? S : length);

  CT2X_XH_STAMP(0);
  float* q_s = xh_lds;                               // [kXhMaxBeam][DH]
  float* probs = q_s + kXhMaxBeam * DH;              // [kXhMaxBeam][s_pad]
  float* part = probs + kXhMaxBeam * s_pad;          // [kXhWaves][kXhMaxBeam][DH]

  for (int e = tid; e < beam * DH; e += kXhThreads) {
    const int j = e / DH, c = e % DH;
    const int64_t row = static_cast<int64_t>(b) * beam + j;
    q_s[j * DH + c] = p.q[row * p.q_row_stride + static_cast<int64_t>(h) * DH + c] * p.scale;
  }
  __syncthreads();

  CT2X_XH_STAMP(1);
  const int64_t kv_time = 2 * p.heads_kv * DH;       // elements per memory position
  const uint16_t* kbase = p.kv + (static_cast<int64_t>(b) * p.s_max + s_lo) * kv_time + static_cast<int64_t>(hk) * DH;
  const uint16_t* vbase = kbase + p.heads_kv * DH;

  // ---- scores on the matrix cores (round 5). The first form - thread = key, dot products on the
  // vector pipe against LDS broadcasts of the queries - spent 80 ds_read_b128 per key on re-reading
  // the queries: 25k of the launch's 72k shader clocks, LDS bound (profiles/r05_xattn_probe.txt).
  // Now S^T = K Q^T per tile of 16 keys: the A operand is the key rows as they lie in memory (lane =
  // (key lr, 8 channels): one 16-byte load per 32-channel step, no LDS), the B operand is Q^T - the
  // hypothesis rows as the 16 columns, zero beyond `beam` - split once into 16-bit terms that
  // carry 22 (float16: hi + lo) / 24 (bfloat16: hi + mid + lo) bits, so the products are the
  // fp32 ones. C layout: lane (hypothesis lr) holds keys 4 * (lane >> 4) + r of the tile.
  using HV = typename XhVec<kBF16>::vec;
  constexpr int KST = DH / 32;
  constexpr int TERMS = kBF16 ? 3 : 2;
  const int lr = lane & 15, lg = lane >> 4;
  HV qt[TERMS][KST];
#pragma unroll
  for (int ks = 0; ks < KST; ++ks)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float x = lr < beam ? q_s[lr * DH + 32 * ks + 8 * lg + e] : 0.f;
#pragma unroll
      for (int t = 0; t < TERMS; ++t) {
        const typename XhVec<kBF16>::elem part16 = static_cast<typename XhVec<kBF16>::elem>(x);
        qt[t][ks][e] = part16;
        x -= static_cast<float>(part16);
      }
    }
  float mx_lane = -FLT_MAX;  // of hypothesis lr over this lane's keys
  {
    constexpr int UNR = DH == 64 ? 6 : 3;  // tiles in flight per wave (DH / 32 x 16 bytes per lane each)
    const int tiles16 = (S + 15) / 16;
    for (int t0 = wave; t0 < tiles16; t0 += kXhWaves * UNR) {
      HV kf[UNR][KST];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int key = (t0 + u * kXhWaves) * 16 + lr;
#pragma unroll
        for (int ks = 0; ks < KST; ++ks) {
          uint4 raw = make_uint4(0u, 0u, 0u, 0u);
          if (key < length)
            raw = *reinterpret_cast<const uint4*>(kbase + static_cast<int64_t>(key) * kv_time + 32 * ks + 8 * lg);
          kf[u][ks] = __builtin_bit_cast(HV, raw);
        }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int t = t0 + u * kXhWaves;
        if (t < tiles16) {  // (wave-uniform)
          v4f acc = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KST; ++ks)
#pragma unroll
            for (int tt = TERMS - 1; tt >= 0; --tt) acc = XhVec<kBF16>::mfma(kf[u][ks], qt[tt][ks], acc);
          if (lr < beam) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int key = t * 16 + lg * 4 + r;
              const float v = key < length ? acc[r] : -FLT_MAX;
              if (key < s_pad) probs[lr * s_pad + key] = v;
              mx_lane = fmaxf(mx_lane, v);
            }
          }
        }
      }
    }
  }
  // the first batch of value tiles of the context phase is requested NOW: its memory round trip runs
  // under the softmax (two exchanges and the exponentials touch no memory); with the memory split
  // over workgroups a wave's whole share of the values is this one batch
  // (the KT lanes of a channel group are adjacent: their partial sums fold on the vector pipe)
  const int kr = lane % KT, cg = lane / KT;  // key within the tile, channel group of 8
  const int tiles = s_pad / KT;
  uint4 vu[kXhUnroll];
  int vs[kXhUnroll];
  auto fetch_values = [&](int t0) {
#pragma unroll
    for (int k = 0; k < kXhUnroll; ++k) {
      const int t = t0 + k * kXhWaves;
      const int s = t * KT + kr;
      vs[k] = (t < tiles && s < length) ? s : -1;
      vu[k] = make_uint4(0u, 0u, 0u, 0u);
      if (vs[k] >= 0)
        vu[k] = *reinterpret_cast<const uint4*>(vbase + static_cast<int64_t>(s) * kv_time + cg * 8);
    }
  };
  fetch_values(wave);
  CT2X_XH_STAMP(2);
  // ---- softmax: one exchange for the maxima of all hypotheses, one for the sums ----------------
  mx_lane = fmaxf(mx_lane, __shfl_xor(mx_lane, 16, 64));
  mx_lane = fmaxf(mx_lane, __shfl_xor(mx_lane, 32, 64));
  if (lane < beam) red[wave][lane] = mx_lane;
  float mx[MB];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < MB; ++j) {
    if (j < beam) {
      float m = red[0][j];
#pragma unroll
      for (int w = 1; w < kXhWaves; ++w) m = fmaxf(m, red[w][j]);
      mx[j] = m;
    }
  }
  if (tid < beam) fin_max[tid] = fmaxf(fmaxf(fmaxf(red[0][tid], red[1][tid]), fmaxf(red[2][tid], red[3][tid])),
                                      fmaxf(fmaxf(red[4][tid], red[5][tid]), fmaxf(red[6][tid], red[7][tid])));
  __syncthreads();  // every thread has read the maxima: `red` is reused for the sums
  CT2X_XH_STAMP(3);
  float sum[MB];
#pragma unroll
  for (int j = 0; j < MB; ++j) sum[j] = 0.f;
  for (int s = tid; s < s_pad; s += kXhThreads) {  // (scores are visible: barrier of the exchange)
#pragma unroll
    for (int j = 0; j < MB; ++j) {
      if (j < beam) {
        const float e = s < length ? expf(probs[j * s_pad + s] - mx[j]) : 0.f;
        probs[j * s_pad + s] = e;
        sum[j] += e;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < MB; ++j) {
    if (j < beam) {
      const float v = wave_sum_dpp(sum[j]);  // (no CPU order to reproduce here: the DPP form)
      if (lane == 0) red[wave][j] = v;
    }
  }
  __syncthreads();  // sums exchanged (they stay in `red` until the end: registers are for the context
                    // phase), probabilities visible to every wave

  CT2X_XH_STAMP(4);
  // ---- context: a wave takes tiles of KT keys, kXhUnroll tiles in flight --------------------------
  float acc[MB][8];
#pragma unroll
  for (int j = 0; j < MB; ++j)
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[j][c] = 0.f;
  for (int t0 = wave; t0 < tiles; t0 += kXhWaves * kXhUnroll) {
    if (t0 != wave) fetch_values(t0);  // (the first batch was requested before the softmax)
#pragma unroll
    for (int k = 0; k < kXhUnroll; ++k) {
      if (vs[k] >= 0) {
        float vf[8];
        unpack8<kBF16>(vu[k], vf);
#pragma unroll
        for (int j = 0; j < MB; ++j) {
          if (j < beam) {
            const float pj = probs[j * s_pad + vs[k]];
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[j][c] += pj * vf[c];
          }
        }
      }
    }
  }
  CT2X_XH_STAMP(5);
  // sum over the KT key lanes of a channel group (3 DPP adds per value; as ds_bpermute butterflies over
  // lane bits 3 - 5 this fold was 7k shader clocks of every workgroup)
#pragma unroll
  for (int j = 0; j < MB; ++j) {
    if (j < beam) {
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[j][c] = sum_adjacent<KT>(acc[j][c]);
      if (kr == 0) {
#pragma unroll
        for (int c = 0; c < 8; ++c) part[(wave * kXhMaxBeam + j) * DH + cg * 8 + c] = acc[j][c];
      }
    }
  }
  __syncthreads();
  CT2X_XH_STAMP(6);
  if (splits == 1) {
    for (int e = tid; e < beam * DH; e += kXhThreads) {
      const int j = e / DH, c = e % DH;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < kXhWaves; ++w) v += part[(w * kXhMaxBeam + j) * DH + c];
      float t = red[0][j];
#pragma unroll
      for (int w = 1; w < kXhWaves; ++w) t += red[w][j];
      const float inv = t > 0.f ? 1.f / t : 0.f;
      const int64_t row = static_cast<int64_t>(b) * beam + j;
      p.out[row * p.out_row_stride + static_cast<int64_t>(h) * DH + c] = v * inv;
    }
    CT2X_XH_STAMP(7);
    return;
  }
  // ---- split memory: publish this chunk's (context, max, sum), the last arrival merges ----------
  constexpr int REC = DH + 2;  // floats per (chunk, hypothesis row): context, max, sum
  float* mine = p.split_scratch + static_cast<int64_t>(blockIdx.x) * beam * REC;
  for (int e = tid; e < beam * DH; e += kXhThreads) {
    const int j = e / DH, c = e % DH;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < kXhWaves; ++w) v += part[(w * kXhMaxBeam + j) * DH + c];
    grid_publish(mine + j * REC + c, v);
  }
  if (tid < beam) {
    float t = red[0][tid];
#pragma unroll
    for (int w = 1; w < kXhWaves; ++w) t += red[w][tid];
    grid_publish(mine + tid * REC + DH, fin_max[tid]);
    grid_publish(mine + tid * REC + DH + 1, t);
  }
  CT2X_XH_STAMP(7);
  if (!grid_last_arrival(p.split_tickets + bh, static_cast<unsigned>(splits))) return;
  CT2X_XH_STAMP_ANY(8);
  // (all records first, then arithmetic: a memory-side load is a ~1 us round trip, and 3 x splits
  // of them one after the other were 40 us of this launch in its first form)
  const float* all = p.split_scratch + static_cast<int64_t>(bh) * splits * beam * REC;
  constexpr int kMaxSplits = 8;
  for (int e = tid; e < beam * DH; e += kXhThreads) {
    const int j = e / DH, c = e % DH;
    float m[kMaxSplits], l[kMaxSplits], o[kMaxSplits];
#pragma unroll
    for (int ch = 0; ch < kMaxSplits; ++ch) {
      m[ch] = -FLT_MAX;
      l[ch] = o[ch] = 0.f;
      if (ch < splits) {
        const float* rec = all + (ch * beam + j) * REC;
        m[ch] = grid_observe(rec + DH);
        l[ch] = grid_observe(rec + DH + 1);
        o[ch] = grid_observe(rec + c);
      }
    }
    float big = -FLT_MAX;
#pragma unroll
    for (int ch = 0; ch < kMaxSplits; ++ch) big = fmaxf(big, m[ch]);
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int ch = 0; ch < kMaxSplits; ++ch) {
      const float w = expf(m[ch] - big);  // an empty chunk: exp(-FLT_MAX - M) = 0 (or its sum is 0)
      den += w * l[ch];
      num += w * o[ch];
    }
    const int64_t row = static_cast<int64_t>(b) * beam + j;
    p.out[row * p.out_row_stride + static_cast<int64_t>(h) * DH + c] = den > 0.f ? num / den : 0.f;
  }
  CT2X_XH_STAMP_ANY(9);
  if (tid == 0) grid_publish(p.split_tickets + bh, 0u);  // replayed from a graph: back to zero
}

size_t xattn_h_lds_bytes(int64_t s_max, int64_t dhead) {
  const int64_t s_pad = (s_max + 7) & ~static_cast<int64_t>(7);
  return static_cast<size_t>(kXhMaxBeam * dhead + kXhMaxBeam * s_pad + kXhWaves * kXhMaxBeam * dhead) *
         sizeof(float);
}

}  // namespace

// How many chunks the memory of a (batch entry, head) is cut into when a scratch buffer is given:
// as many as keep every workgroup resident at once (two per CU up to beam 5 - 128 VGPRs -, one
// beyond; a second round of workgroups costs more than the shorter chunks save:
// profiles/r05_xattn_probe.txt), chunks of whole 8-key tiles and at least 128 keys.
static int64_t resident_workgroups(int64_t beam) { return beam <= 5 ? 512 : 256; }

int cross_attention_h_splits(int64_t batch, int64_t beam, int64_t heads, int64_t s_max) {
  const int64_t units = std::max<int64_t>(1, batch * heads);
  int64_t splits = resident_workgroups(beam) / units;
  splits = std::min<int64_t>(splits, s_max / 128);
  return static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(splits, 8)));
}

static int split_chunk(int64_t s_max, int splits) {
  const int64_t per = (s_max + splits - 1) / splits;
  return static_cast<int>((per + 7) & ~static_cast<int64_t>(7));
}

int64_t cross_attention_h_scratch_floats(int64_t batch, int64_t beam, int64_t heads, int64_t dhead,
                                         int64_t s_max) {
  // a decoding session shrinks as hypotheses finish, and fewer (entry, head) units are cut into
  // MORE chunks: size for the largest count of records over every batch <= this one
  // (units x floor(resident / units) <= resident, and never more than 8 chunks each)
  const int64_t units = batch * heads;
  const int64_t cap = std::min<int64_t>(8, s_max / 128);
  if (cap <= 1 || units >= resident_workgroups(beam)) return 0;
  return std::min(units * cap, resident_workgroups(beam)) * beam * (dhead + 2);
}

#ifdef CT2X_TUNING
void xattn_h_read_stamps(unsigned long long* host16) {
  CT2X_HIP_CHECK(hipDeviceSynchronize());
  CT2X_HIP_CHECK(hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_xh_stamps), sizeof(unsigned long long) * 16));
}
#endif

bool cross_attention_h_applicable(const CrossAttnHParams& p) {
  return (p.dhead == 64 || p.dhead == 128) && p.beam >= 1 && p.beam <= kXhMaxBeam && p.heads_kv >= 1 &&
         p.heads % p.heads_kv == 0 && p.s_max >= 1 &&
         xattn_h_lds_bytes(p.s_max, p.dhead) <= 150 * 1024 &&
         (reinterpret_cast<uintptr_t>(p.kv) & 15) == 0;
}

void launch_cross_attention_h(const CrossAttnHParams& p_in, hipStream_t stream) {
  if (p_in.batch * p_in.heads == 0) return;
  if (!cross_attention_h_applicable(p_in)) throw Error("launch_cross_attention_h: shape not supported");
  CrossAttnHParams p = p_in;
  p.splits = 1;
  p.chunk = static_cast<int>(p.s_max);
  if (p.split_scratch != nullptr && p.split_tickets != nullptr) {
    p.splits = cross_attention_h_splits(p.batch, p.beam, p.heads, p.s_max);
    if (const char* e = tuning_env("CT2X_XATTN_SPLITS"))  // (make TUNING=1 builds only; <= the sized maximum)
      p.splits = std::max(1, std::min(std::atoi(e), p.splits));
    if (p.splits > 1) {
      p.chunk = split_chunk(p.s_max, p.splits);
      // the last chunk must not be empty (chunks are rounded up to whole tiles)
      while (p.splits > 1 && static_cast<int64_t>(p.splits - 1) * p.chunk >= p.s_max) {
        --p.splits;
        p.chunk = split_chunk(p.s_max, p.splits);
      }
    }
  }
  const size_t lds = xattn_h_lds_bytes(p.splits > 1 ? p.chunk : p.s_max, p.dhead);
  const dim3 grid(static_cast<unsigned>(p.batch * p.heads * p.splits));
#define CT2X_XH_MB(BF, DH, MB)                                                                   \
  do {                                                                                            \
    allow_big_lds(reinterpret_cast<const void*>(xattn_h_kernel<BF, DH, MB>), lds);                \
    hipLaunchKernelGGL((xattn_h_kernel<BF, DH, MB>), grid, dim3(kXhThreads), lds, stream, p);     \
  } while (0)
#define CT2X_XH(BF, DH)                          \
  do {                                           \
    if (p.beam <= 1) CT2X_XH_MB(BF, DH, 1);      \
    else if (p.beam <= 2) CT2X_XH_MB(BF, DH, 2); \
    else if (p.beam <= 4) CT2X_XH_MB(BF, DH, 4); \
    else if (p.beam <= 5) CT2X_XH_MB(BF, DH, 5); \
    else CT2X_XH_MB(BF, DH, 8);                  \
  } while (0)
  if (p.dhead == 64) {
    if (p.bf16) CT2X_XH(true, 64);
    else CT2X_XH(false, 64);
  } else {
    if (p.bf16) CT2X_XH(true, 128);
    else CT2X_XH(false, 128);
  }
#undef CT2X_XH
#undef CT2X_XH_MB
}

}  // namespace ct2x
