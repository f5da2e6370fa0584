// Decode attention for d_head = 128 that reproduces the reference CPU path BIT FOR BIT
// (Llama-3-8B geometry: 32 query heads on 8 KV heads, RoPE, caches of hundreds of positions).
//
// What has to be reproduced (src/layers/attention.cc:178-287 on Device::CPU; restated and pinned
// bit for bit against the reference library in oracle/ct2_oracle.c: mkl_dot_main / mkl_dot_tail /
// mkl_pv_m1_col / orc_softmax, tests/test_oracle.py::test_exact_matmul_decode_*):
//   * RotaryEmbeddings::apply on q and the new key (src/ops/rotary_cpu.cc:9-62: two products and a
//     sum, no FMA), cache append in place of ops::Concat (attention.cc:541-546);
//   * QK^T = MKL sgemm with ONE query row per (row, head): every score is a dot product over the
//     128 channels with SIXTEEN lane accumulators (lane (t - 1) % 16 for channel t >= 1, channel 0
//     apart), folded by a 8 / 4 / 2 / 1 tree; the keys in the groups of four take the "main" form
//     (7 chunks accumulate, the 8th goes through a second tree seeded with the first tree's sum),
//     the n % 4 last keys the "tail" forms (type A: thirty-two accumulators over the first 96
//     channels; type B: sixteen throughout; channel 0 added last);
//   * SoftMax = cpu::softmax<AVX2> (exp256_ps, eight strided partial sums, e * (1 / sum));
//   * P.V = MKL's one-row form: blocks of eight keys (C + D) + (A + B), then one FMA per key.
//
// MI355X mapping - the 16 accumulators of MKL's dot product ARE the 16 lanes of a DPP row:
//   * one workgroup (8 waves) per (row, KV head): the G query heads of a GQA group share every K
//     and V byte the workgroup reads (the reference replicates the KV heads, attention.cc:291-295);
//   * QK^T: a 16-lane row owns one key at a time - lane l holds channels l + 1, l + 17, ... of the
//     key in 8 registers, straight from global memory (64-byte segments, no LDS staging, no
//     barrier), q of all G heads in G x 8 registers; the 16-accumulator tree is four row_shl DPP
//     adds. A wave works on 4 keys per pass, the 8 waves on 32, two passes of loads in flight;
//   * SoftMax: one wave per head on the scores in LDS;
//   * P.V: lane = (head of the group, channel) - the G lanes that need V[t][c] read the same
//     address, so V is fetched once; 16 keys per batch, two batches in flight.
// No matrix cores: the order of every addition is prescribed; the kernel is bound by the K / V
// stream (2 * T * 512 B per (row, KV head)) and by the dependent add chains of the order.

#include "device_utils.h"
#include "kernels.h"
#include "runtime.h"

#include <algorithm>
#include <cstdlib>
#include <float.h>

namespace ct2x {

namespace {

constexpr int X128_DH = 128;
constexpr int X128_WAVES = 8;
constexpr int X128_VB = 16;  // V rows per batch

struct AttnX128Args {
  AttnDecodeParams p;
  int group;   // query heads per KV head
  int t_pad;   // score slots per head (multiple of 16, >= keys + 16)
};

// a[l] + a[l + w] for the lanes l < w of every 16-lane row (w = 8, 4, 2, 1): the sum lands in lane
// 0 of the row; the other lanes hold partial garbage that is never used.
__device__ __forceinline__ float row_tree16(float a) {
  a = a + dpp_move<0x108, 0xf>(0.f, a);  // row_shl:8  lane l <- lane l + 8
  a = a + dpp_move<0x104, 0xf>(0.f, a);
  a = a + dpp_move<0x102, 0xf>(0.f, a);
  a = a + dpp_move<0x101, 0xf>(0.f, a);
  return a;
}

// One score of MKL's one-row product in a 16-lane row. kr[i] / qr[i]: channel (16 i + l + 1) & 127
// of the key / the query in lane l - so (i = 7, l = 15) holds channel 0. form: 0 main, 1 tail type
// A, 2 tail type B (oracle/ct2_oracle.c mkl_dot_main / mkl_dot_tail). Result in lane 0 of the row.
__device__ __forceinline__ float x128_dot(const float (&qr)[8], const float (&kr)[8], int l, int form) {
  const float last = qr[7] * kr[7];             // lanes 0..14: channel 113 + l; lane 15: channel 0
  const float p0 = dpp_move<0x121, 0xf>(0.f, last);  // row_ror:1 - lane 0 <- lane 15 (q0 * k0)
  if (form == 0) {
    float acc = (l == 0) ? p0 : 0.f;
#pragma unroll
    for (int i = 0; i < 7; ++i) acc = fmaf(qr[i], kr[i], acc);
    const float s = row_tree16(acc);
    float r = (l == 0) ? fmaf(qr[7], kr[7], s) : (l < 15 ? last : 0.f);
    return row_tree16(r);
  }
  float acc;
  if (form == 1) {
    float lo = qr[0] * kr[0], hi = qr[1] * kr[1];
    lo = fmaf(qr[2], kr[2], lo);
    hi = fmaf(qr[3], kr[3], hi);
    lo = fmaf(qr[4], kr[4], lo);
    hi = fmaf(qr[5], kr[5], hi);
    acc = lo + hi;
    acc = fmaf(qr[6], kr[6], acc);
  } else {
    acc = qr[0] * kr[0];
#pragma unroll
    for (int i = 1; i < 7; ++i) acc = fmaf(qr[i], kr[i], acc);
  }
  if (l < 15) acc = fmaf(qr[7], kr[7], acc);
  const float s = row_tree16(acc);
  return s + p0;
}

// vectorized_reduce_all<AVX2> over e[0 .. size): eight strided partial sums, 4 / 2 / 1 tree, scalar
// tail (device_utils.h ordered_sum8), with the addends of a chain requested ahead of its adds.
__device__ __forceinline__ float x128_ordered_sum8(const float* e, int size) {
  if (size <= 8) return ordered_sum8(e, size);
  const int l = threadIdx.x & 7;
  const int vsize = size & ~7;
  float acc = 0.f;
  int j = l;
  for (; j + 8 * 15 < vsize; j += 8 * 16) {
    float a[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = e[j + 8 * k];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc = acc + a[k];
  }
  for (; j < vsize; j += 8) acc = acc + e[j];
  acc = acc + dpp_xor4(acc);
  acc = acc + dpp_move<0x4E, 0xf>(0.f, acc);
  acc = acc + dpp_move<0xB1, 0xf>(0.f, acc);
  for (int t = vsize; t < size; ++t) acc = acc + e[t];
  return acc;
}

template <int G>
__global__ __launch_bounds__(X128_WAVES * 64) void attn_x128_decode_kernel(AttnX128Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int DH = X128_DH;
  const AttnDecodeParams& p = a.p;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row = blockIdx.x / static_cast<int>(p.heads_kv);
  const int hk = blockIdx.x % static_cast<int>(p.heads_kv);

  float* qs = smem;                    // [G][128] rotated queries
  float* knew_s = qs + G * DH;         // [128] rotated new key
  float* vnew_s = knew_s + DH;         // [128]
  float* sc = vnew_s + DH;             // [G][t_pad] scores, then probabilities
  const int t_pad = a.t_pad;

  int t_cached = p.step_ptr != nullptr ? *p.step_ptr : static_cast<int>(p.length);
  t_cached = max(0, min(t_cached, static_cast<int>(p.t_max)));
  const bool has_new = p.k_new != nullptr;
  const int pos = t_cached;
  const int n_keys = t_cached + (has_new ? 1 : 0);
  const int t_last = max(t_cached - 1, 0);
  const int kvs = static_cast<int>(p.kv_time_stride);
  float* kc = p.k_cache + static_cast<int64_t>(row) * p.kv_row_stride + static_cast<int64_t>(hk) * p.kv_head_stride;
  float* vc = p.v_cache + static_cast<int64_t>(row) * p.kv_row_stride + static_cast<int64_t>(hk) * p.kv_head_stride;

  // ---- prologue: RoPE on q (G heads) and the new key, cache append --------------------------
  {
    const int rd = static_cast<int>(p.rot_dim), middle = rd / 2;
    const bool has_rope = p.rope_sin != nullptr;
    for (int idx = tid; idx < (G + 2) * DH; idx += X128_WAVES * 64) {
      const int which = idx / DH, c = idx % DH;  // 0 .. G-1: query heads, G: new key, G + 1: new value
      if (which > G) {
        if (has_new) {
          const float v = p.v_new[static_cast<int64_t>(row) * p.new_row_stride + hk * DH + c];
          vnew_s[c] = v;
          vc[pos * kvs + c] = v;
        }
        continue;
      }
      if (which == G && !has_new) continue;
      const float* src = which < G
                             ? p.q + static_cast<int64_t>(row) * p.q_row_stride + (hk * G + which) * DH
                             : p.k_new + static_cast<int64_t>(row) * p.new_row_stride + hk * DH;
      float x = src[c];
      if (has_rope && c < rd) {
        int partner;
        bool neg;
        if (p.rope_interleave) {
          partner = c ^ 1;
          neg = (c & 1) == 0;
        } else {
          partner = c < middle ? c + middle : c - middle;
          neg = c < middle;
        }
        const float cs = p.rope_cos[static_cast<int64_t>(pos) * rd + c];
        const float sv = p.rope_sin[static_cast<int64_t>(pos) * rd + c];
        const float sn = neg ? -sv : sv;  // the sign of rot(x) folded into sin: exact
        x = x * cs + src[partner] * sn;
      }
      if (which < G) {
        qs[which * DH + c] = x;
      } else {
        knew_s[c] = x;
        kc[pos * kvs + c] = x;  // ops::Concat of the cache (attention.cc:541-546), in place
      }
    }
  }
  __syncthreads();

  // ---- QK^T: a 16-lane row per key ------------------------------------------------------------
  {
    const int l = lane & 15, grp = lane >> 4;
    float qr[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int i = 0; i < 8; ++i) qr[g][i] = qs[g * DH + ((16 * i + l + 1) & 127)];
    const int main_n = (n_keys / 4) * 4;
    const int n_pass = (n_keys + 4 * X128_WAVES - 1) / (4 * X128_WAVES);
    float kr[2][8];
    auto request = [&](int pass, float (&dst)[8]) {
      const int t = (pass * X128_WAVES + wave) * 4 + grp;
      // rows past the cache are clamped (their scores are never stored); the new key comes from LDS
      const int tc = min(t, t_last);
      const float* kp = kc + static_cast<int64_t>(tc) * kvs;
#pragma unroll
      for (int i = 0; i < 8; ++i) dst[i] = kp[(16 * i + l + 1) & 127];
    };
    auto compute = [&](int pass, const float (&kreg)[8]) {
      const int t = (pass * X128_WAVES + wave) * 4 + grp;
      float kk[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) kk[i] = kreg[i];
      if (has_new && t == pos) {
#pragma unroll
        for (int i = 0; i < 8; ++i) kk[i] = knew_s[(16 * i + l + 1) & 127];
      }
      const int form = t < main_n ? 0 : ((t - main_n) != 1 ? 1 : 2);
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float s = x128_dot(qr[g], kk, l, form) * p.scale;
        if (l == 0 && t < n_keys) sc[g * t_pad + t] = s;
      }
    };
    if (n_pass > 0) request(0, kr[0]);
    for (int pass = 0; pass < n_pass; pass += 2) {  // two passes of loads in flight
      if (pass + 1 < n_pass) request(pass + 1, kr[1]);
      compute(pass, kr[0]);
      if (pass + 2 < n_pass) request(pass + 2, kr[0]);
      if (pass + 1 < n_pass) compute(pass + 1, kr[1]);
    }
  }
  __syncthreads();

  // ---- SoftMax: one wave per head --------------------------------------------------------------
  for (int g = wave; g < G; g += X128_WAVES) {
    float* e = sc + g * t_pad;
    float m = -FLT_MAX;
    for (int t = lane; t < n_keys; t += 64) m = fmaxf(m, e[t]);
    m = wave_max_dpp(m);
    m = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 63));
    for (int t = lane; t < t_pad; t += 64) e[t] = (t < n_keys) ? exp_avx2(e[t] - m) : 0.f;
    __builtin_amdgcn_wave_barrier();
    const float esum = x128_ordered_sum8(e, n_keys);
    const float rcp = 1.f / esum;
    __builtin_amdgcn_wave_barrier();
    for (int t = lane; t < n_keys; t += 64) e[t] = e[t] * rcp;
  }
  __syncthreads();

  // ---- P.V: lane = (head of the group, channel) -----------------------------------------------
  constexpr int CPW = 64 / G;        // channels per wave item
  constexpr int ITEMS = DH / CPW;    // 2 G
  float row_max = 0.f;
  for (int item = wave; item < ITEMS; item += X128_WAVES) {
    const int g = lane / CPW, c = item * CPW + lane % CPW;
    const float* pr = sc + g * t_pad;
    const float vnew = has_new ? vnew_s[c] : 0.f;
    float y = 0.f;
    bool has = false;
    const int n_blocks = n_keys / 8;
#define CT2X_X128_LOAD_V(VR, T0)                                    \
  _Pragma("unroll") for (int j = 0; j < X128_VB; ++j) {             \
    const int t_ = (T0) + j;                                        \
    const int tc_ = min(t_, t_last);                                \
    const float g_ = vc[static_cast<int64_t>(tc_) * kvs + c];       \
    VR[j] = (has_new && t_ == pos) ? vnew : g_;                     \
  }
#define CT2X_X128_BLOCK(VR, J0, T0)                                                          \
  {                                                                                          \
    const v4f pa_ = *reinterpret_cast<const v4f*>(pr + (T0));                                \
    const v4f pb_ = *reinterpret_cast<const v4f*>(pr + (T0) + 4);                            \
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
    float vr[2][X128_VB];
    if (n_keys > 0) CT2X_X128_LOAD_V(vr[0], 0)
    for (int t0 = 0; t0 < n_keys; t0 += 2 * X128_VB) {
      if (t0 + X128_VB < n_keys) CT2X_X128_LOAD_V(vr[1], t0 + X128_VB)
      CT2X_X128_BLOCK(vr[0], 0, t0)
      if (t0 + 8 < n_keys) CT2X_X128_BLOCK(vr[0], 8, t0 + 8)
      if (!(t0 + X128_VB < n_keys)) break;
      if (t0 + 2 * X128_VB < n_keys) CT2X_X128_LOAD_V(vr[0], t0 + 2 * X128_VB)
      CT2X_X128_BLOCK(vr[1], 0, t0 + X128_VB)
      if (t0 + X128_VB + 8 < n_keys) CT2X_X128_BLOCK(vr[1], 8, t0 + X128_VB + 8)
    }
#undef CT2X_X128_LOAD_V
#undef CT2X_X128_BLOCK
    p.out[static_cast<int64_t>(row) * (p.heads * DH) + (hk * G + g) * DH + c] = y;
    row_max = fmaxf(row_max, fabsf(y));
  }
  if (p.row_amax_out != nullptr) {
    // producer-side absmax of the context row for the Quantize that follows (|y| >= 0: the bit
    // patterns order like unsigned integers; exact and order independent)
    row_max = wave_max_dpp(row_max);
    if (lane == 63) atomicMax(reinterpret_cast<unsigned int*>(p.row_amax_out) + row, __float_as_uint(row_max));
  }
}

int x128_t_pad(const AttnDecodeParams& p) {
  return static_cast<int>(((p.t_max + 1 + 15) / 16) * 16 + 16);
}

size_t x128_lds_bytes(const AttnDecodeParams& p) {
  const size_t group = static_cast<size_t>(p.heads / p.heads_kv);
  return ((group + 2) * X128_DH + group * static_cast<size_t>(x128_t_pad(p))) * sizeof(float);
}

bool x128_disabled() {
  static const bool off = [] {
    const char* e = std::getenv("CT2X_FAST_ATTENTION");
    return e != nullptr && e[0] == '1';
  }();
  return off;
}

template <int G>
void launch_x128(const AttnX128Args& a, size_t lds, hipStream_t stream) {
  if (lds > 64 * 1024)
    CT2X_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_x128_decode_kernel<G>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipLaunchKernelGGL(attn_x128_decode_kernel<G>, dim3(static_cast<unsigned>(a.p.rows * a.p.heads_kv)),
                     dim3(X128_WAVES * 64), lds, stream, a);
}

}  // namespace

// Self-attention decode at d_head 128: one query row per cache row (no encoder-memory beams), no
// per-row length vector, no ancestry table, GQA group 1 / 2 / 4 / 8, scores of the group within
// the LDS budget (T up to ~9000 at group 4).
bool attention_exact128_decode_applicable(const AttnDecodeParams& p) {
  if (x128_disabled()) return false;
  if (p.dhead != X128_DH || p.heads < 1 || p.heads_kv < 1 || p.heads % p.heads_kv != 0) return false;
  const int64_t group = p.heads / p.heads_kv;
  if (group != 1 && group != 2 && group != 4 && group != 8) return false;
  if (p.kv_group != 1 || p.lengths != nullptr || p.ancestry != nullptr) return false;
  if (p.rope_sin != nullptr && (p.rope_cos == nullptr || p.rot_dim < 2 || p.rot_dim > p.dhead ||
                                p.rot_dim % 2 != 0 || p.k_new == nullptr))
    return false;
  if (p.k_new != nullptr && p.v_new == nullptr) return false;
  if (p.kv_time_stride < p.dhead || p.t_max < 1) return false;
  return x128_lds_bytes(p) <= 160 * 1024;
}

void launch_attention_exact128_decode(const AttnDecodeParams& p, hipStream_t stream) {
  AttnX128Args a;
  a.p = p;
  a.group = static_cast<int>(p.heads / p.heads_kv);
  a.t_pad = x128_t_pad(p);
  const size_t lds = x128_lds_bytes(p);
  switch (a.group) {
    case 1: launch_x128<1>(a, lds, stream); break;
    case 2: launch_x128<2>(a, lds, stream); break;
    case 4: launch_x128<4>(a, lds, stream); break;
    default: launch_x128<8>(a, lds, stream); break;
  }
}

}  // namespace ct2x
