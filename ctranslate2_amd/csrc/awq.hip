// AWQ int4 weights on gfx950: ops::DequantizeAwq, ops::GemmAwq, ops::GemvAwq
// (include/ctranslate2/ops/awq/*.h; CUDA-only in the reference: src/ops/awq/dequantize_gpu.cu,
// gemm_gpu.cu, gemv_gpu.cu, and `throw "AWQ unsupported with ROCm"` in src/layers/common.cc:403).
// Written for the chip, not translated: the reference stages dequantised tiles through shared
// memory for ldmatrix + mma.sync; here a lane dequantises straight into the MFMA B fragment it
// owns and nothing goes through LDS but the cross-wave sum.
//
// Formats (restated in oracle/awq_oracle.py):
//   GEMM:  qweight [in_c, out_c / 8] int32, nibble i of a word = output channel ORDER[i] of its
//          group of eight, ORDER = 0 2 4 6 1 3 5 7; scales [in_c / G, out_c] f16; qzeros
//          [in_c / G, out_c / 8] packed like qweight.        w[k][n] = (q - z[k/G][n]) * s[k/G][n]
//   GEMV:  qweight [out_c, in_c / 8] int32, nibble i of word w = input channel 8 w + i; qzeros
//          [out_c, zeros_w] (nibble g % 8 of word g / 8 = group g); scales [out_c, sf_w] f16.
// Dequantisation arithmetic of the reference: the subtraction is exact, the product is rounded to
// float16 once (sub.f16x2 + fma.rn.f16x2 with a zero addend, dequantize_gpu.cu:44-52); products
// with the float16 activation accumulate in fp32 (mma.sync ... f32.f16.f16.f32).
//
// The nibble unpack is the classic "or in the exponent of 1024" trick on packed halves: one
// v_and_or_b32 yields two halves (1024 + n_i, 1024 + n_{i+4}) resp. (1024 + 16 n_i, ...), so a word
// costs 1 shift + 4 and-or + 4 packed adds (+ 2 exact packed multiplies by 1/16) + 4 packed
// multiplies by the scale - ~1.7 VALU operations per weight next to a 0.5-byte weight stream.
//   * GEMV format: a word IS eight consecutive K values of one output channel = one MFMA B
//     fragment (v_mfma_f32_16x16x32_f16: lane (col, kb) holds B[8 kb .. 8 kb + 7][col]); the unpack
//     order n0 n4 n1 n5 n2 n6 n3 n7 is absorbed by permuting the ACTIVATION fragment once per K-step
//     (shared by all column tiles).
//   * GEMM format: a word is eight CHANNELS of one K row; lane (cw, kb) loads the words of rows
//     8 kb .. 8 kb + 7 at column word cw, i.e. an 8 x 8 block, and regroups it with v_perm_b32 into
//     eight fragments, fragment c = channel 8 cw + c: MFMA number c of the step computes the 16
//     columns {8 cw + c : cw = 0..15} - a permutation of the 128 output columns, undone in the store.

//
// Round 6 - the bfloat16 arm ("incl. a bf16 variant the reference lacks"; BASELINE
// configs[4] is bf16 + AWQ). Defined here, there being no reference arithmetic to follow: the scales
// are the model's float16 scales rounded to bfloat16 once at load, a weight is (q - z) * s rounded to
// bfloat16 ONCE (the product of a 5-bit integer and an 8-bit significand is exact in fp32), the
// activation is rounded to bfloat16, products accumulate in fp32 on v_mfma_f32_16x16x32_bf16 -
// the float16 arm with the other 16-bit type (oracle/awq_oracle.py dequantize_*(..., bf16=True)).
// The nibble unpack uses the same trick one exponent lower: 0x4300 | n is the bfloat16 128 + n
// (seven mantissa bits resolve integers up to 255), widened to fp32 by a shift, and ONE fma per
// weight forms (128 + n) * s - (128 + z) * s = (n - z) * s exactly; v_cvt_pk_bf16_f32 rounds pairs.

#include "device_utils.h"
#include "kernels.h"
#include "runtime.h"

namespace ct2x {

namespace {

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
typedef __bf16 v8b __attribute__((ext_vector_type(8)));
typedef __bf16 v2b __attribute__((ext_vector_type(2)));
typedef float v4f32 __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v2h as_v2h(unsigned u) { return __builtin_bit_cast(v2h, u); }
__device__ __forceinline__ unsigned as_u32(v2h h) { return __builtin_bit_cast(unsigned, h); }

constexpr unsigned kMagic = 0x64006400u;    // half2 (1024, 1024)
constexpr unsigned kLoMask = 0x000f000fu;   // nibbles i, i + 4 in place
constexpr unsigned kHiMask = 0x00f000f0u;   // nibbles i + 1, i + 5, scaled by 16

// word -> four half2 holding the raw nibbles as 1024 + n (pairs 0 and 2) and 1024 + 16 n (pairs 1
// and 3): (n0, n4), (n1, n5), (n2, n6), (n3, n7)
__device__ __forceinline__ void unpack_raw(unsigned w, unsigned (&h)[4]) {
  const unsigned w8 = w >> 8;
  h[0] = (w & kLoMask) | kMagic;
  h[1] = (w & kHiMask) | kMagic;
  h[2] = (w8 & kLoMask) | kMagic;
  h[3] = (w8 & kHiMask) | kMagic;
}

// (q - z) * s for one word whose eight nibbles share zero point z and scale s (GEMV format)
__device__ __forceinline__ v8h dequant_word_shared(unsigned w, float z, _Float16 s) {
  unsigned h[4];
  unpack_raw(w, h);
  const _Float16 zl = static_cast<_Float16>(1024.f + z);  // exact: integers below 2048
  const _Float16 zh = static_cast<_Float16>(64.f + z);
  const v2h zlo = {zl, zl}, zhi = {zh, zh}, sixteenth = {(_Float16)0.0625f, (_Float16)0.0625f};
  const v2h ss = {s, s};
  const v2h d0 = (as_v2h(h[0]) - zlo) * ss;
  const v2h d1 = (as_v2h(h[1]) * sixteenth - zhi) * ss;  // (1024 + 16 n) / 16 = 64 + n, exact
  const v2h d2 = (as_v2h(h[2]) - zlo) * ss;
  const v2h d3 = (as_v2h(h[3]) * sixteenth - zhi) * ss;
  return v8h{d0.x, d0.y, d1.x, d1.y, d2.x, d2.y, d3.x, d3.y};  // K offsets 0 4 1 5 2 6 3 7
}

// activation fragment (8 consecutive K halves) -> the order the unpack produces
__device__ __forceinline__ v8h permute_0415(v8h x) {
  return v8h{x[0], x[4], x[1], x[5], x[2], x[6], x[3], x[7]};
}
__device__ __forceinline__ v8b permute_0415(v8b x) {
  return v8b{x[0], x[4], x[1], x[5], x[2], x[6], x[3], x[7]};
}

// ---- bfloat16 arm -------------------------------------------------------------------------------
constexpr unsigned kMagicB = 0x43004300u;  // bfloat16 pair (128, 128)
__device__ __forceinline__ float bf16_bits_to_float(unsigned short b) {
  return __uint_as_float(static_cast<unsigned>(b) << 16);
}
// two floats -> packed bfloat16 pair (round to nearest even), `lo` in the low half
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
  const v2b v = {static_cast<__bf16>(lo), static_cast<__bf16>(hi)};
  return __builtin_bit_cast(unsigned, v);
}
// nibbles p and p + 4 of `w` as the floats 128 + n
__device__ __forceinline__ void unpack_pair_b(unsigned w, int p, float& lo, float& hi) {
  const unsigned u = ((w >> (4 * p)) & kLoMask) | kMagicB;
  lo = __uint_as_float(u << 16);
  hi = __uint_as_float(u & 0xffff0000u);
}
// (q - z) * s for a word whose nibbles share zero point and scale (GEMV format); s = the scale,
// c = -(128 + z) * s (exact). K offsets 0 4 1 5 2 6 3 7 as in the float16 arm.
__device__ __forceinline__ v8b dequant_word_shared_b(unsigned w, float s, float c) {
  unsigned d[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    float lo, hi;
    unpack_pair_b(w, p, lo, hi);
    d[p] = pack_bf16(fmaf(lo, s, c), fmaf(hi, s, c));
  }
  const v4u v = {d[0], d[1], d[2], d[3]};
  return __builtin_bit_cast(v8b, v);
}

template <bool BF>
struct AwqT;
template <>
struct AwqT<false> {
  using elem = _Float16;
  using vec = v8h;
  __device__ static __forceinline__ v4f32 mfma(vec a, vec b, v4f32 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};
template <>
struct AwqT<true> {
  using elem = __bf16;
  using vec = v8b;
  __device__ static __forceinline__ v4f32 mfma(vec a, vec b, v4f32 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};

// ---------------------------------------------------------------------------------------------
// DequantizeAwq (GEMM format) -> float16 [in_c, out_c]; one thread per packed word.
// ---------------------------------------------------------------------------------------------
__global__ void awq_dequantize_gemm_kernel(const unsigned* __restrict__ qw,
                                           const _Float16* __restrict__ scales,
                                           const unsigned* __restrict__ qz, int64_t in_c,
                                           int64_t qout, int group, _Float16* __restrict__ out) {
  const int64_t idx = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (idx >= in_c * qout) return;
  const int64_t row = idx / qout, cw = idx - row * qout, g = row / group;
  unsigned h[4], zh[4];
  unpack_raw(qw[idx], h);
  unpack_raw(qz[g * qout + cw], zh);
  const v2h sixteenth = {(_Float16)0.0625f, (_Float16)0.0625f};
  const v2h* s2 = reinterpret_cast<const v2h*>(scales + (g * qout + cw) * 8);
  v2h d[4];
  d[0] = (as_v2h(h[0]) - as_v2h(zh[0])) * s2[0];                // channels 0, 1 (nibbles 0, 4)
  d[1] = ((as_v2h(h[1]) - as_v2h(zh[1])) * sixteenth) * s2[1];  // channels 2, 3 (nibbles 1, 5)
  d[2] = (as_v2h(h[2]) - as_v2h(zh[2])) * s2[2];                // channels 4, 5
  d[3] = ((as_v2h(h[3]) - as_v2h(zh[3])) * sixteenth) * s2[3];  // channels 6, 7
  v2h* o = reinterpret_cast<v2h*>(out + (row * qout + cw) * 8);
  o[0] = d[0]; o[1] = d[1]; o[2] = d[2]; o[3] = d[3];
}

__global__ void awq_dequantize_gemm_b_kernel(const unsigned* __restrict__ qw,
                                             const unsigned short* __restrict__ scales,
                                             const unsigned* __restrict__ qz, int64_t in_c,
                                             int64_t qout, int group, unsigned* __restrict__ out) {
  const int64_t idx = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (idx >= in_c * qout) return;
  const int64_t row = idx / qout, cw = idx - row * qout, g = row / group;
  const unsigned w = qw[idx], zw = qz[g * qout + cw];
  const unsigned short* sp = scales + (g * qout + cw) * 8;
  unsigned* o = out + (row * qout + cw) * 4;
#pragma unroll
  for (int p = 0; p < 4; ++p) {  // nibbles p, p + 4 = channels 2 p, 2 p + 1
    float lo, hi, zlo, zhi;
    unpack_pair_b(w, p, lo, hi);
    unpack_pair_b(zw, p, zlo, zhi);
    const float s0 = bf16_bits_to_float(sp[2 * p]), s1 = bf16_bits_to_float(sp[2 * p + 1]);
    o[p] = pack_bf16(fmaf(lo, s0, -(zlo * s0)), fmaf(hi, s1, -(zhi * s1)));
  }
}

// Dequantised GEMV-format weight -> float16 [out_c, in_c] (tests, and the large-m arm).
__global__ void awq_dequantize_gemv_kernel(const unsigned* __restrict__ qw,
                                           const _Float16* __restrict__ scales,
                                           const unsigned* __restrict__ qz, int64_t out_c,
                                           int64_t words, int group, int zeros_w, int sf_w,
                                           _Float16* __restrict__ out) {
  const int64_t idx = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (idx >= out_c * words) return;
  const int64_t oc = idx / words, w = idx - oc * words;
  const int g = static_cast<int>(w * 8 / group);
  const float z = static_cast<float>((qz[oc * zeros_w + g / 8] >> (4 * (g % 8))) & 15u);
  const v8h d = dequant_word_shared(qw[idx], z, scales[oc * sf_w + g]);
  _Float16* o = out + (oc * words + w) * 8;
  o[0] = d[0]; o[4] = d[1]; o[1] = d[2]; o[5] = d[3]; o[2] = d[4]; o[6] = d[5]; o[3] = d[6]; o[7] = d[7];
}
__global__ void awq_dequantize_gemv_b_kernel(const unsigned* __restrict__ qw,
                                             const unsigned short* __restrict__ scales,
                                             const unsigned* __restrict__ qz, int64_t out_c,
                                             int64_t words, int group, int zeros_w, int sf_w,
                                             __bf16* __restrict__ out) {
  const int64_t idx = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (idx >= out_c * words) return;
  const int64_t oc = idx / words, w = idx - oc * words;
  const int g = static_cast<int>(w * 8 / group);
  const float z = static_cast<float>((qz[oc * zeros_w + g / 8] >> (4 * (g % 8))) & 15u);
  const float s = bf16_bits_to_float(scales[oc * sf_w + g]);
  const v8b d = dequant_word_shared_b(qw[idx], s, -((128.f + z) * s));
  __bf16* o = out + (oc * words + w) * 8;
  o[0] = d[0]; o[4] = d[1]; o[1] = d[2]; o[5] = d[3]; o[2] = d[4]; o[6] = d[5]; o[3] = d[6]; o[7] = d[7];
}

// ---------------------------------------------------------------------------------------------
// Fused decode GEMM, GEMV format: y[m, oc] = x[m, :] . w[oc, :]. A workgroup owns 16 rows
// (grid.y) x NT * 16 output channels; its four waves split K in contiguous quarters, every wave
// walks its quarter in steps of 128 (one 16-byte weight request per lane and column tile, four
// MFMAs), partial tiles are summed through LDS in wave order.
// ---------------------------------------------------------------------------------------------
template <int NT, bool BF>
__global__ __launch_bounds__(256) void awq_gemv_kernel(const typename AwqT<BF>::elem* __restrict__ x,
                                                       const unsigned* __restrict__ qw,
                                                       const typename AwqT<BF>::elem* __restrict__ scales,
                                                       const unsigned* __restrict__ qz, int M,
                                                       int out_c, int in_c, int group, int zeros_w,
                                                       int sf_w, typename AwqT<BF>::elem* __restrict__ y16,
                                                       float* __restrict__ y32) {
  using T = AwqT<BF>;
  using vec = typename T::vec;
  using elem = typename T::elem;
  __shared__ float partial[4][NT][64][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kb = lane >> 4;
  const int m0 = blockIdx.y * 16;
  const int64_t n0 = static_cast<int64_t>(blockIdx.x) * (NT * 16);
  const int words = in_c >> 3;
  // K quarter of this wave, in steps of 128 (in_c % 128 == 0 is a launch precondition)
  const int steps = in_c >> 7;
  const int s_begin = (steps * wave) / 4, s_end = (steps * (wave + 1)) / 4;

  v4f32 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) acc[j] = v4f32{0.f, 0.f, 0.f, 0.f};
  const bool row_ok = m0 + r < M;
  const elem* xr = x + static_cast<int64_t>(row_ok ? m0 + r : 0) * in_c;

  for (int s = s_begin; s < s_end; ++s) {
    const int k0 = s * 128 + kb * 32;  // this lane's 32 K values of the step
    v4u wv[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int64_t oc = n0 + j * 16 + r;
      wv[j] = oc < out_c ? *reinterpret_cast<const v4u*>(qw + oc * words + (k0 >> 3)) : v4u{0, 0, 0, 0};
    }
    vec a[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      a[t] = vec{};
      if (row_ok) a[t] = permute_0415(*reinterpret_cast<const vec*>(xr + k0 + t * 8));
    }
    const int g = k0 / group;  // group >= 32 and a multiple of 32: one group per lane and step
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int64_t oc = n0 + j * 16 + r;
      float z = 0.f;
      elem sc = static_cast<elem>(0.f);
      if (oc < out_c) {
        z = static_cast<float>((qz[oc * zeros_w + (g >> 3)] >> (4 * (g & 7))) & 15u);
        sc = scales[oc * sf_w + g];
      }
      if constexpr (BF) {
        const float sf = static_cast<float>(sc), cf = -((128.f + z) * sf);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[j] = T::mfma(a[t], dequant_word_shared_b(wv[j][t], sf, cf), acc[j]);
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[j] = T::mfma(a[t], dequant_word_shared(wv[j][t], z, sc), acc[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) *reinterpret_cast<v4f32*>(&partial[wave][j][lane][0]) = acc[j];
  __syncthreads();
  // C layout: column = lane & 15, row = (lane >> 4) * 4 + reg
  for (int o = tid; o < NT * 256; o += 256) {
    const int j = o >> 8, within = o & 255, rt = within >> 4, ct = within & 15;
    const int src_lane = ((rt >> 2) << 4) | ct;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) v += partial[w][j][src_lane][rt & 3];
    const int64_t col = n0 + j * 16 + ct;
    if (m0 + rt < M && col < out_c) {
      const int64_t idx = static_cast<int64_t>(m0 + rt) * out_c + col;
      if (y16) y16[idx] = static_cast<elem>(v);
      if (y32) y32[idx] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Fused decode GEMM, GEMM format: y[m, n] = x[m, :] . w[:, n]. A workgroup owns 16 rows x 128
// output channels (16 column words); waves split K in contiguous quarters and walk them in steps of
// 32 rows: lane (cw, kb) loads the eight words of rows k0 + 8 kb .. + 7 at column word cw.
// ---------------------------------------------------------------------------------------------
template <bool BF>
__global__ __launch_bounds__(256) void awq_gemm_kernel(const typename AwqT<BF>::elem* __restrict__ x,
                                                       const unsigned* __restrict__ qw,
                                                       const typename AwqT<BF>::elem* __restrict__ scales,
                                                       const unsigned* __restrict__ qz, int M,
                                                       int in_c, int qout, int group,
                                                       typename AwqT<BF>::elem* __restrict__ y16,
                                                       float* __restrict__ y32,
                                                       float* __restrict__ slices) {
  using T = AwqT<BF>;
  using vec = typename T::vec;
  using elem = typename T::elem;
  // gridDim.z > 1: split K - this workgroup covers steps [z, z + 1) * steps / gridDim.z and writes
  // its fp32 tile to slices[z][m][out_c]; awq_sum_slices_kernel adds the slices in z order (the
  // reference does the same with eight slices and ops::Sum, src/ops/awq/gemm.cc:24-27 - here the
  // split only exists to put >= 256 workgroups on the chip: a 128-column tile per workgroup leaves
  // a 4096-wide projection with 32 of them).
  __shared__ float partial[4][8][64][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kb = lane >> 4;
  const int m0 = blockIdx.y * 16;
  const int64_t cw = static_cast<int64_t>(blockIdx.x) * 16 + r;  // column word of this lane
  const bool cw_ok = cw < qout;
  const int out_c = qout * 8;
  const int steps = in_c >> 5;  // in_c % 32 == 0
  const int z_begin = static_cast<int>(static_cast<int64_t>(steps) * blockIdx.z / gridDim.z);
  const int z_end = static_cast<int>(static_cast<int64_t>(steps) * (blockIdx.z + 1) / gridDim.z);
  const int zsteps = z_end - z_begin;
  const int s_begin = z_begin + (zsteps * wave) / 4, s_end = z_begin + (zsteps * (wave + 1)) / 4;
  const bool row_ok = m0 + r < M;
  const elem* xr = x + static_cast<int64_t>(row_ok ? m0 + r : 0) * in_c;
  const v2h sixteenth = {(_Float16)0.0625f, (_Float16)0.0625f};

  v4f32 acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = v4f32{0.f, 0.f, 0.f, 0.f};

  int g_cached = -1;
  unsigned zh[4] = {kMagic, kMagic, kMagic, kMagic};
  v2h s2[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
  float sb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, cb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int s = s_begin; s < s_end; ++s) {
    const int k0 = s * 32 + kb * 8;
    unsigned w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      w[j] = cw_ok ? qw[static_cast<int64_t>(k0 + j) * qout + cw] : 0u;
    vec a = vec{};
    if (row_ok) a = *reinterpret_cast<const vec*>(xr + k0);
    const int g = k0 / group;  // group % 8 == 0: the eight rows of a lane share their group
    if (g != g_cached) {
      g_cached = g;
      if (cw_ok) {
        if constexpr (BF) {
          const unsigned zw = qz[static_cast<int64_t>(g) * qout + cw];
          const elem* sp = scales + (static_cast<int64_t>(g) * qout + cw) * 8;
#pragma unroll
          for (int p = 0; p < 4; ++p) {  // channels 2 p, 2 p + 1 = nibbles p, p + 4
            float zlo, zhi;
            unpack_pair_b(zw, p, zlo, zhi);
            sb[2 * p] = static_cast<float>(sp[2 * p]);
            sb[2 * p + 1] = static_cast<float>(sp[2 * p + 1]);
            cb[2 * p] = -(zlo * sb[2 * p]);
            cb[2 * p + 1] = -(zhi * sb[2 * p + 1]);
          }
        } else {
          unpack_raw(qz[static_cast<int64_t>(g) * qout + cw], zh);
          const v2h* sp = reinterpret_cast<const v2h*>(scales + (static_cast<int64_t>(g) * qout + cw) * 8);
          s2[0] = sp[0]; s2[1] = sp[1]; s2[2] = sp[2]; s2[3] = sp[3];
        }
      }
    }
    // d[j][p]: row j, channel pair p = channels (2p, 2p + 1)
    unsigned d[8][4];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if constexpr (BF) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          float lo, hi;
          unpack_pair_b(w[j], p, lo, hi);
          d[j][p] = pack_bf16(fmaf(lo, sb[2 * p], cb[2 * p]), fmaf(hi, sb[2 * p + 1], cb[2 * p + 1]));
        }
      } else {
        unsigned h[4];
        unpack_raw(w[j], h);
        d[j][0] = as_u32((as_v2h(h[0]) - as_v2h(zh[0])) * s2[0]);
        d[j][1] = as_u32(((as_v2h(h[1]) - as_v2h(zh[1])) * sixteenth) * s2[1]);
        d[j][2] = as_u32((as_v2h(h[2]) - as_v2h(zh[2])) * s2[2]);
        d[j][3] = as_u32(((as_v2h(h[3]) - as_v2h(zh[3])) * sixteenth) * s2[3]);
      }
    }
    // regroup: fragment of channel c = rows 0..7 of that channel, (row 2q, row 2q + 1) per dword
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int p = c >> 1;
      const unsigned sel = (c & 1) ? 0x07060302u : 0x05040100u;  // high / low halves of (b, a)
      v4u f;
#pragma unroll
      for (int q = 0; q < 4; ++q) f[q] = __builtin_amdgcn_perm(d[2 * q + 1][p], d[2 * q][p], sel);
      acc[c] = T::mfma(a, __builtin_bit_cast(vec, f), acc[c]);
    }
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) *reinterpret_cast<v4f32*>(&partial[wave][c][lane][0]) = acc[c];
  __syncthreads();
  // MFMA number c, column ct of its tile = output channel 8 * (16 blockIdx.x + ct) + c
  for (int o = tid; o < 8 * 256; o += 256) {
    const int c = o >> 8, within = o & 255, rt = within >> 4, ct = within & 15;
    const int src_lane = ((rt >> 2) << 4) | ct;
    float v = 0.f;
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) v += partial[wv][c][src_lane][rt & 3];
    const int64_t col = (static_cast<int64_t>(blockIdx.x) * 16 + ct) * 8 + c;
    if (m0 + rt < M && col < out_c) {
      const int64_t idx = static_cast<int64_t>(m0 + rt) * out_c + col;
      if (slices) {
        slices[static_cast<int64_t>(blockIdx.z) * M * out_c + idx] = v;
      } else {
        if (y16) y16[idx] = static_cast<elem>(v);
        if (y32) y32[idx] = v;
      }
    }
  }
}

template <typename elem>
__global__ void awq_sum_slices_kernel(const float* __restrict__ slices, int splits, int64_t n,
                                      elem* __restrict__ y16, float* __restrict__ y32) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  float v = 0.f;
  for (int z = 0; z < splits; ++z) v += slices[z * n + i];  // fixed order: deterministic
  if (y16) y16[i] = static_cast<elem>(v);
  if (y32) y32[i] = v;
}

}  // namespace

void launch_awq_dequantize_gemm(const int32_t* qweight, const void* scales_f16, const int32_t* qzeros,
                                int64_t in_c, int64_t out_c, int64_t group, void* out_f16,
                                hipStream_t stream, bool bf16) {
  if (in_c == 0 || out_c == 0) return;
  if (out_c % 8 != 0 || group <= 0 || in_c % group != 0)
    throw Error("awq dequantize: out_c must be a multiple of 8 and in_c a multiple of the group size");
  const int64_t n = in_c * (out_c / 8);
  if (bf16) {  // (scales / out: bfloat16)
    hipLaunchKernelGGL(awq_dequantize_gemm_b_kernel, dim3(static_cast<unsigned>((n + 255) / 256)),
                       dim3(256), 0, stream, reinterpret_cast<const unsigned*>(qweight),
                       static_cast<const unsigned short*>(scales_f16),
                       reinterpret_cast<const unsigned*>(qzeros), in_c, out_c / 8,
                       static_cast<int>(group), static_cast<unsigned*>(out_f16));
    return;
  }
  hipLaunchKernelGGL(awq_dequantize_gemm_kernel, dim3(static_cast<unsigned>((n + 255) / 256)),
                     dim3(256), 0, stream, reinterpret_cast<const unsigned*>(qweight),
                     static_cast<const _Float16*>(scales_f16), reinterpret_cast<const unsigned*>(qzeros),
                     in_c, out_c / 8, static_cast<int>(group), static_cast<_Float16*>(out_f16));
}

void launch_awq_dequantize_gemv(const int32_t* qweight, const void* scales_f16, const int32_t* qzeros,
                                int64_t out_c, int64_t in_c, int64_t group, int64_t zeros_w,
                                int64_t sf_w, void* out_f16, hipStream_t stream, bool bf16) {
  if (in_c == 0 || out_c == 0) return;
  if (in_c % 8 != 0 || group <= 0 || group % 8 != 0)
    throw Error("awq dequantize (gemv layout): in_c and the group size must be multiples of 8");
  const int64_t n = out_c * (in_c / 8);
  if (bf16) {
    hipLaunchKernelGGL(awq_dequantize_gemv_b_kernel, dim3(static_cast<unsigned>((n + 255) / 256)),
                       dim3(256), 0, stream, reinterpret_cast<const unsigned*>(qweight),
                       static_cast<const unsigned short*>(scales_f16),
                       reinterpret_cast<const unsigned*>(qzeros), out_c, in_c / 8,
                       static_cast<int>(group), static_cast<int>(zeros_w), static_cast<int>(sf_w),
                       static_cast<__bf16*>(out_f16));
    return;
  }
  hipLaunchKernelGGL(awq_dequantize_gemv_kernel, dim3(static_cast<unsigned>((n + 255) / 256)),
                     dim3(256), 0, stream, reinterpret_cast<const unsigned*>(qweight),
                     static_cast<const _Float16*>(scales_f16), reinterpret_cast<const unsigned*>(qzeros),
                     out_c, in_c / 8, static_cast<int>(group), static_cast<int>(zeros_w),
                     static_cast<int>(sf_w), static_cast<_Float16*>(out_f16));
}

void launch_awq_gemm(const void* x_f16, const int32_t* qweight, const void* scales_f16,
                     const int32_t* qzeros, int64_t m, int64_t in_c, int64_t out_c, int64_t group,
                     void* y_f16, float* y_f32, hipStream_t stream, float* slices_ws, bool bf16) {
  if (m == 0 || out_c == 0) return;
  if (in_c % 32 != 0 || out_c % 8 != 0 || group % 8 != 0 || group <= 0 || in_c % group != 0 ||
      (reinterpret_cast<uintptr_t>(x_f16) & 15) != 0 || (reinterpret_cast<uintptr_t>(scales_f16) & 15) != 0)
    throw Error("awq gemm: needs in_c % 32 == 0, out_c % 8 == 0, group % 8 == 0 and 16-byte aligned x / scales");
  const int64_t qout = out_c / 8;
  const int64_t tiles = ((qout + 15) / 16) * ((m + 15) / 16);
  // K slices so that ~256+ workgroups stream the weight; each slice keeps >= 4 steps per wave
  int64_t splits = 1;
  while (tiles * splits < 256 && (in_c / 32) / (splits * 2) >= 16 && splits < 16) splits *= 2;
  // splits > 1 only while tiles * splits < 512, i.e. splits * m * out_c < 512 * 16 * 128 floats:
  // kAwqGemmWorkspaceFloats always holds the slices. A caller-owned workspace keeps the launch
  // free of allocator calls (a captured decode step must not allocate); without one the slices
  // come from the caching allocator (eager callers of the C ABI).
  float* slices = nullptr;
  if (splits > 1) {
    CT2X_CHECK(static_cast<size_t>(splits) * m * out_c <= kAwqGemmWorkspaceFloats,
               "awq gemm: split-K workspace bound violated");
    slices = slices_ws ? slices_ws
                       : static_cast<float*>(device_malloc(static_cast<size_t>(splits) * m * out_c * sizeof(float)));
  }
  const dim3 grid(static_cast<unsigned>((qout + 15) / 16), static_cast<unsigned>((m + 15) / 16),
                  static_cast<unsigned>(splits));
  if (bf16)
    hipLaunchKernelGGL(awq_gemm_kernel<true>, grid, dim3(256), 0, stream, static_cast<const __bf16*>(x_f16),
                       reinterpret_cast<const unsigned*>(qweight), static_cast<const __bf16*>(scales_f16),
                       reinterpret_cast<const unsigned*>(qzeros), static_cast<int>(m),
                       static_cast<int>(in_c), static_cast<int>(qout), static_cast<int>(group),
                       static_cast<__bf16*>(y_f16), y_f32, slices);
  else
    hipLaunchKernelGGL(awq_gemm_kernel<false>, grid, dim3(256), 0, stream, static_cast<const _Float16*>(x_f16),
                       reinterpret_cast<const unsigned*>(qweight), static_cast<const _Float16*>(scales_f16),
                       reinterpret_cast<const unsigned*>(qzeros), static_cast<int>(m),
                       static_cast<int>(in_c), static_cast<int>(qout), static_cast<int>(group),
                       static_cast<_Float16*>(y_f16), y_f32, slices);
  if (splits > 1) {
    const int64_t n = m * out_c;
    if (bf16)
      hipLaunchKernelGGL(awq_sum_slices_kernel<__bf16>, dim3(static_cast<unsigned>((n + 255) / 256)),
                         dim3(256), 0, stream, slices, static_cast<int>(splits), n,
                         static_cast<__bf16*>(y_f16), y_f32);
    else
      hipLaunchKernelGGL(awq_sum_slices_kernel<_Float16>, dim3(static_cast<unsigned>((n + 255) / 256)),
                         dim3(256), 0, stream, slices, static_cast<int>(splits), n,
                         static_cast<_Float16*>(y_f16), y_f32);
    // stream-ordered allocator: the block is reused only after the work queued on the calling
    // thread's active stream; a launch on some other stream has to finish first
    if (slices_ws == nullptr) {
      if (stream != active_stream()) CT2X_HIP_CHECK(hipStreamSynchronize(stream));
      device_free(slices);
    }
  }
}

void launch_awq_gemv(const void* x_f16, const int32_t* qweight, const void* scales_f16,
                     const int32_t* qzeros, int64_t m, int64_t in_c, int64_t out_c, int64_t group,
                     int64_t zeros_w, int64_t sf_w, void* y_f16, float* y_f32, hipStream_t stream,
                     bool bf16) {
  if (m == 0 || out_c == 0) return;
  if (in_c % 128 != 0 || group % 32 != 0 || group <= 0 ||
      (reinterpret_cast<uintptr_t>(x_f16) & 15) != 0 || (reinterpret_cast<uintptr_t>(qweight) & 15) != 0)
    throw Error("awq gemv: needs in_c % 128 == 0, group % 32 == 0 and 16-byte aligned x / weight");
#define CT2X_AWQ_GEMV(NT, BF, ELEM)                                                              \
  hipLaunchKernelGGL((awq_gemv_kernel<NT, BF>), dim3(static_cast<unsigned>((out_c + 16 * NT - 1) / (16 * NT)), \
                                                     static_cast<unsigned>((m + 15) / 16)),        \
                     dim3(256), 0, stream, static_cast<const ELEM*>(x_f16),                        \
                     reinterpret_cast<const unsigned*>(qweight),                                  \
                     static_cast<const ELEM*>(scales_f16), reinterpret_cast<const unsigned*>(qzeros), \
                     static_cast<int>(m), static_cast<int>(out_c), static_cast<int>(in_c),         \
                     static_cast<int>(group), static_cast<int>(zeros_w), static_cast<int>(sf_w),   \
                     static_cast<ELEM*>(y_f16), y_f32)
  // >= 256 workgroups where the shape allows: two column tiles for wide outputs, one otherwise
  if (bf16) {
    if (out_c >= 8192) CT2X_AWQ_GEMV(2, true, __bf16);
    else CT2X_AWQ_GEMV(1, true, __bf16);
  } else {
    if (out_c >= 8192) CT2X_AWQ_GEMV(2, false, _Float16);
    else CT2X_AWQ_GEMV(1, false, _Float16);
  }
#undef CT2X_AWQ_GEMV
}

}  // namespace ct2x
