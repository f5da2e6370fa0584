// Device-side helpers shared by the gfx950 kernels (wave64 reductions, vector types,
// activation functors). CDNA4 only: wavefront = 64 lanes, no 32-lane assumptions.
#pragma once

#include <float.h>

#include <hip/hip_runtime.h>

#include "activation_exact.h"
#include <stdint.h>

namespace ct2x {

constexpr int kWave = 64;

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

// Wave64 reductions on the DPP network (row_shr 1/2/4/8 inside the 16-lane rows, then row_bcast
// 15 / 31 across rows; total in lane 63, broadcast by v_readlane): ~7 VALU issues instead of six
// dependent ds_bpermute round trips (~130 cycles each) per reduction.
template <int kCtrl, int kRowMask>
__device__ __forceinline__ float dpp_move(float old, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), kCtrl,
                                                    kRowMask, 0xf, false));
}

// The maximum is order independent: every caller gets the DPP form (measured on the norm +
// Quantize launch: the ds_bpermute butterfly cost ~700 of the kernel's 5900 cycles).
__device__ __forceinline__ float wave_max(float v) {
  const float ninf = -__builtin_huge_valf();
  v = fmaxf(v, dpp_move<0x111, 0xf>(ninf, v));
  v = fmaxf(v, dpp_move<0x112, 0xf>(ninf, v));
  v = fmaxf(v, dpp_move<0x114, 0xf>(ninf, v));
  v = fmaxf(v, dpp_move<0x118, 0xf>(ninf, v));
  v = fmaxf(v, dpp_move<0x142, 0xa>(ninf, v));
  v = fmaxf(v, dpp_move<0x143, 0xc>(ninf, v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block-wide reductions through a small LDS scratch (one float per wave, up to 16 waves).
// All threads of the block must call; every thread receives the result.
__device__ __forceinline__ float block_max(float v, float* scratch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nwaves = (blockDim.x + 63) >> 6;
  v = wave_max(v);
  __syncthreads();  // protect scratch reuse
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  float r = scratch[0];
  for (int i = 1; i < nwaves; ++i) r = fmaxf(r, scratch[i]);
  return r;
}

__device__ __forceinline__ float block_sum(float v, float* scratch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nwaves = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  float r = scratch[0];
  for (int i = 1; i < nwaves; ++i) r += scratch[i];
  return r;
}

// (value, index) argmax with lowest-index tie-break == std::max_element
// (reference CPU TopK, src/ops/topk_cpu.cc:27).
struct ValIdx {
  float v;
  int i;
};

__device__ __forceinline__ ValIdx better(ValIdx a, ValIdx b) {
  // a wins if larger, or equal with a lower index. Empty slots carry i = INT_MAX and
  // always lose (even against -inf).
  if (a.i == 0x7fffffff) return b;
  if (b.i == 0x7fffffff) return a;
  if (b.v > a.v || (b.v == a.v && b.i < a.i)) return b;
  return a;
}

__device__ __forceinline__ ValIdx wave_argmax(ValIdx x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ValIdx y;
    y.v = __shfl_xor(x.v, o, 64);
    y.i = __shfl_xor(x.i, o, 64);
    x = better(x, y);
  }
  return x;
}

// Activation epilogues: float math identical in form to the reference CPU functors
// (src/cpu/kernels.cc:146-207).
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += dpp_move<0x111, 0xf>(0.f, v);
  v += dpp_move<0x112, 0xf>(0.f, v);
  v += dpp_move<0x114, 0xf>(0.f, v);
  v += dpp_move<0x118, 0xf>(0.f, v);
  v += dpp_move<0x142, 0xa>(0.f, v);
  v += dpp_move<0x143, 0xc>(0.f, v);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max_dpp(float v) {
  v = fmaxf(v, dpp_move<0x111, 0xf>(-FLT_MAX, v));
  v = fmaxf(v, dpp_move<0x112, 0xf>(-FLT_MAX, v));
  v = fmaxf(v, dpp_move<0x114, 0xf>(-FLT_MAX, v));
  v = fmaxf(v, dpp_move<0x118, 0xf>(-FLT_MAX, v));
  v = fmaxf(v, dpp_move<0x142, 0xa>(-FLT_MAX, v));
  v = fmaxf(v, dpp_move<0x143, 0xc>(-FLT_MAX, v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// (value descending, index ascending) arg-max of a wave in two DPP reductions: the maximum, then
// the lowest index among the lanes that hold it. Empty slots are (v = -inf, i = INT_MAX): a real
// entry with the same value always has the lower index, so it wins (as in better()).
template <int kCtrl, int kRowMask>
__device__ __forceinline__ int dpp_move_i32(int old, int v) {
  return __builtin_amdgcn_update_dpp(old, v, kCtrl, kRowMask, 0xf, false);
}
__device__ __forceinline__ ValIdx wave_argmax_dpp(ValIdx x) {
  const float ninf = -__builtin_huge_valf();
  float m = x.v;
  m = fmaxf(m, dpp_move<0x111, 0xf>(ninf, m));
  m = fmaxf(m, dpp_move<0x112, 0xf>(ninf, m));
  m = fmaxf(m, dpp_move<0x114, 0xf>(ninf, m));
  m = fmaxf(m, dpp_move<0x118, 0xf>(ninf, m));
  m = fmaxf(m, dpp_move<0x142, 0xa>(ninf, m));
  m = fmaxf(m, dpp_move<0x143, 0xc>(ninf, m));
  m = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 63));
  int c = (x.v == m) ? x.i : 0x7fffffff;
  c = min(c, dpp_move_i32<0x111, 0xf>(0x7fffffff, c));
  c = min(c, dpp_move_i32<0x112, 0xf>(0x7fffffff, c));
  c = min(c, dpp_move_i32<0x114, 0xf>(0x7fffffff, c));
  c = min(c, dpp_move_i32<0x118, 0xf>(0x7fffffff, c));
  c = min(c, dpp_move_i32<0x142, 0xa>(0x7fffffff, c));
  c = min(c, dpp_move_i32<0x143, 0xc>(0x7fffffff, c));
  ValIdx r;
  r.v = m;
  r.i = __builtin_amdgcn_readlane(c, 63);
  return r;
}

__device__ __forceinline__ float exp_avx2(float x);

// Correctly rounded t / d in five full-rate VALU operations instead of the 11-instruction IEEE
// sequence the compiler emits (v_div_scale x2, v_rcp, 5 fma, v_div_fmas, v_div_fixup) - for epilogues
// that divide many values by the same per-column scale (the CPU formula order of the reference's
// Dequantize keeps a true division: src/cpu/kernels.cc:666-673). y = 1.f / d is the correctly
// rounded reciprocal (one real division per column); q0 = RN(t y) is within 2 ulp of t / d, one
// residual correction makes it faithful, and by Markstein's theorem (y = RN(1/d), faithful q,
// exact residual by FMA) the second correction returns RN(t / d). No step may underflow or
// overflow: callers guarantee d in [2^-30, 2^30] (exact_div_divisor_ok) and t = int32 * r with
// r in [2^-29, 2^29] (exact_div_factor_ok), i.e. |t| in {0} U [2^-29, 2^60]; anything else takes
// the plain division. Checked against t / d on 10^11 random and adversarial pairs on the CPU
// (oracle/div_check.c, tests/test_oracle.py::test_exact_division_identity).
__device__ __forceinline__ float exact_div(float t, float d, float y) {
  const float q0 = t * y;
  const float r0 = __builtin_fmaf(-d, q0, t);
  const float q1 = __builtin_fmaf(r0, y, q0);
  const float r1 = __builtin_fmaf(-d, q1, t);
  return __builtin_fmaf(r1, y, q1);
}
__device__ __forceinline__ bool exact_div_divisor_ok(float d) {
  return d >= 9.313225746154785e-10f /* 2^-30 */ && d <= 1073741824.f /* 2^30 */;
}
__device__ __forceinline__ bool exact_div_factor_ok(float r) {
  return r >= 1.862645149230957e-09f /* 2^-29 */ && r <= 536870912.f /* 2^29 */;
}

// ops::ActivationType 0..6 exactly as the reference's AVX2 functors compute them (activation_exact.h:
// exp256_ps with FMA contraction, Eigen's rational tanh, the Abramowitz-Stegun erf) - round 2 ran
// GELU / GELUTanh / GELUSigmoid / Tanh / Sigmoid through the device libm ("close"); the same header
// is compiled for the host and pinned against the oracle and the reference library there.
__device__ __forceinline__ float apply_activation(float v, int act) {
  return exact_activation(v, act);
}

// Quantisation mode flags (the `round_before_cast` argument of the C ABI):
//   bit 0: round to nearest even before the cast (binary_version >= 5 models), else truncate;
//   bit 1: reproduce the reference CPU backends' u8 domain shift. On x86 the reference runs
//          s8 x s8 GEMMs as u8 x s8 (MKL / oneDNN, cpu::prefer_u8s8s32_gemm), so Dense quantises
//          with shift_to_uint8: q_u8 = cvtt(round_func(x*scale - (-128))) and compensates
//          -128*sum(W) afterwards (src/cpu/kernels.cc:592-604, src/cpu/primitives.cc:869-898).
//          The effective signed activation is q_u8 - 128. That differs from cvtt(round_func(x*scale))
//          (a) for every negative value in truncation mode (floor vs trunc-toward-zero) and
//          (b) at fp32 rounding boundaries of the +128 add in rounding mode. Token parity with
//          the CPU oracle needs the same integers, so the engine sets this bit by default.
constexpr int kQuantRound = 1;
constexpr int kQuantU8Shift = 2;

// float -> int8 exactly as the reference CPU path: optional round-to-nearest-even
// (Vec::round = _mm256_round_ps NEAREST, src/cpu/vec_avx.h:192-202) then truncating
// conversion (cvttps) and a narrowing store.
// `x`, `scale`: the product is formed here because the u8-shift arm is a fused multiply-add in
// the reference's compiled AVX2 kernels (x*scale + 128 with a single rounding).
__device__ __forceinline__ int8_t to_s8(float x, float scale, int mode) {
  float v;
  if (mode & kQuantU8Shift) {
    v = fmaf(x, scale, 128.f);
    if (mode & kQuantRound) v = rintf(v);
    const int u = static_cast<int>(v) & 0xff;  // narrowing int32 -> uint8 store
    return static_cast<int8_t>(u - 128);
  }
  v = x * scale;
  if (mode & kQuantRound) v = rintf(v);
  return static_cast<int8_t>(static_cast<int>(v));
}

// exp256_ps (third_party/avx_mathfun.h:290-344) as compiled into the reference's AVX2 kernels:
// every mul+add pair of the Cephes polynomial is one FMA.
__device__ __forceinline__ float exp_avx2(float x) { return exact_exp_avx2(x); }
// std::exp of the reference's scalar tails = glibc's expf, restated (activation_exact.h)
__device__ __forceinline__ float expf_glibc(float x) { return exact_expf_glibc(x); }

// logf exactly as glibc computes it (sysdeps/ieee754/flt-32/e_logf.c, the ARM optimized-routines
// algorithm, MIT): 16-entry table of (1/c, log c), a degree-3 polynomial, everything in double,
// one final rounding. The reference's CPU LogSoftMax ends with std::log(exp_sum)
// (src/cpu/kernels.cc:449); glibc's logf is NOT correctly rounded (0.17 percent of inputs differ
// from the rounded double log), and a one-ulp shift of a whole log-probability row is enough to
// swap near-tied beams. The constants were read out of the libm.so.6 of the authoring image
// (glibc 2.35) and the restatement checked bit for bit against it on 5e5 inputs. x must be a
// positive normal float (a softmax denominator is >= 1).
__device__ __forceinline__ float logf_glibc(float x) {
  static constexpr double kTab[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2},
    {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
    {0x1.49539f0f010b0p+0, -0x1.01eae7f513a67p-2},
    {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
    {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3},
    {0x1.25e227b0b8ea0p+0, -0x1.1aa2bc79c8100p-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4},
    {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
    {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5},
    {0x1.0000000000000p+0, 0x0.0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},
    {0x1.ca4b31f026aa0p-1, 0x1.c5e53aa362eb4p-4},
    {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},
    {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d224770p-3},
    {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},
    {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2},
  };
  constexpr double kLn2 = 0x1.62e42fefa39efp-1;
  constexpr double kA0 = -0x1.00ea348b88334p-2, kA1 = 0x1.5575b0be00b6ap-2, kA2 = -0x1.ffffef20a4123p-2;
  const uint32_t ix = __float_as_uint(x);
  const uint32_t tmp = ix - 0x3f330000u;
  const int i = static_cast<int>((tmp >> 19) & 15);
  const int k = static_cast<int32_t>(tmp) >> 23;
  const uint32_t iz = ix - (tmp & 0xff800000u);
  const double z = static_cast<double>(__uint_as_float(iz));
  const double r = z * kTab[i][0] - 1.0;
  const double y0 = kTab[i][1] + static_cast<double>(k) * kLn2;
  const double r2 = r * r;
  double y = kA1 * r + kA2;
  y = kA0 * r2 + y;
  y = y * r2 + (y0 + r);
  return static_cast<float>(y);
}

// Sum of e[0, size) in the order of vectorized_reduce_all<AVX2> (src/cpu/kernels.cc:86-139):
// eight strided partial sums, reduce_m256 (lane l + l^4, + l^2, + l^1), then the scalar tail; a
// row of <= 8 elements is summed front to back. `e` is LDS (or global) memory written by this
// wave; every lane of the wave returns the sum.
__device__ __forceinline__ float dpp_xor4(float v) {
  const float m = dpp_move<0x141, 0xf>(0.f, v);  // row_half_mirror: lane i <- lane 7 - i (per 8)
  return dpp_move<0x1B, 0xf>(0.f, m);             // quad_perm [3,2,1,0]: lane i <- lane i ^ 3
}

// ordered_sum8 for rows of <= 64 elements (the decode attention's scores): the <= 8 addends of
// every chain and the <= 7 tail elements are requested together (one LDS round trip instead of one
// per addend; adding the +0.f of an absent addend leaves a sum of exps unchanged) and the
// 4 / 2 / 1 tree runs on the DPP network. Longer rows take the loop form.
__device__ __forceinline__ float ordered_sum8(const float* e, int size);
__device__ __forceinline__ float ordered_sum8_short(const float* e, int size) {
  if (size > 64 || size <= 8) return ordered_sum8(e, size);
  const int l = threadIdx.x & 7;
  const int vsize = size & ~7;
  float a[8], t[7];
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = (l + 8 * k < vsize) ? e[l + 8 * k] : 0.f;
#pragma unroll
  for (int k = 0; k < 7; ++k) t[k] = (vsize + k < size) ? e[vsize + k] : 0.f;
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) acc = acc + a[k];
  acc = acc + dpp_xor4(acc);
  acc = acc + dpp_move<0x4E, 0xf>(0.f, acc);
  acc = acc + dpp_move<0xB1, 0xf>(0.f, acc);
#pragma unroll
  for (int k = 0; k < 7; ++k) acc = acc + t[k];
  return acc;
}

__device__ __forceinline__ float ordered_sum8(const float* e, int size) {
  const int lane = threadIdx.x & 63;
  float acc = 0.f;
  if (size <= 8) {
    for (int j = 0; j < size; ++j) acc += e[j];  // every lane reads the same addresses
    return acc;
  }
  const int vsize = size & ~7;
  const int l = lane & 7;
  for (int j = l; j < vsize; j += 8) acc += e[j];
  acc = acc + __shfl_xor(acc, 4, 64);
  acc = acc + __shfl_xor(acc, 2, 64);
  acc = acc + __shfl_xor(acc, 1, 64);
  for (int j = vsize; j < size; ++j) acc += e[j];
  return acc;
}

// ---- ordered sums over long rows (vocabulary LogSoftMax) ------------------------------------------
// vectorized_reduce_all<AVX2> is eight strided partial sums, each a dependent chain of size / 8
// additions (4000 for a 32000-word vocabulary: ~8 us at one add per ~4 cycles - the price of the
// reference's order). What must stay off the chain is memory latency: the addends are parked
// TRANSPOSED in LDS - chain l owns the contiguous run e[l], e[l + 8], ... - and walked with
// hand-pipelined 16-byte reads (fenced with sched_barrier, see ordered_row_sums in rowops.hip).
struct ExpSumLayout {
  int lds_floats;  // 0: the row does not fit, use the global park row
  int chain_pad;   // floats per chain in LDS (a multiple of 4, = 4 mod 8: conflict-free 8-lane reads)
};

inline ExpSumLayout exp_sum_layout(int64_t vocab) {
  ExpSumLayout l{0, 0};
  const int64_t chain = vocab > 8 ? vocab / 8 : 0;
  int64_t pad = (chain + 3) / 4 * 4;
  while (pad % 8 != 4) pad += 4;
  const int64_t floats = 8 * pad + 8 + 128;  // chains + tail + read-ahead slack (3 batches of 8 x 4)
  if (chain > 0 && floats * 4 <= 152 * 1024) {
    l.lds_floats = static_cast<int>(floats);
    l.chain_pad = static_cast<int>(pad);
  }
  return l;
}


// Spare workgroup `wg` of `nwg` (256 threads each) of a row-wide launch: request its share of up to
// two weight ranges (kernels.h PrefetchArgs) and drop the data - the point is that the lines are in
// the memory-side cache when the next GEMM streams them.
template <typename Args>
__device__ __forceinline__ void weight_prefetch_wg(const Args& pf, int wg, int nwg) {
  unsigned acc = 0;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const uint4* p = static_cast<const uint4*>(pf.r[k].p);
    if (p == nullptr) continue;
    const int64_t take16 = pf.r[k].take16, unit16 = pf.r[k].unit16, chunks = pf.r[k].chunks;
    const int64_t stride = static_cast<int64_t>(nwg) * 256;
    for (int64_t c = static_cast<int64_t>(wg) * 256 + threadIdx.x; c < chunks; c += 8 * stride) {
      uint4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int64_t cc = c + j * stride;
        v[j] = make_uint4(0, 0, 0, 0);
        if (cc < chunks) {
          const int64_t u = cc / take16;
          v[j] = p[u * unit16 + (cc - u * take16)];
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    }
  }
  asm volatile("" ::"v"(acc));  // the loads are the point
}

// > 64 KiB of dynamic LDS has to be opted into per kernel.
inline void allow_big_lds(const void* fn, size_t bytes) {
  // (the limit covers static + dynamic LDS: ask for what the launch uses, not for the maximum)
  if (bytes > 48 * 1024)
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
}

// Where element j < vsize of the row is parked.
__device__ __forceinline__ int chain_slot(int64_t j, int chain_pad) {
  return static_cast<int>((j & 7) * chain_pad + (j >> 3));
}

// Sum in the reference's order of a row parked by chain_slot (elements [0, vsize), vsize a
// multiple of 8) followed by the scalar tail lds[8 * chain_pad + t] = e[vsize + t]. Called by the
// 64 lanes of ONE wave after a barrier; every lane returns the sum.
__device__ __forceinline__ float ordered_chain_sum_lds(const float* lds, int chain_pad, int64_t vsize,
                                                       int64_t size) {
  const float* tail = lds + 8 * chain_pad;
  float acc = 0.f;
  {
    // One dependent v_add_f32 per element (~10 cycles each on this chip: 4000 elements of a
    // 32000-word row = ~17 us, the floor of the reference's order). The loop is left to the
    // compiler's schedule - contiguous 16-byte LDS reads unrolled eight deep stay ahead of the
    // adds (tools/chain_probe.cpp: a hand-pipelined sched_barrier version was 2x slower) - and
    // carries no per-element bounds: whole float4s first, then the <= 3 leftovers.
    const int n = static_cast<int>(vsize / 8);  // elements per chain
    const float* chain = lds + (threadIdx.x & 7) * chain_pad;
    const v4f* c4 = reinterpret_cast<const v4f*>(chain);
    const int n4 = n / 4;
#pragma unroll 8
    for (int i = 0; i < n4; ++i) {
      const v4f v = c4[i];
      acc = acc + v.x;
      acc = acc + v.y;
      acc = acc + v.z;
      acc = acc + v.w;
    }
    for (int e = 4 * n4; e < n; ++e) acc = acc + chain[e];
  }
  acc = acc + dpp_xor4(acc);
  acc = acc + dpp_move<0x4E, 0xf>(0.f, acc);
  acc = acc + dpp_move<0xB1, 0xf>(0.f, acc);
  for (int64_t j = vsize; j < size; ++j) acc = acc + tail[j - vsize];
  return acc;
}

}  // namespace ct2x
