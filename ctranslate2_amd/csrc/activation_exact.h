// The reference's CPU activation arithmetic, restated once for the device AND for the host: the
// functors of the AVX2 dispatch (src/cpu/kernels.cc:146-207) that the fused Dense epilogue applies
// to every column (kernels.cc:664-680) are explicit polynomial / rational forms, not libm calls -
//   exp   = exp256_ps (third_party/avx_mathfun.h:290-344) with every mul + add pair contracted to
//           an FMA by the reference's compiler (verified against the compiled library),
//   tanh  = vec_tanh (src/cpu/vec.h:155-197, Eigen's clamped rational form, mul_add = FMA),
//   erf   = vec_erf  (src/cpu/vec.h:199-237, Abramowitz-Stegun 7.1.26 on exp256_ps)
// - so a device epilogue built from the same IEEE operations (mul, FMA, correctly rounded division,
// min / max, bit operations) returns the reference's bits. This header compiles for the host too
// (no HIP constructs): tests/test_host.py builds it with g++ and compares it with the oracle's
// independent C restatement, which is pinned bit for bit against the reference library
// (tests/test_oracle.py::test_ref_dequantize_exact) - the device code is checked on the CPU.
#pragma once

#include <math.h>

#if defined(__HIPCC__)
#define CT2X_HD __host__ __device__ __forceinline__
#else
#define CT2X_HD inline
#endif

namespace ct2x {

CT2X_HD float exact_exp_avx2(float x) {
  x = fminf(x, 88.3762626647949f);
  x = fmaxf(x, -88.3762626647949f);
  float fx = fmaf(x, 1.44269504088896341f, 0.5f);
  float tmp = floorf(fx);
  if (tmp > fx) tmp = tmp - 1.f;
  fx = tmp;
  x = fmaf(-fx, 0.693359375f, x);
  x = fmaf(-fx, -2.12194440e-4f, x);
  const float z = x * x;
  float y = 1.9875691500E-4f;
  y = fmaf(y, x, 1.3981999507E-3f);
  y = fmaf(y, x, 8.3334519073E-3f);
  y = fmaf(y, x, 4.1665795894E-2f);
  y = fmaf(y, x, 1.6666665459E-1f);
  y = fmaf(y, x, 5.0000001201E-1f);
  y = fmaf(y, z, x);
  y = y + 1.f;
  const float pow2n = __builtin_bit_cast(float, (static_cast<int>(fx) + 0x7f) << 23);
  return y * pow2n;
}

CT2X_HD float exact_tanh_avx2(float a) {
  float x = fminf(a, 7.90531110763549805f);
  x = fmaxf(x, -7.90531110763549805f);
  const bool tiny = fabsf(a) < 0.0004f;
  const float x2 = x * x;
  float p = fmaf(x2, -2.76076847742355e-16f, 2.00018790482477e-13f);
  p = fmaf(x2, p, -8.60467152213735e-11f);
  p = fmaf(x2, p, 5.12229709037114e-08f);
  p = fmaf(x2, p, 1.48572235717979e-05f);
  p = fmaf(x2, p, 6.37261928875436e-04f);
  p = fmaf(x2, p, 4.89352455891786e-03f);
  p = x * p;
  float q = fmaf(x2, 1.19825839466702e-06f, 1.18534705686654e-04f);
  q = fmaf(x2, q, 2.26843463243900e-03f);
  q = fmaf(x2, q, 4.89352518554385e-03f);
  return tiny ? x : p / q;
}

CT2X_HD float exact_erf_avx2(float a) {
  const unsigned bits = __builtin_bit_cast(unsigned, a);
  const unsigned sign = bits & 0x80000000u;
  const float abs_a = __builtin_bit_cast(float, bits ^ sign);
  const float tmp0 = fmaf(0.3275911f, abs_a, 1.0f);
  const float t = 1.0f / tmp0;
  const float tmp1 = fmaf(1.061405429f, t, -1.453152027f);
  const float tmp2 = fmaf(tmp1, t, 1.421413741f);
  const float tmp3 = fmaf(tmp2, t, -0.284496736f);
  const float r = fmaf(tmp3, t, 0.254829592f);
  const float pow_2 = a * a;
  const float tmp4 = exact_exp_avx2(-pow_2);
  const float tmp6 = (-tmp4) * t;
  const float tmp7 = fmaf(tmp6, r, 1.0f);
  return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, tmp7) ^ sign);
}

// expf exactly as the reference's libm computes it (glibc >= 2.27, sysdeps/ieee754/flt-32/e_expf.c: the
// ARM optimized-routines algorithm - 2^(k/32) table, degree-3 polynomial, everything in double, one
// final rounding - in the FMA build x86-64 machines dispatch to, where the compiler contracted the
// reduction r = InvLn2N * x - kd into one fma). The reference's CPU LogSoftMax sends the last
// depth mod 8 elements of a row (and whole rows of <= 8) through std::exp (src/cpu/kernels.cc:433-455
// scalar tail); this restatement returns libm's bits for EVERY float (compared exhaustively, all
// 2^32 inputs, by tests/test_host.py on the host build of this header). Table: asuint64(2^(i/32)) -
// (i << 47), the values of glibc's __exp2f_data.
CT2X_HD float exact_expf_glibc(float x) {
  const unsigned long long kTab[32] = {
    0x3ff0000000000000ull,
    0x3fefd9b0d3158574ull,
    0x3fefb5586cf9890full,
    0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull,
    0x3fef54873168b9aaull,
    0x3fef387a6e756238ull,
    0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull,
    0x3feef1a7373aa9cbull,
    0x3feedea64c123422ull,
    0x3feece086061892dull,
    0x3feebfdad5362a27ull,
    0x3feeb42b569d4f82ull,
    0x3feeab07dd485429ull,
    0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull,
    0x3fee9f75e8ec5f74ull,
    0x3feea11473eb0187ull,
    0x3feea589994cce13ull,
    0x3feeace5422aa0dbull,
    0x3feeb737b0cdc5e5ull,
    0x3feec49182a3f090ull,
    0x3feed503b23e255dull,
    0x3feee89f995ad3adull,
    0x3feeff76f2fb5e47ull,
    0x3fef199bdd85529cull,
    0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull,
    0x3fef7c97337b9b5full,
    0x3fefa4afa2a490daull,
    0x3fefd0765b6e4540ull
  };
  const double kInvLn2N = 0x1.71547652b82fep+0 * 32, kShift = 0x1.8p+52;
  const double kC0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32, kC1 = 0x1.ebfce50fac4f3p-3 / 32 / 32,
               kC2 = 0x1.62e42ff0c52d6p-1 / 32;
  const unsigned ux = __builtin_bit_cast(unsigned, x);
  const unsigned abstop = (ux >> 20) & 0x7ffu;
  if (abstop >= 0x42bu) {  // |x| >= 88 or NaN
    if (ux == 0xff800000u) return 0.0f;
    if (abstop >= 0x7f8u) return x + x;
    if (x > 0x1.62e42ep6f) return __builtin_huge_valf();
    if (x < -0x1.9fe368p6f) return 0.0f;
  }
  const double xd = static_cast<double>(x);
  const double z = kInvLn2N * xd;
  double kd = z + kShift;
  const unsigned long long ki = __builtin_bit_cast(unsigned long long, kd);
  kd -= kShift;
  const double r = fma(kInvLn2N, xd, -kd);
  const unsigned long long t = kTab[ki & 31u] + (ki << 47);
  const double s = __builtin_bit_cast(double, t);
  const double zz = kC0 * r + kC1;
  const double r2 = r * r;
  double y = kC2 * r + 1.0;
  y = zz * r2 + y;
  y = y * s;
  return static_cast<float>(y);
}

// ops::ActivationType 0..6 (include/ctranslate2/ops/activation.h:9-17) as the AVX2 functors compute
// them; any other value is the identity.
CT2X_HD float exact_activation(float v, int act) {
  switch (act) {
    case 0:  // ReLU
      return fmaxf(v, 0.f);
    case 1: {  // GELUTanh (gelu_tanh_func)
      float u = (v * v) * v;
      u = fmaf(0.044715f, u, v);
      u = 0.7978845608028654f * u;
      u = exact_tanh_avx2(u);
      u = 1.f + u;
      u = v * u;
      return 0.5f * u;
    }
    case 2:  // Swish (swish_func)
      return v / (1.f + exact_exp_avx2(-v));
    case 3: {  // GELU (gelu_func)
      float u = 0.7071067811865475f * v;
      u = 1.f + exact_erf_avx2(u);
      u = v * u;
      return 0.5f * u;
    }
    case 4:  // GELUSigmoid (gelu_sigmoid_func)
      return v / (1.f + exact_exp_avx2(-1.702f * v));
    case 5:  // Tanh
      return exact_tanh_avx2(v);
    case 6:  // Sigmoid (sigmoid_func)
      return 1.f / (1.f + exact_exp_avx2(-v));
    default:
      return v;
  }
}

}  // namespace ct2x
