// Host build of the product's activation arithmetic (ctranslate2_amd/csrc/activation_exact.h) for
// tests/test_host.py: the same source the HIP kernels compile is compared here, on the CPU, with
// the oracle's independent C restatement (which is pinned against the reference library).
//   g++ -O2 -std=c++17 -ffp-contract=off -mfma -shared -fPIC -o activation_host.so activation_host.cc
#include "../../ctranslate2_amd/csrc/activation_exact.h"

#include <cstdint>

extern "C" void host_activation(const float* x, int64_t n, int act, float* y) {
  for (int64_t i = 0; i < n; ++i) y[i] = ct2x::exact_activation(x[i], act);
}
extern "C" void host_exp_avx2(const float* x, int64_t n, float* y) {
  for (int64_t i = 0; i < n; ++i) y[i] = ct2x::exact_exp_avx2(x[i]);
}

// exact_expf_glibc against this machine's libm over EVERY float (NaNs skipped): mismatch count.
#include <cstring>
extern "C" long long host_expf_glibc_exhaustive() {
  long long bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
  for (long long b = 0; b < (1ll << 32); ++b) {
    const uint32_t u = static_cast<uint32_t>(b);
    float x;
    std::memcpy(&x, &u, 4);
    if (x != x) continue;
    const float a = ct2x::exact_expf_glibc(x), e = expf(x);
    uint32_t ua, ue;
    std::memcpy(&ua, &a, 4);
    std::memcpy(&ue, &e, 4);
    bad += ua != ue;
  }
  return bad;
}
