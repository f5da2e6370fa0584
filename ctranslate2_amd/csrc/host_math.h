// Host-side vector sin / cos for the tables the reference builds on the CPU before a decode
// starts (sinusoidal position encodings, src/layers/common.cc:204-229; RoPE tables,
// src/layers/attention_layer.cc:252-343). The reference routes ops::Sin / ops::Cos on
// Device::CPU to Intel MKL's VML (vsSin / vsCos, src/cpu/primitives.cc:378-394) whenever it is
// built with MKL - and VML is neither libm nor correctly rounded: 4-5 % of a position table's
// entries differ from sinf / cosf by one ulp, which is enough to flip int8 activations in every
// sentence (tests/test_parity_taps_gpu.py). So, like the reference, this backend uses an MKL
// runtime for these two functions when one is installed (resolved with dlopen, as RCCL is) and
// libm otherwise. CT2X_HOST_MATH=libm | mkl forces the choice; CT2X_MKL_PATH adds a directory to
// the search.
#pragma once

#include <cstdint>

namespace ct2x {

// y[i] = sin(x[i]) / cos(x[i]) for i < n.
void host_vsin(int64_t n, const float* x, float* y);
void host_vcos(int64_t n, const float* x, float* y);
// "mkl-vml" or "libm".
const char* host_math_backend();

}  // namespace ct2x
