// Host-side vector sin / cos for the tables the reference builds on the CPU before a decode
// starts (sinusoidal position encodings, src/layers/common.cc:204-229; RoPE tables,
// src/layers/attention_layer.cc:252-343). The reference routes ops::Sin / ops::Cos on
// Device::CPU to Intel MKL's VML (vsSin / vsCos, src/cpu/primitives.cc:378-394) whenever it is
// built with MKL - and VML is neither libm nor correctly rounded: 4-5 % of a position table's
// entries differ from sinf / cosf by one ulp, which is enough to flip int8 activations in every
// sentence (tests/test_parity_taps_gpu.py). Round 3: VML's two kernels are RESTATED here (from the
// library's disassembly, pinned bit for bit against it on 38.8 million arguments), so the tables -
// and with them token parity - no longer depend on an MKL runtime being installed on the GPU host.
// CT2X_HOST_MATH=mkl uses an installed MKL instead (dlopen, as RCCL is; CT2X_MKL_PATH adds a
// search directory), CT2X_HOST_MATH=libm the C library (NOT bit-identical to the reference).
#pragma once

#include <cstdint>

namespace ct2x {

// y[i] = sin(x[i]) / cos(x[i]) for i < n.
void host_vsin(int64_t n, const float* x, float* y);
void host_vcos(int64_t n, const float* x, float* y);
// "vml-restated" (default), "mkl-vml" or "libm".
const char* host_math_backend();
// Arguments beyond the restated range (|x| > 10000) that fell back to libm since process start.
int64_t host_math_outside_range();

}  // namespace ct2x
