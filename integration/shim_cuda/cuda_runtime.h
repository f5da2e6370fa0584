// Level-1 integration: the handful of CUDA runtime names that the reference's
// HOST sources leak through src/cuda/utils.h (src/cuda/utils.h:5-28,98-134) and use directly
// (src/devices.cc:81-116, src/utils.cc:60-67), mapped onto the runtime entry points of
// include/ct2_mi355.h. With this directory on the include path the reference's own .cc files
// compile with -DCT2_WITH_CUDA by a plain host compiler - no CUDA / HIP toolkit headers, no
// hipify - and link against libct2_mi355.so. Written for this repository; not reference code.
#pragma once

#include <cstddef>

#include "ct2_mi355.h"

typedef int cudaError_t;
static const cudaError_t cudaSuccess = 0;
typedef void* cudaStream_t;

struct cudaDeviceProp {
  char name[256];
  int major;
  int minor;
  size_t totalGlobalMem;
  int multiProcessorCount;
};

static inline const char* cudaGetErrorString(cudaError_t) { return ct2x_last_error(); }
static inline cudaError_t cudaGetDevice(int* index) { return ct2x_get_device(index); }
static inline cudaError_t cudaSetDevice(int index) { return ct2x_set_device(index); }
static inline cudaError_t cudaStreamSynchronize(cudaStream_t stream) {
  return ct2x_stream_synchronize(stream);
}
static inline cudaError_t cudaDeviceSynchronize() { return ct2x_device_synchronize(); }
