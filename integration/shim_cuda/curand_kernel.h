// See cuda_runtime.h in this directory: src/cuda/random.h names the cuRAND state type in two
// declarations (get_curand_states / free_curand_states). Random sampling on this backend uses the
// counter-based draws of decode_ops.hip; there are no per-thread generator states to keep.
#pragma once
struct curandStatePhilox4_32_10_t {
  unsigned int opaque[16];
};
