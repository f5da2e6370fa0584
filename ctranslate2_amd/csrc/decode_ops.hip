// Device-side pieces of the decoding loop (src/decoding.cc) so that consecutive greedy steps
// need no host round trip: min-length EOS masking, optional log-softmax, arg-max, feeding the
// sampled id to the next step and appending it to a device-resident history.

#include "device_utils.h"
#include "kernels.h"

#include <float.h>
#include <limits.h>

namespace ct2x {

__global__ void greedy_sample_kernel(const float* __restrict__ logits, int64_t vocab,
                                     int32_t disable_id, int32_t min_length, bool with_scores,
                                     const int32_t* __restrict__ step_ptr,
                                     int32_t* __restrict__ next_ids,
                                     int32_t* __restrict__ hist_ids,
                                     float* __restrict__ hist_scores) {
  __shared__ float s_v[16];
  __shared__ int s_i[16];
  __shared__ float scratch[16];
  __shared__ float best_v;
  __shared__ int best_i;
  const int64_t row = blockIdx.x;
  const int64_t rows = gridDim.x;
  const int step = *step_ptr;
  const int masked = (step < min_length) ? disable_id : -1;
  const float* x = logits + row * vocab;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nwaves = (blockDim.x + 63) >> 6;

  ValIdx best;
  best.v = -FLT_MAX;
  best.i = INT_MAX;
  const bool vec = (vocab % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  if (vec) {
    // 16-byte loads, 8 in flight per lane before the first compare (latency, not bandwidth,
    // bounds a 128 KB row scan from one workgroup).
    constexpr int kU = 8;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const int64_t n4 = vocab / 4;
    for (int64_t base = threadIdx.x; base < n4; base += static_cast<int64_t>(blockDim.x) * kU) {
      float4 v[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int64_t i = base + static_cast<int64_t>(u) * blockDim.x;
        v[u] = (i < n4) ? x4[i] : make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX);
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int64_t i = base + static_cast<int64_t>(u) * blockDim.x;
        if (i >= n4) continue;
        const float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int j = static_cast<int>(i * 4 + k);
          ValIdx c;
          c.v = (j == masked) ? -FLT_MAX : e[k];  // DisableTokens: lowest() (decoding_utils.cc:19-35)
          c.i = j;
          best = better(best, c);
        }
      }
    }
  } else {
    for (int64_t j = threadIdx.x; j < vocab; j += blockDim.x) {
      ValIdx c;
      c.v = (j == masked) ? -FLT_MAX : x[j];
      c.i = static_cast<int>(j);
      best = better(best, c);
    }
  }
  best = wave_argmax(best);
  if (lane == 0) {
    s_v[wave] = best.v;
    s_i[wave] = best.i;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    ValIdx r;
    r.v = s_v[0];
    r.i = s_i[0];
    for (int w = 1; w < nwaves; ++w) {
      ValIdx c;
      c.v = s_v[w];
      c.i = s_i[w];
      r = better(r, c);
    }
    best_v = r.v;
    best_i = r.i;
  }
  __syncthreads();
  const float m = best_v;
  float score = m;
  if (with_scores) {
    // LogSoftMax (src/cpu/kernels.cc:441-451): y = x + (-max - log(sum exp(x - max))).
    float s = 0.f;
    for (int64_t j = threadIdx.x; j < vocab; j += blockDim.x) {
      const float v = (j == masked) ? -FLT_MAX : x[j];
      s += expf(v - m);
    }
    s = block_sum(s, scratch);
    score = m + (-m - logf(s));
  }
  if (threadIdx.x == 0) {
    next_ids[row] = best_i;
    hist_ids[static_cast<int64_t>(step) * rows + row] = best_i;
    hist_scores[static_cast<int64_t>(step) * rows + row] = score;
  }
}

void launch_greedy_sample(const float* logits, int64_t rows, int64_t vocab, int32_t disable_id,
                          int32_t min_length, bool with_scores, const int32_t* step_ptr,
                          int32_t* next_ids, int32_t* hist_ids, float* hist_scores,
                          hipStream_t stream) {
  if (rows == 0) return;
  int block = 64;
  while (block < vocab / 4 && block < 1024) block <<= 1;
  hipLaunchKernelGGL(greedy_sample_kernel, dim3(rows), dim3(block), 0, stream, logits, vocab,
                     disable_id, min_length, with_scores, step_ptr, next_ids, hist_ids,
                     hist_scores);
}

__global__ void advance_step_kernel(int32_t* step_ptr) { *step_ptr = *step_ptr + 1; }

void launch_advance_step(int32_t* step_ptr, hipStream_t stream) {
  hipLaunchKernelGGL(advance_step_kernel, dim3(1), dim3(1), 0, stream, step_ptr);
}

__global__ void beam_logprobs_kernel(const float* __restrict__ logits,
                                     const float* __restrict__ beam_scores, int64_t vocab,
                                     int32_t disable_id, int32_t min_length,
                                     const int32_t* __restrict__ step_ptr,
                                     float* __restrict__ y) {
  __shared__ float scratch[16];
  const int64_t row = blockIdx.x;
  const int step = *step_ptr;
  const int masked = (step < min_length) ? disable_id : -1;
  const float* x = logits + row * vocab;
  float* yr = y + row * vocab;
  float m = -FLT_MAX;
  for (int64_t j = threadIdx.x; j < vocab; j += blockDim.x)
    m = fmaxf(m, (j == masked) ? -FLT_MAX : x[j]);
  m = block_max(m, scratch);
  float s = 0.f;
  for (int64_t j = threadIdx.x; j < vocab; j += blockDim.x)
    s += expf(((j == masked) ? -FLT_MAX : x[j]) - m);
  s = block_sum(s, scratch);
  const float c = -m - logf(s);
  const float b = beam_scores ? beam_scores[row] : 0.f;
  for (int64_t j = threadIdx.x; j < vocab; j += blockDim.x) {
    const float v = ((j == masked) ? -FLT_MAX : x[j]) + c;
    // add_depth_broadcast of the cumulative beam score (src/decoding.cc:548-554)
    yr[j] = beam_scores ? b + v : v;
  }
}

void launch_beam_logprobs(const float* logits, const float* beam_scores, int64_t rows,
                          int64_t vocab, int32_t disable_id, int32_t min_length,
                          const int32_t* step_ptr, float* y, hipStream_t stream) {
  if (rows == 0) return;
  int block = 64;
  while (block < vocab / 4 && block < 1024) block <<= 1;
  hipLaunchKernelGGL(beam_logprobs_kernel, dim3(rows), dim3(block), 0, stream, logits,
                     beam_scores, vocab, disable_id, min_length, step_ptr, y);
}

}  // namespace ct2x
