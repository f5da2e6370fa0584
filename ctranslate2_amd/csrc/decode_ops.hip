// Device-side pieces of the decoding loop (src/decoding.cc) so that consecutive greedy steps
// need no host round trip: min-length EOS masking, optional log-softmax, arg-max, feeding the
// sampled id to the next step and appending it to a device-resident history.

#include "device_utils.h"
#include "grid_sync.h"
#include "kernels.h"
#include "runtime.h"

#include <cstdlib>
#include <algorithm>
#include <float.h>
#include <limits.h>

namespace ct2x {

// Hard prefix (update_sample_with_prefix, src/decoding.cc:21-62, greedy arm): while the step,
// counted from prefix.first_step, is inside a row's prefix the sampled id is replaced by the prefix
// token and its score by 0; EOS stays disabled for prefix_len + min_length steps
// (apply_min_length, src/decoding.cc:384-408).
struct SamplePrefix {
  const int32_t* ids;   // [rows, stride] or nullptr
  const int32_t* lens;  // [rows]
  int32_t stride;
  int32_t first_step;
  int32_t extends_min_length;  // return_prefix == false: EOS stays disabled prefix_len steps longer
};

// Last-row-out advances the device-resident step counter (one launch less per step than a
// separate advance_step kernel). Every workgroup reads *step_ptr at its start and the counter is
// only written once ALL rows have finished, so no row can observe the incremented value; the
// kernel boundary publishes it to the next step's kernels. Called by thread 0 of each row.
// No fence: nothing written here is read again inside this launch (an agent-scope fence on
// gfx950 writes back and invalidates the XCD's whole L2 - measured 2x on the decode step when
// every workgroup of a launch executes one).
struct StepAdvance {
  int32_t* step_ptr;  // nullptr: the caller advances the step itself
  int32_t* done;      // zero between launches
};

__device__ __forceinline__ void row_finished(const StepAdvance& adv, int64_t rows, int step) {
  if (adv.step_ptr == nullptr) return;
  if (atomicAdd(adv.done, 1) == static_cast<int>(rows) - 1) {
    *adv.done = 0;
    *adv.step_ptr = step + 1;
  }
}

__device__ __forceinline__ bool forced_by_prefix(const SamplePrefix& pf, const StepAdvance& adv,
                                                 int64_t row, int64_t rows,
                                                 int step, int32_t* next_ids, int32_t* hist_ids,
                                                 float* hist_scores) {
  if (pf.ids == nullptr) return false;
  const int local = step - pf.first_step;
  if (local < 0 || local >= pf.lens[row]) return false;
  if (threadIdx.x == 0) {
    const int32_t id = pf.ids[row * pf.stride + local];
    next_ids[row] = id;
    hist_ids[static_cast<int64_t>(step) * rows + row] = id;
    hist_scores[static_cast<int64_t>(step) * rows + row] = 0.f;
    row_finished(adv, rows, step);
  }
  return true;
}

// Denominator of the reference's LogSoftMax over one vocabulary row, bit for bit
// (cpu::softmax<AVX2> log arm, src/cpu/kernels.cc:430-451; oracle orc_softmax): exp256_ps of
// x - max for the vector body, libm expf for the scalar tail (vocab % 8 elements - vocabularies
// here are padded to multiples of 8 or 16, where the tail is empty and the result exact), eight
// strided partial sums + reduce_m256 + tail. Each partial sum is a dependent chain of vocab / 8
// additions (4000 for a 32000-word vocabulary: ~8 us at one add per ~4 cycles - the price of the
// reference's order); what must stay off the chain is memory latency, so the exps are parked
// TRANSPOSED in LDS - chain l owns the contiguous run e[l], e[l + 8], ... - and walked with
// hand-pipelined 16-byte reads. Rows too long for LDS park in global memory (`park`) and walk it
// with ordered_sum8 (slow, exact). All threads of the block call; every thread receives the sum.
__device__ __forceinline__ float ordered_row_exp_sum(const float* __restrict__ x, int64_t vocab,
                                                     int masked, float m, float* __restrict__ park,
                                                     float* lds, int chain_pad, float* scratch) {
  const int64_t vsize = vocab > 8 ? vocab - vocab % 8 : 0;
  if (lds == nullptr) {
    for (int64_t j = threadIdx.x; j < vocab; j += blockDim.x) {
      const float v = ((j == masked) ? -FLT_MAX : x[j]) - m;
      park[j] = (j < vsize) ? exp_avx2(v) : expf_glibc(v);
    }
    __syncthreads();
    if (threadIdx.x < 64) {
      const float s = ordered_sum8(park, static_cast<int>(vocab));
      if (threadIdx.x == 0) *scratch = s;
    }
    __syncthreads();
    return *scratch;
  }
  float* tail = lds + 8 * chain_pad;
  for (int64_t j = threadIdx.x; j < vocab; j += blockDim.x) {
    const float v = ((j == masked) ? -FLT_MAX : x[j]) - m;
    if (j < vsize)
      lds[(j & 7) * chain_pad + (j >> 3)] = exp_avx2(v);
    else
      tail[j - vsize] = expf_glibc(v);
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const float acc = ordered_chain_sum_lds(lds, chain_pad, vsize, vocab);
    if (threadIdx.x == 0) *scratch = acc;
  }
  __syncthreads();
  return *scratch;
}

__global__ void greedy_sample_kernel(const float* __restrict__ logits, int64_t vocab,
                                     int32_t disable_id, int32_t min_length, bool with_scores,
                                     float* __restrict__ exp_scratch, int chain_pad,
                                     SamplePrefix pf, StepAdvance adv,
                                     const int32_t* step_ptr, int32_t* __restrict__ next_ids,
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
  if (forced_by_prefix(pf, adv, row, rows, step, next_ids, hist_ids, hist_scores)) return;
  const int row_min = min_length + ((pf.ids && pf.extends_min_length) ? pf.lens[row] : 0);
  const int masked = (step < row_min) ? disable_id : -1;
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
    // LogSoftMax (src/cpu/kernels.cc:441-451): y = x + (-max - log(sum exp(x - max))), with the
    // reference's exp polynomial, sum order and libm logf
    extern __shared__ __attribute__((aligned(16))) float exp_lds[];
    const float s = ordered_row_exp_sum(x, vocab, masked, m, exp_scratch + row * vocab,
                                        chain_pad > 0 ? exp_lds : nullptr, chain_pad, scratch);
    score = m + (-m - logf_glibc(s));
  }
  if (threadIdx.x == 0) {
    next_ids[row] = best_i;
    hist_ids[static_cast<int64_t>(step) * rows + row] = best_i;
    hist_scores[static_cast<int64_t>(step) * rows + row] = score;
    row_finished(adv, rows, step);
  }
}

void launch_greedy_sample(const float* logits, int64_t rows, int64_t vocab, int32_t disable_id,
                          int32_t min_length, bool with_scores, const int32_t* step_ptr,
                          int32_t* next_ids, int32_t* hist_ids, float* hist_scores,
                          hipStream_t stream, const int32_t* prefix_ids, const int32_t* prefix_lens,
                          int32_t prefix_stride, int32_t prefix_first_step, bool prefix_extends_min,
                          int32_t* advance_step_ptr, int32_t* done_counter, float* exp_scratch) {
  if (rows == 0) return;
  int block = 64;
  while (block < vocab / 4 && block < 1024) block <<= 1;
  const SamplePrefix pf{prefix_ids, prefix_lens, prefix_stride, prefix_first_step,
                        prefix_extends_min ? 1 : 0};
  const StepAdvance adv{advance_step_ptr, done_counter};
  CT2X_CHECK(!with_scores || exp_scratch != nullptr, "greedy sampler: scores need a scratch row buffer");
  const ExpSumLayout el = with_scores ? exp_sum_layout(vocab) : ExpSumLayout{0, 0};
  const size_t lds = static_cast<size_t>(el.lds_floats) * sizeof(float);
  allow_big_lds(reinterpret_cast<const void*>(greedy_sample_kernel), lds);
  hipLaunchKernelGGL(greedy_sample_kernel, dim3(rows), dim3(block), lds, stream, logits, vocab,
                     disable_id, min_length, with_scores, exp_scratch, el.chain_pad, pf, adv, step_ptr,
                     next_ids, hist_ids, hist_scores);
}

// return_logits_vocab (src/decoding.cc:868-872, 905-908): the logits a greedy step samples from -
// after the logits processors and DisableTokens::apply (the end token below the minimum length is
// lowest(), src/decoding.cc:384-408) - kept per step in a ring [ring][rows][out_vocab] indexed by
// the device step counter. The samplers apply the same mask on the fly and never write it back.
__global__ void record_logits_kernel(const float* __restrict__ logits, int64_t vocab,
                                     int64_t out_vocab, int32_t disable_id, int32_t min_length,
                                     const int32_t* __restrict__ prefix_lens, int extends_min,
                                     const int32_t* __restrict__ step_ptr, int32_t ring,
                                     float* __restrict__ hist) {
  const int64_t row = blockIdx.x, rows = gridDim.x;
  const int step = *step_ptr;
  const int row_min = min_length + ((prefix_lens && extends_min) ? prefix_lens[row] : 0);
  const int masked = (step < row_min) ? disable_id : -1;
  const float* x = logits + row * vocab;
  float* y = hist + (static_cast<int64_t>(step % ring) * rows + row) * out_vocab;
  for (int64_t j = static_cast<int64_t>(blockIdx.y) * blockDim.x + threadIdx.x; j < out_vocab;
       j += static_cast<int64_t>(gridDim.y) * blockDim.x)
    y[j] = (j == masked) ? -FLT_MAX : x[j];
}

void launch_record_logits(const float* logits, int64_t rows, int64_t vocab, int64_t out_vocab,
                          int32_t disable_id, int32_t min_length, const int32_t* prefix_lens,
                          bool prefix_extends_min, const int32_t* step_ptr, int32_t ring, float* hist,
                          hipStream_t stream) {
  if (rows == 0) return;
  const unsigned chunks = static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>(64, out_vocab / 1024)));
  hipLaunchKernelGGL(record_logits_kernel, dim3(static_cast<unsigned>(rows), chunks), dim3(256), 0,
                     stream, logits, vocab, out_vocab, disable_id, min_length, prefix_lens,
                     prefix_extends_min ? 1 : 0, step_ptr, ring, hist);
}

// ---------------------------------------------------------------------------------------------
// RandomSampler (src/sampling.cc:41-101) for one sample per row, fused like the greedy sampler:
// min-length EOS masking -> restrict to the best `topk` candidates (ops::TopK order: value
// descending, ties by lowest index) -> divide by the temperature -> TopPMask (keep a candidate
// while the probability mass ranked before it is < topp, src/ops/topp_mask_cpu.cc:24-31) ->
// SoftMax -> Multinomial (inverse CDF on a uniform draw) -> id to the next step + history; the
// returned score is the log-probability of the id under the ORIGINAL distribution
// (ops::Gather(scores, sampled_ids)). The uniform draw is a counter-based hash of
// (seed, step, row): reproducible for a seed, independent of batch composition and of
// hipGraph replay (the step counter lives in device memory).
// ---------------------------------------------------------------------------------------------
constexpr int kMaxSampleTopK = 64;

__device__ __forceinline__ float uniform01(uint32_t seed, uint32_t step, uint32_t row) {
  uint64_t z = (static_cast<uint64_t>(seed) << 32) ^ (static_cast<uint64_t>(step) << 20) ^ row;
  z += 0x9E3779B97F4A7C15ull;  // splitmix64
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return static_cast<float>(z >> 40) * (1.0f / 16777216.0f);  // 24 bits -> [0, 1)
}

__global__ void random_sample_kernel(const float* __restrict__ logits, int64_t vocab,
                                     int32_t disable_id, int32_t min_length, bool with_scores,
                                     int topk, float topp, float temperature, uint32_t seed,
                                     SamplePrefix pf, StepAdvance adv,
                                     const int32_t* step_ptr, int32_t* __restrict__ next_ids,
                                     int32_t* __restrict__ hist_ids,
                                     float* __restrict__ hist_scores,
                                     const float* __restrict__ uniforms) {
  __shared__ float s_v[16];
  __shared__ int s_i[16];
  __shared__ float scratch[16];
  __shared__ float cand_v[kMaxSampleTopK];
  __shared__ int cand_i[kMaxSampleTopK];
  const int64_t row = blockIdx.x;
  const int64_t rows = gridDim.x;
  const int step = *step_ptr;
  if (forced_by_prefix(pf, adv, row, rows, step, next_ids, hist_ids, hist_scores)) return;
  const int row_min = min_length + ((pf.ids && pf.extends_min_length) ? pf.lens[row] : 0);
  const int masked = (step < row_min) ? disable_id : -1;
  const float* x = logits + row * vocab;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nwaves = (blockDim.x + 63) >> 6;

  // log-softmax constant of the original distribution (only needed for the returned score)
  float lse_c = 0.f;
  if (with_scores) {
    float m = -FLT_MAX;
    for (int64_t j = threadIdx.x; j < vocab; j += blockDim.x)
      m = fmaxf(m, (j == masked) ? -FLT_MAX : x[j]);
    m = block_max(m, scratch);
    float s = 0.f;
    for (int64_t j = threadIdx.x; j < vocab; j += blockDim.x)
      s += expf(((j == masked) ? -FLT_MAX : x[j]) - m);
    s = block_sum(s, scratch);
    lse_c = -m - logf(s);
  }

  // top-k by repeated arg-max: pass p selects the best element ranked after the previous pick
  float last_v = FLT_MAX;
  int last_i = -1;
  for (int pass = 0; pass < topk; ++pass) {
    ValIdx best;
    best.v = -FLT_MAX;
    best.i = INT_MAX;
    for (int64_t j = threadIdx.x; j < vocab; j += blockDim.x) {
      const float v = (j == masked) ? -FLT_MAX : x[j];
      const bool after = (v < last_v) || (v == last_v && static_cast<int>(j) > last_i);
      if (!after) continue;
      ValIdx c;
      c.v = v;
      c.i = static_cast<int>(j);
      best = better(best, c);
    }
    best = wave_argmax(best);
    if (lane == 0) {
      s_v[wave] = best.v;
      s_i[wave] = best.i;
    }
    __syncthreads();
    ValIdx r;
    r.v = s_v[0];
    r.i = s_i[0];
    for (int w = 1; w < nwaves; ++w) {
      ValIdx c;
      c.v = s_v[w];
      c.i = s_i[w];
      r = better(r, c);
    }
    if (threadIdx.x == 0) {
      cand_v[pass] = r.v;
      cand_i[pass] = r.i;
    }
    last_v = r.v;
    last_i = r.i;
    __syncthreads();
  }

  if (threadIdx.x == 0) {
    // candidates are sorted by value, descending: cand_v[0] is the maximum
    const float inv_t = 1.f / temperature;
    float probs[kMaxSampleTopK];
    float total = 0.f;
    for (int c = 0; c < topk; ++c) {
      probs[c] = (cand_i[c] == INT_MAX) ? 0.f : expf((cand_v[c] - cand_v[0]) * inv_t);
      total += probs[c];
    }
    int kept = topk;
    float kept_total = total;
    if (topp < 1.f) {  // TopPMask over the (temperature-scaled) softmax of the candidates
      float mass = 0.f;
      kept = 0;
      kept_total = 0.f;
      for (int c = 0; c < topk; ++c) {
        if (!(mass < topp)) break;
        mass += probs[c] / total;
        kept_total += probs[c];
        ++kept;
      }
    }
    // (tests inject the draw: uniforms[row] in [0, 1) replaces the hash)
    const float u01 = uniforms ? uniforms[row]
                               : uniform01(seed, static_cast<uint32_t>(step), static_cast<uint32_t>(row));
    const float u = u01 * kept_total;
    int pick = kept - 1;
    float acc = 0.f;
    for (int c = 0; c < kept; ++c) {
      acc += probs[c];
      if (u < acc) {
        pick = c;
        break;
      }
    }
    const int id = cand_i[pick];
    next_ids[row] = id;
    hist_ids[static_cast<int64_t>(step) * rows + row] = id;
    hist_scores[static_cast<int64_t>(step) * rows + row] = with_scores ? cand_v[pick] + lse_c : 0.f;
    row_finished(adv, rows, step);
  }
}

// RandomSampler over the WHOLE vocabulary (sampling_topk = 0 in the reference: no TopK
// restriction, src/sampling.cc:54-66) or over the best `topk` tokens for a topk too large for the
// candidate lists of random_sample_kernel (65 .. vocabulary - 1):
// [TopK] -> temperature -> [TopPMask] -> SoftMax -> Multinomial.
//  * TopK without a sort: the value of the topk-th largest token is found by bisection on the
//    ordered bit pattern (32 counting passes); tokens above it are in, and of the tokens that TIE
//    with it the ones with the lowest indices, as many as are missing (ops::TopK's order:
//    src/ops/topk_cpu.cc:13-55). TopPMask and the draw then see only those tokens;
//  * the multinomial draw is an inverse CDF in index order: every thread owns a contiguous
//    segment of the vocabulary, a block scan of the segment masses finds the owner of the
//    uniform draw, which then walks its segment;
//  * TopPMask (src/ops/topp_mask_cpu.cc:24-31: in value order, keep a token while the mass ranked
//    before it is < topp) needs no sort here: the kept set is {i : M(x_i) < topp} with
//    M(v) = mass of the tokens with a value > v, a threshold found by bisection on the ordered
//    bit pattern of the float (32 vocabulary passes). Tokens whose value TIES with the threshold
//    are all kept (the reference keeps those its sort happens to rank first).
__device__ __forceinline__ uint32_t ordered_key(float v) {
  const uint32_t b = __float_as_uint(v);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__global__ __launch_bounds__(1024) void random_sample_full_kernel(
    const float* __restrict__ logits, int64_t vocab, int32_t disable_id, int32_t min_length,
    bool with_scores, int topk, float topp, float temperature, uint32_t seed, SamplePrefix pf,
    StepAdvance adv, const int32_t* step_ptr, int32_t* __restrict__ next_ids,
    int32_t* __restrict__ hist_ids, float* __restrict__ hist_scores,
    const float* __restrict__ uniforms) {
  __shared__ float scratch[16];
  __shared__ float seg_incl[1024];
  __shared__ int s_pick;
  __shared__ int s_tie_limit;
  const int64_t row = blockIdx.x;
  const int64_t rows = gridDim.x;
  const int step = *step_ptr;
  if (forced_by_prefix(pf, adv, row, rows, step, next_ids, hist_ids, hist_scores)) return;
  const int row_min = min_length + ((pf.ids && pf.extends_min_length) ? pf.lens[row] : 0);
  const int masked = (step < row_min) ? disable_id : -1;
  const float* x = logits + row * vocab;
  const int tid = threadIdx.x, nthreads = blockDim.x;
  auto val = [&](int64_t j) { return (j == masked) ? -FLT_MAX : x[j]; };

  float m = -FLT_MAX;
  for (int64_t j = tid; j < vocab; j += nthreads) m = fmaxf(m, val(j));
  m = block_max(m, scratch);
  const float inv_t = 1.f / temperature;
  // weights of the sampling distribution: exp((x - max) / T); of the original one: exp(x - max)
  float lse_c = 0.f;
  if (with_scores) {
    float s = 0.f;
    for (int64_t j = tid; j < vocab; j += nthreads) s += expf(val(j) - m);
    s = block_sum(s, scratch);
    lse_c = -m - logf(s);
  }
  const int64_t seg = (vocab + nthreads - 1) / nthreads;
  const int64_t j0 = min(vocab, static_cast<int64_t>(tid) * seg), j1 = min(vocab, j0 + seg);
  // TopK: candidates are the tokens with a key above k_key, and those with key == k_key up to
  // index tie_limit (counts as floats: exact below 2^24 tokens)
  uint32_t k_key = 0;
  int64_t tie_limit = vocab;
  if (topk > 0 && topk < vocab) {
    uint32_t lo = 0, hi = ordered_key(m);
    while (lo < hi) {  // the smallest key with fewer than topk tokens above it
      const uint32_t mid = lo + ((hi - lo) >> 1);
      float above = 0.f;
      for (int64_t j = tid; j < vocab; j += nthreads) above += (ordered_key(val(j)) > mid) ? 1.f : 0.f;
      above = block_sum(above, scratch);
      if (above < static_cast<float>(topk))
        hi = mid;
      else
        lo = mid + 1;
    }
    k_key = lo;
    float above = 0.f;
    for (int64_t j = tid; j < vocab; j += nthreads) above += (ordered_key(val(j)) > k_key) ? 1.f : 0.f;
    above = block_sum(above, scratch);
    const int need = topk - static_cast<int>(above);  // >= 1 tied tokens complete the list
    float ties = 0.f;
    for (int64_t j = j0; j < j1; ++j) ties += (ordered_key(val(j)) == k_key) ? 1.f : 0.f;
    seg_incl[tid] = ties;
    if (tid == 0) s_tie_limit = static_cast<int>(vocab);
    __syncthreads();
    for (int off = 1; off < nthreads; off <<= 1) {
      const float add = (tid >= off) ? seg_incl[tid - off] : 0.f;
      __syncthreads();
      seg_incl[tid] += add;
      __syncthreads();
    }
    const int before = static_cast<int>(tid ? seg_incl[tid - 1] : 0.f);
    if (before < need && need <= static_cast<int>(seg_incl[tid])) {  // the need-th tie is in this segment
      int seen = before;
      for (int64_t j = j0; j < j1; ++j)
        if (ordered_key(val(j)) == k_key && ++seen == need) s_tie_limit = static_cast<int>(j);
    }
    __syncthreads();
    tie_limit = s_tie_limit;
    __syncthreads();
  }
  auto candidate = [&](int64_t j, uint32_t key) {
    return key > k_key || (key == k_key && j <= tie_limit);
  };
  uint32_t keep_key = 0;  // candidates with ordered_key >= keep_key take part
  if (topp < 1.f) {
    float total = 0.f;
    for (int64_t j = tid; j < vocab; j += nthreads) {
      const float v = val(j);
      if (candidate(j, ordered_key(v))) total += expf((v - m) * inv_t);
    }
    total = block_sum(total, scratch);
    uint32_t lo = 0, hi = ordered_key(m);
    while (lo < hi) {
      const uint32_t mid = lo + ((hi - lo) >> 1);
      float above = 0.f;
      for (int64_t j = tid; j < vocab; j += nthreads) {
        const float v = val(j);
        const uint32_t key = ordered_key(v);
        if (key > mid && candidate(j, key)) above += expf((v - m) * inv_t);
      }
      above = block_sum(above, scratch);
      if (above / total < topp)
        hi = mid;
      else
        lo = mid + 1;
    }
    keep_key = lo;
  }
  auto weight = [&](int64_t j) {
    const float v = val(j);
    const uint32_t key = ordered_key(v);
    return (key >= keep_key && candidate(j, key)) ? expf((v - m) * inv_t) : 0.f;
  };

  float mass = 0.f;
  for (int64_t j = j0; j < j1; ++j) mass += weight(j);
  seg_incl[tid] = mass;
  if (tid == 0) s_pick = -1;
  __syncthreads();
  for (int off = 1; off < nthreads; off <<= 1) {  // Hillis-Steele inclusive scan
    const float add = (tid >= off) ? seg_incl[tid - off] : 0.f;
    __syncthreads();
    seg_incl[tid] += add;
    __syncthreads();
  }
  const float total = seg_incl[nthreads - 1];
  float u = (uniforms ? uniforms[row]
                      : uniform01(seed, static_cast<uint32_t>(step), static_cast<uint32_t>(row))) * total;
  u = fminf(u, total * (1.f - 1.f / 16777216.f));  // strictly below the total: exactly one owner
  const float excl = tid ? seg_incl[tid - 1] : 0.f;
  if (excl <= u && u < seg_incl[tid]) {
    float acc = excl;
    int pick = -1, last = -1;
    for (int64_t j = j0; j < j1 && pick < 0; ++j) {
      const float w = weight(j);
      if (w > 0.f) last = static_cast<int>(j);
      acc += w;
      if (u < acc) pick = static_cast<int>(j);
    }
    s_pick = pick >= 0 ? pick : last;  // the scan and the walk associate the sums differently
  }
  __syncthreads();
  if (tid == 0) {
    const int id = s_pick;
    next_ids[row] = id;
    hist_ids[static_cast<int64_t>(step) * rows + row] = id;
    hist_scores[static_cast<int64_t>(step) * rows + row] = with_scores ? x[id] + lse_c : 0.f;
    row_finished(adv, rows, step);
  }
}

// ---------------------------------------------------------------------------------------------
// ops::TopPMask standalone (src/ops/topp_mask_cpu.cc:12-39, src/ops/topp_mask_gpu.cu): in order of
// decreasing probability, an element keeps its input value while the mass ranked before it is
// < p, else it becomes `mask`. No sort: kept = { i : mass of the elements with a LARGER
// probability < p }, the threshold found by bisection on the ordered bit pattern of the
// probability (32 passes over the row; elements that tie with the threshold are all kept - the
// reference keeps those its unstable sort happens to rank first). One workgroup per row.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void topp_mask_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ probs, float p,
                                                        float mask, int64_t depth,
                                                        float* __restrict__ y) {
  __shared__ float scratch[16];
  const int64_t row = blockIdx.x;
  const float* pr = probs + row * depth;
  float top = -FLT_MAX;
  for (int64_t j = threadIdx.x; j < depth; j += blockDim.x) top = fmaxf(top, pr[j]);
  top = block_max(top, scratch);
  uint32_t lo = 0, hi = ordered_key(top);
  while (lo < hi) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    float above = 0.f;
    for (int64_t j = threadIdx.x; j < depth; j += blockDim.x)
      if (ordered_key(pr[j]) > mid) above += pr[j];
    above = block_sum(above, scratch);
    if (above < p)
      hi = mid;
    else
      lo = mid + 1;
  }
  for (int64_t j = threadIdx.x; j < depth; j += blockDim.x)
    y[row * depth + j] = (ordered_key(pr[j]) >= lo) ? x[row * depth + j] : mask;
}

void launch_topp_mask(const float* x, const float* probs, int64_t rows, int64_t depth, float p,
                      float mask, float* y, hipStream_t stream) {
  if (rows == 0 || depth == 0) return;
  hipLaunchKernelGGL(topp_mask_kernel, dim3(rows), dim3(256), 0, stream, x, probs, p, mask, depth, y);
}

// ---------------------------------------------------------------------------------------------
// ops::Multinomial standalone (src/ops/multinomial_cpu.cc / _gpu.cu): `samples` independent draws
// per row of a (not necessarily normalised) weight row, by inverse CDF in index order; draw s of
// row r of call c uses the counter-based hash of (seed, c, r * samples + s). One thread per draw.
// ---------------------------------------------------------------------------------------------
__global__ void multinomial_kernel(const float* __restrict__ weights, int64_t rows, int64_t classes,
                                   int64_t samples, uint32_t seed, uint32_t call,
                                   int32_t* __restrict__ out) {
  const int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (e >= rows * samples) return;
  const int64_t row = e / samples;
  const float* w = weights + row * classes;
  float total = 0.f;
  for (int64_t j = 0; j < classes; ++j) total += w[j];
  const float u = uniform01(seed, call, static_cast<uint32_t>(e)) * total;
  float acc = 0.f;
  int32_t pick = -1, last = 0;
  for (int64_t j = 0; j < classes && pick < 0; ++j) {
    if (w[j] > 0.f) last = static_cast<int32_t>(j);
    acc += w[j];
    if (u < acc) pick = static_cast<int32_t>(j);
  }
  out[e] = pick >= 0 ? pick : last;
}

// ops::GumbelMax::add_gumbel_noise (src/ops/gumbel_max_gpu.cu:12-38; CPU: gumbel_max_cpu.cc:12-26):
// y = x - log(u), u uniform in (0, 1] - the noise the reference adds before its TopK (an
// exponential variate; the op keeps the reference's name). One draw per element from the
// counter-based hash on (seed, call number, element): reproducible for a seed, fresh per call.
__global__ void gumbel_noise_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n,
                                    uint32_t seed, uint32_t call) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  const uint32_t lo = static_cast<uint32_t>(i), hi = static_cast<uint32_t>(i >> 32);
  const float u = 1.f - uniform01(seed ^ (hi * 0x9E3779B9u), call, lo);  // (0, 1]
  y[i] = x[i] + (-logf(u));
}

void launch_gumbel_noise(const float* x, float* y, int64_t n, uint32_t seed, uint32_t call,
                         hipStream_t stream) {
  if (n == 0) return;
  hipLaunchKernelGGL(gumbel_noise_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0,
                     stream, x, y, n, seed, call);
}

void launch_multinomial(const float* weights, int64_t rows, int64_t classes, int64_t samples,
                        uint32_t seed, uint32_t call, int32_t* out, hipStream_t stream) {
  const int64_t n = rows * samples;
  if (n == 0) return;
  hipLaunchKernelGGL(multinomial_kernel, dim3((n + 127) / 128), dim3(128), 0, stream, weights, rows,
                     classes, samples, seed, call, out);
}

void launch_random_sample(const float* logits, int64_t rows, int64_t vocab, int32_t disable_id,
                          int32_t min_length, bool with_scores, int topk, float topp,
                          float temperature, uint32_t seed, const int32_t* step_ptr,
                          int32_t* next_ids, int32_t* hist_ids, float* hist_scores,
                          hipStream_t stream, const int32_t* prefix_ids, const int32_t* prefix_lens,
                          int32_t prefix_stride, int32_t prefix_first_step, bool prefix_extends_min,
                          int32_t* advance_step_ptr, int32_t* done_counter, const float* uniforms) {
  if (rows == 0) return;
  int block = 64;
  while (block < vocab / 4 && block < 1024) block <<= 1;
  const SamplePrefix pf{prefix_ids, prefix_lens, prefix_stride, prefix_first_step,
                        prefix_extends_min ? 1 : 0};
  const StepAdvance adv{advance_step_ptr, done_counter};
  if (topk <= 0 || topk > kMaxSampleTopK) {  // the whole vocabulary, or a long candidate list
    hipLaunchKernelGGL(random_sample_full_kernel, dim3(rows), dim3(1024), 0, stream, logits, vocab,
                       disable_id, min_length, with_scores, (topk <= 0 || topk >= vocab) ? 0 : topk, topp,
                       temperature, seed, pf, adv,
                       step_ptr, next_ids, hist_ids, hist_scores, uniforms);
    return;
  }
  hipLaunchKernelGGL(random_sample_kernel, dim3(rows), dim3(block), 0, stream, logits, vocab,
                     disable_id, min_length, with_scores, topk, topp, temperature, seed, pf, adv,
                     step_ptr, next_ids, hist_ids, hist_scores, uniforms);
}

int max_sampling_topk() { return kMaxSampleTopK; }  // longest candidate LIST; larger k: threshold form

// ---------------------------------------------------------------------------------------------
// Logits processors (src/decoding_utils.cc): RepetitionPenalty -> NoRepeatNgram -> SuppressTokens /
// SuppressSequences, in the order make_logits_processors builds them (src/decoding.cc:1091-1120),
// applied in place on the step's logits BEFORE log-softmax / sampling (src/decoding.cc:863-866).
// One workgroup per row; `seq` is the row's generated history (prefix tokens included, start
// token excluded: alive_seq, src/decoding.cc:886-892) addressed as seq[t * t_stride + row *
// row_stride] so that both the device-resident greedy history ([step, rows]) and an uploaded
// beam history ([rows, len]) fit.
//  * RepetitionPenalty (:43-66 + primitives::penalize_previous_tokens, src/cpu/primitives.cc:
//    411-428): the previous tokens' scores are GATHERED first and then scattered back as
//    s < 0 ? s * p : s / p, so a token that occurs twice is penalised once - hence the two
//    phases around a barrier.
//  * NoRepeatNgram (:74-105): every earlier occurrence of the last n-1 tokens bans the token
//    that followed it.
//  * SuppressSequences (:108-153): length-1 sequences are always banned (this is also how
//    disable_unk / disable_ids arrive); a longer one bans its last token when the history ends
//    with the rest.
// Banned = numeric_limits<float>::lowest() (DisableTokens, decoding_utils.h:36-71).
struct LogitsProcessParams {
  float* logits;
  int64_t rows, vocab;
  const int32_t* seq;
  int64_t t_stride, row_stride;
  const int32_t* step_ptr;  // history length = *step_ptr - first_step (or `length` when null)
  int32_t first_step;
  int32_t length;
  float penalty;            // 1 = off
  int32_t ngram;            // 0 = off
  const int32_t* sup_ids;   // flattened suppressed sequences
  const int32_t* sup_off;   // [n_sup + 1]
  int32_t n_sup;
};

__global__ __launch_bounds__(256) void process_logits_kernel(LogitsProcessParams p) {
  extern __shared__ float s_prev[];  // [length] gathered scores
  const int64_t row = blockIdx.x;
  const int tid = threadIdx.x;
  const int len = p.step_ptr ? (*p.step_ptr - p.first_step) : p.length;
  float* lg = p.logits + row * p.vocab;
  const int32_t* sq = p.seq + row * p.row_stride;
  auto tok = [&](int t) { return sq[static_cast<int64_t>(t) * p.t_stride]; };

  if (p.penalty != 1.f && len > 0) {
    for (int j = tid; j < len; j += blockDim.x) s_prev[j] = lg[tok(j)];
    __syncthreads();
    for (int j = tid; j < len; j += blockDim.x) {
      const float s = s_prev[j];
      lg[tok(j)] = s < 0.f ? s * p.penalty : s / p.penalty;
    }
    __syncthreads();  // bans below must land after the penalties
  }
  if (p.ngram > 0 && len >= p.ngram) {
    const int cmp = p.ngram - 1;
    for (int i = tid; i + p.ngram <= len; i += blockDim.x) {
      bool same = true;
      for (int c = 0; c < cmp && same; ++c) same = tok(i + c) == tok(len - cmp + c);
      if (same) lg[tok(i + cmp)] = -FLT_MAX;
    }
  }
  for (int s = tid; s < p.n_sup; s += blockDim.x) {
    const int32_t* ids = p.sup_ids + p.sup_off[s];
    const int cmp = p.sup_off[s + 1] - p.sup_off[s] - 1;
    bool ban = len >= cmp;
    for (int c = 0; c < cmp && ban; ++c) ban = tok(len - cmp + c) == ids[c];
    if (ban) lg[ids[cmp]] = -FLT_MAX;
  }
}

void launch_process_logits(float* logits, int64_t rows, int64_t vocab, const int32_t* seq,
                           int64_t t_stride, int64_t row_stride, const int32_t* step_ptr,
                           int32_t first_step, int32_t length, int32_t max_length, float penalty,
                           int32_t ngram, const int32_t* sup_ids, const int32_t* sup_off,
                           int32_t n_sup, hipStream_t stream) {
  if (rows == 0) return;
  const LogitsProcessParams p{logits, rows, vocab, seq, t_stride, row_stride, step_ptr, first_step,
                              length, penalty, ngram, sup_ids, sup_off, n_sup};
  const size_t lds = penalty != 1.f ? static_cast<size_t>(max_length) * sizeof(float) : 0;
  hipLaunchKernelGGL(process_logits_kernel, dim3(rows), dim3(256), lds, stream, p);
}

__global__ void advance_step_kernel(int32_t* step_ptr) { *step_ptr = *step_ptr + 1; }

void launch_advance_step(int32_t* step_ptr, hipStream_t stream) {
  hipLaunchKernelGGL(advance_step_kernel, dim3(1), dim3(1), 0, stream, step_ptr);
}

// Biased decoding (BiasedDecoder::decode, src/decoding.cc:263-323): while a beam has followed the
// target prefix so far, its next-token distribution is (1 - beta) * softmax(logits) + beta * onehot
// (prefix token of this step); a beam that left the prefix (`diverged`, kept by beam_select_kernel)
// and every step past the prefix use the plain LogSoftMax.
struct BeamBias {
  const int32_t* ids;       // [batch, stride] target prefixes, nullptr = no biased decoding
  const int32_t* lens;      // [batch]
  int32_t stride;
  int32_t first_step;
  float beta;
  const uint8_t* diverged;  // [batch * beam]
};

constexpr int kBeamCandCap = 256;  // candidates of a row's top k the list in LDS holds (beam_logprobs_kernel)
// maximum of a wave, any lane count, every lane gets it
__device__ __forceinline__ float wave_max_any(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// f(j, x[j]) for every element of a row, each thread taking 8-byte pieces sixteen at a time (a
// 1 MB row scanned by ONE workgroup is bound by what it keeps in flight: 4-byte loads a few deep
// read 2.4 TB/s over 256 rows). Rows of the 256 206-word vocabulary start 8 bytes off a 16-byte
// boundary every other row, so 8-byte pieces are what every row allows.
template <typename F>
__device__ __forceinline__ void row_scan(const float* __restrict__ x, int64_t vocab, F&& f) {
  int64_t done = 0;
  if ((reinterpret_cast<uintptr_t>(x) & 7) == 0) {
    const int64_t n2 = vocab / 2;
    const float2* x2 = reinterpret_cast<const float2*>(x);
    constexpr int U = 16;
    for (int64_t i = threadIdx.x; i < n2; i += static_cast<int64_t>(blockDim.x) * U) {
      float2 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t idx = i + static_cast<int64_t>(u) * blockDim.x;
        v[u] = idx < n2 ? x2[idx] : make_float2(0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t idx = i + static_cast<int64_t>(u) * blockDim.x;
        if (idx < n2) {
          f(2 * idx, v[u].x);
          f(2 * idx + 1, v[u].y);
        }
      }
    }
    done = 2 * n2;
  }
  for (int64_t j = done + threadIdx.x; j < vocab; j += blockDim.x) f(j, x[j]);
}

// CAND: the candidate form (part_v != nullptr) as an instantiation of its own - its row registers
// die after the exps (round 6, last change: with both forms in one kernel the 32 row registers
// lived across the 4000-step chain, 47 of them in scratch memory, and the candidate listing that
// followed reloaded them one dependent scratch round trip at a time: 11 us of a 54 us launch).
template <bool CAND>
__global__ void beam_logprobs_kernel(const float* __restrict__ logits,
                                     const float* __restrict__ beam_scores, int64_t vocab,
                                     int32_t disable_id, int32_t min_length,
                                     const int32_t* __restrict__ step_ptr,
                                     const int32_t* __restrict__ extra_min, int32_t beam,
                                     int chain_pad, float* __restrict__ y, int flags, BeamBias bias,
                                     float* __restrict__ part_v, int32_t* __restrict__ part_i, int topk) {
  extern __shared__ __attribute__((aligned(16))) float exp_lds[];
  __shared__ float scratch[16];
  __shared__ int scratch_i[16];
  __shared__ float cand_v[kBeamCandCap];
  __shared__ int cand_i[kBeamCandCap];
  __shared__ int cand_count;
  __shared__ float wave_max_s[16];
  const int64_t row = blockIdx.x;
  const int step = *step_ptr;
  // apply_min_length with a hidden prefix (src/decoding.cc:384-408): EOS stays disabled
  // prefix_length steps longer for that batch entry. The reference adds the entry's end token with
  // disable_tokens.add(i, end_id), i the index of the live BATCH ENTRY - a row index of the
  // [entries * beam, vocab] logits once the beams are expanded (every step after the first on
  // Device::CPU, src/decoding.cc:494): row i < entries is masked by entry i's threshold, the rows
  // from `entries` on are not masked at all. Reproduced as it is (the goldens of a hidden prefix
  // under beam search with min_length > 0 need it); flags & 2 = every beam of the entry by the
  // entry's own threshold (CT2X_HIDDEN_PREFIX_MIN_LENGTH_PER_ENTRY=1).
  int row_min = min_length;
  if (extra_min != nullptr) {
    const int64_t entries = gridDim.x / beam;
    // (min_length == 0: the rule only ever masks rows inside their own forced prefix steps when the
    // entries are in the reference's order, longest first - where scores are reset anyway,
    // update_sample_with_prefix; the per-entry form is the same thing for any order)
    if ((flags & 2) != 0 || step <= bias.first_step || min_length <= bias.first_step)
      row_min = min_length + extra_min[row / beam];
    else
      row_min = row < entries ? min_length + extra_min[row] : INT_MIN;
  }
  const int masked = (step < row_min) ? disable_id : -1;
  const int fast = flags & 1;
  const float* x = logits + row * vocab;
  float* yr = y + row * vocab;
  if (bias.ids != nullptr) {
    const int pstep = step - bias.first_step;
    const int64_t b = row / beam;
    if (pstep >= 0 && pstep < bias.lens[b] && !bias.diverged[row]) {
      // SoftMax -> * (1 - beta) -> [prefix token] += beta -> Log (the reference's op sequence; its
      // Log is MKL's vsLn, here logf: the scores agree to float rounding, not bit for bit)
      const int32_t biased = bias.ids[b * bias.stride + pstep];
      float m = -FLT_MAX;
      for (int64_t j = threadIdx.x; j < vocab; j += blockDim.x)
        m = fmaxf(m, (j == masked) ? -FLT_MAX : x[j]);
      m = block_max(m, scratch);
      const float s = ordered_row_exp_sum(x, vocab, masked, m, yr, chain_pad > 0 ? exp_lds : nullptr,
                                          chain_pad, scratch);
      const float bs = beam_scores ? beam_scores[row] : 0.f;
      const float discount = 1.f - bias.beta;
      for (int64_t j = threadIdx.x; j < vocab; j += blockDim.x) {
        const float xv = (j == masked) ? -FLT_MAX : x[j];
        float pr = (expf(xv - m) / s) * discount;
        if (j == biased) pr = pr + bias.beta;
        const float v = logf(pr);
        yr[j] = beam_scores ? bs + v : v;
      }
      return;
    }
  }
  constexpr int XR = 32;  // row elements a thread can keep in registers
  if (!fast && chain_pad > 0 && vocab <= static_cast<int64_t>(XR) * blockDim.x) {
    // The row is read ONCE: each thread keeps its <= 32 elements in registers for the maximum,
    // the exps (parked transposed in LDS for the ordered sum) and the final x + c.
    float xv[XR];
#pragma unroll
    for (int u = 0; u < XR; ++u) {
      const int64_t j = threadIdx.x + static_cast<int64_t>(u) * blockDim.x;
      xv[u] = (j < vocab && j != masked) ? x[j] : -FLT_MAX;
    }
    float m = -FLT_MAX;
#pragma unroll
    for (int u = 0; u < XR; ++u) m = fmaxf(m, xv[u]);
    const float thread_max_x = m;
    m = block_max(m, scratch);
    const float b = beam_scores ? beam_scores[row] : 0.f;
    bool listed = false;
    if (CAND) {
      // The candidates are listed HERE, by x, while the row is in registers anyway: y = b + (x + c)
      // is monotone in x, so the row's top k by y are among the elements at or above the topk-th
      // largest of the waves' maxima (topk elements of the row sit at or above it, one per wave) -
      // lowered by a margin that covers every x the two roundings of y can collapse onto the
      // bound's y (c = -m - log(sum), sum in [1, vocab], is not known yet: |c| <= |m| + 17). The
      // list keeps x; wave 0 forms y once c is known and picks the exact order (value descending,
      // flat index ascending) - what topk_partial / topk_merge would return over the written row.
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
      const int nwaves = (blockDim.x + 63) >> 6;
      const float wm_own = wave_max_any(thread_max_x);
      if (lane == 0) wave_max_s[wave] = wm_own;
      if (threadIdx.x == 0) cand_count = 0;
      __syncthreads();
      if (nwaves >= topk) {
        // rank of each wave's maximum among the waves' maxima (ties by wave index): no rounds
        const float mine = lane < nwaves ? wave_max_s[lane] : -__builtin_huge_valf();
        int rank = 0;
        for (int q = 0; q < nwaves; ++q) {
          const float o = wave_max_s[q];
          rank += (o > mine || (o == mine && q < lane)) ? 1 : 0;
        }
        const unsigned long long holder = __builtin_amdgcn_ballot_w64(lane < nwaves && rank == topk - 1);
        const float bound_x = __uint_as_float(static_cast<unsigned>(
            __builtin_amdgcn_readlane(static_cast<int>(__float_as_uint(mine)), __builtin_ctzll(holder))));
        if (bound_x > -FLT_MAX) {
          const float mag = fabsf(b) + fabsf(bound_x) + fabsf(m) + 17.f;
          const float thr = bound_x - mag * 9.5367431640625e-07f;  // 8 ulps at that magnitude (2^-20)
          const int flat0 = static_cast<int>((row % beam) * vocab) + static_cast<int>(threadIdx.x);
#pragma unroll
          for (int u = 0; u < XR; ++u) {
            const int64_t j = threadIdx.x + static_cast<int64_t>(u) * blockDim.x;
            if (j < vocab && xv[u] >= thr) {
              const int pos = atomicAdd(&cand_count, 1);
              if (pos < kBeamCandCap) {
                cand_v[pos] = xv[u];
                cand_i[pos] = flat0 + u * static_cast<int>(blockDim.x);
              }
            }
          }
          listed = true;
        }
      }
    }
    const int64_t vsize = vocab > 8 ? vocab - vocab % 8 : 0;
    float* tail = exp_lds + 8 * chain_pad;
#pragma unroll
    for (int u = 0; u < XR; ++u) {
      const int64_t j = threadIdx.x + static_cast<int64_t>(u) * blockDim.x;
      if (j < vocab) {
        const float v = xv[u] - m;
        if (j < vsize)
          exp_lds[(j & 7) * chain_pad + (j >> 3)] = exp_avx2(v);
        else
          tail[j - vsize] = expf_glibc(v);
      }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
      const float acc = ordered_chain_sum_lds(exp_lds, chain_pad, vsize, vocab);
      if (threadIdx.x == 0) scratch[0] = acc;
    }
    __syncthreads();
    const float c = -m - logf_glibc(scratch[0]);
    if (CAND) {
      const float ninf = -__builtin_huge_valf();
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
      const int nwaves = (blockDim.x + 63) >> 6;
      const int n = cand_count;  // (written before the exps: two barriers ago)
      auto score_of = [&](float xj) {
        const float v = xj + c;
        return beam_scores ? b + v : v;
      };
      if (listed && n <= kBeamCandCap) {
        if (wave == 0) {
          constexpr int PER = kBeamCandCap / 64;
          float lv[PER];
          int li[PER];
#pragma unroll
          for (int e = 0; e < PER; ++e) {
            const int at = lane + e * 64;
            li[e] = at < n ? cand_i[at] : INT_MAX;
            lv[e] = at < n ? score_of(cand_v[at]) : ninf;
          }
          for (int r = 0; r < topk; ++r) {
            ValIdx best;
            best.v = ninf;
            best.i = INT_MAX;
#pragma unroll
            for (int e = 0; e < PER; ++e)
              if (li[e] != INT_MAX && (best.i == INT_MAX || lv[e] > best.v || (lv[e] == best.v && li[e] < best.i))) {
                best.v = lv[e];
                best.i = li[e];
              }
            if (best.i == INT_MAX) best.v = ninf;
            const ValIdx w = wave_argmax_dpp(best);
#pragma unroll
            for (int e = 0; e < PER; ++e)
              if (li[e] == w.i) {  // retired (flat indices are unique; INT_MAX stays empty)
                li[e] = INT_MAX;
                lv[e] = ninf;
              }
            if (lane == 0) {
              part_v[row * topk + r] = w.v;
              part_i[row * topk + r] = w.i;
            }
          }
        }
        return;
      }
      // (no bound - fewer waves than candidates - or more elements near it than the list holds:
      // topk rounds of a block-wide arg-max over the row, re-read from memory; the winners so far
      // sit in cand_i[0 .. r))
      const int flat_row = static_cast<int>((row % beam) * vocab);
      __syncthreads();
      for (int r = 0; r < topk; ++r) {
        ValIdx best;
        best.v = ninf;
        best.i = INT_MAX;
        for (int64_t j = threadIdx.x; j < vocab; j += blockDim.x) {  // (ascending index: strict compare)
          const int fi = flat_row + static_cast<int>(j);
          bool taken = false;
          for (int q = 0; q < r; ++q) taken = taken || (cand_i[q] == fi);
          if (taken) continue;
          const float y = score_of((j == masked) ? -FLT_MAX : x[j]);
          if (best.i == INT_MAX || y > best.v) {
            best.v = y;
            best.i = fi;
          }
        }
        if (best.i == INT_MAX) best.v = ninf;
        best = wave_argmax_dpp(best);
        __syncthreads();
        if (lane == 0) {
          scratch[wave] = best.v;
          scratch_i[wave] = best.i;
        }
        __syncthreads();
        ValIdx w;
        w.v = scratch[0];
        w.i = scratch_i[0];
        for (int q = 1; q < nwaves; ++q) {
          const float v = scratch[q];
          const int i = scratch_i[q];
          if (i != INT_MAX && (w.i == INT_MAX || v > w.v || (v == w.v && i < w.i))) {
            w.v = v;
            w.i = i;
          }
        }
        if (threadIdx.x == 0) {
          cand_i[r] = w.i;
          part_v[row * topk + r] = w.v;
          part_i[row * topk + r] = w.i;
        }
        __syncthreads();
      }
      return;
    }
#pragma unroll
    for (int u = 0; u < XR; ++u) {
      const int64_t j = threadIdx.x + static_cast<int64_t>(u) * blockDim.x;
      if (j < vocab) {
        const float v = xv[u] + c;
        yr[j] = beam_scores ? b + v : v;
      }
    }
    return;
  }
  float m = -FLT_MAX;
  if (fast) {
    row_scan(x, vocab, [&](int64_t j, float xj) { m = fmaxf(m, (j == masked) ? -FLT_MAX : xj); });
  } else {
    for (int64_t j = threadIdx.x; j < vocab; j += blockDim.x)
      m = fmaxf(m, (j == masked) ? -FLT_MAX : x[j]);
  }
  const float thread_max = m;  // (the fused candidates' bound, below)
  m = block_max(m, scratch);
  if (fast) {
    // 16-bit compute types (no CPU counterpart to reproduce bit for bit): a parallel sum instead of
    // the reference's eight dependent chains - 32 000 sequential adds per row of a 256 206-word
    // vocabulary cost 4.8 ms per beam step of the NLLB shape (profiles/r03_nllb_kernel_stats_b.csv)
    float part = 0.f;
    row_scan(x, vocab, [&](int64_t j, float xj) {
      if (j != masked) part += __expf(xj - m);
    });
    part = wave_sum(part);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = part;
    __syncthreads();
    float tot = 0.f;
    for (int w = 0; w < static_cast<int>((blockDim.x + 63) >> 6); ++w) tot += scratch[w];
    const float c = -m - logf(tot);
    const float b = beam_scores ? beam_scores[row] : 0.f;
    if (CAND) {
      // Round 6 (second half): the candidates of the register form above for rows of ANY length
      // (the 256 206-word vocabulary of the NLLB shape): the row is read a third time instead of
      // being written as 1 MB of log-probabilities that TopK's partial pass reads back (441 + 156 +
      // 12 us per step there). Same values y = b + (x + c) and the same total order as topk_partial /
      // topk_merge (value descending, flat index ascending). The bound: the topk-th largest of the
      // waves' maxima has topk elements of the row at or above it; it is applied to y, which is
      // monotone in x, so elements whose y collapses onto the bound's by rounding are listed too.
      const float ninf = -__builtin_huge_valf();
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
      const int nwaves = (blockDim.x + 63) >> 6;
      const int flat_row = static_cast<int>((row % beam) * vocab);
      auto score_of = [&](int64_t j, float xj) {
        const float v = ((j == masked) ? -FLT_MAX : xj) + c;
        return beam_scores ? b + v : v;
      };
      auto score = [&](int64_t j) { return score_of(j, x[j]); };
      const float wmax = wave_max_any(thread_max);
      __syncthreads();  // (scratch held the waves' sums)
      if (lane == 0) scratch[wave] = wmax;
      if (threadIdx.x == 0) cand_count = 0;
      __syncthreads();
      float bound_x = ninf;
      {
        float wm = lane < nwaves ? scratch[lane] : ninf;
        for (int r = 0; r < topk; ++r) {
          bound_x = wave_max_any(wm);
          const unsigned long long holders = __builtin_amdgcn_ballot_w64(wm == bound_x);
          if (holders != 0 && lane == __builtin_ctzll(holders)) wm = ninf;
        }
      }
      const bool bounded = nwaves >= topk && bound_x > -FLT_MAX;
      if (bounded) {
        const float vb = bound_x + c;
        const float y_bound = beam_scores ? b + vb : vb;
        row_scan(x, vocab, [&](int64_t j, float xj) {
          const float y = score_of(j, xj);
          if (y >= y_bound) {
            const int pos = atomicAdd(&cand_count, 1);
            if (pos < kBeamCandCap) {
              cand_v[pos] = y;
              cand_i[pos] = flat_row + static_cast<int>(j);
            }
          }
        });
      }
      __syncthreads();
      const int n = cand_count;
      if (bounded && n <= kBeamCandCap) {
        if (wave == 0) {
          constexpr int PER = kBeamCandCap / 64;
          float lv[PER];
          int li[PER];
#pragma unroll
          for (int e = 0; e < PER; ++e) {
            const int at = lane + e * 64;
            li[e] = at < n ? cand_i[at] : INT_MAX;
            lv[e] = at < n ? cand_v[at] : ninf;
          }
          for (int r = 0; r < topk; ++r) {
            ValIdx best;
            best.v = ninf;
            best.i = INT_MAX;
#pragma unroll
            for (int e = 0; e < PER; ++e)
              if (li[e] != INT_MAX && (best.i == INT_MAX || lv[e] > best.v || (lv[e] == best.v && li[e] < best.i))) {
                best.v = lv[e];
                best.i = li[e];
              }
            if (best.i == INT_MAX) best.v = ninf;
            const ValIdx w = wave_argmax_dpp(best);
#pragma unroll
            for (int e = 0; e < PER; ++e)
              if (li[e] == w.i) {
                li[e] = INT_MAX;
                lv[e] = ninf;
              }
            if (lane == 0) {
              part_v[row * topk + r] = w.v;
              part_i[row * topk + r] = w.i;
            }
          }
        }
        return;
      }
      // (no bound - fewer waves than candidates - or more ties at it than the list holds: topk
      // rounds of a block-wide arg-max over the row; the winners so far sit in cand_i[0 .. r))
      for (int r = 0; r < topk; ++r) {
        ValIdx best;
        best.v = ninf;
        best.i = INT_MAX;
        for (int64_t j = threadIdx.x; j < vocab; j += blockDim.x) {  // (ascending index: strict compare)
          const int fi = flat_row + static_cast<int>(j);
          bool taken = false;
          for (int q = 0; q < r; ++q) taken = taken || (cand_i[q] == fi);
          if (taken) continue;
          const float y = score(j);
          if (best.i == INT_MAX || y > best.v) {
            best.v = y;
            best.i = fi;
          }
        }
        if (best.i == INT_MAX) best.v = ninf;
        best = wave_argmax_dpp(best);
        __syncthreads();
        if (lane == 0) {
          scratch[wave] = best.v;
          scratch_i[wave] = best.i;
        }
        __syncthreads();
        ValIdx w;
        w.v = scratch[0];
        w.i = scratch_i[0];
        for (int q = 1; q < nwaves; ++q) {
          const float v = scratch[q];
          const int i = scratch_i[q];
          if (i != INT_MAX && (w.i == INT_MAX || v > w.v || (v == w.v && i < w.i))) {
            w.v = v;
            w.i = i;
          }
        }
        if (threadIdx.x == 0) {
          cand_i[r] = w.i;
          part_v[row * topk + r] = w.v;
          part_i[row * topk + r] = w.i;
        }
        __syncthreads();
      }
      return;
    }
    row_scan(x, vocab, [&](int64_t j, float xj) {
      const float v = ((j == masked) ? -FLT_MAX : xj) + c;
      yr[j] = beam_scores ? b + v : v;
    });
    return;
  }
  // the reference's LogSoftMax bit for bit: exp polynomial, sum order, libm logf (the output row
  // parks the exps until the sum is known)
  const float s = ordered_row_exp_sum(x, vocab, masked, m, yr, chain_pad > 0 ? exp_lds : nullptr,
                                      chain_pad, scratch);
  const float c = -m - logf_glibc(s);
  const float b = beam_scores ? beam_scores[row] : 0.f;
  for (int64_t j = threadIdx.x; j < vocab; j += blockDim.x) {
    const float v = ((j == masked) ? -FLT_MAX : x[j]) + c;
    // add_depth_broadcast of the cumulative beam score (src/decoding.cc:548-554)
    yr[j] = beam_scores ? b + v : v;
  }
}

void launch_beam_logprobs(const float* logits, const float* beam_scores, int64_t rows,
                          int64_t vocab, int32_t disable_id, int32_t min_length,
                          const int32_t* step_ptr, float* y, hipStream_t stream,
                          const int32_t* extra_min_per_batch, int32_t beam, bool exact,
                          const int32_t* bias_ids, const int32_t* bias_lens, int32_t bias_stride,
                          int32_t bias_first_step, float bias_beta, const uint8_t* diverged,
                          float* part_v, int32_t* part_i, int32_t topk) {
  if (rows == 0) return;
  int block = 64;
  while (block < vocab / 4 && block < 1024) block <<= 1;
  CT2X_CHECK(part_v == nullptr || (beam_logprobs_topk_applicable(vocab, exact, beam > 0 ? beam : 1, topk) &&
                                   !(bias_beta > 0.f)),
             "beam_logprobs: the fused candidates need the exact one-read form or the parallel-sum form");
  const BeamBias bias{bias_beta > 0.f ? bias_ids : nullptr, bias_lens, bias_stride, bias_first_step,
                      bias_beta, diverged};
  const ExpSumLayout el = exp_sum_layout(vocab);
  const size_t lds = static_cast<size_t>(el.lds_floats) * sizeof(float);
  // (kernel flags: 1 = the parallel-sum LogSoftMax of the 16-bit compute types; 2 = a hidden
  // prefix's min_length per entry instead of the reference's row rule, see the kernel)
  static const bool per_entry_min = [] {
    const char* e = std::getenv("CT2X_HIDDEN_PREFIX_MIN_LENGTH_PER_ENTRY");
    return e != nullptr && e[0] == '1';
  }();
  const int flags = (exact ? 0 : 1) | (per_entry_min ? 2 : 0);
  allow_big_lds(part_v != nullptr ? reinterpret_cast<const void*>(beam_logprobs_kernel<true>)
                                  : reinterpret_cast<const void*>(beam_logprobs_kernel<false>), lds);
  if (part_v != nullptr)
    hipLaunchKernelGGL(beam_logprobs_kernel<true>, dim3(rows), dim3(block), lds, stream, logits,
                       beam_scores, vocab, disable_id, min_length, step_ptr, extra_min_per_batch,
                       beam > 0 ? beam : 1, el.chain_pad, y, flags, bias, part_v, part_i, topk);
  else
    hipLaunchKernelGGL(beam_logprobs_kernel<false>, dim3(rows), dim3(block), lds, stream, logits,
                       beam_scores, vocab, disable_id, min_length, step_ptr, extra_min_per_batch,
                       beam > 0 ? beam : 1, el.chain_pad, y, flags, bias, part_v, part_i, topk);
}

// The fused candidates exist where beam_logprobs_kernel keeps the row in registers (the exact
// LogSoftMax of the 32-bit compute types, vocabularies of up to 32 x 1024 words) and the entry's
// beam x topk candidates fit the one wave of beam_select_kernel.
bool beam_logprobs_topk_applicable(int64_t vocab, bool exact, int32_t beam, int32_t topk) {
  int block = 64;
  while (block < vocab / 4 && block < 1024) block <<= 1;
  if (!(topk >= 1 && topk <= kBeamCandCap && static_cast<int64_t>(beam) * topk <= 64 &&
        static_cast<int64_t>(beam) * vocab < INT_MAX))
    return false;
  // 16-bit compute types (the parallel-sum form, any row length: the row is re-read, round 6)
  static const bool fast_fused = [] {
    const char* e = std::getenv("CT2X_BEAM_FUSED_FAST");
    return e == nullptr || e[0] != '0';
  }();
  if (!exact) return fast_fused;
  return exp_sum_layout(vocab).chain_pad > 0 && vocab <= 32 * static_cast<int64_t>(block);
}

// ---------------------------------------------------------------------------------------------
// Device side of one beam-search step after TopK (src/decoding.cc:559-700): unflatten the
// 2*beam candidates of a batch entry, apply the hard target prefix (update_sample_with_prefix,
// beam arm, :21-66), record them for the host's hypothesis bookkeeping, and choose the beams of
// the next step exactly as the reference does (:596-676: a candidate among the first `beam` that
// ended is replaced by the next secondary candidate that did not) - ids, cumulative scores and
// parent rows for the cache reorder never leave the device, so consecutive steps can be queued
// (and replayed from a hipGraph) without a host round trip. The host replays the same rule on
// the recorded candidates later (decoding.cc beam_search) to build the hypotheses.
// One workgroup (64 lanes) per batch entry; 2 * beam <= 64.
// ---------------------------------------------------------------------------------------------
struct BeamSelectParams {
  const float* topk_scores;   // [batch, 2*beam] descending
  const int32_t* topk_ids;    // [batch, 2*beam] flat (beam * vocab + word)
  int32_t beam, vocab, eos;
  const int32_t* prefix_ids;  // [batch, prefix_stride] or nullptr
  const int32_t* prefix_lens; // [batch]
  int32_t prefix_stride;
  const int32_t* step_ptr;
  int32_t first_step;
  // history of candidates: [step - first_step][batch, 2*beam]
  int32_t* hist_word;
  float* hist_score;
  int32_t* hist_origin;
  // next step
  int32_t* next_ids;          // [batch*beam]
  int32_t* parents;           // [batch*beam] parent row
  float* beam_scores;         // [batch*beam]
  // generated tokens per active beam (for the logits processors), ping-pong; may be nullptr
  const int32_t* seq_src;     // [batch*beam, seq_stride]
  int32_t* seq_dst;
  int32_t seq_stride;
  // ancestry of the self-attention caches (AttnDecodeParams::ancestry), ping-pong; may be nullptr:
  // [positions][batch*beam] -> cache row holding that position of the beam's history
  const int32_t* anc_src;
  int32_t* anc_dst;
  // biased decoding: the prefix does not constrain; diverged[batch * beam + k] records that the
  // candidate at position k has left it (get_beams_divergence_from_prefix, src/decoding.cc:325-342 -
  // the reference keeps the flag per POSITION, not per origin beam; reproduced as is)
  uint8_t* diverged;
  // round 6: per-row candidates of beam_logprobs_kernel, [batch, n_part = beam * 2 * beam] (value,
  // flat index), merged here (nullptr: topk_scores / topk_ids hold the entry's TopK already)
  const float* part_v;
  const int32_t* part_i;
  int32_t n_part;
  // round 6: the step counter advances in this launch - by the LAST workgroup to finish, every
  // other one has read it by then (nullptr: a launch of its own follows)
  int32_t* advance_step;
  unsigned* ticket;
};

__global__ __launch_bounds__(64) void beam_select_kernel(BeamSelectParams p) {
  __shared__ int32_t s_word[64], s_origin[64], s_next[32];
  __shared__ float s_score[64];
  const int b = blockIdx.x, j = threadIdx.x;
  const int nc = 2 * p.beam;
  const int step = *p.step_ptr - p.first_step;
  const bool biased = p.diverged != nullptr;
  const int plen_given = p.prefix_ids ? p.prefix_lens[b] : 0;
  const int plen = biased ? 0 : plen_given;  // (use_hard_prefix == false: no forced tokens)
  // TopK over the entry's flattened [beam * vocab] scores from the rows' candidates: nc rounds of
  // wave arg-max, (value descending, flat index ascending) as topk_merge_kernel; lane r keeps the
  // r-th best.
  float merged_v = -FLT_MAX;
  int32_t merged_i = 0;
  if (p.part_v != nullptr) {
    ValIdx c;
    c.i = j < p.n_part ? p.part_i[b * p.n_part + j] : INT_MAX;
    c.v = (j < p.n_part && c.i != INT_MAX) ? p.part_v[b * p.n_part + j] : -__builtin_huge_valf();
    for (int r = 0; r < nc; ++r) {
      const ValIdx w = wave_argmax_dpp(c);
      if (w.i != INT_MAX && c.i == w.i) {  // retired (flat indices are unique)
        c.i = INT_MAX;
        c.v = -__builtin_huge_valf();
      }
      if (j == r && w.i != INT_MAX) {      // (fewer than nc candidates: -FLT_MAX / 0 as topk_merge_kernel)
        merged_v = w.v;
        merged_i = w.i;
      }
    }
  }
  if (j < nc) {
    const int32_t flat = p.part_v != nullptr ? merged_i : p.topk_ids[b * nc + j];
    int32_t origin = flat / p.vocab, word = flat % p.vocab;   // unflatten_ids, :95-116
    float sc = p.part_v != nullptr ? merged_v : p.topk_scores[b * nc + j];
    if (p.prefix_ids && !biased && step <= plen) {
      if (step < plen) {        // inside the prefix: every candidate is the prefix token
        word = p.prefix_ids[b * p.prefix_stride + step];
        sc = (j == 0) ? 0.f : -1e10f;
        origin = 0;
      } else if (j > 0 && word == p.eos) {   // first free step: no EOS in a secondary candidate
        word = 0;
        sc = -1e10f;
        origin = 0;
      }
    }
    s_word[j] = word;
    s_origin[j] = origin;
    s_score[j] = sc;
    const int64_t h = (static_cast<int64_t>(step) * gridDim.x + b) * nc + j;
    p.hist_word[h] = word;
    p.hist_score[h] = sc;
    p.hist_origin[h] = origin;
  }
  __syncthreads();
  if (biased && j < p.beam) {
    const int row = b * p.beam + j;
    p.diverged[row] = (step >= plen_given || p.diverged[row] ||
                       s_word[j] != p.prefix_ids[b * p.prefix_stride + step]) ? 1 : 0;
  }
  if (j == 0) {
    int secondary = p.beam;
    for (int k = 0; k < p.beam; ++k) {
      int next = k;
      if (s_word[k] == p.eos && step >= plen) {
        for (int c = secondary; c < nc; ++c) {
          if (s_word[c] != p.eos) {
            next = c;
            secondary = c + 1;
            break;
          }
        }
      }
      s_next[k] = next;
    }
  }
  __syncthreads();
  if (j < p.beam) {
    const int c = s_next[j];
    const int row = b * p.beam + j;
    p.next_ids[row] = s_word[c];
    p.beam_scores[row] = s_score[c];
    p.parents[row] = b * p.beam + s_origin[c];
  }
  if (p.anc_dst != nullptr) {
    // gather_beam_flat on the caches (src/layers/decoder.cc:33-56) WITHOUT moving them: beam k
    // inherits its parent's table of cache rows for every position written so far (0 .. *step_ptr;
    // the parent's own row holds the position it just appended - the tables start as the identity)
    const int rows = static_cast<int>(gridDim.x) * p.beam;
    const int written = *p.step_ptr + 1;
    for (int e = j; e < written * p.beam; e += 64) {
      const int t = e / p.beam, k = e % p.beam;
      const int parent = b * p.beam + s_origin[s_next[k]];
      p.anc_dst[static_cast<int64_t>(t) * rows + b * p.beam + k] =
          p.anc_src[static_cast<int64_t>(t) * rows + parent];
    }
  }
  if (p.seq_dst != nullptr) {
    // append_step_output + gather_beam_flat (:588, :686): beam k inherits its parent's tokens
    for (int k = 0; k < p.beam; ++k) {
      const int c = s_next[k];
      const int32_t* src = p.seq_src + static_cast<int64_t>(b * p.beam + s_origin[c]) * p.seq_stride;
      int32_t* dst = p.seq_dst + static_cast<int64_t>(b * p.beam + k) * p.seq_stride;
      for (int t = j; t < step; t += 64) dst[t] = src[t];
      if (j == 0 && step < p.seq_stride) dst[step] = s_word[c];
    }
  }
  if (p.advance_step != nullptr && grid_last_arrival(p.ticket, gridDim.x) && j == 0) {
    *p.advance_step = *p.advance_step + 1;
    __hip_atomic_store(p.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (graph replays)
  }
}

void launch_beam_select(const float* topk_scores, const int32_t* topk_ids, int64_t batch,
                        int32_t beam, int64_t vocab, int32_t eos, const int32_t* prefix_ids,
                        const int32_t* prefix_lens, int32_t prefix_stride, const int32_t* step_ptr,
                        int32_t first_step, int32_t* hist_word, float* hist_score,
                        int32_t* hist_origin, int32_t* next_ids, int32_t* parents,
                        float* beam_scores, const int32_t* seq_src, int32_t* seq_dst,
                        int32_t seq_stride, hipStream_t stream, const int32_t* ancestry_src,
                        int32_t* ancestry_dst, uint8_t* diverged, const float* part_v,
                        const int32_t* part_i, int32_t n_part, int32_t* advance_step, unsigned* ticket) {
  if (batch == 0) return;
  CT2X_CHECK(part_v == nullptr || (n_part >= 1 && n_part <= 64), "beam_select: at most 64 candidates to merge");
  CT2X_CHECK(advance_step == nullptr || ticket != nullptr, "beam_select: the advance needs its ticket");
  const BeamSelectParams p{topk_scores, topk_ids, beam, static_cast<int32_t>(vocab), eos, prefix_ids,
                           prefix_lens, prefix_stride, step_ptr, first_step, hist_word, hist_score,
                           hist_origin, next_ids, parents, beam_scores, seq_src, seq_dst, seq_stride,
                           ancestry_src, ancestry_dst, diverged, part_v, part_i, n_part, advance_step,
                           ticket};
  hipLaunchKernelGGL(beam_select_kernel, dim3(batch), dim3(64), 0, stream, p);
}

// return_attention (src/decoding.cc:503, 586-593; TransformerDecoder::decode collecting the
// alignment heads, src/layers/transformer.cc:770-828): the encoder-decoder attention distribution
// of the first `align_heads` heads of the alignment layer, averaged (ops::Mean over the heads), for
// every row of the step: out[*step_ptr][row][0 .. s_max) (zero past the source length). The
// probabilities are recomputed here from the layer's queries and the projected memory keys - the
// attention kernels keep theirs in registers / LDS - in plain fp32 (softmax(q.k * scale)).
// One workgroup per row.
__global__ __launch_bounds__(256) void attention_probs_kernel(
    const float* __restrict__ q, int64_t q_row_stride, const float* __restrict__ k,
    int64_t k_batch_stride, int64_t k_time_stride, const int32_t* __restrict__ lengths,
    int32_t kv_group, int32_t heads_per_kv, int32_t dhead, int32_t align_heads, int32_t s_max,
    float scale, const int32_t* __restrict__ step_ptr, int64_t rows, float* __restrict__ out) {
  extern __shared__ float probs[];  // [s_max] scores of one head
  __shared__ float scratch[16];
  const int64_t row = blockIdx.x;
  const int64_t b = row / kv_group;
  const int len = lengths ? min(lengths[b], s_max) : s_max;
  float* o = out + (static_cast<int64_t>(*step_ptr) * rows + row) * s_max;
  for (int t = threadIdx.x; t < s_max; t += blockDim.x) o[t] = 0.f;
  const float inv_heads = 1.f / static_cast<float>(align_heads);
  for (int h = 0; h < align_heads; ++h) {
    const float* qh = q + row * q_row_stride + static_cast<int64_t>(h) * dhead;
    const float* kh = k + b * k_batch_stride + static_cast<int64_t>(h / heads_per_kv) * dhead;
    float m = -FLT_MAX;
    for (int t = threadIdx.x; t < len; t += blockDim.x) {
      const float* kt = kh + static_cast<int64_t>(t) * k_time_stride;
      float acc = 0.f;
      for (int d = 0; d < dhead; ++d) acc = fmaf(qh[d], kt[d], acc);
      acc *= scale;
      probs[t] = acc;
      m = fmaxf(m, acc);
    }
    m = block_max(m, scratch);
    float sum = 0.f;
    for (int t = threadIdx.x; t < len; t += blockDim.x) {
      const float e = expf(probs[t] - m);
      probs[t] = e;
      sum += e;
    }
    sum = block_sum(sum, scratch);
    __syncthreads();
    for (int t = threadIdx.x; t < len; t += blockDim.x) o[t] += (probs[t] / sum) * inv_heads;
    __syncthreads();
  }
}

void launch_attention_probs(const float* q, int64_t q_row_stride, const float* k, int64_t k_batch_stride,
                            int64_t k_time_stride, const int32_t* lengths, int64_t rows,
                            int32_t kv_group, int32_t heads, int32_t heads_kv, int32_t dhead,
                            int32_t align_heads, int32_t s_max, float scale, const int32_t* step_ptr,
                            float* out, hipStream_t stream) {
  if (rows == 0) return;
  const size_t lds = sizeof(float) * static_cast<size_t>(s_max);
  allow_big_lds(reinterpret_cast<const void*>(attention_probs_kernel), lds);
  hipLaunchKernelGGL(attention_probs_kernel, dim3(rows), dim3(256), lds, stream, q, q_row_stride, k,
                     k_batch_stride, k_time_stride, lengths, kv_group, heads / heads_kv, dhead,
                     align_heads, s_max, scale, step_ptr, rows, out);
}

// score_sequences (src/scoring.cc:42-48): LogSoftMax over the step's logits, then Gather of the
// given target id - one launch, the log-probability of targets[row] under logits[row, :]
// (x + (-max - log(sum exp(x - max))), the reference's log-softmax form, src/cpu/kernels.cc:452).
__global__ void score_tokens_kernel(const float* __restrict__ logits, int64_t vocab,
                                    const int32_t* __restrict__ targets,
                                    float* __restrict__ out) {
  __shared__ float scratch[16];
  const int64_t row = blockIdx.x;
  const float* x = logits + row * vocab;
  float m = -FLT_MAX;
  for (int64_t j = threadIdx.x; j < vocab; j += blockDim.x) m = fmaxf(m, x[j]);
  m = block_max(m, scratch);
  float s = 0.f;
  for (int64_t j = threadIdx.x; j < vocab; j += blockDim.x) s += expf(x[j] - m);
  s = block_sum(s, scratch);
  if (threadIdx.x == 0) {
    const float c = -m - logf(s);
    out[row] = x[targets[row]] + c;
  }
}

void launch_score_tokens(const float* logits, int64_t rows, int64_t vocab, const int32_t* targets,
                         float* out, hipStream_t stream) {
  if (rows == 0) return;
  int block = 64;
  while (block < vocab / 4 && block < 1024) block <<= 1;
  hipLaunchKernelGGL(score_tokens_kernel, dim3(rows), dim3(block), 0, stream, logits, vocab,
                     targets, out);
}

}  // namespace ct2x
