// Device side of the Whisper glue (src/models/whisper.cc, src/layers/whisper.cc):
//   * im2col for the two kernel-3 convolutions of the encoder front end: the convolution itself
//     then runs as a Dense on the matrix pipe (int8 / 16-bit MFMA GEMM with the bias + GELU
//     epilogue), which is also how the reference's CPU path computes it
//     (Conv1D::compute_with_gemm, src/ops/conv1d_cpu.cc:139-207) - and the [time, channel] output
//     of that GEMM is already the layout the encoder layers want (no Transpose launch);
//   * the learned position table added to the convolution output (PositionEmbedding);
//   * ApplyTimestampRules + SuppressTokensBegin + GetNoSpeechProbs (whisper.cc:202-231, 741-861,
//     decoding_utils.cc:172-194) as one launch inside the decoding step's graph;
//   * the raw (pre-softmax) encoder-decoder attention scores of the alignment heads and the
//     post-processing of Whisper::align (whisper.cc:388-423, 520-553).
// HBM-bound row work throughout: one workgroup per row, coalesced along the fastest axis.

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>

#include "device_utils.h"
#include "kernels.h"
#include "runtime.h"

namespace ct2x {

// ---------------------------------------------------------------------------------------------
// im2col_transposed (src/ops/conv1d_cpu.cc:209-243), kernel 3, any stride / padding:
// out[(b * out_len + t) * (C * 3) + c * 3 + k] = x[b, c, t * stride - padding + k] (0 outside).
// `time_major`: the input is [batch, length, channels] (the previous convolution's GEMM output)
// instead of the [batch, channels, length] feature layout.
// ---------------------------------------------------------------------------------------------
__global__ void im2col_k3_kernel(const float* __restrict__ x, int64_t channels, int64_t length,
                                 int64_t out_len, int stride, int padding, bool time_major,
                                 float* __restrict__ out) {
  const int64_t row = blockIdx.x;  // b * out_len + t
  const int64_t b = row / out_len, t = row % out_len;
  const int64_t cols = channels * 3;
  const float* xb = x + b * channels * length;
  float* o = out + row * cols;
  for (int64_t j = threadIdx.x; j < cols; j += blockDim.x) {
    const int64_t c = j / 3, k = j % 3;
    const int64_t pos = t * stride - padding + k;
    float v = 0.f;
    if (pos >= 0 && pos < length) v = time_major ? xb[pos * channels + c] : xb[c * length + pos];
    o[j] = v;
  }
}

void launch_im2col_k3(const float* x, int64_t batch, int64_t channels, int64_t length,
                      int64_t out_len, int stride, int padding, bool time_major, float* out,
                      hipStream_t stream) {
  if (batch * out_len == 0) return;
  hipLaunchKernelGGL(im2col_k3_kernel, dim3(batch * out_len), dim3(256), 0, stream, x, channels,
                     length, out_len, stride, padding, time_major, out);
}

// x[b, t, :] += table[t, :] (PositionEncoder::operator(), src/layers/common.cc:150-175)
__global__ void add_positions_kernel(float* __restrict__ x, const float* __restrict__ table,
                                     int64_t time, int64_t depth) {
  const int64_t row = blockIdx.x;
  const int64_t t = row % time;
  float* xr = x + row * depth;
  const float* tr = table + t * depth;
  for (int64_t j = threadIdx.x; j < depth; j += blockDim.x) xr[j] = xr[j] + tr[j];
}

void launch_add_positions(float* x, const float* table, int64_t batch, int64_t time, int64_t depth,
                          hipStream_t stream) {
  if (batch * time == 0) return;
  hipLaunchKernelGGL(add_positions_kernel, dim3(batch * time), dim3(256), 0, stream, x, table, time,
                     depth);
}

// ---------------------------------------------------------------------------------------------
// Whisper's logits processors, one workgroup per row, in make_logits_processors order
// (src/decoding.cc:1091-1120) AFTER RepetitionPenalty / NoRepeatNgram / SuppressTokens
// (process_logits_kernel): SuppressTokensBegin, then ApplyTimestampRules. GetNoSpeechProbs
// (apply_first) reads the untouched logits, so the engine launches no_speech_prob_kernel before
// any other processor.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void no_speech_prob_kernel(const float* __restrict__ logits,
                                                             int64_t vocab, int32_t no_speech_id,
                                                             const int32_t* __restrict__ step_ptr,
                                                             int32_t first_step,
                                                             float* __restrict__ out) {
  __shared__ float scratch[16];
  if (*step_ptr != first_step) return;  // step 0 of the decoding loop only (whisper.cc:218)
  const int64_t row = blockIdx.x;
  const float* x = logits + row * vocab;
  float m = -FLT_MAX;
  for (int64_t j = threadIdx.x; j < vocab; j += blockDim.x) m = fmaxf(m, x[j]);
  m = block_max(m, scratch);
  float s = 0.f;
  for (int64_t j = threadIdx.x; j < vocab; j += blockDim.x) s += expf(x[j] - m);
  s = block_sum(s, scratch);
  if (threadIdx.x == 0) out[row] = expf(x[no_speech_id] - m) / s;
}

void launch_no_speech_prob(const float* logits, int64_t rows, int64_t vocab, int32_t no_speech_id,
                           const int32_t* step_ptr, int32_t first_step, float* out,
                           hipStream_t stream) {
  if (rows == 0) return;
  hipLaunchKernelGGL(no_speech_prob_kernel, dim3(rows), dim3(256), 0, stream, logits, vocab,
                     no_speech_id, step_ptr, first_step, out);
}

__global__ __launch_bounds__(256) void whisper_rules_kernel(WhisperRulesParams p) {
  __shared__ float scratch[16];
  __shared__ int s_flag;
  const int64_t row = blockIdx.x;
  const int tid = threadIdx.x;
  float* lg = p.logits + row * p.vocab;
  const int step = *p.step_ptr - p.first_step;  // step of the decoding loop
  const int sample_begin = p.prefix_len ? p.prefix_len[row / p.prefix_div] : 0;
  const int32_t* sq = p.seq + row * p.row_stride;
  auto tok = [&](int t) { return sq[static_cast<int64_t>(t) * p.t_stride]; };

  // SuppressTokensBegin (decoding_utils.cc:177-194)
  if (step == sample_begin)
    for (int i = tid; i < p.n_begin; i += blockDim.x) lg[p.begin_ids[i]] = -FLT_MAX;
  if (!p.timestamp_rules) return;

  const int ts_begin = p.timestamp_begin, ts_end = p.timestamp_end;
  if (tid == 0) lg[p.no_timestamps_id] = -FLT_MAX;
  bool check_prob = false;
  if (step == sample_begin && step == 0) {
    for (int i = tid; i < ts_begin; i += blockDim.x) lg[i] = -FLT_MAX;
    for (int i = p.max_initial_timestamp + 1 + tid; i <= ts_end; i += blockDim.x) lg[i] = -FLT_MAX;
  } else if (step > sample_begin) {
    const int last = tok(step - 1);
    if (last >= ts_begin) {
      const int penultimate = (step - 1 > sample_begin) ? tok(step - 2) : last;
      if (penultimate >= ts_begin) {  // has to be non-timestamp
        for (int i = ts_begin + tid; i <= ts_end; i += blockDim.x) lg[i] = -FLT_MAX;
      } else {                        // cannot be normal text tokens
        for (int i = tid; i < p.eot_id; i += blockDim.x) lg[i] = -FLT_MAX;
        for (int i = ts_begin + tid; i < last; i += blockDim.x) lg[i] = -FLT_MAX;
        check_prob = true;
      }
    } else {
      check_prob = true;
      // timestamps shouldn't decrease: forbid timestamp tokens up to the last one produced
      if (tid == 0) {
        s_flag = -1;
        for (int t = step - 1; t >= sample_begin; --t) {
          const int token = tok(t);
          if (token >= ts_begin) {
            s_flag = token;
            break;
          }
        }
      }
      __syncthreads();
      const int bound = s_flag;
      for (int i = ts_begin + tid; i <= bound; i += blockDim.x) lg[i] = -FLT_MAX;
    }
  }
  if (!check_prob) return;
  __syncthreads();  // the bans above are part of the distribution (disable_tokens.apply())
  // should_sample_timestamp (whisper.cc:843-857): logsumexp over the timestamp log-probabilities
  // against the best text log-probability. Both sides carry the same -logsumexp(all) term, so the
  // comparison is made on the logits.
  float mt = -FLT_MAX;
  for (int i = tid; i < ts_begin; i += blockDim.x) mt = fmaxf(mt, lg[i]);
  mt = block_max(mt, scratch);
  float ms = -FLT_MAX;
  for (int i = ts_begin + tid; i <= ts_end; i += blockDim.x) ms = fmaxf(ms, lg[i]);
  ms = block_max(ms, scratch);
  float s = 0.f;
  for (int i = ts_begin + tid; i <= ts_end; i += blockDim.x) s += expf(lg[i] - ms);
  s = block_sum(s, scratch);
  const float timestamp_lse = ms + logf(s);
  if (timestamp_lse > mt)
    for (int i = tid; i < ts_begin; i += blockDim.x) lg[i] = -FLT_MAX;
}

void launch_whisper_rules(const WhisperRulesParams& p, hipStream_t stream) {
  if (p.rows == 0) return;
  hipLaunchKernelGGL(whisper_rules_kernel, dim3(p.rows), dim3(256), 0, stream, p);
}

// ---------------------------------------------------------------------------------------------
// Whisper::align pieces.
// ---------------------------------------------------------------------------------------------

// Raw scores of selected heads: out[b, i, t, s] = scale * q[b, t, head_i, :] . k[b, s, head_i, :]
// (dot_product_attention before its SoftMax, src/layers/attention.cc:178-270, with
// return_normalized_attention() == false, layers/whisper.h). q rows have stride q_row_stride,
// memory keys [batch, s_max, k_time_stride] (the cross-attention K/V in GEMM-output layout).
__global__ __launch_bounds__(256) void attention_scores_kernel(
    const float* __restrict__ q, int64_t q_batch_stride, int64_t q_time_stride,
    const float* __restrict__ k, int64_t k_batch_stride, int64_t k_time_stride,
    const int32_t* __restrict__ heads, int64_t n_sel, int64_t tq, int64_t tk, int64_t dhead,
    float scale, float* __restrict__ out) {
  extern __shared__ float s_q[];
  // block = (b, i, t)
  const int64_t idx = blockIdx.x;
  const int64_t t = idx % tq, i = (idx / tq) % n_sel, b = idx / (tq * n_sel);
  const int64_t h = heads[i];
  const float* qr = q + b * q_batch_stride + t * q_time_stride + h * dhead;
  for (int64_t j = threadIdx.x; j < dhead; j += blockDim.x) s_q[j] = qr[j] * scale;
  __syncthreads();
  float* o = out + idx * tk;
  for (int64_t s = threadIdx.x; s < tk; s += blockDim.x) {
    const float* kr = k + b * k_batch_stride + s * k_time_stride + h * dhead;
    float acc = 0.f;
    for (int64_t j = 0; j < dhead; ++j) acc += s_q[j] * kr[j];
    o[s] = acc;
  }
}

void launch_attention_scores(const float* q, int64_t q_batch_stride, int64_t q_time_stride,
                             const float* k, int64_t k_batch_stride, int64_t k_time_stride,
                             const int32_t* heads, int64_t batch, int64_t n_sel, int64_t tq,
                             int64_t tk, int64_t dhead, float scale, float* out,
                             hipStream_t stream) {
  if (batch * n_sel * tq == 0) return;
  hipLaunchKernelGGL(attention_scores_kernel, dim3(batch * n_sel * tq), dim3(256),
                     sizeof(float) * dhead, stream, q, q_batch_stride, q_time_stride, k,
                     k_batch_stride, k_time_stride, heads, n_sel, tq, tk, dhead, scale, out);
}

// ops::LayerNorm(axis = -2, epsilon = 0) without gamma / beta on [outer, axis, inner]
// (compute_alignments, whisper.cc:396; src/ops/layer_norm_cpu.cc axis form): every (outer, inner)
// column is normalised over `axis` with the population variance.
__global__ void layer_norm_axis_kernel(const float* __restrict__ x, int64_t axis, int64_t inner,
                                       int64_t inner_used, float* __restrict__ y) {
  const int64_t col = blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t o = blockIdx.y;
  if (col >= inner_used) return;
  const float* xo = x + o * axis * inner + col;
  float* yo = y + o * axis * inner + col;
  // cpu::layer_norm_axis (src/cpu/kernels.cc:499-539) AS COMPILED: the function sits inside
  // CT2_FFAST_MATH_BEGIN / END (`#pragma GCC optimize("-ffast-math")`, kernels.cc:461-574), so GCC
  // vectorises the reduction - four partial sums over k = l, l + 4, ... folded as
  // (p0 + p2) + (p1 + p3), then the axis % 4 last elements one by one (axis <= 4: one chain) - and
  // rewrites the divisions: mean = sum * (1 / n), variance = fma(sum_squares, 1 / n, -(mean * mean)).
  // Read off the reference's object code, restated in oracle/whisper_oracle.py and pinned bit for
  // bit against the library there (tests/test_whisper_oracle.py).
  float sum = 0.f, sum_squares = 0.f;
  int64_t first_tail = 0;
  if (axis > 4) {
    float s4[4] = {0.f, 0.f, 0.f, 0.f}, q4[4] = {0.f, 0.f, 0.f, 0.f};
    const int64_t n4 = axis / 4;
    for (int64_t g = 0; g < n4; ++g) {
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        const float v = xo[(4 * g + l) * inner];
        q4[l] = fmaf(v, v, q4[l]);
        s4[l] = s4[l] + v;
      }
    }
    sum = (s4[1] + s4[3]) + (s4[0] + s4[2]);
    sum_squares = (q4[1] + q4[3]) + (q4[0] + q4[2]);
    first_tail = 4 * n4;
  }
  for (int64_t a = first_tail; a < axis; ++a) {
    const float v = xo[a * inner];
    sum_squares = fmaf(v, v, sum_squares);
    sum = sum + v;
  }
  const float inv_n = 1.f / static_cast<float>(axis);
  const float mean = sum * inv_n;
  const float mean2 = mean * mean;
  const float variance = fmaxf(fmaf(sum_squares, inv_n, -mean2), 0.f);
  const float rstd = 1.f / sqrtf(variance);
  for (int64_t a = 0; a < axis; ++a) yo[a * inner] = (xo[a * inner] - mean) * rstd;
}

void launch_layer_norm_axis(const float* x, int64_t outer, int64_t axis, int64_t inner,
                            int64_t inner_used, float* y, hipStream_t stream) {
  if (outer * axis * inner_used == 0) return;
  hipLaunchKernelGGL(layer_norm_axis_kernel, dim3((inner_used + 63) / 64, outer), dim3(64), 0,
                     stream, x, axis, inner, inner_used, y);
}

// log-free softmax probability of one target token per row over the first `classes` logits
// (Whisper::align: SoftMax over the text vocabulary [0, eot), then Gather of the output ids,
// whisper.cc:497-506); targets < 0 or >= classes give 0.
__global__ __launch_bounds__(256) void token_probs_kernel(const float* __restrict__ logits,
                                                          int64_t vocab, int64_t classes,
                                                          const int32_t* __restrict__ targets,
                                                          float* __restrict__ out) {
  __shared__ float scratch[16];
  const int64_t row = blockIdx.x;
  const float* x = logits + row * vocab;
  float m = -FLT_MAX;
  for (int64_t j = threadIdx.x; j < classes; j += blockDim.x) m = fmaxf(m, x[j]);
  m = block_max(m, scratch);
  float s = 0.f;
  for (int64_t j = threadIdx.x; j < classes; j += blockDim.x) s += expf(x[j] - m);
  s = block_sum(s, scratch);
  if (threadIdx.x == 0) {
    const int32_t t = targets[row];
    out[row] = (t >= 0 && t < classes) ? expf(x[t] - m) / s : 0.f;
  }
}

void launch_token_probs(const float* logits, int64_t rows, int64_t vocab, int64_t classes,
                        const int32_t* targets, float* out, hipStream_t stream) {
  if (rows == 0) return;
  hipLaunchKernelGGL(token_probs_kernel, dim3(rows), dim3(256), 0, stream, logits, vocab, classes,
                     targets, out);
}

// out[b, j] = softmax over the gathered logits[b, ids[j]] (Whisper::detect_language,
// whisper.cc:600-603); one workgroup per row, n <= 1024 ids.
__global__ __launch_bounds__(256) void gather_softmax_kernel(const float* __restrict__ logits,
                                                             int64_t vocab,
                                                             const int32_t* __restrict__ ids,
                                                             int64_t n, float* __restrict__ out) {
  __shared__ float scratch[16];
  const int64_t row = blockIdx.x;
  const float* x = logits + row * vocab;
  float m = -FLT_MAX;
  for (int64_t j = threadIdx.x; j < n; j += blockDim.x) m = fmaxf(m, x[ids[j]]);
  m = block_max(m, scratch);
  float s = 0.f;
  for (int64_t j = threadIdx.x; j < n; j += blockDim.x) s += expf(x[ids[j]] - m);
  s = block_sum(s, scratch);
  for (int64_t j = threadIdx.x; j < n; j += blockDim.x) out[row * n + j] = expf(x[ids[j]] - m) / s;
}

void launch_gather_softmax(const float* logits, int64_t rows, int64_t vocab, const int32_t* ids,
                           int64_t n, float* out, hipStream_t stream) {
  if (rows * n == 0) return;
  hipLaunchKernelGGL(gather_softmax_kernel, dim3(rows), dim3(256), 0, stream, logits, vocab, ids, n,
                     out);
}

// One batch entry of Whisper::align's attention post-processing input (whisper.cc:520-553):
// out[i, t, 0:frames] = softmax(scores[i, min(t, length - 1), 0:frames]) for t < rows_out - the
// padded query rows of the reference's batch are copies of the entry's last row (Padder::
// add_padding, src/padder.cc:37-40), and the softmax runs over the entry's own frames.
__global__ __launch_bounds__(256) void align_softmax_kernel(const float* __restrict__ scores,
                                                            int64_t tq, int64_t tk, int64_t length,
                                                            int64_t rows_out, int64_t frames,
                                                            float* __restrict__ out) {
  __shared__ float scratch[16];
  const int64_t t = blockIdx.x % rows_out, i = blockIdx.x / rows_out;
  const int64_t ts = t < length ? t : length - 1;
  const float* x = scores + (i * tq + ts) * tk;
  float* o = out + (i * rows_out + t) * frames;
  float m = -FLT_MAX;
  for (int64_t j = threadIdx.x; j < frames; j += blockDim.x) m = fmaxf(m, x[j]);
  m = block_max(m, scratch);
  float s = 0.f;
  for (int64_t j = threadIdx.x; j < frames; j += blockDim.x) s += expf(x[j] - m);
  s = block_sum(s, scratch);
  for (int64_t j = threadIdx.x; j < frames; j += blockDim.x) o[j] = expf(x[j] - m) / s;
}

void launch_align_softmax(const float* scores, int64_t n_sel, int64_t tq, int64_t tk, int64_t length,
                          int64_t rows_out, int64_t frames, float* out, hipStream_t stream) {
  if (n_sel * rows_out * frames == 0) return;
  hipLaunchKernelGGL(align_softmax_kernel, dim3(n_sel * rows_out), dim3(256), 0, stream, scores, tq,
                     tk, length, rows_out, frames, out);
}

// Dequantize of the int8 convolution as the reference's CPU path evaluates it: its GEMM is
// weight x im2col^T, so the WEIGHT scale is the row scale of cpu::dequantize_gemm_output
// (src/cpu/kernels.cc:653-685: multiplied in as a reciprocal) and the input row's scale the
// column scale (divided by) - the other way round from a Dense. Then bias and activation
// (apply_bias_and_activation, src/ops/gemm.cc:11-25). c [rows, n] int32 -> y [rows, n].
__global__ void conv_dequantize_kernel(const int32_t* __restrict__ c, const float* __restrict__ x_scale,
                                       const float* __restrict__ w_scale,
                                       const float* __restrict__ bias, int64_t n, int act,
                                       int64_t total, float* __restrict__ y) {
  const int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (e >= total) return;
  const int64_t row = e / n, col = e % n;
  float v = static_cast<float>(c[e]) * (1.f / w_scale[col]);
  v = v / x_scale[row];
  if (bias) v = v + bias[col];
  y[e] = act >= 0 ? apply_activation(v, act) : v;
}

void launch_conv_dequantize(const int32_t* c, const float* x_scale, const float* w_scale,
                            const float* bias, int64_t rows, int64_t n, int act, float* y,
                            hipStream_t stream) {
  const int64_t total = rows * n;
  if (total == 0) return;
  hipLaunchKernelGGL(conv_dequantize_kernel, dim3(static_cast<unsigned>((total + 255) / 256)),
                     dim3(256), 0, stream, c, x_scale, w_scale, bias, n, act, total, y);
}

}  // namespace ct2x
