// TEST INFRASTRUCTURE ONLY -- never linked into or imported by the product.
//
// Thin C ABI over the REAL reference CPU library (oracle/_ref/libctranslate2_ref.so,
// compiled unmodified from /root/reference by oracle/Makefile). It exists so that the
// Python test-suite can (a) pin our plain-C restatement (ct2_oracle.c) against the
// reference's own op implementations, (b) generate the golden fixtures committed under
// tests/golden/, and (c) time the reference's CPU path as the `cpu_baseline`.
//
// Every entry point drives the reference through its public C++ API:
//   ops::*            include/ctranslate2/ops/*.h
//   Translator        include/ctranslate2/translator.h:27-60
//   layers::Transformer{Encoder,Decoder}  include/ctranslate2/layers/transformer.h
// and returns 0 on success, -1 on exception (message via ct2ref_last_error()).

#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "dtw.h"
#include <ctranslate2/layers/transformer.h>
#include <ctranslate2/models/model.h>
#include <ctranslate2/models/sequence_to_sequence.h>
#include <ctranslate2/models/language_model.h>
#include <ctranslate2/models/whisper.h>
#include <ctranslate2/ops/ops.h>
#include <ctranslate2/translator.h>
#include <ctranslate2/generator.h>
#include <ctranslate2/utils.h>
#include <ctranslate2/decoding_utils.h>

using namespace ctranslate2;

static thread_local std::string g_last_error;

#define CT2REF_TRY try {
#define CT2REF_CATCH                                   \
  }                                                    \
  catch (const std::exception& e) {                    \
    g_last_error = e.what();                           \
    return -1;                                         \
  }                                                    \
  return 0;

template <typename T>
static StorageView view(const T* data, Shape shape) {
  return StorageView(std::move(shape), const_cast<T*>(data), Device::CPU);
}

template <typename T>
static void copy_out(const StorageView& x, T* out) {
  std::memcpy(out, x.data<T>(), x.size() * sizeof(T));
}

extern "C" {

const char* ct2ref_last_error() { return g_last_error.c_str(); }

int ct2ref_set_num_threads(int n) {
  CT2REF_TRY
  set_num_threads(n);
  CT2REF_CATCH
}

int ct2ref_mayiuse_int8() { return mayiuse_int8(Device::CPU) ? 1 : 0; }

// ops::Quantize (src/ops/quantize.cc:21-51), INT8 per-row.
int ct2ref_quantize_s8(const float* x, int64_t rows, int64_t depth, int round_before_cast,
                       int8_t* q, float* scales) {
  CT2REF_TRY
  StorageView in = view(x, {rows, depth});
  StorageView out(DataType::INT8), scale(DataType::FLOAT32);
  const bool shift = (round_before_cast & 2) != 0;
  ops::Quantize(ops::Quantize::ScaleType::GLOBAL, /*shift_to_uint8=*/shift,
                (round_before_cast & 1) != 0)(in, out, scale);
  copy_out(out, q);
  if (shift)  // the op stored uint8 bit patterns: report the effective signed operand
    for (int64_t i = 0; i < rows * depth; ++i)
      q[i] = static_cast<int8_t>(static_cast<int>(static_cast<uint8_t>(q[i])) - 128);
  copy_out(scale, scales);
  CT2REF_CATCH
}

// ops::Gemm s8*s8->s32 with trans_b (src/ops/gemm.cc:45-107).
int ct2ref_gemm_s8(const int8_t* a, const int8_t* b, int64_t m, int64_t n, int64_t k,
                   int32_t* c) {
  CT2REF_TRY
  StorageView A = view(a, {m, k});
  StorageView B = view(b, {n, k});
  StorageView C(DataType::INT32);
  ops::Gemm(1.f, 0.f, false, true)(A, B, C);
  copy_out(C, c);
  CT2REF_CATCH
}

// ops::Gemm f32 with trans_b, optional bias / residual / activation.
int ct2ref_gemm_f32(const float* a, const float* b, int64_t m, int64_t n, int64_t k,
                    int trans_b, const float* bias, const float* residual, int act, float* c) {
  CT2REF_TRY
  StorageView A = view(a, {m, k});
  StorageView B = trans_b ? view(b, {n, k}) : view(b, {k, n});
  StorageView C(DataType::FLOAT32);
  ops::ActivationType act_type = static_cast<ops::ActivationType>(act < 0 ? 0 : act);
  std::unique_ptr<StorageView> bias_v, res_v;
  if (bias) bias_v = std::make_unique<StorageView>(view(bias, {n}));
  if (residual) res_v = std::make_unique<StorageView>(view(residual, {m, n}));
  ops::Gemm(1.f, 0.f, false, trans_b != 0, false, false, act < 0 ? nullptr : &act_type)(
      A, B, C, nullptr, bias_v.get(), res_v.get());
  copy_out(C, c);
  CT2REF_CATCH
}

// ops::Dequantize GEMM-output form (src/ops/dequantize.cc:46-59). act < 0 => none,
// otherwise the integer value of ops::ActivationType (include/ctranslate2/ops/activation.h).
int ct2ref_dequantize_gemm_output(const int32_t* c, const float* a_scale, const float* b_scale,
                                  const float* bias, int64_t m, int64_t n, int act, float* y) {
  CT2REF_TRY
  StorageView C = view(c, {m, n});
  StorageView sa = view(a_scale, {m});
  StorageView sb = view(b_scale, {n});
  std::unique_ptr<StorageView> bias_v;
  if (bias) bias_v = std::make_unique<StorageView>(view(bias, {n}));
  ops::ActivationType act_type = static_cast<ops::ActivationType>(act < 0 ? 0 : act);
  StorageView Y(DataType::FLOAT32);
  ops::Dequantize(act < 0 ? nullptr : &act_type)(C, sa, sb, false, true, Y, bias_v.get());
  copy_out(Y, y);
  CT2REF_CATCH
}

// ops::Dequantize int8 rows -> float (embedding path).
int ct2ref_dequantize_rows(const int8_t* x, const float* scale, int64_t rows, int64_t depth,
                           float* y) {
  CT2REF_TRY
  StorageView X = view(x, {rows, depth});
  StorageView S = view(scale, {rows});
  StorageView Y(DataType::FLOAT32);
  ops::Dequantize()(X, S, Y);
  copy_out(Y, y);
  CT2REF_CATCH
}

int ct2ref_layer_norm(const float* x, const float* gamma, const float* beta, int64_t rows,
                      int64_t depth, float eps, float* y) {
  CT2REF_TRY
  StorageView X = view(x, {rows, depth});
  StorageView G = view(gamma, {depth});
  StorageView B = view(beta, {depth});
  StorageView Y(DataType::FLOAT32);
  ops::LayerNorm(-1, eps)(B, G, X, Y);
  copy_out(Y, y);
  CT2REF_CATCH
}

int ct2ref_rms_norm(const float* x, const float* gamma, int64_t rows, int64_t depth, float eps,
                    int use_residual, float* y) {
  CT2REF_TRY
  StorageView X = view(x, {rows, depth});
  StorageView G = view(gamma, {depth});
  StorageView Y(DataType::FLOAT32);
  ops::RMSNorm(eps, use_residual != 0)(G, X, Y);
  copy_out(Y, y);
  CT2REF_CATCH
}

int ct2ref_softmax(const float* x, const int32_t* lengths, int64_t rows, int64_t depth, int log,
                   float* y) {
  CT2REF_TRY
  StorageView X = view(x, {rows, depth});
  StorageView Y(DataType::FLOAT32);
  std::unique_ptr<StorageView> L;
  if (lengths) L = std::make_unique<StorageView>(view(lengths, {rows}));
  ops::SoftMax(log != 0)(X, L.get(), Y);
  copy_out(Y, y);
  CT2REF_CATCH
}

// ops::Rotary on [batch, heads, time, dhead]; sin/cos are [time, ndims] (already offset).
int ct2ref_rotary(const float* x, const float* sin, const float* cos, int64_t batch,
                  int64_t heads, int64_t time, int64_t dhead, int64_t ndims, int interleave,
                  float* y) {
  CT2REF_TRY
  StorageView X = view(x, {batch, heads, time, dhead});
  StorageView S = view(sin, {time, ndims});
  StorageView C = view(cos, {time, ndims});
  StorageView Y(DataType::FLOAT32);
  ops::Rotary(ndims, interleave != 0)(X, S, C, Y, true);
  copy_out(Y, y);
  CT2REF_CATCH
}

int ct2ref_topk(const float* x, int64_t rows, int64_t depth, int64_t k, float* values,
                int32_t* indices) {
  CT2REF_TRY
  StorageView X = view(x, {rows, depth});
  StorageView V(DataType::FLOAT32), I(DataType::INT32);
  const ops::TopK topk_op(k);
  topk_op(X, V, I);
  copy_out(V, values);
  copy_out(I, indices);
  CT2REF_CATCH
}

int ct2ref_gather_rows(const float* data, int64_t n_rows, int64_t row_size, const int32_t* idx,
                       int64_t n_idx, float* out) {
  CT2REF_TRY
  StorageView D = view(data, {n_rows, row_size});
  StorageView I = view(idx, {n_idx});
  StorageView O(DataType::FLOAT32);
  ops::Gather()(D, I, O);
  copy_out(O, out);
  CT2REF_CATCH
}

// ops::MatMul on [batch, m, k] x [batch, k|n, n|k] (src/ops/matmul.cc:14-84).
int ct2ref_matmul(const float* a, const float* b, int64_t batch, int64_t m, int64_t n, int64_t k,
                  int trans_b, float alpha, float* c) {
  CT2REF_TRY
  StorageView A = view(a, {batch, m, k});
  StorageView B = trans_b ? view(b, {batch, n, k}) : view(b, {batch, k, n});
  StorageView C(DataType::FLOAT32);
  ops::MatMul(false, trans_b != 0, alpha)(A, B, C);
  copy_out(C, c);
  CT2REF_CATCH
}

// The reference's own LogitsProcessor classes (include/ctranslate2/decoding_utils.h) applied in
// make_logits_processors order on a CPU logits tensor [rows, vocab], history seq [rows, length].
int ct2ref_process_logits(float* logits, int64_t rows, int64_t vocab, const int32_t* seq,
                          int32_t length, float penalty, int32_t ngram, const int32_t* sup_ids,
                          const int32_t* sup_off, int32_t n_sup) {
  CT2REF_TRY
  StorageView lg({rows, vocab}, logits, Device::CPU);  // view over the caller's buffer
  StorageView sequences(DataType::INT32);
  if (length > 0) {
    std::vector<int32_t> flat(seq, seq + rows * length);
    sequences = StorageView({rows, static_cast<dim_t>(length)}, flat, Device::CPU);
  }
  std::vector<std::shared_ptr<LogitsProcessor>> processors;
  if (penalty != 1.f) processors.emplace_back(std::make_shared<RepetitionPenalty>(penalty));
  if (ngram > 0) processors.emplace_back(std::make_shared<NoRepeatNgram>(ngram));
  if (n_sup > 0) {
    std::vector<std::vector<size_t>> banned;
    for (int32_t s = 0; s < n_sup; ++s)
      banned.emplace_back(sup_ids + sup_off[s], sup_ids + sup_off[s + 1]);
    processors.emplace_back(std::make_shared<SuppressSequences>(std::move(banned)));
  }
  DisableTokens disable_tokens(lg);
  const std::vector<dim_t> batch_offset(rows, 0);
  for (const auto& p : processors)
    p->apply(length, lg, disable_tokens, sequences, batch_offset, nullptr);
  disable_tokens.apply();
  CT2REF_CATCH
}

// The reference's own negative_dtw (src/dtw.cc) on a host matrix: pins ct2x_host_negative_dtw.
int64_t ct2ref_negative_dtw(const float* x, int64_t n, int64_t m, int64_t* path_i, int64_t* path_j) {
  try {
    std::vector<float> flat(x, x + n * m);
    StorageView X({n, m}, flat, Device::CPU);
    const auto path = negative_dtw(X);
    for (size_t k = 0; k < path.size(); ++k) {
      path_i[k] = path[k].first;
      path_j[k] = path[k].second;
    }
    return static_cast<int64_t>(path.size());
  } catch (const std::exception& e) {
    std::fprintf(stderr, "ct2ref_negative_dtw: %s\n", e.what());
    return -1;
  }
}

// ---------------------------------------------------------------------------------------
// Model-level entry points.
// ---------------------------------------------------------------------------------------

struct RefModel {
  std::shared_ptr<const models::Model> model;
  std::unique_ptr<Translator> translator;
  std::unique_ptr<Generator> generator;
  int intra_threads = 0;
};

// device: "cpu", or "cuda" in the Level-1 integration build (oracle/Makefile target mi355: the
// reference's own front-end on the MI355X kernels; Device::CUDA is the reference's name for it).
void* ct2ref_model_load_device(const char* path, const char* device, const char* compute_type,
                               int inter_threads, int intra_threads) {
  try {
    auto handle = std::make_unique<RefModel>();
    handle->intra_threads = intra_threads;
    models::ModelLoader loader(path);
    loader.device = str_to_device(device);
    loader.device_indices = {0};
    loader.num_replicas_per_device = inter_threads > 0 ? inter_threads : 1;
    loader.compute_type = str_to_compute_type(compute_type);
    auto models = loader.load();
    handle->model = models[0];
    ReplicaPoolConfig config;
    config.num_threads_per_replica = intra_threads;
    if (dynamic_cast<const models::SequenceToSequenceModel*>(handle->model.get()))
      handle->translator = std::make_unique<Translator>(models, config);
    else
      handle->generator = std::make_unique<Generator>(models, config);
    return handle.release();
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return nullptr;
  }
}

void* ct2ref_model_load(const char* path, const char* compute_type, int inter_threads,
                        int intra_threads) {
  return ct2ref_model_load_device(path, "cpu", compute_type, inter_threads, intra_threads);
}

void ct2ref_model_free(void* h) { delete static_cast<RefModel*>(h); }

int ct2ref_model_compute_type(void* h, char* out, int out_size) {
  CT2REF_TRY
  auto* m = static_cast<RefModel*>(h);
  const std::string s = compute_type_to_str(m->model->effective_compute_type());
  std::strncpy(out, s.c_str(), out_size - 1);
  out[out_size - 1] = 0;
  CT2REF_CATCH
}

static std::vector<std::vector<std::string>>
ids_to_tokens(const Vocabulary& vocab, const int32_t* ids, const int32_t* lengths, int batch,
              int stride) {
  std::vector<std::vector<std::string>> out(batch);
  for (int b = 0; b < batch; ++b) {
    out[b].reserve(lengths[b]);
    for (int t = 0; t < lengths[b]; ++t)
      out[b].push_back(vocab.to_token(ids[b * stride + t]));
  }
  return out;
}

// Translator::translate_batch (src/translator.cc:8-28) on integer ids.
// src_ids: [batch, src_stride] raw token ids WITHOUT the implicit </s> (the reference adds
// source BOS/EOS itself per the model config). Outputs: out_ids [batch, num_hyp, max_len],
// out_lengths [batch, num_hyp], out_scores [batch, num_hyp] (if return_scores).
int ct2ref_translate(void* h, const int32_t* src_ids, const int32_t* src_lengths, int batch,
                     int src_stride, int beam_size, int max_len, int min_len,
                     float length_penalty, int num_hyp, int return_scores, int max_batch_size,
                     int32_t* out_ids, int32_t* out_lengths, float* out_scores) {
  CT2REF_TRY
  auto* m = static_cast<RefModel*>(h);
  const auto* s2s = dynamic_cast<const models::SequenceToSequenceModel*>(m->model.get());
  if (!s2s || !m->translator) throw std::runtime_error("not a sequence-to-sequence model");
  auto source = ids_to_tokens(s2s->get_source_vocabulary(), src_ids, src_lengths, batch, src_stride);
  TranslationOptions opt;
  opt.beam_size = beam_size;
  opt.max_decoding_length = max_len;
  opt.min_decoding_length = min_len;
  opt.length_penalty = length_penalty;
  opt.num_hypotheses = num_hyp;
  opt.return_scores = return_scores != 0;
  opt.max_input_length = 0;
  auto results = m->translator->translate_batch(source, opt, max_batch_size);
  const auto& tv = s2s->get_target_vocabulary();
  for (int b = 0; b < batch; ++b) {
    for (int k = 0; k < num_hyp; ++k) {
      int32_t* dst = out_ids + (static_cast<int64_t>(b) * num_hyp + k) * max_len;
      int len = 0;
      if (k < static_cast<int>(results[b].hypotheses.size())) {
        const auto& hyp = results[b].hypotheses[k];
        len = static_cast<int>(hyp.size());
        for (int t = 0; t < len && t < max_len; ++t) dst[t] = tv.to_id(hyp[t]);
        if (return_scores && out_scores) out_scores[b * num_hyp + k] = results[b].scores[k];
      }
      out_lengths[b * num_hyp + k] = len;
    }
  }
  CT2REF_CATCH
}

// Translator::translate_batch(source, target_prefix, options) on integer ids.
int ct2ref_translate_prefix(void* h, const int32_t* src_ids, const int32_t* src_lengths, int batch,
                            int src_stride, const int32_t* prefix_ids,
                            const int32_t* prefix_lengths, int prefix_stride, int beam_size,
                            int max_len, int min_len, float length_penalty, int return_scores,
                            int32_t* out_ids, int32_t* out_lengths, float* out_scores) {
  CT2REF_TRY
  auto* m = static_cast<RefModel*>(h);
  const auto* s2s = dynamic_cast<const models::SequenceToSequenceModel*>(m->model.get());
  if (!s2s || !m->translator) throw std::runtime_error("not a sequence-to-sequence model");
  auto source = ids_to_tokens(s2s->get_source_vocabulary(), src_ids, src_lengths, batch, src_stride);
  auto prefix = ids_to_tokens(s2s->get_target_vocabulary(), prefix_ids, prefix_lengths, batch,
                              prefix_stride);
  TranslationOptions opt;
  opt.beam_size = beam_size;
  opt.max_decoding_length = max_len;
  opt.min_decoding_length = min_len;
  opt.length_penalty = length_penalty;
  opt.num_hypotheses = 1;
  opt.return_scores = return_scores != 0;
  opt.max_input_length = 0;
  auto results = m->translator->translate_batch(source, prefix, opt);
  const auto& tv = s2s->get_target_vocabulary();
  for (int b = 0; b < batch; ++b) {
    const auto& hyp = results[b].hypotheses[0];
    const int len = static_cast<int>(hyp.size());
    for (int t = 0; t < len && t < max_len; ++t)
      out_ids[static_cast<int64_t>(b) * max_len + t] = tv.to_id(hyp[t]);
    out_lengths[b] = len;
    if (return_scores && out_scores) out_scores[b] = results[b].scores[0];
  }
  CT2REF_CATCH
}

// Translator::translate_batch with the logits-processor options (TranslationOptions::
// repetition_penalty / no_repeat_ngram_size / suppress_sequences, include/ctranslate2/
// translation.h:25-31). suppress_ids: num_suppress sequences back to back, lengths in
// suppress_lengths (ids are mapped to the target vocabulary's tokens).
int ct2ref_translate_processors(void* h, const int32_t* src_ids, const int32_t* src_lengths,
                                int batch, int src_stride, int beam_size, int max_len, int min_len,
                                float length_penalty, int return_scores, float repetition_penalty,
                                int no_repeat_ngram_size, const int32_t* suppress_ids,
                                const int32_t* suppress_lengths, int num_suppress,
                                int32_t* out_ids, int32_t* out_lengths, float* out_scores) {
  CT2REF_TRY
  auto* m = static_cast<RefModel*>(h);
  const auto* s2s = dynamic_cast<const models::SequenceToSequenceModel*>(m->model.get());
  if (!s2s || !m->translator) throw std::runtime_error("not a sequence-to-sequence model");
  auto source = ids_to_tokens(s2s->get_source_vocabulary(), src_ids, src_lengths, batch, src_stride);
  const auto& tv = s2s->get_target_vocabulary();
  TranslationOptions opt;
  opt.beam_size = beam_size;
  opt.max_decoding_length = max_len;
  opt.min_decoding_length = min_len;
  opt.length_penalty = length_penalty;
  opt.num_hypotheses = 1;
  opt.return_scores = return_scores != 0;
  opt.max_input_length = 0;
  opt.repetition_penalty = repetition_penalty;
  opt.no_repeat_ngram_size = static_cast<size_t>(no_repeat_ngram_size);
  for (int s = 0; s < num_suppress; ++s) {
    std::vector<std::string> seq;
    for (int t = 0; t < suppress_lengths[s]; ++t) seq.push_back(tv.to_token(*suppress_ids++));
    opt.suppress_sequences.push_back(std::move(seq));
  }
  auto results = m->translator->translate_batch(source, opt);
  for (int b = 0; b < batch; ++b) {
    const auto& hyp = results[b].hypotheses[0];
    const int len = static_cast<int>(hyp.size());
    for (int t = 0; t < len && t < max_len; ++t)
      out_ids[static_cast<int64_t>(b) * max_len + t] = tv.to_id(hyp[t]);
    out_lengths[b] = len;
    if (return_scores && out_scores) out_scores[b] = results[b].scores[0];
  }
  CT2REF_CATCH
}

// Generator::generate_batch on integer ids (decoder-only models).
int ct2ref_generate(void* h, const int32_t* prompt_ids, const int32_t* prompt_lengths, int batch,
                    int stride, int beam_size, int max_len, int min_len, int return_scores,
                    int32_t* out_ids, int32_t* out_lengths, float* out_scores) {
  CT2REF_TRY
  auto* m = static_cast<RefModel*>(h);
  const auto* lm = dynamic_cast<const models::LanguageModel*>(m->model.get());
  if (!lm || !m->generator) throw std::runtime_error("not a language model");
  auto prompts = ids_to_tokens(lm->get_vocabulary(), prompt_ids, prompt_lengths, batch, stride);
  GenerationOptions opt;
  opt.beam_size = beam_size;
  opt.max_length = max_len;
  opt.min_length = min_len;
  opt.sampling_topk = 1;
  opt.return_scores = return_scores != 0;
  opt.include_prompt_in_result = false;
  auto futures = m->generator->generate_batch_async(prompts, opt);
  std::vector<GenerationResult> results;
  for (auto& f : futures) results.emplace_back(f.get());
  for (int b = 0; b < batch; ++b) {
    const auto& ids = results[b].sequences_ids[0];
    const int len = static_cast<int>(ids.size());
    for (int t = 0; t < len && t < max_len; ++t) out_ids[b * max_len + t] = ids[t];
    out_lengths[b] = len;
    if (return_scores && out_scores) out_scores[b] = results[b].scores[0];
  }
  CT2REF_CATCH
}

// Generator::generate_batch with include_prompt_in_result = true (the reference's default: the
// prompt is decoded step by step as a returned hard prefix, src/models/language_model.cc:150-282;
// the sequences begin with the prompt). out_ids [batch, out_stride], out_stride >= max_len + 1.
int ct2ref_generate_static(void* h, const int32_t* prompt_ids, const int32_t* prompt_lengths,
                           int batch, int stride, int static_len, int include_prompt, int beam_size, int max_len,
                           int min_len, int return_scores, int out_stride, int32_t* out_ids,
                           int32_t* out_lengths, float* out_scores);

int ct2ref_generate_with_prompt(void* h, const int32_t* prompt_ids, const int32_t* prompt_lengths,
                                int batch, int stride, int beam_size, int max_len, int min_len,
                                int return_scores, int out_stride, int32_t* out_ids,
                                int32_t* out_lengths, float* out_scores) {
  return ct2ref_generate_static(h, prompt_ids, prompt_lengths, batch, stride, 0, 1, beam_size, max_len, min_len,
                                return_scores, out_stride, out_ids, out_lengths, out_scores);
}

// ... and with GenerationOptions::static_prompt = the first static_len tokens of prompt row 0 (every
// row must begin with them; src/models/language_model.cc:184-206), include_prompt_in_result as given.
int ct2ref_generate_static(void* h, const int32_t* prompt_ids, const int32_t* prompt_lengths,
                           int batch, int stride, int static_len, int include_prompt, int beam_size, int max_len,
                           int min_len, int return_scores, int out_stride, int32_t* out_ids,
                           int32_t* out_lengths, float* out_scores) {
  CT2REF_TRY
  auto* m = static_cast<RefModel*>(h);
  const auto* lm = dynamic_cast<const models::LanguageModel*>(m->model.get());
  if (!lm || !m->generator) throw std::runtime_error("not a language model");
  auto prompts = ids_to_tokens(lm->get_vocabulary(), prompt_ids, prompt_lengths, batch, stride);
  GenerationOptions opt;
  opt.beam_size = beam_size;
  opt.max_length = max_len;
  opt.min_length = min_len;
  opt.sampling_topk = 1;
  opt.return_scores = return_scores != 0;
  opt.include_prompt_in_result = include_prompt != 0;
  if (static_len > 0) {
    opt.static_prompt.assign(prompts[0].begin(), prompts[0].begin() + static_len);
    opt.cache_static_prompt = false;
    for (auto& p : prompts) p.erase(p.begin(), p.begin() + static_len);
  }
  auto futures = m->generator->generate_batch_async(prompts, opt);
  std::vector<GenerationResult> results;
  for (auto& f : futures) results.emplace_back(f.get());
  for (int b = 0; b < batch; ++b) {
    const auto& ids = results[b].sequences_ids[0];
    const int len = static_cast<int>(ids.size());
    for (int t = 0; t < len && t < out_stride; ++t) out_ids[b * out_stride + t] = ids[t];
    out_lengths[b] = len;
    if (return_scores && out_scores) out_scores[b] = results[b].scores[0];
  }
  CT2REF_CATCH
}

// Generator::generate_batch with num_hypotheses / return_alternatives /
// min_alternative_expansion_prob (GenerationOptions, include/ctranslate2/generation.h;
// decode_alternatives, src/decoding.cc:1122-1290 - reached from DecoderReplica::run_generation,
// src/models/language_model.cc:241-245). out_ids [batch, num_hyp, max_len], out_lengths /
// out_scores [batch, num_hyp] (length -1: the reference returned fewer sequences).
int ct2ref_generate_ex(void* h, const int32_t* prompt_ids, const int32_t* prompt_lengths, int batch,
                       int stride, int beam_size, int max_len, int min_len, int num_hyp,
                       int return_alternatives, float min_alternative_expansion_prob,
                       int32_t* out_ids, int32_t* out_lengths, float* out_scores) {
  CT2REF_TRY
  auto* m = static_cast<RefModel*>(h);
  const auto* lm = dynamic_cast<const models::LanguageModel*>(m->model.get());
  if (!lm || !m->generator) throw std::runtime_error("not a language model");
  auto prompts = ids_to_tokens(lm->get_vocabulary(), prompt_ids, prompt_lengths, batch, stride);
  GenerationOptions opt;
  opt.beam_size = beam_size;
  opt.max_length = max_len;
  opt.min_length = min_len;
  opt.sampling_topk = 1;
  opt.num_hypotheses = num_hyp;
  opt.return_scores = true;
  opt.return_alternatives = return_alternatives != 0;
  opt.min_alternative_expansion_prob = min_alternative_expansion_prob;
  opt.include_prompt_in_result = false;
  auto futures = m->generator->generate_batch_async(prompts, opt);
  std::vector<GenerationResult> results;
  for (auto& f : futures) results.emplace_back(f.get());
  for (int b = 0; b < batch; ++b)
    for (int k = 0; k < num_hyp; ++k) {
      const int64_t slot = static_cast<int64_t>(b) * num_hyp + k;
      out_lengths[slot] = -1;
      if (k >= static_cast<int>(results[b].sequences_ids.size())) continue;
      const auto& ids = results[b].sequences_ids[k];
      const int len = static_cast<int>(ids.size());
      for (int t = 0; t < len && t < max_len; ++t) out_ids[slot * max_len + t] = static_cast<int32_t>(ids[t]);
      out_lengths[slot] = len;
      if (out_scores && k < static_cast<int>(results[b].scores.size())) out_scores[slot] = results[b].scores[k];
    }
  CT2REF_CATCH
}

// Generator::generate_batch with the logits-processor options (GenerationOptions::
// repetition_penalty / no_repeat_ngram_size / suppress_sequences, include/ctranslate2/generation.h).
int ct2ref_generate_processors(void* h, const int32_t* prompt_ids, const int32_t* prompt_lengths,
                               int batch, int stride, int beam_size, int max_len, int min_len,
                               int return_scores, float repetition_penalty,
                               int no_repeat_ngram_size, const int32_t* suppress_ids,
                               const int32_t* suppress_lengths, int num_suppress,
                               int32_t* out_ids, int32_t* out_lengths, float* out_scores) {
  CT2REF_TRY
  auto* m = static_cast<RefModel*>(h);
  const auto* lm = dynamic_cast<const models::LanguageModel*>(m->model.get());
  if (!lm || !m->generator) throw std::runtime_error("not a language model");
  const auto& vocab = lm->get_vocabulary();
  auto prompts = ids_to_tokens(vocab, prompt_ids, prompt_lengths, batch, stride);
  GenerationOptions opt;
  opt.beam_size = beam_size;
  opt.max_length = max_len;
  opt.min_length = min_len;
  opt.sampling_topk = 1;
  opt.return_scores = return_scores != 0;
  opt.include_prompt_in_result = false;
  opt.repetition_penalty = repetition_penalty;
  opt.no_repeat_ngram_size = static_cast<size_t>(no_repeat_ngram_size);
  for (int s = 0; s < num_suppress; ++s) {
    std::vector<std::string> seq;
    for (int t = 0; t < suppress_lengths[s]; ++t) seq.push_back(vocab.to_token(*suppress_ids++));
    opt.suppress_sequences.push_back(std::move(seq));
  }
  auto futures = m->generator->generate_batch_async(prompts, opt);
  std::vector<GenerationResult> results;
  for (auto& f : futures) results.emplace_back(f.get());
  for (int b = 0; b < batch; ++b) {
    const auto& ids = results[b].sequences_ids[0];
    const int len = static_cast<int>(ids.size());
    for (int t = 0; t < len && t < max_len; ++t) out_ids[b * max_len + t] = ids[t];
    out_lengths[b] = len;
    if (return_scores && out_scores) out_scores[b] = results[b].scores[0];
  }
  CT2REF_CATCH
}

// Translator::translate_batch with every decoding option the fused engine's "remaining decoding"
// row covers (include/ctranslate2/translation.h): target prefix, return_alternatives +
// min_alternative_expansion_prob (decode_alternatives, src/decoding.cc:1122-1290), biased decoding
// (prefix_bias_beta, BiasedDecoder :256-343), return_attention, coverage_penalty.
// out_ids [batch, num_hyp, max_len], out_lengths / out_scores [batch, num_hyp] (length -1: the
// reference returned fewer hypotheses), out_attention [batch, num_hyp, max_len, att_stride] or NULL.
int ct2ref_translate_ex(void* h, const int32_t* src_ids, const int32_t* src_lengths, int batch,
                        int src_stride, const int32_t* prefix_ids, const int32_t* prefix_lengths,
                        int prefix_stride, int beam_size, int max_len, int min_len,
                        float length_penalty, float coverage_penalty, int num_hyp,
                        int return_alternatives, float min_alternative_expansion_prob,
                        float prefix_bias_beta, int32_t* out_ids, int32_t* out_lengths,
                        float* out_scores, float* out_attention, int att_stride) {
  CT2REF_TRY
  auto* m = static_cast<RefModel*>(h);
  const auto* s2s = dynamic_cast<const models::SequenceToSequenceModel*>(m->model.get());
  if (!s2s || !m->translator) throw std::runtime_error("not a sequence-to-sequence model");
  auto source = ids_to_tokens(s2s->get_source_vocabulary(), src_ids, src_lengths, batch, src_stride);
  std::vector<std::vector<std::string>> prefix;
  if (prefix_ids)
    prefix = ids_to_tokens(s2s->get_target_vocabulary(), prefix_ids, prefix_lengths, batch,
                           prefix_stride);
  TranslationOptions opt;
  opt.beam_size = beam_size;
  opt.max_decoding_length = max_len;
  opt.min_decoding_length = min_len;
  opt.length_penalty = length_penalty;
  opt.coverage_penalty = coverage_penalty;
  opt.num_hypotheses = num_hyp;
  opt.return_scores = true;
  opt.return_attention = out_attention != nullptr;
  opt.return_alternatives = return_alternatives != 0;
  opt.min_alternative_expansion_prob = min_alternative_expansion_prob;
  opt.prefix_bias_beta = prefix_bias_beta;
  opt.max_input_length = 0;
  auto results = m->translator->translate_batch(source, prefix, opt);
  const auto& tv = s2s->get_target_vocabulary();
  for (int b = 0; b < batch; ++b) {
    for (int k = 0; k < num_hyp; ++k) {
      const int64_t slot = static_cast<int64_t>(b) * num_hyp + k;
      if (k >= static_cast<int>(results[b].hypotheses.size())) {
        out_lengths[slot] = -1;
        continue;
      }
      const auto& hyp = results[b].hypotheses[k];
      const int len = static_cast<int>(hyp.size());
      for (int t = 0; t < len && t < max_len; ++t) out_ids[slot * max_len + t] = tv.to_id(hyp[t]);
      out_lengths[slot] = len;
      out_scores[slot] = k < static_cast<int>(results[b].scores.size()) ? results[b].scores[k] : 0.f;
      if (out_attention && k < static_cast<int>(results[b].attention.size())) {
        const auto& att = results[b].attention[k];
        for (int t = 0; t < static_cast<int>(att.size()) && t < max_len; ++t)
          for (int j = 0; j < static_cast<int>(att[t].size()) && j < att_stride; ++j)
            out_attention[(slot * max_len + t) * att_stride + j] = att[t][j];
      }
    }
  }
  CT2REF_CATCH
}

// Generator::score_batch (include/ctranslate2/generator.h:21-29; models::DecoderReplica::
// run_scoring -> score_sequences, src/scoring.cc): log-probability of tokens[1:] given the tokens
// before them. out_scores [batch, stride]; out_lengths[b] = scored tokens (0: skipped, < 2 tokens).
int ct2ref_generator_score(void* h, const int32_t* ids, const int32_t* lengths, int batch,
                           int stride, int max_input_length, float* out_scores,
                           int32_t* out_lengths) {
  CT2REF_TRY
  auto* m = static_cast<RefModel*>(h);
  const auto* lm = dynamic_cast<const models::LanguageModel*>(m->model.get());
  if (!lm || !m->generator) throw std::runtime_error("not a language model");
  auto tokens = ids_to_tokens(lm->get_vocabulary(), ids, lengths, batch, stride);
  ScoringOptions opt;
  opt.max_input_length = max_input_length;
  auto futures = m->generator->score_batch_async(tokens, opt);
  for (int b = 0; b < batch; ++b) {
    const ScoringResult r = futures[b].get();
    out_lengths[b] = static_cast<int32_t>(r.tokens_score.size());
    for (size_t t = 0; t < r.tokens_score.size() && t < static_cast<size_t>(stride); ++t)
      out_scores[static_cast<int64_t>(b) * stride + t] = r.tokens_score[t];
  }
  CT2REF_CATCH
}

// Generator::forward_batch (include/ctranslate2/generator.h:31-35; SequenceGeneratorReplica::
// forward, src/models/language_model.cc:66-100): logits (or log-probabilities) of every position,
// out [batch, max(lengths), vocab]; rows are padded to the longest sequence.
int ct2ref_generator_forward(void* h, const int32_t* ids, const int32_t* lengths, int batch,
                             int stride, int return_log_probs, float* out, int64_t* vocab_out) {
  CT2REF_TRY
  auto* m = static_cast<RefModel*>(h);
  const auto* lm = dynamic_cast<const models::LanguageModel*>(m->model.get());
  if (!lm || !m->generator) throw std::runtime_error("not a language model");
  std::vector<std::vector<size_t>> seqs(batch);
  for (int b = 0; b < batch; ++b)
    seqs[b].assign(ids + b * stride, ids + b * stride + lengths[b]);
  StorageView logits = m->generator->forward_batch_async(seqs, return_log_probs != 0).get();
  StorageView lf = logits.to_float32();
  if (vocab_out) *vocab_out = lf.dim(-1);
  std::memcpy(out, lf.data<float>(), sizeof(float) * lf.size());
  CT2REF_CATCH
}

// Teacher-forced decoding through layers::TransformerEncoder / TransformerDecoder
// (src/layers/transformer.cc:427-471, 621-871): runs the encoder on `src_ids`
// ([batch, src_stride] INCLUDING any special tokens, lengths in src_lengths; pass
// batch==0 sources for decoder-only models), then feeds tgt_in[:, t] at step t and writes the
// step logits to logits_out [T, batch, vocab]. This is the "logits within 1e-3" oracle.
int ct2ref_step_logits(void* h, const int32_t* src_ids, const int32_t* src_lengths, int batch,
                       int src_stride, const int32_t* tgt_in, int T, float* logits_out,
                       int64_t* vocab_out) {
  CT2REF_TRY
  auto* m = static_cast<RefModel*>(h);
  if (m->intra_threads > 0) set_num_threads(m->intra_threads);
  const auto& model = *m->model;
  const bool s2s = dynamic_cast<const models::SequenceToSequenceModel*>(m->model.get()) != nullptr;
  layers::TransformerDecoder decoder(model, "decoder");
  layers::DecoderState state = decoder.initial_state();
  if (s2s) {
    layers::TransformerEncoder encoder(model, "encoder");
    std::vector<std::vector<size_t>> ids(batch);
    for (int b = 0; b < batch; ++b)
      ids[b].assign(src_ids + b * src_stride, src_ids + b * src_stride + src_lengths[b]);
    StorageView lengths(DataType::INT32);
    std::vector<StorageView> inputs;
    inputs.emplace_back(layers::make_sequence_inputs(ids, Device::CPU, 1, &lengths));
    StorageView memory(encoder.output_type(), Device::CPU);
    encoder(inputs, &lengths, memory);
    state.emplace("memory", std::move(memory));
    state.emplace("memory_lengths", std::move(lengths));
  }
  int64_t vocab = 0;
  for (int t = 0; t < T; ++t) {
    StorageView step_ids({batch}, DataType::INT32);
    for (int b = 0; b < batch; ++b) step_ids.at<int32_t>(b) = tgt_in[b * T + t];
    StorageView logits(decoder.output_type(), Device::CPU);
    decoder(t, step_ids, state, &logits);
    vocab = logits.dim(-1);
    StorageView lf = logits.to_float32();
    std::memcpy(logits_out + static_cast<int64_t>(t) * batch * vocab, lf.data<float>(),
                sizeof(float) * batch * vocab);
  }
  if (vocab_out) *vocab_out = vocab;
  CT2REF_CATCH
}

// The reference's sinusoidal position table (generate_sinusoidal_position_encoding,
// src/layers/common.cc:204-229: logf / expf, then ops::Sin / ops::Cos = MKL VML vsSin / vsCos on
// this build): out [positions, depth]. Obtained by adding the encodings to zeros.
int ct2ref_sinusoidal_table(int64_t depth, int64_t positions, float* out) {
  CT2REF_TRY
  layers::SinusoidalPositionEncoder enc(depth);
  StorageView x({1, positions, depth}, 0.f);
  enc(x);
  copy_out(x, out);
  CT2REF_CATCH
}

// The reference's rotary tables (RotaryEmbeddings::initialize, src/layers/attention_layer.cc:
// 254-330: inv_freq with None / Linear / Llama3 scaling, MatMul, ops::Sin / ops::Cos): sin_out,
// cos_out [positions, dim]. The tables are private; they are read back by applying the layer to
// unit patterns: ops::Rotary computes y[i] = x[i] * c[i] + rot(x)[i] * s[i]
// (src/ops/rotary_cpu.cc:30-35), so x = 1 on the first half (even slots when interleaved) and 0
// on the other returns c on the ones and s on the zeros, and the complementary pattern returns
// the rest (-s on the zeros) - all exactly (products with 0 and 1, one addend +-0).
int ct2ref_rotary_tables(int64_t dim, int interleave, int scaling_type, float scaling_factor,
                         float base, int64_t num_initial_positions, float low_freq_factor,
                         float high_freq_factor, int64_t original_max_position_embeddings,
                         int64_t positions, float* sin_out, float* cos_out) {
  CT2REF_TRY
  layers::RotaryEmbeddings rot(dim, interleave != 0,
                               static_cast<layers::RotaryScalingType>(scaling_type), scaling_factor,
                               base, num_initial_positions, nullptr, nullptr, low_freq_factor,
                               high_freq_factor, original_max_position_embeddings, 0, true);
  const int64_t half = dim / 2;
  for (int pass = 0; pass < 2; ++pass) {
    std::vector<float> pattern(static_cast<size_t>(positions * dim));
    for (int64_t p = 0; p < positions; ++p)
      for (int64_t i = 0; i < dim; ++i) {
        const bool first = interleave ? (i % 2 == 0) : (i < half);
        pattern[p * dim + i] = (first == (pass == 0)) ? 1.f : 0.f;
      }
    StorageView x({1, 1, positions, dim}, pattern);
    rot.apply(x, 0, false);
    const float* y = x.data<float>();
    for (int64_t p = 0; p < positions; ++p)
      for (int64_t i = 0; i < dim; ++i) {
        const bool first = interleave ? (i % 2 == 0) : (i < half);
        const bool one = first == (pass == 0);
        if (one) {
          cos_out[p * dim + i] = y[p * dim + i];
        } else {
          // zero slot: y = rot(x) * s with rot(x) = +1 on the second kind of slot, -1 on the first
          const float v = y[p * dim + i];
          sin_out[p * dim + i] = first ? -v : v;
        }
      }
  }
  CT2REF_CATCH
}

// ---------------------------------------------------------------------------------------
// Whisper (include/ctranslate2/models/whisper.h): the reference's own models::Whisper pool
// (encode / generate / detect_language / align, src/models/whisper.cc:80-630) on float features
// [batch, n_mels, frames]. With device "cuda" in the Level-1 build the same calls run on the
// MI355X kernels.
// ---------------------------------------------------------------------------------------
struct RefWhisper {
  std::unique_ptr<models::Whisper> pool;
};

// Field for field models::WhisperOptions (whisper.h:11-58); return_logits_vocab is not exposed.
struct ct2ref_whisper_options {
  int32_t beam_size;
  float patience;
  float length_penalty;
  float repetition_penalty;
  int32_t no_repeat_ngram_size;
  int32_t max_length;
  int32_t sampling_topk;
  float sampling_temperature;
  int32_t num_hypotheses;
  int32_t return_scores;
  int32_t return_no_speech_prob;
  int32_t max_initial_timestamp_index;
  int32_t suppress_blank;
  int32_t num_suppress_tokens;
  const int32_t* suppress_tokens;
};

void* ct2ref_whisper_load(const char* path, const char* device, const char* compute_type,
                          int intra_threads) {
  try {
    models::ModelLoader loader(path);
    loader.device = str_to_device(device);
    loader.device_indices = {0};
    loader.num_replicas_per_device = 1;
    loader.compute_type = str_to_compute_type(compute_type);
    ReplicaPoolConfig config;
    config.num_threads_per_replica = intra_threads;
    auto handle = std::make_unique<RefWhisper>();
    handle->pool = std::make_unique<models::Whisper>(loader, config);
    return handle.release();
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return nullptr;
  }
}

void ct2ref_whisper_free(void* h) { delete static_cast<RefWhisper*>(h); }

int ct2ref_whisper_info(void* h, int32_t* is_multilingual, int32_t* n_mels, int32_t* num_languages) {
  CT2REF_TRY
  auto* w = static_cast<RefWhisper*>(h);
  *is_multilingual = w->pool->is_multilingual() ? 1 : 0;
  *n_mels = static_cast<int32_t>(w->pool->n_mels());
  *num_languages = static_cast<int32_t>(w->pool->num_languages());
  CT2REF_CATCH
}

static StorageView whisper_features(const float* features, int batch, int n_mels, int frames) {
  StorageView f({batch, n_mels, frames}, DataType::FLOAT32, Device::CPU);
  std::memcpy(f.data<float>(), features, sizeof(float) * batch * n_mels * frames);
  return f;
}

// Whisper::encode (whisper.cc:80-106): out [batch, frames / 2, d_model] as float32;
// out_dims receives the three dimensions.
int ct2ref_whisper_encode(void* h, const float* features, int batch, int n_mels, int frames,
                          float* out, int64_t out_capacity, int64_t* out_dims) {
  CT2REF_TRY
  auto* w = static_cast<RefWhisper*>(h);
  StorageView y = w->pool->encode(whisper_features(features, batch, n_mels, frames), true).get();
  if (y.dtype() != DataType::FLOAT32) y = y.to_float32();
  for (int i = 0; i < 3; ++i) out_dims[i] = y.dim(i);
  if (y.size() > out_capacity) throw std::runtime_error("ct2ref_whisper_encode: output too small");
  copy_out<float>(y, out);
  CT2REF_CATCH
}

// Whisper::generate (whisper.cc:232-371) on id prompts [batch, prompt_stride].
// out_ids [batch, num_hypotheses, max_len], out_lengths / out_scores [batch, num_hypotheses],
// out_no_speech [batch].
int ct2ref_whisper_generate(void* h, const float* features, int batch, int n_mels, int frames,
                            const int32_t* prompts, const int32_t* prompt_lengths,
                            int prompt_stride, const ct2ref_whisper_options* o, int max_len,
                            int32_t* out_ids, int32_t* out_lengths, float* out_scores,
                            float* out_no_speech) {
  CT2REF_TRY
  auto* w = static_cast<RefWhisper*>(h);
  models::WhisperOptions opt;
  opt.beam_size = o->beam_size;
  opt.patience = o->patience;
  opt.length_penalty = o->length_penalty;
  opt.repetition_penalty = o->repetition_penalty;
  opt.no_repeat_ngram_size = o->no_repeat_ngram_size;
  opt.max_length = o->max_length;
  opt.sampling_topk = o->sampling_topk;
  opt.sampling_temperature = o->sampling_temperature;
  opt.num_hypotheses = o->num_hypotheses;
  opt.return_scores = o->return_scores != 0;
  opt.return_no_speech_prob = o->return_no_speech_prob != 0;
  opt.max_initial_timestamp_index = o->max_initial_timestamp_index;
  opt.suppress_blank = o->suppress_blank != 0;
  opt.suppress_tokens.assign(o->suppress_tokens, o->suppress_tokens + o->num_suppress_tokens);
  std::vector<std::vector<size_t>> p(batch);
  for (int b = 0; b < batch; ++b)
    for (int i = 0; i < prompt_lengths[b]; ++i) p[b].push_back(prompts[b * prompt_stride + i]);
  auto futures = w->pool->generate(whisper_features(features, batch, n_mels, frames), std::move(p), opt);
  const int nh = o->num_hypotheses;
  for (int b = 0; b < batch; ++b) {
    const models::WhisperGenerationResult r = futures[b].get();
    for (int k = 0; k < nh; ++k) {
      out_lengths[b * nh + k] = -1;
      if (k >= static_cast<int>(r.sequences_ids.size())) continue;
      const auto& ids = r.sequences_ids[k];
      out_lengths[b * nh + k] = static_cast<int32_t>(ids.size());
      for (size_t i = 0; i < ids.size() && static_cast<int>(i) < max_len; ++i)
        out_ids[(static_cast<int64_t>(b) * nh + k) * max_len + i] = static_cast<int32_t>(ids[i]);
      if (out_scores && k < static_cast<int>(r.scores.size())) out_scores[b * nh + k] = r.scores[k];
    }
    if (out_no_speech) out_no_speech[b] = r.no_speech_prob;
  }
  CT2REF_CATCH
}

// Whisper::detect_language (whisper.cc:562-630): per batch entry the language token ids sorted by
// decreasing probability. out_ids / out_probs [batch, num_languages].
int ct2ref_whisper_detect_language(void* h, const float* features, int batch, int n_mels,
                                   int frames, int num_languages, int32_t* out_ids,
                                   float* out_probs) {
  CT2REF_TRY
  auto* w = static_cast<RefWhisper*>(h);
  auto futures = w->pool->detect_language(whisper_features(features, batch, n_mels, frames));
  const auto& vocab = dynamic_cast<const models::WhisperModel&>(*w->pool->get_first_replica().model())
                          .get_vocabulary();
  for (int b = 0; b < batch; ++b) {
    const auto r = futures[b].get();
    if (static_cast<int>(r.size()) != num_languages)
      throw std::runtime_error("ct2ref_whisper_detect_language: num_languages mismatch");
    for (int j = 0; j < num_languages; ++j) {
      out_ids[b * num_languages + j] = static_cast<int32_t>(vocab.to_id(r[j].first));
      out_probs[b * num_languages + j] = r[j].second;
    }
  }
  CT2REF_CATCH
}

// Whisper::align (whisper.cc:425-560). text_tokens [batch, text_stride]; out_pairs
// [batch, pair_capacity, 2] = (text index, time index), out_num_pairs [batch], out_token_probs
// [batch, text_stride].
int ct2ref_whisper_align(void* h, const float* features, int batch, int n_mels, int frames,
                         const int32_t* start_sequence, int num_start, const int32_t* text_tokens,
                         const int32_t* text_lengths, int text_stride, const int32_t* num_frames,
                         int median_filter_width, int pair_capacity, int32_t* out_pairs,
                         int32_t* out_num_pairs, float* out_token_probs) {
  CT2REF_TRY
  auto* w = static_cast<RefWhisper*>(h);
  std::vector<size_t> start(start_sequence, start_sequence + num_start);
  std::vector<std::vector<size_t>> text(batch);
  std::vector<size_t> nf(batch);
  for (int b = 0; b < batch; ++b) {
    for (int i = 0; i < text_lengths[b]; ++i) text[b].push_back(text_tokens[b * text_stride + i]);
    nf[b] = num_frames[b];
  }
  auto futures = w->pool->align(whisper_features(features, batch, n_mels, frames), start, text, nf,
                                median_filter_width);
  for (int b = 0; b < batch; ++b) {
    const models::WhisperAlignmentResult r = futures[b].get();
    out_num_pairs[b] = static_cast<int32_t>(r.alignments.size());
    if (static_cast<int>(r.alignments.size()) > pair_capacity)
      throw std::runtime_error("ct2ref_whisper_align: pair capacity too small");
    for (size_t i = 0; i < r.alignments.size(); ++i) {
      out_pairs[(static_cast<int64_t>(b) * pair_capacity + i) * 2] = static_cast<int32_t>(r.alignments[i].first);
      out_pairs[(static_cast<int64_t>(b) * pair_capacity + i) * 2 + 1] = static_cast<int32_t>(r.alignments[i].second);
    }
    for (size_t i = 0; i < r.text_token_probs.size(); ++i)
      out_token_probs[b * text_stride + i] = r.text_token_probs[i];
  }
  CT2REF_CATCH
}

// ops::Conv1D on Device::CPU (src/ops/conv1d_cpu.cc:128-207: im2col + Gemm; with qscale the int8
// path - im2col rows quantised, s8 x s8 -> s32, Dequantize, then bias + activation).
// x [batch, in_c, length] float, w [out_c, in_c, kernel] float or (w_scale != NULL) int8,
// y [batch, out_c, out_len]. act < 0: none.
int ct2ref_conv1d(const float* x, const void* w, const float* w_scale, const float* bias, int64_t batch,
                  int64_t in_c, int64_t length, int64_t out_c, int64_t kernel, int64_t stride,
                  int64_t padding, int act, float* y, int64_t* out_len) {
  CT2REF_TRY
  const ops::ActivationType type = static_cast<ops::ActivationType>(act);
  const ops::Conv1D op(stride, padding, 1, 1, act >= 0 ? &type : nullptr);
  StorageView input = view(x, {batch, in_c, length});
  StorageView output(DataType::FLOAT32, Device::CPU);
  StorageView b = bias ? view(bias, {out_c}) : StorageView();
  if (w_scale) {
    StorageView weight = view(static_cast<const int8_t*>(w), {out_c, in_c, kernel});
    StorageView qscale = view(w_scale, {out_c});
    if (bias) op(input, weight, b, output, &qscale);
    else op(input, weight, output, &qscale);
  } else {
    StorageView weight = view(static_cast<const float*>(w), {out_c, in_c, kernel});
    if (bias) op(input, weight, b, output, nullptr);
    else op(input, weight, output, nullptr);
  }
  *out_len = output.dim(2);
  copy_out<float>(output, y);
  CT2REF_CATCH
}

// ops::LayerNorm(axis, epsilon) without gamma / beta on [outer, axis_size, inner]
// (compute_alignments, src/models/whisper.cc:396).
int ct2ref_layer_norm_axis(const float* x, int64_t outer, int64_t axis_size, int64_t inner, float eps,
                           float* y) {
  CT2REF_TRY
  const StorageView input = view(x, {outer, axis_size, inner});
  StorageView output(DataType::FLOAT32, Device::CPU);
  ops::LayerNorm(-2, eps)(input, output);
  copy_out<float>(output, y);
  CT2REF_CATCH
}

}  // extern "C"
