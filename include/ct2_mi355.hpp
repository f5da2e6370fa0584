// C++ host API of the MI355X backend: the surface of ctranslate2::Translator / ctranslate2::Generator
// (include/ctranslate2/translator.h:26-60, generator.h:10-30, translation.h:11-135,
// generation.h:12-110) restated over the C ABI of ct2_mi355.h, header only. An application
// written against the reference replaces
//     ctranslate2::Translator t(path, Device::CUDA, ComputeType::INT8, {0});
// by
//     ct2_mi355::Translator   t(path, Device::CUDA, ComputeType::INT8, {0});
// and keeps its calls: option and result members carry the reference's names, defaults and
// meaning, errors are std::invalid_argument / std::runtime_error as in the reference. Options the
// backend does not implement are present (so code compiles unchanged) and rejected with
// std::invalid_argument when they are set to a non-default value - nothing is silently ignored.
// There is no CPU arm: Device::CPU throws.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <exception>
#include <functional>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "ct2_mi355.h"

namespace ct2_mi355 {

// include/ctranslate2/devices.h:9-12 - Device::CUDA is the (HIP) GPU, as in the reference's own
// ROCm build.
enum class Device { CPU, CUDA };

// include/ctranslate2/types.h:26-37 (the subset with a GPU arm here)
enum class ComputeType { DEFAULT, AUTO, FLOAT32, INT8, INT8_FLOAT32, INT8_FLOAT16, INT8_BFLOAT16, FLOAT16, BFLOAT16 };

inline const char* compute_type_name(ComputeType t) {
  switch (t) {
    case ComputeType::DEFAULT: return "default";
    case ComputeType::AUTO: return "auto";
    case ComputeType::FLOAT32: return "float32";
    case ComputeType::INT8: return "int8";
    case ComputeType::INT8_FLOAT32: return "int8_float32";
    case ComputeType::INT8_FLOAT16: return "int8_float16";
    case ComputeType::INT8_BFLOAT16: return "int8_bfloat16";
    case ComputeType::FLOAT16: return "float16";
    case ComputeType::BFLOAT16: return "bfloat16";
  }
  return "default";
}

// generation.h:93-118: what TranslationOptions / GenerationOptions::callback receives for every
// generated token (the per-step `logits` member is absent: return_logits_vocab delivers them with the result)
struct GenerationStepResult {
  size_t step = 0;
  size_t batch_id = 0;
  size_t token_id = 0;
  size_t hypothesis_id = 0;
  std::string token;
  std::optional<float> score;
  bool is_last = false;
};

// translation.h:11-96
struct TranslationOptions {
  size_t beam_size = 2;
  float patience = 1;
  float length_penalty = 1;
  float coverage_penalty = 0;
  float repetition_penalty = 1;
  size_t no_repeat_ngram_size = 0;
  bool disable_unk = false;
  std::vector<std::vector<std::string>> suppress_sequences;
  float prefix_bias_beta = 0;            // biased decoding towards target_prefix (beam search)
  std::string end_token;                 // empty = the model's end token (single token form)
  bool return_end_token = false;
  size_t max_input_length = 1024;
  size_t max_decoding_length = 256;
  size_t min_decoding_length = 1;
  size_t sampling_topk = 1;              // 0 = the whole vocabulary
  float sampling_topp = 1;
  float sampling_temperature = 1;
  size_t num_hypotheses = 1;
  bool return_scores = false;
  bool return_attention = false;
  bool return_logits_vocab = false;      // beam_size 1 (greedy search, random sampling)
  bool return_alternatives = false;
  float min_alternative_expansion_prob = 0;
  bool replace_unknowns = false;         // unknown target tokens -> the source token with the highest attention
  // called for every generated token when beam_size is 1; returning true stops that batch entry
  std::function<bool(GenerationStepResult)> callback = nullptr;
  // backend additions
  unsigned int seed = 0;                 // random sampling is reproducible for a seed
  bool use_graph = true;                 // replay decode steps from a hipGraph
};

// translation.h:98-135
struct TranslationResult {
  std::vector<std::vector<std::string>> hypotheses;
  std::vector<float> scores;
  std::vector<std::vector<std::vector<float>>> attention;  // [hypothesis][token][source position]
  // return_logits_vocab: [hypothesis][step][vocabulary] (the reference: a StorageView per step)
  std::vector<std::vector<std::vector<float>>> logits;
  bool has_attention() const { return !attention.empty(); }

  const std::vector<std::string>& output() const {
    if (hypotheses.empty()) throw std::runtime_error("This result is empty");
    return hypotheses[0];
  }
  float score() const {
    if (scores.empty()) throw std::runtime_error("This result has no scores");
    return scores[0];
  }
  size_t num_hypotheses() const { return hypotheses.size(); }
  bool has_scores() const { return !scores.empty(); }
};

// scoring.h:9-40
struct ScoringOptions {
  size_t max_input_length = 1024;
};

struct ScoringResult {
  std::vector<std::string> tokens;
  std::vector<float> tokens_score;
  float cumulated_score() const {
    float total = 0;
    for (const float s : tokens_score) total += s;
    return total;
  }
  float normalized_score() const {
    return tokens_score.empty() ? 0.f : cumulated_score() / static_cast<float>(tokens_score.size());
  }
};

// generation.h:12-76
struct GenerationOptions {
  size_t beam_size = 1;
  float patience = 1;
  float length_penalty = 1;
  float repetition_penalty = 1;
  size_t no_repeat_ngram_size = 0;
  bool disable_unk = false;
  std::vector<std::vector<std::string>> suppress_sequences;
  std::string end_token;                 // empty = the model's end token
  bool return_end_token = false;
  size_t max_length = 512;
  size_t min_length = 0;
  size_t sampling_topk = 1;
  float sampling_topp = 1;
  float sampling_temperature = 1;
  size_t num_hypotheses = 1;
  bool return_scores = false;
  bool return_logits_vocab = false;      // beam_size 1 (greedy search, random sampling)
  bool return_alternatives = false;      // round 5 (greedy / beam search; no callback, logits)
  float min_alternative_expansion_prob = 0;
  std::vector<std::string> static_prompt;
  bool cache_static_prompt = true;       // accepted; the prompt is simply forwarded
  bool include_prompt_in_result = true;
  // called for every generated token when beam_size is 1; returning true stops that batch entry
  std::function<bool(GenerationStepResult)> callback = nullptr;
  unsigned int seed = 0;
  bool use_graph = true;
};

// generation.h:78-110
struct GenerationResult {
  std::vector<std::vector<std::string>> sequences;
  std::vector<std::vector<size_t>> sequences_ids;
  std::vector<float> scores;
  // return_logits_vocab: [sequence][step][vocabulary], one vector per decoding step: the steps
  // that force what a longer prompt has beyond the shortest one come first, then the generated
  // tokens' steps (the C entry decodes with include_prompt_in_result = false)
  std::vector<std::vector<std::vector<float>>> logits;
  size_t num_sequences() const { return sequences.size(); }
  bool has_scores() const { return !scores.empty(); }
};

namespace detail {

inline void check(int rc) {
  if (rc != 0) throw std::runtime_error(ct2x_last_error());
}

inline void reject(bool set, const char* what) {
  if (set) throw std::invalid_argument(std::string(what) + " is not supported by the MI355X backend");
}

// Owns the C handle of one model on one device.
class Handle {
public:
  Handle(const std::string& model_path, Device device, ComputeType compute_type,
         const std::vector<int>& device_indices) {
    if (device != Device::CUDA)
      throw std::invalid_argument("the MI355X backend has no CPU arm: use Device::CUDA");
    if (device_indices.size() != 1)
      throw std::invalid_argument("one device index per Translator/Generator (one process per GPU; "
                                  "see ct2x_translator_create_tp for tensor parallelism)");
    h_ = ct2x_translator_create(model_path.c_str(), device_indices[0], compute_type_name(compute_type));
    if (!h_) throw std::runtime_error(ct2x_last_error());
    int32_t s2s = 0, vocab = 0, d = 0, layers = 0, heads = 0, bos = 0, eos = 0;
    char ct[32];
    check(ct2x_model_info(h_, &s2s, &vocab, &d, &layers, &heads, &bos, &eos, ct, sizeof(ct)));
    is_seq2seq_ = s2s != 0;
    vocab_size_ = static_cast<size_t>(vocab);
    bos_id_ = bos;
    compute_type_ = ct;
  }
  ~Handle() {
    if (h_) ct2x_translator_destroy(h_);
  }
  Handle(const Handle&) = delete;
  Handle& operator=(const Handle&) = delete;

  void* get() const { return h_; }
  bool is_seq2seq() const { return is_seq2seq_; }
  size_t vocab_size() const { return vocab_size_; }
  int32_t bos_id() const { return bos_id_; }
  const std::string& compute_type() const { return compute_type_; }

  // Vocabulary::to_id / to_token (include/ctranslate2/vocabulary.h:16-70); which: 0 source, 1 target
  int32_t to_id(int which, const std::string& token) const {
    return ct2x_vocab_id(h_, which, token.c_str());
  }
  // Vocabulary::to_id with allow_unk = false (src/vocabulary.cc): out-of-vocabulary is an error
  int32_t known_id(int which, const std::string& token, const char* what) const {
    const int32_t id = to_id(which, token);
    if (to_token(which, id) != token)
      throw std::invalid_argument(std::string(what) + " token '" + token + "' is not in the vocabulary");
    return id;
  }
  std::string to_token(int which, int32_t id) const {
    const char* t = ct2x_vocab_token(h_, which, id);
    if (!t) throw std::runtime_error(ct2x_last_error());
    return t;
  }

private:
  void* h_ = nullptr;
  bool is_seq2seq_ = false;
  size_t vocab_size_ = 0;
  int32_t bos_id_ = -1;
  std::string compute_type_;
};

// Ragged token batch -> padded id matrix.
struct IdBatch {
  std::vector<int32_t> ids, lengths;
  int32_t stride = 1;
};

inline IdBatch to_ids(const Handle& h, int which, const std::vector<std::vector<std::string>>& batch) {
  IdBatch out;
  size_t longest = 1;
  for (const auto& s : batch) longest = std::max(longest, s.size());
  out.stride = static_cast<int32_t>(longest);
  out.ids.assign(batch.size() * longest, 0);
  out.lengths.resize(batch.size());
  for (size_t b = 0; b < batch.size(); ++b) {
    out.lengths[b] = static_cast<int32_t>(batch[b].size());
    for (size_t t = 0; t < batch[b].size(); ++t) out.ids[b * longest + t] = h.to_id(which, batch[b][t]);
  }
  return out;
}

// suppress_sequences + disable_unk -> the id lists of ct2x_translation_options
struct Suppressed {
  std::vector<int32_t> ids, lengths;
};

inline Suppressed suppressed_ids(const Handle& h, int which,
                                 const std::vector<std::vector<std::string>>& sequences,
                                 bool disable_unk) {
  Suppressed s;
  for (const auto& seq : sequences) {
    if (seq.empty()) continue;
    for (const auto& token : seq) s.ids.push_back(h.known_id(which, token, "suppress_sequences"));
    s.lengths.push_back(static_cast<int32_t>(seq.size()));
  }
  if (disable_unk) {  // Vocabulary::unk_id(): the id an unknown token maps to
    s.ids.push_back(h.to_id(which, "<unk>"));
    s.lengths.push_back(1);
  }
  return s;
}

inline void set_processors(ct2x_translation_options& o, float repetition_penalty,
                           size_t no_repeat_ngram_size, const Suppressed& s) {
  o.repetition_penalty = repetition_penalty;
  o.no_repeat_ngram_size = static_cast<int32_t>(no_repeat_ngram_size);
  o.num_suppress_sequences = static_cast<int32_t>(s.lengths.size());
  o.suppress_ids = s.ids.empty() ? nullptr : s.ids.data();
  o.suppress_lengths = s.lengths.empty() ? nullptr : s.lengths.data();
}

inline int32_t sampling_topk_field(size_t topk) {
  // the C struct keeps 0 = arg-max for zero-initialised options; -1 = the whole vocabulary
  return topk == 0 ? -1 : static_cast<int32_t>(topk);
}

// Carries a std::function callback through the C ABI's (function pointer, user data) pair. An
// exception thrown by the callback must not unwind through the C frames: it is kept, every entry is
// stopped, and rethrow() raises it once the C call has returned.
class StepCallback {
public:
  StepCallback(const Handle& h, const std::function<bool(GenerationStepResult)>& fn, size_t step_offset)
      : h_(h), fn_(fn), step_offset_(step_offset) {}

  void attach(ct2x_translation_options& o) {
    if (!fn_) return;
    o.step_callback = &StepCallback::trampoline;
    o.step_callback_data = this;
  }
  void rethrow() {
    if (error_) std::rethrow_exception(error_);
  }

private:
  static int trampoline(void* user, int64_t step, int64_t batch_id, int64_t hypothesis_id,
                        int32_t token_id, int32_t has_score, float score, int32_t is_last) {
    auto* self = static_cast<StepCallback*>(user);
    if (self->error_) return 1;
    try {
      GenerationStepResult r;
      r.step = static_cast<size_t>(step) + self->step_offset_;
      r.batch_id = static_cast<size_t>(batch_id);
      r.hypothesis_id = static_cast<size_t>(hypothesis_id);
      r.token_id = static_cast<size_t>(token_id);
      r.token = self->h_.to_token(1, token_id);
      if (has_score) r.score = score;
      r.is_last = is_last != 0;
      return self->fn_(std::move(r)) ? 1 : 0;
    } catch (...) {
      self->error_ = std::current_exception();
      return 1;
    }
  }

  const Handle& h_;
  const std::function<bool(GenerationStepResult)>& fn_;
  size_t step_offset_;
  std::exception_ptr error_;
};

// return_logits_vocab: host buffers for the C entry and their conversion into the result member
struct LogitsSink {
  std::vector<float> data;
  std::vector<int32_t> counts;
  size_t hyp = 0, steps = 0, vocab = 0;

  void attach(ct2x_translation_options& o, const Handle& h, size_t batch, size_t hypotheses, size_t max_steps) {
    hyp = hypotheses;
    steps = max_steps;
    vocab = h.vocab_size();
    data.assign(batch * hyp * steps * vocab, 0.f);
    counts.assign(batch * hyp, 0);
    o.logits_out = data.data();
    o.logits_counts_out = counts.data();
    o.logits_steps = static_cast<int32_t>(steps);
  }
  std::vector<std::vector<float>> take(size_t b, size_t k) const {
    std::vector<std::vector<float>> out(static_cast<size_t>(counts[b * hyp + k]));
    for (size_t s = 0; s < out.size(); ++s) {
      const float* row = data.data() + ((b * hyp + k) * steps + s) * vocab;
      out[s].assign(row, row + vocab);
    }
    return out;
  }
};

}  // namespace detail

// ctranslate2::Translator (include/ctranslate2/translator.h:26-60)
class Translator {
public:
  Translator(const std::string& model_path, Device device = Device::CUDA,
             ComputeType compute_type = ComputeType::DEFAULT,
             const std::vector<int>& device_indices = {0})
      : h_(model_path, device, compute_type, device_indices) {
    if (!h_.is_seq2seq())
      throw std::invalid_argument("The model is not a sequence-to-sequence model: use Generator");
  }

  std::vector<TranslationResult>
  translate_batch(const std::vector<std::vector<std::string>>& source,
                  const TranslationOptions& options = TranslationOptions(),
                  size_t max_batch_size = 0) {
    return translate_batch(source, {}, options, max_batch_size);
  }

  std::vector<TranslationResult>
  translate_batch(const std::vector<std::vector<std::string>>& source,
                  const std::vector<std::vector<std::string>>& target_prefix,
                  const TranslationOptions& options = TranslationOptions(),
                  size_t max_batch_size = 0) {
    if (!target_prefix.empty() && target_prefix.size() != source.size())
      throw std::invalid_argument("Batch size mismatch: got " + std::to_string(source.size()) +
                                  " for source and " + std::to_string(target_prefix.size()) +
                                  " for target_prefix");
    const size_t batch = source.size();
    if (batch == 0) return {};

    ct2x_translation_options o;
    ct2x_translation_options_default(&o);
    o.beam_size = static_cast<int32_t>(options.beam_size);
    o.patience = options.patience;
    o.length_penalty = options.length_penalty;
    o.return_end_token = options.return_end_token;
    o.max_input_length = static_cast<int32_t>(options.max_input_length);
    o.max_decoding_length = static_cast<int32_t>(options.max_decoding_length);
    o.min_decoding_length = static_cast<int32_t>(options.min_decoding_length);
    o.sampling_topk = detail::sampling_topk_field(options.sampling_topk);
    o.sampling_topp = options.sampling_topp;
    o.sampling_temperature = options.sampling_temperature;
    o.num_hypotheses = static_cast<int32_t>(options.num_hypotheses);
    o.return_scores = options.return_scores;
    o.max_batch_size = static_cast<int32_t>(max_batch_size);
    o.seed = options.seed;
    o.use_graph = options.use_graph;
    const detail::Suppressed sup =
        detail::suppressed_ids(h_, 1, options.suppress_sequences, options.disable_unk);
    detail::set_processors(o, options.repetition_penalty, options.no_repeat_ngram_size, sup);
    if (!options.end_token.empty())
      o.end_token_plus_one = h_.known_id(1, options.end_token, "end_token") + 1;

    const detail::IdBatch src = detail::to_ids(h_, 0, source);
    const size_t hyp = options.num_hypotheses, len = options.max_decoding_length;
    std::vector<int32_t> out(batch * hyp * len, 0), out_len(batch * hyp, 0);
    std::vector<float> scores(batch * hyp, 0.f);
    o.return_alternatives = options.return_alternatives;
    o.min_alternative_expansion_prob = options.min_alternative_expansion_prob;
    o.prefix_bias_beta = options.prefix_bias_beta;
    o.coverage_penalty = options.coverage_penalty;
    std::vector<float> attention;
    // replace_unknowns decodes with the attention (src/models/sequence_to_sequence.cc:352)
    const bool want_attention = options.return_attention || options.replace_unknowns;
    if (want_attention) {
      attention.assign(batch * hyp * len * static_cast<size_t>(src.stride), 0.f);
      o.attention_out = attention.data();
      o.attention_stride = src.stride;
    }
    detail::StepCallback step_callback(h_, options.callback, 0);
    step_callback.attach(o);
    detail::LogitsSink logits;
    if (options.return_logits_vocab) logits.attach(o, h_, batch, hyp, len);
    if (target_prefix.empty()) {
      detail::check(ct2x_translate_batch(h_.get(), src.ids.data(), src.lengths.data(),
                                         static_cast<int32_t>(batch), src.stride, &o, out.data(),
                                         out_len.data(), scores.data()));
    } else {
      const detail::IdBatch pfx = detail::to_ids(h_, 1, target_prefix);
      detail::check(ct2x_translate_batch_prefix(h_.get(), src.ids.data(), src.lengths.data(),
                                                static_cast<int32_t>(batch), src.stride,
                                                pfx.ids.data(), pfx.lengths.data(), pfx.stride, &o,
                                                out.data(), out_len.data(), scores.data()));
    }
    step_callback.rethrow();
    std::vector<TranslationResult> results(batch);
    for (size_t b = 0; b < batch; ++b) {
      for (size_t k = 0; k < hyp; ++k) {
        const int32_t n = out_len[b * hyp + k];
        if (n < 0 || (k > 0 && n == 0 && !options.return_alternatives))
          continue;  // fewer finished hypotheses (or alternatives) than requested
        std::vector<std::string> tokens;
        for (int32_t t = 0; t < n; ++t) tokens.push_back(h_.to_token(1, out[(b * hyp + k) * len + t]));
        if (want_attention) {
          std::vector<std::vector<float>> att(static_cast<size_t>(n));
          for (int32_t t = 0; t < n; ++t) {
            const float* row = attention.data() + ((b * hyp + k) * len + t) * static_cast<size_t>(src.stride);
            att[t].assign(row, row + source[b].size());
          }
          if (options.replace_unknowns && !source[b].empty()) {
            // replace_unknown_tokens (src/models/sequence_to_sequence.cc:288-302): the source token
            // under the first maximum of the token's attention vector
            // (the id an absent token maps to IS the unknown token's, whatever the model calls it)
            const int32_t unk_id = h_.to_id(1, "<unk>");
            for (int32_t t = 0; t < n; ++t)
              if (out[(b * hyp + k) * len + t] == unk_id) {
                const auto& v = att[static_cast<size_t>(t)];
                size_t pos = 0;
                for (size_t j = 1; j < v.size(); ++j)
                  if (v[j] > v[pos]) pos = j;
                tokens[static_cast<size_t>(t)] = source[b][pos];
              }
          }
          if (options.return_attention) results[b].attention.push_back(std::move(att));
        }
        results[b].hypotheses.push_back(std::move(tokens));
        if (options.return_logits_vocab) results[b].logits.push_back(logits.take(b, k));
        if (options.return_scores) results[b].scores.push_back(scores[b * hyp + k]);
      }
    }
    return results;
  }

  // Translator::score_batch (include/ctranslate2/translator.h:62-76)
  std::vector<ScoringResult>
  score_batch(const std::vector<std::vector<std::string>>& source,
              const std::vector<std::vector<std::string>>& target,
              const ScoringOptions& options = ScoringOptions()) {
    if (source.size() != target.size())
      throw std::invalid_argument("Batch size mismatch: got " + std::to_string(source.size()) +
                                  " for source and " + std::to_string(target.size()) + " for target");
    const size_t batch = source.size();
    if (batch == 0) return {};
    const detail::IdBatch src = detail::to_ids(h_, 0, source);
    const detail::IdBatch tgt = detail::to_ids(h_, 1, target);
    const int32_t out_stride = tgt.stride + 1;
    std::vector<int32_t> out_ids(batch * out_stride, 0), out_len(batch, 0);
    std::vector<float> out_scores(batch * out_stride, 0.f);
    detail::check(ct2x_score_batch(h_.get(), src.ids.data(), src.lengths.data(),
                                   static_cast<int32_t>(batch), src.stride, tgt.ids.data(),
                                   tgt.lengths.data(), tgt.stride,
                                   static_cast<int32_t>(options.max_input_length), out_ids.data(),
                                   out_scores.data(), out_stride, out_len.data()));
    std::vector<ScoringResult> results(batch);
    for (size_t b = 0; b < batch; ++b)
      for (int32_t i = 0; i < out_len[b]; ++i) {
        results[b].tokens.push_back(h_.to_token(1, out_ids[b * out_stride + i]));
        results[b].tokens_score.push_back(out_scores[b * out_stride + i]);
      }
    return results;
  }

  // Frees the decoding session kept for reuse by the next call of the same shape (ct2_mi355.h,
  // "Session reuse"); it is rebuilt by the next call.
  void release_session() { detail::check(ct2x_release_session(h_.get())); }

  // ExecutionStats-like counters of the last call (include/ctranslate2/translator.h:9-13)
  ct2x_exec_stats last_stats() const {
    ct2x_exec_stats st;
    detail::check(ct2x_last_stats(h_.get(), &st));
    return st;
  }
  const std::string& compute_type() const { return h_.compute_type(); }

private:
  detail::Handle h_;
};

// ctranslate2::Generator (include/ctranslate2/generator.h:10-30)
class Generator {
public:
  Generator(const std::string& model_path, Device device = Device::CUDA,
            ComputeType compute_type = ComputeType::DEFAULT,
            const std::vector<int>& device_indices = {0})
      : h_(model_path, device, compute_type, device_indices) {
    if (h_.is_seq2seq())
      throw std::invalid_argument("The model is a sequence-to-sequence model: use Translator");
  }

  std::vector<GenerationResult>
  generate_batch(const std::vector<std::vector<std::string>>& start_tokens,
                 const GenerationOptions& options = GenerationOptions()) {
    const size_t batch = start_tokens.size();
    if (batch == 0) return {};
    // static_prompt + start_tokens is what the model consumes (language_model.cc:184-206)
    std::vector<std::vector<std::string>> prompts(batch);
    for (size_t b = 0; b < batch; ++b) {
      prompts[b] = options.static_prompt;
      prompts[b].insert(prompts[b].end(), start_tokens[b].begin(), start_tokens[b].end());
      if (prompts[b].empty()) throw std::invalid_argument("a prompt needs at least one token");
    }

    ct2x_translation_options o;
    ct2x_translation_options_default(&o);
    o.beam_size = static_cast<int32_t>(options.beam_size);
    o.patience = options.patience;
    o.length_penalty = options.length_penalty;
    o.return_end_token = options.return_end_token;
    o.max_decoding_length = static_cast<int32_t>(options.max_length);
    o.min_decoding_length = static_cast<int32_t>(options.min_length);
    o.sampling_topk = detail::sampling_topk_field(options.sampling_topk);
    o.sampling_topp = options.sampling_topp;
    o.sampling_temperature = options.sampling_temperature;
    o.num_hypotheses = static_cast<int32_t>(options.num_hypotheses);
    o.return_scores = options.return_scores;
    o.max_input_length = 0;
    o.seed = options.seed;
    o.use_graph = options.use_graph;
    // decode_alternatives behind run_generation (language_model.cc:241-245, decoding.cc:1122-1290)
    o.return_alternatives = options.return_alternatives;
    o.min_alternative_expansion_prob = options.min_alternative_expansion_prob;
    const detail::Suppressed sup =
        detail::suppressed_ids(h_, 1, options.suppress_sequences, options.disable_unk);
    detail::set_processors(o, options.repetition_penalty, options.no_repeat_ngram_size, sup);
    if (!options.end_token.empty())
      o.end_token_plus_one = h_.known_id(1, options.end_token, "end_token") + 1;

    const detail::IdBatch in = detail::to_ids(h_, 1, prompts);
    const size_t len = options.max_length, hyp = options.num_hypotheses;
    std::vector<int32_t> out(batch * hyp * len, 0), out_len(batch * hyp, 0);
    std::vector<float> scores(batch * hyp, 0.f);
    // include_prompt_in_result (the reference's default): the C entry decodes the prompt step by
    // step as a returned hard prefix, exactly as DecoderReplica::run_generation does
    // (src/models/language_model.cc:150-282) - lengths, scores and callbacks are the reference's.
    // Together with return_alternatives or return_logits_vocab the entry decodes
    // with include_prompt_in_result = false (one sequence-mode pass over the prompt, the first
    // generated token of the shortest prompt is step 0) and the prompt is put in front of the
    // sequences here: the start tokens are handed to the callback before the decoding and the steps
    // continue from them (a forced token's log-probability is not computed on that path:
    // return_scores is rejected).
    const bool exact_prompt = options.include_prompt_in_result && !options.return_alternatives &&
                              !options.return_logits_vocab;
    o.include_prompt_in_result = exact_prompt ? 1 : 0;
    o.static_prompt_length = static_cast<int32_t>(options.static_prompt.size());
    size_t step_offset = 0;
    if (options.callback && options.include_prompt_in_result && !exact_prompt) {
      detail::reject(options.return_scores, "a callback with return_scores and include_prompt_in_result");
      size_t shortest = start_tokens[0].size();
      for (const auto& s : start_tokens) shortest = std::min(shortest, s.size());
      if (shortest == 0) throw std::invalid_argument("a callback with include_prompt_in_result needs start tokens");
      step_offset = shortest - 1;
      for (size_t b = 0; b < batch; ++b)
        for (size_t t = 1; t < start_tokens[b].size(); ++t) {
          GenerationStepResult r;
          r.step = t - 1;
          r.batch_id = b;
          r.token = start_tokens[b][t];
          r.token_id = static_cast<size_t>(h_.to_id(1, r.token));
          options.callback(std::move(r));  // (a stop request during the prompt is not honoured)
        }
    }
    detail::StepCallback step_callback(h_, options.callback, step_offset);
    step_callback.attach(o);
    detail::LogitsSink logits;
    if (options.return_logits_vocab) {
      int32_t shortest = in.lengths[0], longest = in.lengths[0];
      for (const int32_t n : in.lengths) {
        shortest = std::min(shortest, n);
        longest = std::max(longest, n);
      }
      logits.attach(o, h_, batch, hyp, len + static_cast<size_t>(longest - shortest));
    }
    detail::check(ct2x_generate_batch(h_.get(), in.ids.data(), in.lengths.data(),
                                      static_cast<int32_t>(batch), in.stride, &o, out.data(),
                                      out_len.data(), scores.data()));
    step_callback.rethrow();
    std::vector<GenerationResult> results(batch);
    for (size_t b = 0; b < batch; ++b)
      for (size_t k = 0; k < hyp; ++k) {
        const int32_t n = out_len[b * hyp + k];
        if (options.return_alternatives ? n < 0 : (k > 0 && n == 0))
          continue;  // fewer alternatives expanded / fewer finished hypotheses than requested
        std::vector<std::string> tokens;
        std::vector<size_t> ids;
        if (exact_prompt) {
          // "Forward the start token to the output if it is not the special BOS token"
          // (language_model.cc:262-267); the rest of the prompt is in the decoded sequence
          const size_t first = b * in.stride + options.static_prompt.size();  // (behind the static prompt)
          if (in.ids[first] != h_.bos_id()) {
            ids.push_back(static_cast<size_t>(in.ids[first]));
            tokens.push_back(start_tokens[b][0]);
          }
        } else if (options.include_prompt_in_result) {
          // (alternatives: the sequences of a longer prompt of a ragged batch already begin with
          // its tokens beyond the shortest prompt - as the reference returns them)
          int32_t upto = in.lengths[b];
          if (options.return_alternatives)
            for (size_t i = 0; i < batch; ++i) upto = std::min(upto, in.lengths[i]);
          for (int32_t t = static_cast<int32_t>(options.static_prompt.size()); t < upto; ++t) {
            ids.push_back(static_cast<size_t>(in.ids[b * in.stride + t]));
            tokens.push_back(start_tokens[b][static_cast<size_t>(t) - options.static_prompt.size()]);
          }
        }
        for (int32_t t = 0; t < n; ++t) {
          const int32_t id = out[(b * hyp + k) * len + t];
          ids.push_back(static_cast<size_t>(id));
          tokens.push_back(h_.to_token(1, id));
        }
        results[b].sequences.push_back(std::move(tokens));
        results[b].sequences_ids.push_back(std::move(ids));
        if (options.return_logits_vocab) results[b].logits.push_back(logits.take(b, k));
        if (options.return_scores) results[b].scores.push_back(scores[b * hyp + k]);
      }
    return results;
  }

  // Generator::score_batch (include/ctranslate2/generator.h:21-25): log P(token k + 1 | tokens ..k);
  // sequences of fewer than 2 tokens give an empty result.
  std::vector<ScoringResult>
  score_batch(const std::vector<std::vector<std::string>>& tokens,
              const ScoringOptions& options = ScoringOptions()) {
    const size_t batch = tokens.size();
    if (batch == 0) return {};
    const detail::IdBatch in = detail::to_ids(h_, 1, tokens);
    std::vector<float> scores(batch * static_cast<size_t>(in.stride), 0.f);
    std::vector<int32_t> out_len(batch, 0);
    detail::check(ct2x_generator_score_batch(h_.get(), in.ids.data(), in.lengths.data(),
                                             static_cast<int32_t>(batch), in.stride,
                                             static_cast<int32_t>(options.max_input_length),
                                             scores.data(), out_len.data()));
    std::vector<ScoringResult> results(batch);
    for (size_t b = 0; b < batch; ++b)
      for (int32_t i = 0; i < out_len[b]; ++i) {
        // (the scored tokens are tokens[1:], or their truncation to max_input_length)
        results[b].tokens.push_back(tokens[b][static_cast<size_t>(i) + 1]);
        results[b].tokens_score.push_back(scores[b * static_cast<size_t>(in.stride) + i]);
      }
    return results;
  }

  // Generator::forward_batch (include/ctranslate2/generator.h:27-35): the logits (or
  // log-probabilities) of every position, flattened [batch, longest sequence, vocabulary].
  std::vector<float> forward_batch(const std::vector<std::vector<std::string>>& tokens,
                                   bool return_log_probs, size_t* time_out = nullptr,
                                   size_t* vocab_out = nullptr) {
    const size_t batch = tokens.size();
    if (batch == 0) return {};
    const detail::IdBatch in = detail::to_ids(h_, 1, tokens);
    size_t T = 0;
    for (size_t b = 0; b < batch; ++b) T = std::max<size_t>(T, static_cast<size_t>(in.lengths[b]));
    const size_t V = static_cast<size_t>(ct2x_vocab_size(h_.get(), 1));
    std::vector<float> out(batch * T * V);
    detail::check(ct2x_forward_batch(h_.get(), in.ids.data(), in.lengths.data(),
                                     static_cast<int32_t>(batch), in.stride, return_log_probs ? 1 : 0,
                                     out.data()));
    if (time_out) *time_out = T;
    if (vocab_out) *vocab_out = V;
    return out;
  }

  const std::string& compute_type() const { return h_.compute_type(); }

private:
  detail::Handle h_;
};

// ---- ctranslate2::models::Whisper (include/ctranslate2/models/whisper.h) ---------------------
namespace models {

// whisper.h:11-58 (return_logits_vocab is present and must stay false)
struct WhisperOptions {
  size_t beam_size = 5;
  float patience = 1;
  float length_penalty = 1;
  float repetition_penalty = 1;
  size_t no_repeat_ngram_size = 0;
  size_t max_length = 448;
  size_t sampling_topk = 1;
  float sampling_temperature = 1;
  size_t num_hypotheses = 1;
  bool return_scores = false;
  bool return_logits_vocab = false;  // not built
  bool return_no_speech_prob = false;
  size_t max_initial_timestamp_index = 50;
  bool suppress_blank = true;
  std::vector<int> suppress_tokens = {-1};
};

// whisper.h:60-74
struct WhisperGenerationResult {
  std::vector<std::vector<std::string>> sequences;
  std::vector<std::vector<size_t>> sequences_ids;
  std::vector<float> scores;
  float no_speech_prob = 0;
  size_t num_sequences() const { return sequences.size(); }
  bool has_scores() const { return !scores.empty(); }
};

// whisper.h:76-79
struct WhisperAlignmentResult {
  std::vector<std::pair<int64_t, int64_t>> alignments;
  std::vector<float> text_token_probs;
};

// Features are float32 [batch, n_mels, frames], row major (the reference's StorageView on CPU).
struct WhisperFeatures {
  const float* data = nullptr;
  size_t batch = 0, n_mels = 0, frames = 0;
};

// whisper.h:152-189: the synchronous form of the replica pool's methods (one device, no futures).
class Whisper {
public:
  Whisper(const std::string& model_path, Device device = Device::CUDA,
          ComputeType compute_type = ComputeType::DEFAULT, const std::vector<int>& device_indices = {0}) {
    if (device != Device::CUDA)
      throw std::invalid_argument("the MI355X backend has no CPU arm: use Device::CUDA");
    if (device_indices.size() != 1)
      throw std::invalid_argument("one device index per Whisper object (one process per GPU)");
    h_ = ct2x_whisper_create(model_path.c_str(), device_indices[0], compute_type_name(compute_type));
    if (!h_) throw std::runtime_error(ct2x_last_error());
    int32_t multi = 0, mels = 0, langs = 0;
    detail::check(ct2x_whisper_info(h_, &multi, &mels, &langs));
    is_multilingual_ = multi != 0;
    n_mels_ = static_cast<size_t>(mels);
    num_languages_ = static_cast<size_t>(langs);
  }
  ~Whisper() {
    if (h_) ct2x_whisper_destroy(h_);
  }
  Whisper(const Whisper&) = delete;
  Whisper& operator=(const Whisper&) = delete;

  bool is_multilingual() const { return is_multilingual_; }
  size_t n_mels() const { return n_mels_; }
  size_t num_languages() const { return num_languages_; }

  // Whisper::encode: [batch, (frames - 1) / 2 + 1, d_model] float32 on the host; dims receives the shape
  std::vector<float> encode(const WhisperFeatures& f, std::vector<int64_t>* dims = nullptr) {
    const int64_t d = ct2x_whisper_d_model(h_);
    std::vector<float> out(f.batch * ((f.frames + 1) / 2 + 1) * static_cast<size_t>(d));
    int64_t shape[3] = {0, 0, 0};
    detail::check(ct2x_whisper_encode(h_, f.data, static_cast<int32_t>(f.batch),
                                      static_cast<int32_t>(f.n_mels), static_cast<int32_t>(f.frames),
                                      out.data(), static_cast<int64_t>(out.size()), shape));
    out.resize(static_cast<size_t>(shape[0] * shape[1] * shape[2]));
    if (dims) dims->assign(shape, shape + 3);
    return out;
  }

  std::vector<WhisperGenerationResult> generate(const WhisperFeatures& f,
                                                const std::vector<std::vector<size_t>>& prompts,
                                                const WhisperOptions& options = WhisperOptions()) {
    detail::reject(options.return_logits_vocab, "return_logits_vocab");
    const size_t batch = prompts.size();
    if (batch == 0) return {};
    if (batch != f.batch) throw std::invalid_argument("one prompt per batch entry is required");
    size_t stride = 1;
    for (const auto& p : prompts) stride = std::max(stride, p.size());
    std::vector<int32_t> ids(batch * stride, 0), lengths(batch, 0);
    for (size_t b = 0; b < batch; ++b) {
      lengths[b] = static_cast<int32_t>(prompts[b].size());
      for (size_t i = 0; i < prompts[b].size(); ++i) ids[b * stride + i] = static_cast<int32_t>(prompts[b][i]);
    }
    std::vector<int32_t> suppress(options.suppress_tokens.begin(), options.suppress_tokens.end());
    ct2x_whisper_options o;
    ct2x_whisper_options_default(&o);
    o.beam_size = static_cast<int32_t>(options.beam_size);
    o.patience = options.patience;
    o.length_penalty = options.length_penalty;
    o.repetition_penalty = options.repetition_penalty;
    o.no_repeat_ngram_size = static_cast<int32_t>(options.no_repeat_ngram_size);
    o.max_length = static_cast<int32_t>(options.max_length);
    o.sampling_topk = static_cast<int32_t>(options.sampling_topk);
    o.sampling_temperature = options.sampling_temperature;
    o.num_hypotheses = static_cast<int32_t>(options.num_hypotheses);
    o.return_scores = options.return_scores;
    o.return_no_speech_prob = options.return_no_speech_prob;
    o.max_initial_timestamp_index = static_cast<int32_t>(options.max_initial_timestamp_index);
    o.suppress_blank = options.suppress_blank;
    o.num_suppress_tokens = static_cast<int32_t>(suppress.size());
    o.suppress_tokens = suppress.data();
    const size_t nh = options.num_hypotheses, max_len = std::max<size_t>(options.max_length, 1);
    std::vector<int32_t> out(batch * nh * max_len, 0), out_len(batch * nh, 0);
    std::vector<float> scores(batch * nh, 0.f), no_speech(batch, 0.f);
    detail::check(ct2x_whisper_generate(h_, f.data, static_cast<int32_t>(batch),
                                        static_cast<int32_t>(f.n_mels), static_cast<int32_t>(f.frames),
                                        ids.data(), lengths.data(), static_cast<int32_t>(stride), &o,
                                        static_cast<int32_t>(max_len), out.data(), out_len.data(),
                                        scores.data(), no_speech.data()));
    std::vector<WhisperGenerationResult> results(batch);
    for (size_t b = 0; b < batch; ++b) {
      WhisperGenerationResult& r = results[b];
      for (size_t k = 0; k < nh; ++k) {
        const int32_t len = out_len[b * nh + k];
        if (len < 0) continue;  // fewer hypotheses than asked for
        std::vector<size_t> seq;
        std::vector<std::string> tokens;
        for (int32_t i = 0; i < len && static_cast<size_t>(i) < max_len; ++i) {
          const int32_t id = out[(b * nh + k) * max_len + i];
          seq.push_back(static_cast<size_t>(id));
          const char* t = ct2x_whisper_vocab_token(h_, id);
          tokens.emplace_back(t ? t : "");
        }
        r.sequences_ids.emplace_back(std::move(seq));
        r.sequences.emplace_back(std::move(tokens));
        if (options.return_scores) r.scores.push_back(scores[b * nh + k]);
      }
      if (options.return_no_speech_prob) r.no_speech_prob = no_speech[b];
    }
    return results;
  }

  // prompts as tokens (Whisper::generate's other overload, whisper.h:170-173)
  std::vector<WhisperGenerationResult> generate(const WhisperFeatures& f,
                                                const std::vector<std::vector<std::string>>& prompts,
                                                const WhisperOptions& options = WhisperOptions()) {
    std::vector<std::vector<size_t>> ids(prompts.size());
    for (size_t b = 0; b < prompts.size(); ++b)
      for (const auto& t : prompts[b]) ids[b].push_back(static_cast<size_t>(ct2x_whisper_vocab_id(h_, t.c_str())));
    return generate(f, ids, options);
  }

  // Whisper::detect_language: (language token, probability) by decreasing probability
  std::vector<std::vector<std::pair<std::string, float>>> detect_language(const WhisperFeatures& f) {
    const size_t n = num_languages_;
    std::vector<int32_t> ids(f.batch * n, 0);
    std::vector<float> probs(f.batch * n, 0.f);
    detail::check(ct2x_whisper_detect_language(h_, f.data, static_cast<int32_t>(f.batch),
                                               static_cast<int32_t>(f.n_mels),
                                               static_cast<int32_t>(f.frames), static_cast<int32_t>(n),
                                               ids.data(), probs.data()));
    std::vector<std::vector<std::pair<std::string, float>>> results(f.batch);
    for (size_t b = 0; b < f.batch; ++b)
      for (size_t j = 0; j < n; ++j) {
        const char* t = ct2x_whisper_vocab_token(h_, ids[b * n + j]);
        results[b].emplace_back(t ? t : "", probs[b * n + j]);
      }
    return results;
  }

  // Whisper::align (whisper.h:181-187)
  std::vector<WhisperAlignmentResult> align(const WhisperFeatures& f,
                                            const std::vector<size_t>& start_sequence,
                                            const std::vector<std::vector<size_t>>& text_tokens,
                                            const std::vector<size_t>& num_frames,
                                            int64_t median_filter_width) {
    const size_t batch = text_tokens.size();
    if (batch == 0) return {};
    if (num_frames.size() != batch) throw std::invalid_argument("Invalid batch size for argument num_frames");
    size_t stride = 1;
    for (const auto& t : text_tokens) stride = std::max(stride, t.size());
    std::vector<int32_t> start(start_sequence.begin(), start_sequence.end());
    std::vector<int32_t> ids(batch * stride, 0), lengths(batch, 0), frames(batch, 0);
    for (size_t b = 0; b < batch; ++b) {
      lengths[b] = static_cast<int32_t>(text_tokens[b].size());
      frames[b] = static_cast<int32_t>(num_frames[b]);
      for (size_t i = 0; i < text_tokens[b].size(); ++i)
        ids[b * stride + i] = static_cast<int32_t>(text_tokens[b][i]);
    }
    const size_t cap = stride + 2 + f.frames;
    std::vector<int32_t> pairs(batch * cap * 2, 0), n_pairs(batch, 0);
    std::vector<float> probs(batch * stride, 0.f);
    detail::check(ct2x_whisper_align(h_, f.data, static_cast<int32_t>(batch),
                                     static_cast<int32_t>(f.n_mels), static_cast<int32_t>(f.frames),
                                     start.data(), static_cast<int32_t>(start.size()), ids.data(),
                                     lengths.data(), static_cast<int32_t>(stride), frames.data(),
                                     static_cast<int32_t>(median_filter_width), static_cast<int32_t>(cap),
                                     pairs.data(), n_pairs.data(), probs.data()));
    std::vector<WhisperAlignmentResult> results(batch);
    for (size_t b = 0; b < batch; ++b) {
      for (int32_t i = 0; i < n_pairs[b]; ++i)
        results[b].alignments.emplace_back(pairs[(b * cap + i) * 2], pairs[(b * cap + i) * 2 + 1]);
      results[b].text_token_probs.assign(probs.begin() + b * stride,
                                         probs.begin() + b * stride + text_tokens[b].size());
    }
    return results;
  }

private:
  void* h_ = nullptr;
  bool is_multilingual_ = false;
  size_t n_mels_ = 0, num_languages_ = 0;
};

}  // namespace models

}  // namespace ct2_mi355
