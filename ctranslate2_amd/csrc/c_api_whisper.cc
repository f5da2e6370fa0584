// extern "C" surface of ctranslate2::models::Whisper on the MI355X engine (include/ct2_mi355.h,
// "Whisper" section): the host composition of WhisperReplica (src/models/whisper.cc:46-630) over
// Engine::start_features / prefill / the decoding loops of decoding.cc.

#include "../../include/ct2_mi355.h"

#include <algorithm>
#include <cstring>
#include <mutex>
#include <numeric>
#include <stdexcept>

#include "decoding.h"
#include "engine.h"
#include "model.h"
#include "runtime.h"

using namespace ct2x;

namespace {

struct Whisper {
  std::shared_ptr<ModelFile> model;
  std::unique_ptr<Engine> engine;
  std::shared_ptr<Vocabulary> vocab;
  // WhisperReplica ctor (whisper.cc:62-77)
  int32_t sot_id = 0, eot_id = 0, no_timestamps_id = 0, no_speech_id = 0;
  bool is_multilingual = false;
  int32_t num_languages = 0;
  std::vector<int32_t> suppress_ids, suppress_ids_begin, lang_ids;
  std::vector<std::pair<int32_t, int32_t>> alignment_heads;
  bool has_alignment_heads = false;
  std::mutex mutex;
};

std::vector<int32_t> config_ints(const ModelFile& m, const char* key) {
  std::vector<int32_t> out;
  const Json* v = m.config.get(key);
  if (v && v->kind == Json::kArray)
    for (const Json& e : v->arr)
      if (e.kind == Json::kNumber) out.push_back(static_cast<int32_t>(e.num));
  return out;
}

// get_sot_index / get_prompt_length / check_prompts (whisper.cc:149-200)
size_t sot_index_of(const std::vector<int32_t>& prompt, int32_t sot_id) {
  const auto it = std::find(prompt.begin(), prompt.end(), sot_id);
  if (it == prompt.end())
    throw std::invalid_argument("<|startoftranscript|> token was not found in the prompt");
  return static_cast<size_t>(it - prompt.begin());
}

size_t prompt_length_of(const std::vector<int32_t>& prompt, int32_t sot_id, int32_t no_timestamps_id) {
  size_t index = sot_index_of(prompt, sot_id);
  while (index < prompt.size() && prompt[index] >= sot_id && prompt[index] <= no_timestamps_id) index++;
  return index;
}

void check_prompts(const std::vector<std::vector<int32_t>>& prompts, int32_t sot_id,
                   int32_t no_timestamps_id, size_t& sot_index, size_t& prompt_length) {
  bool first = true;
  for (const auto& prompt : prompts) {
    const size_t batch_sot_index = sot_index_of(prompt, sot_id);
    const size_t batch_prompt_length = prompt_length_of(prompt, sot_id, no_timestamps_id);
    if (first) {
      sot_index = batch_sot_index;
      prompt_length = batch_prompt_length;
    } else if (batch_sot_index != sot_index) {
      throw std::invalid_argument("The generate method currently requires the "
                                  "<|startoftranscript|> token to be at the same position "
                                  "in all batches. To work around this limitation, "
                                  "simply adapt the number of previous text tokens in each "
                                  "batch.");
    } else if (batch_prompt_length != prompt_length) {
      throw std::invalid_argument("The generate method currently requires each batch to have "
                                  "the same number of task tokens after <|startoftranscript|>.");
    }
    first = false;
  }
}

void check_features(const Whisper& w, int32_t batch, int32_t n_mels, int32_t frames) {
  // WhisperEncoder::operator() (src/layers/whisper.cc:33-50)
  if (batch < 1 || n_mels != w.engine->n_mels() || frames < 1 ||
      frames > 2 * w.engine->max_source_positions())
    throw std::invalid_argument(
        "Invalid input features shape: expected an input with shape (" + std::to_string(batch) +
        ", " + std::to_string(w.engine->n_mels()) + ", " +
        std::to_string(std::min<int64_t>(frames, 2 * w.engine->max_source_positions())) +
        "), but got an input with shape (" + std::to_string(batch) + ", " + std::to_string(n_mels) +
        ", " + std::to_string(frames) + ") instead");
}

}  // namespace

#define API_TRY try {
#define API_CATCH                                     \
  }                                                   \
  catch (const std::exception& e) {                   \
    set_last_error(e.what());                         \
    return 1;                                         \
  }                                                   \
  return 0;

extern "C" {

int ct2x_host_whisper_check_prompts(const int32_t* prompts, const int32_t* lengths, int32_t batch,
                                    int32_t stride, int32_t sot_id, int32_t no_timestamps_id,
                                    int32_t* sot_index, int32_t* prompt_length) {
  API_TRY
  CT2X_CHECK(prompts && lengths && sot_index && prompt_length && batch >= 1 && stride >= 0,
             "ct2x_host_whisper_check_prompts: bad arguments");
  std::vector<std::vector<int32_t>> p(batch);
  for (int32_t b = 0; b < batch; ++b) {
    CT2X_CHECK(lengths[b] >= 0 && lengths[b] <= stride, "prompt length out of range");
    p[b].assign(prompts + static_cast<int64_t>(b) * stride,
                prompts + static_cast<int64_t>(b) * stride + lengths[b]);
  }
  size_t si = 0, pl = 0;
  check_prompts(p, sot_id, no_timestamps_id, si, pl);
  *sot_index = static_cast<int32_t>(si);
  *prompt_length = static_cast<int32_t>(pl);
  API_CATCH
}

void ct2x_whisper_options_default(struct ct2x_whisper_options* o) {
  static const int32_t kDefaultSuppress[1] = {-1};
  std::memset(o, 0, sizeof(*o));
  o->beam_size = 5;
  o->patience = 1.f;
  o->length_penalty = 1.f;
  o->repetition_penalty = 1.f;
  o->no_repeat_ngram_size = 0;
  o->max_length = 448;
  o->sampling_topk = 1;
  o->sampling_temperature = 1.f;
  o->num_hypotheses = 1;
  o->max_initial_timestamp_index = 50;
  o->suppress_blank = 1;
  o->num_suppress_tokens = 1;
  o->suppress_tokens = kDefaultSuppress;
}

void* ct2x_whisper_create(const char* model_dir, int device_index, const char* compute_type) {
  try {
    auto w = std::make_unique<Whisper>();
    w->model = ModelFile::load(model_dir);
    if (w->model->spec != "WhisperSpec") throw std::invalid_argument("The model is not a Whisper model");
    w->engine = std::make_unique<Engine>(w->model, compute_type ? compute_type : "default",
                                         device_index);
    // WhisperModel::initialize (whisper.cc:25-35)
    w->vocab = Vocabulary::load(w->model->dir, "vocabulary", "<|endoftext|>", "<|startoftranscript|>",
                                "<|endoftext|>");
    if (!w->vocab) throw std::runtime_error("Cannot load the vocabulary from the model directory");
    if (static_cast<int64_t>(w->vocab->size()) != w->engine->vocab_size())
      throw std::runtime_error("The vocabulary has " + std::to_string(w->vocab->size()) +
                               " tokens but the model output layer has " +
                               std::to_string(w->engine->vocab_size()));
    const Vocabulary& v = *w->vocab;
    w->sot_id = static_cast<int32_t>(v.bos_id());
    w->eot_id = static_cast<int32_t>(v.eos_id());
    w->no_timestamps_id = static_cast<int32_t>(v.to_id("<|notimestamps|>"));
    w->no_speech_id = static_cast<int32_t>(v.to_id("<|nospeech|>"));
    if (w->no_speech_id == static_cast<int32_t>(v.unk_id()))
      w->no_speech_id = static_cast<int32_t>(v.to_id("<|nocaptions|>"));
    w->is_multilingual = v.to_id("") != v.unk_id();  // (as this copy of whisper.cc:72 reads)
    w->num_languages = w->no_speech_id - w->sot_id - 5;
    w->suppress_ids = config_ints(*w->model, "suppress_ids");
    w->suppress_ids_begin = config_ints(*w->model, "suppress_ids_begin");
    w->lang_ids = config_ints(*w->model, "lang_ids");
    if (const Json* ah = w->model->config.get("alignment_heads")) {
      if (ah->kind == Json::kArray) {
        w->has_alignment_heads = true;
        for (const Json& e : ah->arr)
          if (e.kind == Json::kArray && e.arr.size() == 2)
            w->alignment_heads.emplace_back(static_cast<int32_t>(e.arr[0].num),
                                            static_cast<int32_t>(e.arr[1].num));
      }
    }
    return w.release();
  } catch (const std::exception& e) {
    set_last_error(e.what());
    return nullptr;
  }
}

void ct2x_whisper_destroy(void* whisper) { delete static_cast<Whisper*>(whisper); }

int ct2x_whisper_info(void* whisper, int32_t* is_multilingual, int32_t* n_mels,
                      int32_t* num_languages) {
  API_TRY
  auto* w = static_cast<Whisper*>(whisper);
  CT2X_CHECK(w != nullptr, "null whisper handle");
  *is_multilingual = w->is_multilingual ? 1 : 0;
  *n_mels = static_cast<int32_t>(w->engine->n_mels());
  *num_languages = w->num_languages;
  API_CATCH
}

int ct2x_whisper_vocab_size(void* whisper) {
  auto* w = static_cast<Whisper*>(whisper);
  return w ? static_cast<int>(w->vocab->size()) : -1;
}

int64_t ct2x_whisper_d_model(void* whisper) {
  auto* w = static_cast<Whisper*>(whisper);
  return w ? w->engine->d_model() : -1;
}

const char* ct2x_whisper_vocab_token(void* whisper, int32_t id) {
  auto* w = static_cast<Whisper*>(whisper);
  if (!w || id < 0 || id >= static_cast<int32_t>(w->vocab->size())) return nullptr;
  return w->vocab->to_token(static_cast<size_t>(id)).c_str();
}

int32_t ct2x_whisper_vocab_id(void* whisper, const char* token) {
  auto* w = static_cast<Whisper*>(whisper);
  if (!w || !token) return -1;
  return static_cast<int32_t>(w->vocab->to_id(token));
}

int ct2x_whisper_encode(void* whisper, const float* features, int32_t batch, int32_t n_mels,
                        int32_t frames, float* out, int64_t out_capacity, int64_t* out_dims) {
  API_TRY
  auto* w = static_cast<Whisper*>(whisper);
  CT2X_CHECK(w != nullptr && features && out && out_dims, "ct2x_whisper_encode: null argument");
  std::lock_guard<std::mutex> lock(w->mutex);
  check_features(*w, batch, n_mels, frames);
  auto st = w->engine->start_features(features, batch, frames, 1, 1);
  out_dims[0] = batch;
  out_dims[1] = st->s_max;
  out_dims[2] = w->engine->d_model();
  CT2X_CHECK(out_dims[0] * out_dims[1] * out_dims[2] <= out_capacity,
             "ct2x_whisper_encode: output buffer too small");
  w->engine->read_encoder_output(*st, out);
  API_CATCH
}

int ct2x_whisper_generate(void* whisper, const float* features, int32_t batch, int32_t n_mels,
                          int32_t frames, const int32_t* prompts, const int32_t* prompt_lengths,
                          int32_t prompt_stride, const struct ct2x_whisper_options* o,
                          int32_t max_len, int32_t* out_ids, int32_t* out_lengths,
                          float* out_scores, float* out_no_speech) {
  API_TRY
  auto* w = static_cast<Whisper*>(whisper);
  CT2X_CHECK(w != nullptr && o != nullptr, "ct2x_whisper_generate: null argument");
  if (batch == 0) return 0;  // "if (prompts.empty()) return {}" (whisper.cc:239)
  CT2X_CHECK(batch > 0 && features && prompts && prompt_lengths && out_ids && out_lengths &&
                 max_len >= 0 && prompt_stride >= 0,
             "ct2x_whisper_generate: null or negative argument");
  CT2X_CHECK(o->num_suppress_tokens >= 0 && (o->num_suppress_tokens == 0 || o->suppress_tokens),
             "ct2x_whisper_generate: suppress_tokens missing");
  CT2X_CHECK(!o->return_scores || out_scores, "ct2x_whisper_generate: return_scores without out_scores");
  for (int32_t b = 0; b < batch; ++b)
    CT2X_CHECK(prompt_lengths[b] >= 0 && prompt_lengths[b] <= prompt_stride,
               "ct2x_whisper_generate: prompt length out of range");
  std::lock_guard<std::mutex> lock(w->mutex);
  Engine& engine = *w->engine;
  check_features(*w, batch, n_mels, frames);
  const int32_t vocab_size = static_cast<int32_t>(engine.vocab_size());
  std::vector<std::vector<int32_t>> p(batch);
  for (int32_t b = 0; b < batch; ++b) {
    p[b].assign(prompts + static_cast<int64_t>(b) * prompt_stride,
                prompts + static_cast<int64_t>(b) * prompt_stride + prompt_lengths[b]);
    for (int32_t id : p[b])
      CT2X_CHECK(id >= 0 && id < vocab_size, "Invalid prompt token ID " + std::to_string(id));
  }
  size_t sot_index = 0, prompt_length = 0;  // length of the prompt before the text tokens
  check_prompts(p, w->sot_id, w->no_timestamps_id, sot_index, prompt_length);
  const bool sot_is_start_token = (sot_index == prompt_length - 1);

  // start tokens = the last task token and whatever follows it (a hard prefix of the search,
  // split_start_tokens in decode(), src/decoding.cc:1345-1347); the tokens before go through one
  // sequence-mode pass (forward_prompt)
  const int64_t start_step = static_cast<int64_t>(prompt_length) - 1;
  std::vector<int32_t> start_ids(batch);
  std::vector<std::vector<int32_t>> prefix(batch);
  size_t longest_prefix = 0;
  for (int32_t b = 0; b < batch; ++b) {
    start_ids[b] = p[b][prompt_length - 1];
    prefix[b].assign(p[b].begin() + prompt_length, p[b].end());
    longest_prefix = std::max(longest_prefix, prefix[b].size());
  }

  DecodingOptions dopt;
  dopt.start_step = static_cast<int32_t>(start_step);
  dopt.beam_size = o->beam_size;
  dopt.patience = o->patience;
  dopt.length_penalty = o->length_penalty;
  dopt.logits_processing.repetition_penalty = o->repetition_penalty;
  dopt.logits_processing.no_repeat_ngram_size = o->no_repeat_ngram_size;
  const int64_t total_max_length = o->max_length;
  dopt.max_length = static_cast<int32_t>(std::min(total_max_length / 2, total_max_length - start_step));
  dopt.min_length = 0;
  dopt.sampling_topk = o->sampling_topk;
  dopt.sampling_temperature = o->sampling_temperature;
  dopt.num_hypotheses = o->num_hypotheses;
  dopt.return_scores = o->return_scores != 0;
  dopt.include_eos_in_hypotheses = false;
  dopt.return_prefix = true;
  validate_decoding_options(dopt);
  CT2X_CHECK(dopt.max_length > 0, "max_length leaves no room after the prompt");
  CT2X_CHECK(start_step + dopt.max_length <= engine.max_target_positions() ||
                 engine.max_target_positions() == 0,
             "No position encodings are defined for positions >= " +
                 std::to_string(engine.max_target_positions()));
  for (int32_t i = 0; i < o->num_suppress_tokens; ++i) {
    const int32_t id = o->suppress_tokens[i];
    if (id >= 0)
      dopt.logits_processing.suppress_sequences.push_back({id});
    else if (id == -1)
      for (int32_t d : w->suppress_ids) dopt.logits_processing.suppress_sequences.push_back({d});
  }
  WhisperProcessing wp;
  if (o->suppress_blank) wp.begin_ids = w->suppress_ids_begin;
  if (o->return_no_speech_prob && sot_is_start_token) wp.no_speech_id = w->no_speech_id;
  if (p[0][prompt_length - 1] != w->no_timestamps_id) {
    wp.timestamp_rules = true;
    wp.eot_id = w->eot_id;
    wp.no_timestamps_id = w->no_timestamps_id;
    wp.timestamp_begin = w->no_timestamps_id + 1;
    wp.timestamp_end = vocab_size - 1;
    wp.max_initial_timestamp = wp.timestamp_begin + o->max_initial_timestamp_index;
  }
  dopt.whisper = &wp;

  const int32_t beam = dopt.beam_size;
  const int64_t rows = static_cast<int64_t>(batch) * beam;
  const int64_t t_max = start_step + dopt.max_length + 1;
  auto st = engine.start_features(features, batch, frames, beam, t_max, start_step);
  std::vector<float> no_speech(batch, 0.f);
  if (start_step > 0) {
    std::vector<int32_t> prefill_ids(static_cast<size_t>(rows) * start_step);
    for (int32_t b = 0; b < batch; ++b)
      for (int32_t k = 0; k < beam; ++k)
        std::copy(p[b].begin(), p[b].begin() + start_step,
                  prefill_ids.begin() + (static_cast<size_t>(b) * beam + k) * start_step);
    if (o->return_no_speech_prob && !sot_is_start_token) {
      // "Get the probability of the no speech token at the start of transcript step"
      // (whisper.cc:283-293): the logits of prompt position sot_index
      float* logits_all = static_cast<float*>(device_malloc(sizeof(float) * rows * start_step * vocab_size));
      st->owned.push_back(logits_all);
      engine.prefill(*st, prefill_ids.data(), start_step, logits_all);
      std::vector<int32_t> ids{w->no_speech_id};
      // softmax over the whole vocabulary, probability of the no-speech token, per batch entry
      int32_t* d_t = static_cast<int32_t*>(device_malloc(sizeof(int32_t) * batch));
      float* d_p = static_cast<float*>(device_malloc(sizeof(float) * batch));
      st->owned.push_back(d_t);
      st->owned.push_back(d_p);
      std::vector<int32_t> targets(batch, w->no_speech_id);
      CT2X_HIP_CHECK(hipMemcpyAsync(d_t, targets.data(), sizeof(int32_t) * batch,
                                    hipMemcpyHostToDevice, engine.cur_stream()));
      for (int32_t b = 0; b < batch; ++b)  // hypothesis row 0 of entry b, position sot_index
        launch_token_probs(logits_all + ((static_cast<int64_t>(b) * beam) * start_step +
                                         static_cast<int64_t>(sot_index)) * vocab_size,
                           1, vocab_size, vocab_size, d_t + b, d_p + b, engine.cur_stream());
      CT2X_HIP_CHECK(hipMemcpyAsync(no_speech.data(), d_p, sizeof(float) * batch,
                                    hipMemcpyDeviceToHost, engine.cur_stream()));
      engine.synchronize();
    } else {
      engine.prefill(*st, prefill_ids.data(), start_step);
    }
  }
  const std::vector<std::vector<int32_t>>* pf = longest_prefix > 0 ? &prefix : nullptr;
  DecodeStats ds;
  std::vector<DecodingResult> results =
      beam > 1 ? beam_search(engine, *st, start_ids, w->eot_id, dopt, &ds, pf)
               : greedy_search(engine, *st, start_ids, w->eot_id, dopt, &ds, pf);
  if (wp.no_speech_id >= 0) {
    std::vector<float> per_row(rows);
    engine.read_no_speech_probs(*st, per_row.data());
    for (int32_t b = 0; b < batch; ++b) no_speech[b] = per_row[static_cast<size_t>(b) * beam];
  }
  const int32_t nh = o->num_hypotheses;
  for (int32_t b = 0; b < batch; ++b) {
    const DecodingResult& r = results[b];
    for (int32_t k = 0; k < nh; ++k) {
      out_lengths[b * nh + k] = -1;
      if (k >= static_cast<int32_t>(r.hypotheses.size())) continue;
      const auto& hyp = r.hypotheses[k];
      out_lengths[b * nh + k] = static_cast<int32_t>(hyp.size());
      std::copy(hyp.begin(), hyp.begin() + std::min<size_t>(hyp.size(), max_len),
                out_ids + (static_cast<int64_t>(b) * nh + k) * max_len);
      if (out_scores && k < static_cast<int32_t>(r.scores.size())) out_scores[b * nh + k] = r.scores[k];
    }
    if (out_no_speech && o->return_no_speech_prob) out_no_speech[b] = no_speech[b];
  }
  API_CATCH
}

int ct2x_whisper_detect_language(void* whisper, const float* features, int32_t batch,
                                 int32_t n_mels, int32_t frames, int32_t num_languages,
                                 int32_t* out_ids, float* out_probs) {
  API_TRY
  auto* w = static_cast<Whisper*>(whisper);
  CT2X_CHECK(w != nullptr && features && out_ids && out_probs, "ct2x_whisper_detect_language: null argument");
  if (!w->is_multilingual)
    throw std::runtime_error("detect_language can only be called on multilingual models");
  std::lock_guard<std::mutex> lock(w->mutex);
  Engine& engine = *w->engine;
  check_features(*w, batch, n_mels, frames);
  const int64_t n = static_cast<int64_t>(w->lang_ids.size());
  CT2X_CHECK(n == num_languages, "ct2x_whisper_detect_language: the model lists " +
                                     std::to_string(n) + " languages");
  // one decoder step from <|startoftranscript|>, the logits of the language tokens, SoftMax
  auto st = engine.start_features(features, batch, frames, 1, 2);
  std::vector<int32_t> start(batch, w->sot_id);
  engine.set_ids(*st, start.data());
  engine.set_step(*st, 0);
  engine.step(*st, /*graph=*/false);
  int32_t* d_ids = static_cast<int32_t*>(device_malloc(sizeof(int32_t) * n));
  float* d_probs = static_cast<float*>(device_malloc(sizeof(float) * batch * n));
  st->owned.push_back(d_ids);
  st->owned.push_back(d_probs);
  CT2X_HIP_CHECK(hipMemcpyAsync(d_ids, w->lang_ids.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice,
                                engine.cur_stream()));
  launch_gather_softmax(st->logits, batch, engine.vocab_size(), d_ids, n, d_probs, engine.cur_stream());
  std::vector<float> probs(static_cast<size_t>(batch) * n);
  CT2X_HIP_CHECK(hipMemcpyAsync(probs.data(), d_probs, sizeof(float) * batch * n, hipMemcpyDeviceToHost,
                                engine.cur_stream()));
  engine.synchronize();
  for (int32_t b = 0; b < batch; ++b) {
    std::vector<int32_t> order(n);
    std::iota(order.begin(), order.end(), 0);
    // std::sort by decreasing probability (whisper.cc:617-621); ties keep the config order
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) {
      return probs[b * n + x] > probs[b * n + y];
    });
    for (int64_t j = 0; j < n; ++j) {
      out_ids[b * n + j] = w->lang_ids[order[j]];
      out_probs[b * n + j] = probs[b * n + order[j]];
    }
  }
  API_CATCH
}

int ct2x_whisper_align(void* whisper, const float* features, int32_t batch, int32_t n_mels,
                       int32_t frames, const int32_t* start_sequence, int32_t num_start,
                       const int32_t* text_tokens, const int32_t* text_lengths, int32_t text_stride,
                       const int32_t* num_frames, int32_t median_filter_width,
                       int32_t pair_capacity, int32_t* out_pairs, int32_t* out_num_pairs,
                       float* out_token_probs) {
  API_TRY
  auto* w = static_cast<Whisper*>(whisper);
  CT2X_CHECK(w != nullptr, "null whisper handle");
  if (batch == 0) return 0;
  CT2X_CHECK(batch > 0 && features && text_tokens && text_lengths && num_frames && out_pairs &&
                 out_num_pairs && out_token_probs && num_start >= 0 && (num_start == 0 || start_sequence),
             "ct2x_whisper_align: null or negative argument");
  for (int32_t b = 0; b < batch; ++b)
    CT2X_CHECK(text_lengths[b] >= 0 && text_lengths[b] <= text_stride,
               "ct2x_whisper_align: text length out of range");
  std::lock_guard<std::mutex> lock(w->mutex);
  Engine& engine = *w->engine;
  check_features(*w, batch, n_mels, frames);
  if (!w->has_alignment_heads)
    throw std::runtime_error("The model configuration does not contain the field "
                             "'alignment_heads' which lists the cross-attention heads "
                             "that are highly correlated to the word-level timing. "
                             "Please reconvert this model with the current version "
                             "of ctranslate2.");
  CT2X_CHECK(median_filter_width >= 1 && median_filter_width % 2 == 1 && median_filter_width <= 63,
             "median_filter_width must be odd and at most 63");
  const int32_t vocab_size = static_cast<int32_t>(engine.vocab_size());
  // input = start_sequence + <|notimestamps|> + text + <|endoftext|>; output = input shifted by
  // one (+ a 0), whisper.cc:456-468
  std::vector<std::vector<int32_t>> inputs(batch);
  int64_t time = 0;
  for (int32_t b = 0; b < batch; ++b) {
    auto& in = inputs[b];
    in.assign(start_sequence, start_sequence + num_start);
    in.push_back(w->no_timestamps_id);
    in.insert(in.end(), text_tokens + static_cast<int64_t>(b) * text_stride,
              text_tokens + static_cast<int64_t>(b) * text_stride + text_lengths[b]);
    in.push_back(w->eot_id);
    for (int32_t id : in)
      CT2X_CHECK(id >= 0 && id < vocab_size, "Invalid token ID " + std::to_string(id));
    time = std::max<int64_t>(time, static_cast<int64_t>(in.size()));
  }
  CT2X_CHECK(time <= engine.max_target_positions() || engine.max_target_positions() == 0,
             "No position encodings are defined for positions >= " +
                 std::to_string(engine.max_target_positions()));
  std::vector<int32_t> ids(static_cast<size_t>(batch) * time, 0), targets(static_cast<size_t>(batch) * time, 0);
  std::vector<int32_t> lengths(batch), half_frames(batch);
  const int64_t S = Engine::whisper_memory_length(frames);
  for (int32_t b = 0; b < batch; ++b) {
    const auto& in = inputs[b];
    lengths[b] = static_cast<int32_t>(in.size());
    for (size_t t = 0; t < in.size(); ++t) {
      ids[b * time + t] = in[t];
      // a padded row of the reference repeats the entry's last row; its output id is 0
      if (t + 1 < in.size()) targets[b * time + t] = in[t + 1];
    }
    for (int64_t t = static_cast<int64_t>(in.size()); t < time; ++t) ids[b * time + t] = 0;
    // "The second convolution layer uses a stride of 2" (whisper.cc:510-514)
    CT2X_CHECK(num_frames[b] >= 0, "Invalid batch size for argument num_frames");
    half_frames[b] = num_frames[b] / 2;
    CT2X_CHECK(half_frames[b] <= S, "num_frames exceeds the encoded frames");
  }
  auto st = engine.start_features(features, batch, frames, 1, time + 1, time);
  std::vector<float> token_probs(static_cast<size_t>(batch) * time);
  std::vector<float> weights(static_cast<size_t>(batch) * time * st->s_max);
  const bool all_zero = std::all_of(half_frames.begin(), half_frames.end(),
                                    [](int32_t f) { return f == 0; });
  // (with no frame left to align against the attention post-processing is skipped, the token
  // probabilities are still computed: whisper.cc:518-523)
  std::vector<int32_t> frames_arg = half_frames;
  if (all_zero) std::fill(frames_arg.begin(), frames_arg.end(), 0);
  engine.whisper_align_forward(*st, ids.data(), targets.data(), time, lengths, frames_arg,
                               w->alignment_heads, w->eot_id, median_filter_width,
                               token_probs.data(), weights.data());
  bool variable = false;
  for (int32_t b = 0; b < batch; ++b) variable = variable || half_frames[b] != half_frames[0];
  for (int32_t b = 0; b < batch; ++b) {
    const int64_t text_length = text_lengths[b], sot_length = num_start, F = half_frames[b];
    out_num_pairs[b] = 0;
    for (int64_t t = 0; t < text_length; ++t)  // result.text_token_probs (whisper.cc:545-546)
      out_token_probs[static_cast<int64_t>(b) * text_stride + t] = token_probs[b * time + sot_length + t];
    if (F == 0) continue;
    // compute_alignments (whisper.cc:410-420): rows [sot_length, sot_length + text_length] of the
    // entry's [rows, F] matrix go through the negated DTW
    const int64_t n = text_length + 1;
    const float* matrix = weights.data() + static_cast<int64_t>(b) * time * st->s_max + sot_length * F;
    (void)variable;
    std::vector<int64_t> pi(static_cast<size_t>(n + F)), pj(static_cast<size_t>(n + F));
    const int64_t len = ct2x_host_negative_dtw(matrix, n, F, pi.data(), pj.data());
    if (len < 0) throw std::runtime_error(ct2x_last_error());
    CT2X_CHECK(len <= pair_capacity, "ct2x_whisper_align: pair capacity too small");
    out_num_pairs[b] = static_cast<int32_t>(len);
    for (int64_t i = 0; i < len; ++i) {
      out_pairs[(static_cast<int64_t>(b) * pair_capacity + i) * 2] = static_cast<int32_t>(pi[i]);
      out_pairs[(static_cast<int64_t>(b) * pair_capacity + i) * 2 + 1] = static_cast<int32_t>(pj[i]);
    }
  }
  API_CATCH
}

}  // extern "C"
