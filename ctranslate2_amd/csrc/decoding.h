// Host side of the decoding loops: GreedySearch::search and BeamSearch::search
// (src/decoding.cc:733-974, 425-720) re-hosted over the device-resident engine. The
// bookkeeping (hypothesis registration, EOS handling, candidate replacement, length penalty,
// result ordering) follows the reference line by line; the difference is mechanical:
//  - the batch is static (finished sentences stay in place and are ignored) instead of being
//    physically gathered out of every cache tensor, which is row-independent and therefore
//    result-neutral;
//  - greedy steps are enqueued `sync_every` at a time: ids are sampled and fed back on the device,
//    the host only reads the id history to detect EOS.
#pragma once

#include <cstdint>
#include <functional>
#include <vector>

#include "engine.h"

namespace ct2x {

struct DecodingStepResult {
  int64_t step = 0;
  int64_t batch_id = 0;
  int64_t hypothesis_id = 0;  // > 0 only for the samples of greedy_search_samples
  int32_t token_id = 0;
  bool has_score = false;
  float score = 0.f;
  bool is_last = false;
};

struct DecodingOptions {
  int32_t beam_size = 1;
  int32_t max_length = 256;
  int32_t min_length = 0;
  int32_t num_hypotheses = 1;
  bool return_scores = false;
  bool include_eos_in_hypotheses = true;
  float length_penalty = 0.f;
  float patience = 1.f;
  bool use_graph = true;
  int32_t sync_every = 8;
  int32_t start_step = 0;
  // RandomSampler (src/sampling.cc:41-101), greedy search only: sampling_topk == 1 is arg-max,
  // 0 samples from the whole vocabulary.
  int32_t sampling_topk = 1;
  float sampling_topp = 1.f;
  float sampling_temperature = 1.f;
  uint32_t seed = 0;
  // DecodingOptions::return_prefix (include/ctranslate2/decoding.h): true for translation (the
  // target prefix is part of the hypothesis and of max_length), false for generation.
  bool return_prefix = true;
  // repetition_penalty / no_repeat_ngram_size / disable_ids + disable_sequences
  // (include/ctranslate2/decoding.h DecodingOptions; make_logits_processors,
  // src/decoding.cc:1091-1120). Single-token entries of suppress_sequences are the reference's
  // disable_ids (disable_unk) list.
  LogitsProcessing logits_processing;
  // Biased decoding (DecodingOptions::prefix_bias_beta, BiasedDecoder, src/decoding.cc:256-343):
  // beam search only; the prefix passed to beam_search biases instead of constraining.
  float prefix_bias_beta = 0.f;
  // return_alternatives (composition in c_api_engine.cc decode_alternatives; here: validation)
  bool return_alternatives = false;
  float min_alternative_expansion_prob = 0.f;
  // return_attention / coverage_penalty (src/decoding.cc:176-207, 503, 586-593): the searches
  // record, per step and row, the attention vector of the alignment heads over the source.
  bool return_attention = false;
  float coverage_penalty = 0.f;
  // return_logits_vocab (src/decoding.cc:868-872, 905-908), greedy search / random sampling: the
  // logits every step of an entry sampled from (after the logits processors and DisableTokens)
  bool return_logits_vocab = false;
  // Whisper's logits processors (DecodingOptions::logits_processors + disable_ids_begin of
  // WhisperReplica::generate, src/models/whisper.cc:312-343); nullptr = none
  const WhisperProcessing* whisper = nullptr;
  // DecodingOptions::callback (include/ctranslate2/decoding.h; src/decoding.cc:925-939): greedy
  // search calls it for every token a row generates (hidden prefix tokens excepted); true ends the
  // row. batch_id is the row's index in THIS search's batch.
  std::function<bool(const DecodingStepResult&)> callback;
};

struct DecodingResult {
  std::vector<std::vector<int32_t>> hypotheses;
  std::vector<float> scores;
  // return_attention: attention[h][t] = the source attention vector of token t of hypothesis h
  std::vector<std::vector<std::vector<float>>> attention;
  // return_logits_vocab: logits_vocab[h][s] = the vocabulary logits of step s of hypothesis h
  // (one entry per step the row ran - forced prefix steps and the end token's step included)
  std::vector<std::vector<std::vector<float>>> logits_vocab;
};

struct DecodeStats {
  int64_t steps = 0;
  int64_t compactions = 0;  // times the greedy loop dropped its finished rows (Engine::compact_rows)
};

void validate_decoding_options(const DecodingOptions& options);

// prefix_ids (optional, one entry per row): tokens forced at the first steps of a row (hard
// prefix of decode(), src/decoding.cc:21-62 / 842-925 with return_prefix = false): they are not
// part of the returned hypothesis, contribute 0 to its score, and a row may use max_length steps
// AFTER its prefix.
std::vector<DecodingResult> greedy_search(Engine& engine, DecodeState& state,
                                          const std::vector<int32_t>& start_ids, int32_t end_id,
                                          const DecodingOptions& options, DecodeStats* stats,
                                          const std::vector<std::vector<int32_t>>* prefix_ids = nullptr);

// GreedySearch::search with num_hypotheses > 1 (src/decoding.cc:751-814): random sampling returns
// several hypotheses by decoding every batch entry num_hypotheses times and sorting an entry's
// samples by score. `state` holds start_ids.size() * num_hypotheses rows, entry-major (the
// reference's repeat_batch); start_ids / prefix_ids have one entry per batch entry.
std::vector<DecodingResult> greedy_search_samples(Engine& engine, DecodeState& state,
                                                  const std::vector<int32_t>& start_ids, int32_t end_id,
                                                  const DecodingOptions& options, DecodeStats* stats,
                                                  const std::vector<std::vector<int32_t>>* prefix_ids = nullptr);

std::vector<DecodingResult> beam_search(Engine& engine, DecodeState& state,
                                        const std::vector<int32_t>& start_ids, int32_t end_id,
                                        const DecodingOptions& options, DecodeStats* stats,
                                        const std::vector<std::vector<int32_t>>* prefix_ids = nullptr);

// Host-only pieces, exported for the CPU test-suite -------------------------------------------

// finalize_hypothesis_score without coverage penalty (src/decoding.cc:192-207).
float finalize_score(float score, float length, float length_penalty);

// One batch entry of BeamSearch's post-TopK bookkeeping (src/decoding.cc:596-676): given the
// 2*beam candidates (word ids, cumulative scores, origin beams) decide which hypotheses finish
// and which candidates become the next active beams. Pure function of its inputs.
struct BeamStepInput {
  int32_t beam_size = 1;
  int32_t step = 0;
  bool is_last_step = false;
  int32_t end_id = 0;
  const int32_t* word_ids = nullptr;   // [2*beam]
  const float* scores = nullptr;       // [2*beam]
  int32_t prefix_length = 0;           // an EOS before this step does not finish the hypothesis
};
struct BeamStepOutput {
  std::vector<int32_t> finished;       // candidate slots k < beam whose hypothesis finished
  std::vector<int32_t> next;           // [beam] candidate index that occupies beam k next step
};
BeamStepOutput beam_step_select(const BeamStepInput& in);

}  // namespace ct2x
