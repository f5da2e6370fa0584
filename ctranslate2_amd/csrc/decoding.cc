#include "decoding.h"

#include <mutex>
#include "kernels.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <numeric>
#include <stdexcept>

namespace ct2x {

// validate_decoding_options (src/decoding.cc:1023-1063), the subset that applies here.
void validate_decoding_options(const DecodingOptions& o) {
  if (o.beam_size <= 0) throw std::invalid_argument("The beam size must be > 0");
  if (o.patience <= 0) throw std::invalid_argument("The patience factor must be > 0");
  if (o.num_hypotheses <= 0) throw std::invalid_argument("The number of hypotheses must be > 0");
  const int64_t max_candidates = std::llround(float(o.beam_size) * o.patience);
  if (o.num_hypotheses > max_candidates && !o.return_alternatives &&
      !(o.beam_size == 1 && o.sampling_topk != 1))
    throw std::invalid_argument("The number of hypotheses cannot be greater than "
                                "beam_size * patience");
  if (o.min_length > o.max_length)
    throw std::invalid_argument("The minimum decoding length is greater than "
                                "the maximum decoding length");
  if (o.max_length <= 0) throw std::invalid_argument("The maximum decoding length must be > 0");
  if (o.sampling_topk != 1) {
    if (o.beam_size != 1)
      throw std::invalid_argument("Random sampling (sampling_topk != 1) requires beam_size 1");
    // 0 = the whole vocabulary (the reference's sampling_topk <= 0), else the best k (any k:
    // candidate lists up to max_sampling_topk(), the threshold form of the sampler above)
    if (o.sampling_topk < 0)
      throw std::invalid_argument("sampling_topk must be 0 (whole vocabulary) or positive");
  }
  if (!(o.sampling_topp > 0.f && o.sampling_topp <= 1.f))
    throw std::invalid_argument("sampling_topp must be in (0, 1]");
  if (!(o.sampling_temperature > 0.f))
    throw std::invalid_argument("sampling_temperature must be > 0");
  // (src/decoding.cc:1040-1051)
  if (o.prefix_bias_beta >= 1)
    throw std::invalid_argument("The beta value in biased decoding must be smaller than 1");
  if (o.prefix_bias_beta > 0 && o.return_alternatives)
    throw std::invalid_argument("Biased decoding is not compatible with the return_alternatives mode");
  if (o.return_alternatives &&
      (o.min_alternative_expansion_prob < 0 || o.min_alternative_expansion_prob > 1))
    throw std::invalid_argument("The minimum alternative expansion probability must be between 0 and 1");
  if (o.prefix_bias_beta > 0 && o.sampling_topk != 1)
    throw std::invalid_argument("Biased decoding runs a beam search: random sampling is not available with it");
  // NOT a rule of the reference (its beam search records logits too: src/decoding.cc:529, 633-666,
  // and so does decode_alternatives, :1198, :1262): a gap of this backend, rejected loudly and
  // listed in README.md "Not supported" (what the reference's beam search records is one vector
  // per finished hypothesis - row i * k of the finishing step, :633-636 - not per-step logits)
  if (o.return_logits_vocab && (o.beam_size != 1 || o.prefix_bias_beta > 0 || o.return_alternatives))
    throw std::invalid_argument("return_logits_vocab is built for greedy search and random sampling "
                                "(beam_size 1) in this backend");
  // (src/decoding.cc:1052-1054)
  if (o.callback && (o.beam_size != 1 || o.prefix_bias_beta > 0))
    throw std::invalid_argument("The callback function is not compatible with "
                                "beam_size > 1 or prefix_bias_beta > 0");
}

float finalize_score(float score, float length, float length_penalty) {
  // finalize_hypothesis_score (src/decoding.cc:192-207): score /= pow(length, length_penalty).
  return score / std::pow(length, length_penalty);
}

namespace {

// sort_hypotheses (src/decoding.cc:209-240).
void sort_and_trim(DecodingResult& r, size_t max_hypotheses, bool keep_scores, bool keep_attention) {
  std::vector<size_t> idx(r.hypotheses.size());
  std::iota(idx.begin(), idx.end(), 0);
  std::sort(idx.begin(), idx.end(), [&r](size_t a, size_t b) { return r.scores[a] > r.scores[b]; });
  if (max_hypotheses < idx.size()) idx.resize(max_hypotheses);
  DecodingResult out;
  for (size_t i : idx) {
    out.hypotheses.emplace_back(std::move(r.hypotheses[i]));
    if (keep_scores) out.scores.push_back(r.scores[i]);
    if (keep_attention && i < r.attention.size()) out.attention.emplace_back(std::move(r.attention[i]));
    if (i < r.logits_vocab.size()) out.logits_vocab.emplace_back(std::move(r.logits_vocab[i]));
  }
  r = std::move(out);
}

// compute_coverage_penalty (src/decoding.cc:176-188).
float coverage_term(const std::vector<std::vector<float>>& attention, float beta) {
  float penalty = 0;
  if (attention.empty()) return 0.f;
  for (size_t column = 0; column < attention[0].size(); ++column) {
    float coverage = 0;
    for (size_t row = 0; row < attention.size(); ++row) coverage += attention[row][column];
    if (coverage > 0) penalty += std::log(std::min(coverage, 1.f));
  }
  return beta * penalty;
}

// finalize_result (src/decoding.cc:242-262).
void finalize(DecodingResult& r, size_t max_hypotheses, const DecodingOptions& opt) {
  for (size_t i = 0; i < r.scores.size(); ++i) {
    r.scores[i] = finalize_score(r.scores[i], static_cast<float>(r.hypotheses[i].size()),
                                 opt.length_penalty);
    if (opt.coverage_penalty != 0 && i < r.attention.size())
      r.scores[i] += coverage_term(r.attention[i], opt.coverage_penalty);
  }
  sort_and_trim(r, max_hypotheses, opt.return_scores, opt.return_attention);
}

}  // namespace

// GreedySearch::search (src/decoding.cc:733-974), num_hypotheses == 1, no prefix.
// Row count from which the greedy loop drops finished rows (0 = never). Default 64: below that a
// decode step is launch / weight-stream bound and costs the same for any row count (measured: §2 of
// DESIGN.md), so a smaller batch buys nothing and costs a graph capture. CT2X_COMPACT_MIN_ROWS
// overrides (tests run the goldens with 2).
static int64_t compact_min_rows_setting() {
  static const int64_t v = [] {
    const char* e = std::getenv("CT2X_COMPACT_MIN_ROWS");
    return e != nullptr ? static_cast<int64_t>(std::atoll(e)) : int64_t(64);
  }();
  return v;
}

std::vector<DecodingResult> greedy_search(Engine& engine, DecodeState& st,
                                          const std::vector<int32_t>& start_ids, int32_t end_id,
                                          const DecodingOptions& opt, DecodeStats* stats,
                                          const std::vector<std::vector<int32_t>>* prefix_ids) {
  const int64_t batch = static_cast<int64_t>(start_ids.size());
  if (batch != st.rows) throw std::runtime_error("greedy_search: state/batch size mismatch");
  // (the decoder-stack launch only while this loop holds the device's gate; with a step callback -
  // which a re-run after a ClusterTimeout would call twice - the launch-per-op step)
  ClusterLease cluster_lease(engine, st, /*allowed=*/!opt.callback);
  std::vector<DecodingResult> results(batch);
  std::vector<float> score_sum(batch, 0.f);
  std::vector<char> alive(batch, 1);
  for (auto& r : results) r.hypotheses.resize(1);
  int64_t count_alive = batch;
  // gather_attention (src/decoding.cc:819): the attention also feeds the coverage penalty
  const bool gather_attention =
      opt.return_attention || (opt.return_scores && opt.coverage_penalty != 0);
  std::vector<float> att_steps;
  if (gather_attention) {
    engine.enable_attention(st);
    for (auto& r : results) r.attention.resize(1);
  }
  std::vector<float> logit_steps;
  const int64_t V = engine.vocab_size();
  if (opt.return_logits_vocab) {
    engine.enable_logits_history(st);
    for (auto& r : results) r.logits_vocab.resize(1);
  }

  engine.set_ids(st, start_ids.data());
  engine.set_step(st, opt.start_step);
  st.fuse_greedy = true;
  st.g_eos = end_id;
  st.g_min_length = opt.start_step + opt.min_length;  // device step counts from start_step
  st.g_scores = opt.return_scores;
  st.g_topk = opt.sampling_topk;
  st.g_topp = opt.sampling_topp;
  st.g_temperature = opt.sampling_temperature;
  st.g_seed = opt.seed;
  engine.set_logits_processing(st, opt.logits_processing);
  // Hard prefix: forced on the device by the fused sampler; the host only has to know how many
  // leading steps of each row to skip (get_max_step / is_last_step, src/decoding.cc:357-382).
  std::vector<int64_t> prefix_len(batch, 0);
  int64_t longest_prefix = 0;
  if (prefix_ids) {
    for (int64_t i = 0; i < batch; ++i) {
      prefix_len[i] = static_cast<int64_t>((*prefix_ids)[i].size());
      longest_prefix = std::max(longest_prefix, prefix_len[i]);
    }
    engine.set_prefix(st, *prefix_ids, opt.start_step, !opt.return_prefix);
  } else {
    engine.set_prefix(st, {}, opt.start_step);
  }
  if (opt.whisper) engine.set_whisper_processing(st, *opt.whisper);
  // return_prefix: the prefix is part of the hypothesis and counts towards max_length
  const int64_t extra = opt.return_prefix ? 0 : 1;
  const int64_t max_step = opt.max_length + extra * longest_prefix;
  const int64_t sync_every = std::max<int32_t>(1, opt.sync_every);
  int64_t launched = 0;   // steps enqueued on the device
  int64_t examined = 0;   // steps whose ids the host has looked at

  // Software pipeline of depth 2: group k+1 is queued on the device before the host waits for
  // the ids of group k, so the GPU never idles on the host's EOS scan. At most one group of
  // steps is computed past the point where every sentence had finished (their results are
  // simply not read).
  struct Group { int64_t first, count; int slot; };
  std::vector<Group> inflight;
  int next_slot = 0;
  auto launch_group = [&]() {
    const int64_t target = std::min(max_step, launched + sync_every);
    if (target == launched) return false;
    Group g{launched, target - launched, next_slot};
    engine.step_group(st, static_cast<int>(g.count), opt.use_graph);
    launched = target;
    engine.read_history_async(st, opt.start_step + g.first, g.count, g.slot);
    next_slot ^= 1;
    inflight.push_back(g);
    return true;
  };

  // Finished sentences leave the batch (the reference gathers the alive rows out of every state
  // tensor at every step that finishes one: src/decoding.cc:916-956). Here the step is a captured
  // graph of a fixed row count and the host runs up to two groups of steps behind the device, so
  // rows are dropped in bulk: once at most half of the session's rows are alive, the groups in
  // flight are examined (no new one is queued), Engine::compact_rows packs the alive rows and the
  // loop goes on with a graph of the smaller batch. Row-wise arithmetic is batch-composition
  // neutral (tests: test_batch_composition_is_result_neutral), so results do not change. Worth it
  // only where a step's cost follows the row count (min_rows: above the skinny-GEMM regime, where a
  // decode step costs the same for 1 or 32 rows).
  const int64_t compact_min_rows = compact_min_rows_setting();
  // (random sampling draws from a hash of (seed, step, ROW): renumbering the rows would change the
  // sample a seed produces - those sessions keep their rows)
  const bool may_compact = !gather_attention && !opt.return_logits_vocab && compact_min_rows > 0 &&
                           opt.sampling_topk == 1 &&
                           engine.compactable(st);
  std::vector<int64_t> origin(batch);  // session row -> index into results / alive / prefix_len
  for (int64_t i = 0; i < batch; ++i) origin[i] = i;
  bool compact_pending = false;

  launch_group();
  while ((!inflight.empty() || compact_pending) && count_alive > 0) {
    if (compact_pending && inflight.empty()) {
      std::vector<int32_t> keep;
      std::vector<int64_t> new_origin;
      for (int64_t r = 0; r < st.rows; ++r)
        if (alive[origin[r]]) {
          keep.push_back(static_cast<int32_t>(r));
          new_origin.push_back(origin[r]);
        }
      engine.compact_rows(st, keep, launched);
      origin.swap(new_origin);
      compact_pending = false;
      if (stats) ++stats->compactions;
      if (!launch_group()) break;
      continue;
    }
    if (!compact_pending && inflight.size() < 2) launch_group();  // keep the device busy while we scan group k
    const Group g = inflight.front();
    inflight.erase(inflight.begin());
    const int32_t* h_ids = nullptr;
    const float* h_scores = nullptr;
    engine.wait_history(st, g.slot, &h_ids, &h_scores);
    const int64_t rows_now = st.rows;  // (no compaction while a group is in flight)
    if (gather_attention) {
      att_steps.resize(static_cast<size_t>(g.count) * batch * st.s_max);
      engine.read_attention(st, opt.start_step + g.first, g.count, att_steps.data());
    }
    if (opt.return_logits_vocab) {
      logit_steps.resize(static_cast<size_t>(g.count) * batch * V);
      engine.read_logits(st, opt.start_step + g.first, g.count, logit_steps.data());
    }
    for (int64_t s = 0; s < g.count && count_alive > 0; ++s) {
      const int64_t step = g.first + s;
      for (int64_t row = 0; row < rows_now; ++row) {
        const int64_t i = origin[row];
        if (!alive[i]) continue;
        if (opt.return_logits_vocab) {  // every step of a live row, whatever becomes of its token
          const float* lg = logit_steps.data() + (static_cast<size_t>(s) * rows_now + row) * V;
          results[i].logits_vocab[0].emplace_back(lg, lg + V);
        }
        // is_last_step (src/decoding.cc:376-382): max_length steps after the row's own prefix
        const bool last = (step + 1 == opt.max_length + extra * prefix_len[i]);
        const int32_t word = h_ids[s * rows_now + row];
        const bool in_prefix = step < prefix_len[i];
        const bool eos = (word == end_id) && !in_prefix;
        if ((!in_prefix || opt.return_prefix) && (!eos || opt.include_eos_in_hypotheses)) {
          results[i].hypotheses[0].push_back(word);
          if (gather_attention) {
            const float* a = att_steps.data() + (static_cast<size_t>(s) * rows_now + row) * st.s_max;
            results[i].attention[0].emplace_back(a, a + st.s_max);
          }
        }
        if (opt.return_scores) score_sum[i] += h_scores[s * rows_now + row];  // forced steps add 0
        bool stop = false;
        if (opt.callback && (opt.return_prefix || !in_prefix)) {  // (src/decoding.cc:925-939)
          DecodingStepResult r;
          r.step = step;
          r.batch_id = i;
          r.token_id = word;
          r.has_score = opt.return_scores;
          r.score = opt.return_scores ? h_scores[s * rows_now + row] : 0.f;
          r.is_last = eos || last;
          stop = opt.callback(r);
        }
        if (eos || last || stop) {
          alive[i] = 0;
          --count_alive;
          results[i].scores.assign(1, score_sum[i]);
          finalize(results[i], 1, opt);
        }
      }
    }
    examined = g.first + g.count;
    if (may_compact && !compact_pending && count_alive > 0 && st.rows >= compact_min_rows &&
        count_alive * 2 <= st.rows)
      compact_pending = true;
  }
  (void)examined;
  engine.synchronize();  // drain a speculative group before the state is torn down
  engine.check_cluster(st);  // (decoder-stack launch: a cluster that waited in vain voids the results)
  st.fuse_greedy = false;
  if (stats) stats->steps = launched;
  return results;
}

std::vector<DecodingResult> greedy_search_samples(Engine& engine, DecodeState& st,
                                                  const std::vector<int32_t>& start_ids, int32_t end_id,
                                                  const DecodingOptions& opt, DecodeStats* stats,
                                                  const std::vector<std::vector<int32_t>>* prefix_ids) {
  const int64_t nh = opt.num_hypotheses;
  if (nh <= 1) return greedy_search(engine, st, start_ids, end_id, opt, stats, prefix_ids);
  const size_t batch = start_ids.size();
  std::vector<int32_t> rep_start;
  std::vector<std::vector<int32_t>> rep_prefix;
  for (size_t i = 0; i < batch; ++i)
    for (int64_t k = 0; k < nh; ++k) {
      rep_start.push_back(start_ids[i]);
      if (prefix_ids) rep_prefix.push_back((*prefix_ids)[i]);
    }
  DecodingOptions inner = opt;
  inner.num_hypotheses = 1;
  inner.return_scores = true;  // the samples are ordered by score (:789)
  if (opt.callback)
    inner.callback = [&opt, nh](const DecodingStepResult& r) {  // (:766-771)
      DecodingStepResult m = r;
      m.hypothesis_id = r.batch_id % nh;
      m.batch_id = r.batch_id / nh;
      return opt.callback(m);
    };
  std::vector<DecodingResult> rows =
      greedy_search(engine, st, rep_start, end_id, inner, stats, prefix_ids ? &rep_prefix : nullptr);
  std::vector<DecodingResult> results(batch);
  for (size_t r = 0; r < rows.size(); ++r) {
    DecodingResult& out = results[r / static_cast<size_t>(nh)];
    out.hypotheses.emplace_back(std::move(rows[r].hypotheses[0]));
    out.scores.push_back(rows[r].scores[0]);
    if (opt.return_attention) out.attention.emplace_back(std::move(rows[r].attention[0]));
    if (opt.return_logits_vocab) out.logits_vocab.emplace_back(std::move(rows[r].logits_vocab[0]));
  }
  for (auto& r : results) sort_and_trim(r, static_cast<size_t>(nh), opt.return_scores, opt.return_attention);
  return results;
}

BeamStepOutput beam_step_select(const BeamStepInput& in) {
  BeamStepOutput out;
  const int32_t beam = in.beam_size, num_candidates = 2 * in.beam_size;
  out.next.resize(beam);
  int32_t secondary_candidates_offset = beam;
  for (int32_t k = 0; k < beam; ++k) {
    const int32_t last_id = in.word_ids[k];
    int32_t next_beam_id = k;
    if ((last_id == in.end_id && in.step >= in.prefix_length) || in.is_last_step) {
      out.finished.push_back(k);
      // Move another active beam to this position (src/decoding.cc:638-646).
      for (int32_t j = secondary_candidates_offset; j < num_candidates; ++j) {
        if (in.word_ids[j] != in.end_id) {
          next_beam_id = j;
          secondary_candidates_offset = j + 1;
          break;
        }
      }
    }
    out.next[k] = next_beam_id;
  }
  return out;
}

// BeamSearch::search (src/decoding.cc:425-720) on the GPU arm of the reference: the batch is
// expanded before the first step (expand_after_first_step is CPU-only, :453-454) and secondary
// beams start at lowest() (:84-93).
// The step itself - log-softmax + beam scores, TopK, hard prefix, candidate selection, cache
// reorder - runs on the device (Engine::launch_step with fuse_beam, beam_select_kernel) and
// records its 2*beam candidates per batch entry; steps are queued `sync_every` at a time like the
// greedy loop. The host reads the recorded candidates one group behind and replays the
// reference's bookkeeping on them (which hypotheses finished, which candidates stay alive,
// when a sentence is complete). Finished sentences keep stepping on the device (static batch);
// their rows are ignored.
std::vector<DecodingResult> beam_search(Engine& engine, DecodeState& st,
                                        const std::vector<int32_t>& start_ids, int32_t end_id,
                                        const DecodingOptions& opt, DecodeStats* stats,
                                        const std::vector<std::vector<int32_t>>* prefix_ids) {
  const int64_t batch = static_cast<int64_t>(start_ids.size());
  const int32_t beam = opt.beam_size;
  const int32_t num_candidates = 2 * beam;
  const int64_t rows = batch * beam;
  if (rows != st.rows || st.beam != beam)
    throw std::runtime_error("beam_search: state/batch size mismatch");
  ClusterLease cluster_lease(engine, st, /*allowed=*/!opt.callback);
  const size_t max_candidates = static_cast<size_t>(std::llround(float(beam) * opt.patience));
  const bool allow_early_exit = (opt.length_penalty == 0 && opt.coverage_penalty == 0);

  std::vector<DecodingResult> results(batch);
  std::vector<char> finished(batch, 0), top_beam_finished(batch, 0);
  int64_t remaining = batch;

  {
    std::vector<int32_t> first_ids(rows);
    std::vector<float> beam_scores(rows);
    for (int64_t i = 0; i < batch; ++i)
      for (int32_t k = 0; k < beam; ++k) {
        first_ids[i * beam + k] = start_ids[i];
        beam_scores[i * beam + k] = (k == 0) ? 0.f : -FLT_MAX;  // numeric_limits<float>::lowest()
      }
    engine.set_ids(st, first_ids.data());
    engine.set_step(st, opt.start_step);
    engine.set_beam_scores(st, beam_scores.data());
  }
  st.fuse_greedy = false;
  st.fuse_beam = true;
  st.g_eos = end_id;
  st.g_min_length = opt.start_step + opt.min_length;  // device step counts from start_step
  engine.set_logits_processing(st, opt.logits_processing);
  // return_prefix == false (generation): the prefix is forced but hidden - it is not part of the
  // hypotheses, and max_length / min_length count from its end (get_max_step, is_last_step,
  // apply_min_length, src/decoding.cc:357-408).
  const int64_t extra = opt.return_prefix ? 0 : 1;
  int64_t longest_prefix = 0;
  // biased decoding (src/decoding.cc:478-483): the prefix biases the distributions on the device
  // (beam_logprobs_kernel) and constrains nothing - no forced tokens, no effect on the lengths
  const std::vector<std::vector<int32_t>>* const prefix_ids_in = prefix_ids;
  const bool bias_towards_prefix = prefix_ids && opt.prefix_bias_beta > 0;
  if (bias_towards_prefix) {
    engine.set_beam_prefix(st, *prefix_ids, opt.start_step, !opt.return_prefix, opt.prefix_bias_beta);
    prefix_ids = nullptr;
  } else if (prefix_ids) {
    for (const auto& pfx : *prefix_ids)
      longest_prefix = std::max<int64_t>(longest_prefix, static_cast<int64_t>(pfx.size()));
    engine.set_beam_prefix(st, *prefix_ids, opt.start_step, !opt.return_prefix);
  } else {
    engine.set_beam_prefix(st, {}, opt.start_step);
  }
  if (opt.whisper) engine.set_whisper_processing(st, *opt.whisper);

  // alive_seq[i][k]: tokens of active beam k of batch i.
  std::vector<std::vector<std::vector<int32_t>>> alive_seq(
      batch, std::vector<std::vector<int32_t>>(beam));
  // alive_attention[i][k]: one source attention vector per token of that beam (:586-593)
  const bool gather_attention = opt.return_attention || opt.coverage_penalty != 0;
  std::vector<std::vector<std::vector<std::vector<float>>>> alive_attention;
  std::vector<float> att_steps;
  if (gather_attention) {
    engine.enable_attention(st);
    alive_attention.assign(batch, std::vector<std::vector<std::vector<float>>>(beam));
  }

  const int64_t max_step = opt.max_length + extra * longest_prefix;
  // A hidden prefix with min_length > 0: the reference disables the end token by BATCH index in the
  // rows of the expanded logits (apply_min_length, src/decoding.cc:392-402: row i < live entries by
  // live entry i's threshold, no other row) - which rows are masked depends on how many entries are
  // still alive AT THAT STEP. beam_logprobs_kernel applies that rule to the session's entries; for
  // it to see the reference's live entries the loop runs in lock step in this one configuration:
  // one step per group, nothing launched ahead of the replay, an entry leaves the session at the
  // step it finishes (as src/decoding.cc:678-694 does). Sessions that cannot be compacted keep every
  // entry: exact until the first entry finishes.
  static const bool per_entry_min = [] {
    const char* e = std::getenv("CT2X_HIDDEN_PREFIX_MIN_LENGTH_PER_ENTRY");
    return e != nullptr && e[0] == '1';
  }();
  const bool row_rule_lockstep = prefix_ids_in != nullptr && !opt.return_prefix && beam > 1 &&
                                 opt.min_length > 0 && batch > 1 && !per_entry_min;
  const int64_t sync_every = row_rule_lockstep ? 1 : std::max<int32_t>(1, opt.sync_every);
  int64_t launched = 0, step = 0;
  struct Group { int64_t first, count; int slot; };
  std::vector<Group> inflight;
  int next_slot = 0;
  auto launch_group = [&]() {
    const int64_t target = std::min(max_step, launched + sync_every);
    if (target == launched) return false;
    Group g{launched, target - launched, next_slot};
    engine.step_group(st, static_cast<int>(g.count), opt.use_graph);
    launched = target;
    engine.read_beam_history_async(st, g.first, g.count, g.slot);
    next_slot ^= 1;
    inflight.push_back(g);
    return true;
  };

  // Finished batch entries leave the session (src/decoding.cc:599-694 removes them at every step; see
  // greedy_search for the bulk policy): the entries in flight are replayed, then
  // Engine::compact_beam_entries packs what is per entry / per row and the ancestry tables.
  const int64_t compact_min_rows = compact_min_rows_setting();
  const bool may_compact = !gather_attention && compact_min_rows > 0 && engine.beam_compactable(st);
  std::vector<int64_t> origin(batch);  // session entry -> index into results / finished / alive_seq
  for (int64_t i = 0; i < batch; ++i) origin[i] = i;
  bool compact_pending = false;

  launch_group();
  while ((!inflight.empty() || compact_pending) && remaining > 0) {
    if (compact_pending && inflight.empty()) {
      std::vector<int32_t> keep;
      std::vector<int64_t> new_origin;
      for (int64_t e = 0; e < st.batch; ++e)
        if (!finished[origin[e]]) {
          keep.push_back(static_cast<int32_t>(e));
          new_origin.push_back(origin[e]);
        }
      engine.compact_beam_entries(st, keep, opt.start_step + launched);
      origin.swap(new_origin);
      compact_pending = false;
      if (stats) ++stats->compactions;
      if (!launch_group()) break;
      continue;
    }
    if (!compact_pending && inflight.size() < 2 && !row_rule_lockstep)
      launch_group();  // keep the device busy while this group is replayed
    const Group g = inflight.front();
    inflight.erase(inflight.begin());
    const int32_t* h_word = nullptr;
    const float* h_score = nullptr;
    const int32_t* h_origin = nullptr;
    engine.wait_beam_history(st, g.slot, &h_word, &h_score, &h_origin);
    const int64_t entries_now = st.batch;  // (no compaction while a group is in flight)
    if (gather_attention) {
      att_steps.resize(static_cast<size_t>(g.count) * rows * st.s_max);
      engine.read_attention(st, opt.start_step + g.first, g.count, att_steps.data());
    }
    for (int64_t s = 0; s < g.count && remaining > 0; ++s, ++step) {
      for (int64_t e = 0; e < entries_now; ++e) {
        const int64_t i = origin[e];
        if (finished[i]) continue;
        // candidates as the device selected from them: unflattened, hard prefix applied
        const int64_t base = (s * entries_now + e) * num_candidates;
        const int32_t* cand_word = h_word + base;
        const int32_t* cand_origin = h_origin + base;
        const float* sc = h_score + base;
        const int64_t plen = prefix_ids ? static_cast<int64_t>((*prefix_ids)[i].size()) : 0;
        const bool last = (step + 1 == opt.max_length + extra * plen);  // is_last_step
        // append_step_output: every candidate inherits its origin beam's history + its word.
        std::vector<std::vector<int32_t>> cand_seq(num_candidates);
        for (int32_t j = 0; j < num_candidates; ++j) {
          cand_seq[j] = alive_seq[i][cand_origin[j]];
          cand_seq[j].push_back(cand_word[j]);
        }
        // the attention of a candidate = its origin beam's history + that beam's vector of this step
        std::vector<std::vector<std::vector<float>>> cand_att;
        if (gather_attention) {
          cand_att.resize(num_candidates);
          for (int32_t j = 0; j < num_candidates; ++j) {
            cand_att[j] = alive_attention[i][cand_origin[j]];
            const float* a = att_steps.data() +
                             (static_cast<size_t>(s) * rows + e * beam + cand_origin[j]) * st.s_max;
            cand_att[j].emplace_back(a, a + st.s_max);
          }
        }

        BeamStepInput in;
        in.beam_size = beam;
        in.step = static_cast<int32_t>(step);
        in.is_last_step = last;
        in.end_id = end_id;
        in.word_ids = cand_word;
        in.scores = sc;
        in.prefix_length = static_cast<int32_t>(plen);
        const BeamStepOutput sel = beam_step_select(in);

        DecodingResult& result = results[i];
        for (int32_t k : sel.finished) {
          if (k == 0) top_beam_finished[i] = 1;
          const bool ignore_last = (cand_word[k] == end_id) && !opt.include_eos_in_hypotheses;
          // build_hypothesis(alive_seq, i, k, start, end): a hidden prefix is cut off
          std::vector<int32_t> hyp(cand_seq[k].begin() + std::min<int64_t>(extra * plen, cand_seq[k].size()),
                                   cand_seq[k].end());
          if (ignore_last && !hyp.empty()) hyp.pop_back();
          result.scores.push_back(sc[k]);
          if (gather_attention) {  // build_attention with the same start / end as the hypothesis
            const size_t a0 = std::min<size_t>(extra * plen, cand_att[k].size());
            std::vector<std::vector<float>> att(cand_att[k].begin() + a0, cand_att[k].end());
            if (ignore_last && !att.empty()) att.pop_back();
            result.attention.emplace_back(std::move(att));
          }
          result.hypotheses.emplace_back(std::move(hyp));
        }

        bool is_finished = false;
        if (last)
          is_finished = true;
        else if (allow_early_exit)
          is_finished = top_beam_finished[i] &&
                        result.hypotheses.size() >= static_cast<size_t>(opt.num_hypotheses);
        else
          is_finished = result.hypotheses.size() >= max_candidates;

        if (is_finished) {
          finalize(result, opt.num_hypotheses, opt);
          finished[i] = 1;
          --remaining;
          continue;
        }

        // the device made the same choice (beam_select_kernel); keep the host copy of the tokens
        std::vector<std::vector<int32_t>> new_alive(beam);
        for (int32_t k = 0; k < beam; ++k) new_alive[k] = std::move(cand_seq[sel.next[k]]);
        alive_seq[i] = std::move(new_alive);
        if (gather_attention) {
          std::vector<std::vector<std::vector<float>>> new_att(beam);
          for (int32_t k = 0; k < beam; ++k) new_att[k] = cand_att[sel.next[k]];
          alive_attention[i] = std::move(new_att);
        }
      }
    }
    if (may_compact && !compact_pending && remaining > 0 && st.rows >= compact_min_rows &&
        remaining * 2 <= st.batch)
      compact_pending = true;
    if (row_rule_lockstep && remaining > 0) {
      if (engine.beam_compactable(st) && !gather_attention && remaining < st.batch && remaining >= 1)
        compact_pending = true;
      else if (inflight.empty() && !compact_pending)
        launch_group();
    }
  }
  engine.synchronize();  // drain a speculative group before the state is torn down
  engine.check_cluster(st);
  st.fuse_beam = false;
  if (stats) stats->steps = step;
  return results;
}

}  // namespace ct2x
