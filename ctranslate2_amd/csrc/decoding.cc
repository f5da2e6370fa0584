#include "decoding.h"
#include "kernels.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <numeric>
#include <stdexcept>

namespace ct2x {

// validate_decoding_options (src/decoding.cc:1023-1063), the subset that applies here.
void validate_decoding_options(const DecodingOptions& o) {
  if (o.beam_size <= 0) throw std::invalid_argument("The beam size must be > 0");
  if (o.patience <= 0) throw std::invalid_argument("The patience factor must be > 0");
  if (o.num_hypotheses <= 0) throw std::invalid_argument("The number of hypotheses must be > 0");
  const int64_t max_candidates = std::llround(float(o.beam_size) * o.patience);
  if (o.num_hypotheses > max_candidates)
    throw std::invalid_argument("The number of hypotheses cannot be greater than "
                                "beam_size * patience");
  if (o.min_length > o.max_length)
    throw std::invalid_argument("The minimum decoding length is greater than "
                                "the maximum decoding length");
  if (o.max_length <= 0) throw std::invalid_argument("The maximum decoding length must be > 0");
  if (o.sampling_topk != 1) {
    if (o.beam_size != 1)
      throw std::invalid_argument("Random sampling (sampling_topk != 1) requires beam_size 1");
    // 0 = the whole vocabulary (the reference's sampling_topk <= 0), else the best k <= 64
    if (o.sampling_topk < 0 || o.sampling_topk > max_sampling_topk())
      throw std::invalid_argument("sampling_topk must be 0 (whole vocabulary) or between 1 and " +
                                  std::to_string(max_sampling_topk()) + " in this backend");
  }
  if (!(o.sampling_topp > 0.f && o.sampling_topp <= 1.f))
    throw std::invalid_argument("sampling_topp must be in (0, 1]");
  if (!(o.sampling_temperature > 0.f))
    throw std::invalid_argument("sampling_temperature must be > 0");
}

float finalize_score(float score, float length, float length_penalty) {
  // finalize_hypothesis_score (src/decoding.cc:192-207): score /= pow(length, length_penalty).
  return score / std::pow(length, length_penalty);
}

namespace {

// sort_hypotheses (src/decoding.cc:209-240).
void sort_and_trim(DecodingResult& r, size_t max_hypotheses, bool keep_scores) {
  std::vector<size_t> idx(r.hypotheses.size());
  std::iota(idx.begin(), idx.end(), 0);
  std::sort(idx.begin(), idx.end(), [&r](size_t a, size_t b) { return r.scores[a] > r.scores[b]; });
  if (max_hypotheses < idx.size()) idx.resize(max_hypotheses);
  DecodingResult out;
  for (size_t i : idx) {
    out.hypotheses.emplace_back(std::move(r.hypotheses[i]));
    if (keep_scores) out.scores.push_back(r.scores[i]);
  }
  r = std::move(out);
}

// finalize_result (src/decoding.cc:242-262).
void finalize(DecodingResult& r, size_t max_hypotheses, float length_penalty, bool keep_scores) {
  for (size_t i = 0; i < r.scores.size(); ++i)
    r.scores[i] = finalize_score(r.scores[i], static_cast<float>(r.hypotheses[i].size()),
                                 length_penalty);
  sort_and_trim(r, max_hypotheses, keep_scores);
}

}  // namespace

// GreedySearch::search (src/decoding.cc:733-974), num_hypotheses == 1, no prefix.
std::vector<DecodingResult> greedy_search(Engine& engine, DecodeState& st,
                                          const std::vector<int32_t>& start_ids, int32_t end_id,
                                          const DecodingOptions& opt, DecodeStats* stats,
                                          const std::vector<std::vector<int32_t>>* prefix_ids) {
  const int64_t batch = static_cast<int64_t>(start_ids.size());
  if (batch != st.rows) throw std::runtime_error("greedy_search: state/batch size mismatch");
  std::vector<DecodingResult> results(batch);
  std::vector<float> score_sum(batch, 0.f);
  std::vector<char> alive(batch, 1);
  for (auto& r : results) r.hypotheses.resize(1);
  int64_t count_alive = batch;

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
    for (; launched < target; ++launched) engine.step(st, opt.use_graph);
    engine.read_history_async(st, opt.start_step + g.first, g.count, g.slot);
    next_slot ^= 1;
    inflight.push_back(g);
    return true;
  };

  launch_group();
  while (!inflight.empty() && count_alive > 0) {
    if (inflight.size() < 2) launch_group();  // keep the device busy while we scan group k
    const Group g = inflight.front();
    inflight.erase(inflight.begin());
    const int32_t* h_ids = nullptr;
    const float* h_scores = nullptr;
    engine.wait_history(st, g.slot, &h_ids, &h_scores);
    for (int64_t s = 0; s < g.count && count_alive > 0; ++s) {
      const int64_t step = g.first + s;
      for (int64_t i = 0; i < batch; ++i) {
        if (!alive[i]) continue;
        // is_last_step (src/decoding.cc:376-382): max_length steps after the row's own prefix
        const bool last = (step + 1 == opt.max_length + extra * prefix_len[i]);
        const int32_t word = h_ids[s * batch + i];
        const bool in_prefix = step < prefix_len[i];
        const bool eos = (word == end_id) && !in_prefix;
        if ((!in_prefix || opt.return_prefix) && (!eos || opt.include_eos_in_hypotheses))
          results[i].hypotheses[0].push_back(word);
        if (opt.return_scores) score_sum[i] += h_scores[s * batch + i];  // forced steps add 0
        if (eos || last) {
          alive[i] = 0;
          --count_alive;
          results[i].scores.assign(1, score_sum[i]);
          finalize(results[i], 1, opt.length_penalty, opt.return_scores);
        }
      }
    }
    examined = g.first + g.count;
  }
  (void)examined;
  engine.synchronize();  // drain a speculative group before the state is torn down
  st.fuse_greedy = false;
  if (stats) stats->steps = launched;
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
std::vector<DecodingResult> beam_search(Engine& engine, DecodeState& st,
                                        const std::vector<int32_t>& start_ids, int32_t end_id,
                                        const DecodingOptions& opt, DecodeStats* stats,
                                        const std::vector<std::vector<int32_t>>* prefix_ids) {
  const int64_t batch = static_cast<int64_t>(start_ids.size());
  const int32_t beam = opt.beam_size;
  if (prefix_ids && !opt.return_prefix)
    throw std::invalid_argument("beam search with a hidden prefix (generation) is not supported");
  const int32_t num_candidates = 2 * beam;
  const int64_t rows = batch * beam;
  if (rows != st.rows || st.beam != beam)
    throw std::runtime_error("beam_search: state/batch size mismatch");
  const int64_t vocab = engine.vocab_size();
  const size_t max_candidates = static_cast<size_t>(std::llround(float(beam) * opt.patience));
  const bool allow_early_exit = (opt.length_penalty == 0);

  std::vector<DecodingResult> results(batch);
  std::vector<char> finished(batch, 0), top_beam_finished(batch, 0);
  int64_t remaining = batch;

  std::vector<int32_t> next_ids(rows), parents(rows);
  std::vector<float> beam_scores(rows);
  for (int64_t i = 0; i < batch; ++i)
    for (int32_t k = 0; k < beam; ++k) {
      next_ids[i * beam + k] = start_ids[i];
      beam_scores[i * beam + k] = (k == 0) ? 0.f : -FLT_MAX;  // numeric_limits<float>::lowest()
    }
  engine.set_ids(st, next_ids.data());
  engine.set_step(st, opt.start_step);
  engine.set_beam_scores(st, beam_scores.data());
  st.fuse_greedy = false;
  engine.set_logits_processing(st, opt.logits_processing);
  const bool processing = opt.logits_processing.active();
  std::vector<int32_t> flat_seq;

  // alive_seq[i][k]: tokens of active beam k of batch i.
  std::vector<std::vector<std::vector<int32_t>>> alive_seq(
      batch, std::vector<std::vector<int32_t>>(beam));
  std::vector<float> cand_scores(batch * num_candidates);
  std::vector<int32_t> cand_flat(batch * num_candidates);
  std::vector<int32_t> cand_word(num_candidates), cand_origin(num_candidates);
  const int32_t min_length_abs = opt.start_step + opt.min_length;

  int64_t step = 0;
  for (; step < opt.max_length && remaining > 0; ++step) {
    engine.step(st, opt.use_graph);
    if (processing) {
      // merge_batch_beam(alive_seq) (src/decoding.cc:518-525): every active beam has `step` tokens
      flat_seq.assign(static_cast<size_t>(rows) * step, 0);
      for (int64_t i = 0; i < batch; ++i)
        for (int32_t k = 0; k < beam; ++k) {
          const auto& seq = alive_seq[i][k];
          if (static_cast<int64_t>(seq.size()) == step)
            std::copy(seq.begin(), seq.end(), flat_seq.begin() + (i * beam + k) * step);
        }
      engine.process_logits_beam(st, flat_seq.data(), static_cast<int32_t>(step));
    }
    engine.beam_candidates(st, end_id, min_length_abs, cand_scores.data(), cand_flat.data());
    const bool last = (step + 1 == opt.max_length);

    for (int64_t i = 0; i < batch; ++i) {
      if (finished[i]) {
        for (int32_t k = 0; k < beam; ++k) {
          parents[i * beam + k] = static_cast<int32_t>(i * beam + k);
          // ids / scores of finished batches are irrelevant: keep the rows computable.
          next_ids[i * beam + k] = end_id;
          beam_scores[i * beam + k] = 0.f;
        }
        continue;
      }
      float* sc = cand_scores.data() + i * num_candidates;
      const int32_t* flat = cand_flat.data() + i * num_candidates;
      // unflatten_ids (src/decoding.cc:95-116)
      for (int32_t j = 0; j < num_candidates; ++j) {
        cand_origin[j] = static_cast<int32_t>(flat[j] / vocab);
        cand_word[j] = static_cast<int32_t>(flat[j] % vocab);
      }
      // update_sample_with_prefix, beam arm (src/decoding.cc:21-66): inside the prefix every
      // candidate becomes the prefix token, only the first keeps a live score (0: the prefix does
      // not count) and all descend from beam 0; at the first free step an EOS in a secondary
      // candidate is neutralised.
      const int64_t plen = prefix_ids ? static_cast<int64_t>((*prefix_ids)[i].size()) : 0;
      if (prefix_ids && step <= plen) {
        for (int32_t j = 0; j < num_candidates; ++j) {
          if (step < plen) {
            cand_word[j] = (*prefix_ids)[i][step];
            sc[j] = (j == 0) ? 0.f : -1e10f;
            cand_origin[j] = 0;
          } else if (j > 0 && cand_word[j] == end_id) {
            cand_word[j] = 0;
            sc[j] = -1e10f;
            cand_origin[j] = 0;
          }
        }
      }
      // append_step_output: every candidate inherits its origin beam's history + its word.
      std::vector<std::vector<int32_t>> cand_seq(num_candidates);
      for (int32_t j = 0; j < num_candidates; ++j) {
        cand_seq[j] = alive_seq[i][cand_origin[j]];
        cand_seq[j].push_back(cand_word[j]);
      }

      BeamStepInput in;
      in.beam_size = beam;
      in.step = static_cast<int32_t>(step);
      in.is_last_step = last;
      in.end_id = end_id;
      in.word_ids = cand_word.data();
      in.scores = sc;
      in.prefix_length = static_cast<int32_t>(plen);
      const BeamStepOutput sel = beam_step_select(in);

      DecodingResult& result = results[i];
      for (int32_t k : sel.finished) {
        if (k == 0) top_beam_finished[i] = 1;
        const bool ignore_last = (cand_word[k] == end_id) && !opt.include_eos_in_hypotheses;
        std::vector<int32_t> hyp = cand_seq[k];
        if (ignore_last) hyp.pop_back();
        result.scores.push_back(sc[k]);
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
        finalize(result, opt.num_hypotheses, opt.length_penalty, opt.return_scores);
        finished[i] = 1;
        --remaining;
        for (int32_t k = 0; k < beam; ++k) {
          parents[i * beam + k] = static_cast<int32_t>(i * beam + k);
          next_ids[i * beam + k] = end_id;
          beam_scores[i * beam + k] = 0.f;
        }
        continue;
      }

      std::vector<std::vector<int32_t>> new_alive(beam);
      for (int32_t k = 0; k < beam; ++k) {
        const int32_t j = sel.next[k];
        new_alive[k] = std::move(cand_seq[j]);
        next_ids[i * beam + k] = cand_word[j];
        beam_scores[i * beam + k] = sc[j];
        parents[i * beam + k] = static_cast<int32_t>(i * beam + cand_origin[j]);
        // a candidate selected twice cannot happen: `next` indices are distinct by construction
      }
      alive_seq[i] = std::move(new_alive);
    }

    if (remaining == 0) {
      ++step;
      break;
    }
    engine.beam_update(st, next_ids.data(), parents.data(), beam_scores.data());
  }
  if (stats) stats->steps = step;
  return results;
}

}  // namespace ct2x
