// Level 1.5: the reference's OWN ctranslate2::Translator on the FUSED engine.
//
// Level 1 (ops_mi355_stub.cc & co.) lets the reference's front-end run op by op on this backend:
// identical API, ~400 launches per decode step. This translation unit keeps the API identical and
// moves the model-level entry - EncoderDecoderReplica::run_translation
// (src/models/sequence_to_sequence.cc:305-432) - onto ct2x_translate_batch, the device-resident
// engine the benchmark measures: `ctranslate2::Translator(model, Device::CUDA)` then decodes at the
// engine's speed with TranslationOptions / TranslationResult as they are.
//
// How, without touching a reference source: run_translation is a virtual of
// SequenceToSequenceReplica, and the replica object is made in ONE place,
// TransformerModel::as_sequence_to_sequence (src/models/transformer.cc:82-90). This unit is compiled
// IN PLACE OF src/models/transformer.cc: it includes that file where it lies, with the class name in
// its `std::make_unique<EncoderDecoderReplica>(...)` mapped to the subclass below. In a real
// integration the same is a three-line change in transformer.cc.
//
// The subclass hands the model it finds in the reference's memory (already converted to the compute
// type by Model::load) to ct2x_translator_create_from_memory once per (model, device), and serves
// every call whose options the fused engine implements with the reference's own semantics. Round 6:
// that is the whole of TranslationOptions - target prefixes (hard, or a bias with
// prefix_bias_beta), random sampling, return_attention, replace_unknowns, the step callback,
// num_hypotheses / return_alternatives, a custom end token, return_logits_vocab of a greedy search.
// What still falls through to the reference's own op-by-op path, unchanged: a vocabulary map
// (use_vmap with a map loaded), SEVERAL end tokens, return_logits_vocab under beam search, decoder
// start tokens given by the user, several source features, tensor parallelism.
// CT2X_LEVEL1_FUSED=0 sends every call there (A/B: tools/level1_speed.py);
// CT2X_LEVEL15_REPORT=<file> writes "served <n> fallback <m>" when the process ends.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "ctranslate2/layers/transformer.h"
#include "ctranslate2/models/language_model.h"
#include "ctranslate2/models/sequence_to_sequence.h"
#include "ctranslate2/models/transformer.h"

#include "ct2_mi355.h"

namespace ctranslate2 {
  namespace models {

    namespace {

      struct FusedHandle {
        void* h = nullptr;
        ~FusedHandle() {
          if (h) ct2x_translator_destroy(h);
        }
      };

      // one engine per (model object, device): the replicas of a pool share their model, so they
      // share the engine as well (its calls are serialised per handle, like a replica's)
      std::mutex g_registry_mutex;
      std::map<const Model*, std::weak_ptr<FusedHandle>> g_registry;

      const char* compute_type_name(ComputeType t) {
        switch (t) {
          case ComputeType::FLOAT32: return "float32";
          case ComputeType::INT8: return "int8";
          case ComputeType::INT8_FLOAT32: return "int8_float32";
          case ComputeType::INT8_FLOAT16: return "int8_float16";
          case ComputeType::INT8_BFLOAT16: return "int8_bfloat16";
          case ComputeType::FLOAT16: return "float16";
          case ComputeType::BFLOAT16: return "bfloat16";
          default: return nullptr;
        }
      }

      // `source` may be null (decoder-only models, or a shared vocabulary)
      std::shared_ptr<FusedHandle> make_handle(const Model& model, const char* spec, const Vocabulary* source,
                                               const Vocabulary& target) {
        const char* ct = compute_type_name(model.effective_compute_type());
        if (ct == nullptr) return nullptr;
        // the variables as Model::load left them (converted to the compute type), copied to the host
        std::vector<std::string> names;
        std::vector<StorageView> host;
        std::vector<std::vector<int64_t>> shapes;
        for (auto& item : model.get_variables()) {
          names.push_back(item.first);
          host.emplace_back(item.second.to(Device::CPU));
          const auto& shape = host.back().shape();
          shapes.emplace_back(shape.begin(), shape.end());
        }
        std::vector<ct2x_memory_variable> vars(names.size());
        for (size_t i = 0; i < names.size(); ++i) {
          vars[i].name = names[i].c_str();
          vars[i].dtype = static_cast<int32_t>(host[i].dtype());
          vars[i].rank = static_cast<int32_t>(shapes[i].size());
          vars[i].shape = shapes[i].data();
          vars[i].data = host[i].buffer();
          vars[i].nbytes = static_cast<uint64_t>(host[i].size()) * host[i].item_size();
        }
        auto tokens_of = [](const Vocabulary& v, std::vector<std::string>& store,
                            std::vector<const char*>& ptrs) {
          store.reserve(v.size());
          for (size_t i = 0; i < v.size(); ++i) store.push_back(v.to_token(i));
          for (const auto& s : store) ptrs.push_back(s.c_str());
        };
        std::vector<std::string> src_store, tgt_store;
        std::vector<const char*> src, tgt;
        if (source != nullptr) tokens_of(*source, src_store, src);
        tokens_of(target, tgt_store, tgt);
        const std::string config = model.config.dump();
        auto handle = std::make_shared<FusedHandle>();
        handle->h = ct2x_translator_create_from_memory(
            spec, static_cast<uint32_t>(model.spec_revision()), static_cast<uint32_t>(model.binary_version()),
            config.c_str(), vars.data(), static_cast<int64_t>(vars.size()), source ? src.data() : nullptr,
            static_cast<int64_t>(src.size()), tgt.data(), static_cast<int64_t>(tgt.size()), model.device_index(), ct);
        if (handle->h == nullptr) return nullptr;  // (a model this engine does not take: Level 1 serves it)
        return handle;
      }

      // the engine of a model object: made once, shared by the replicas of a pool
      template <typename Make>
      std::shared_ptr<FusedHandle> handle_of(const Model& base, Make make) {
        std::lock_guard<std::mutex> lock(g_registry_mutex);
        auto& slot = g_registry[&base];
        auto handle = slot.lock();
        if (!handle) {
          const auto scoped_device_setter = base.get_scoped_device_setter();
          handle = make();
          slot = handle;
        }
        return handle;
      }

      std::atomic<int64_t> g_served_calls{0};
      std::atomic<int64_t> g_fallback_calls{0};
      std::atomic<uint32_t> g_sampling_calls{0};

      struct ExitReport {
        ~ExitReport() {
          const char* path = std::getenv("CT2X_LEVEL15_REPORT");
          if (path == nullptr || path[0] == 0) return;
          if (FILE* f = std::fopen(path, "w")) {
            std::fprintf(f, "served %lld fallback %lld\n", static_cast<long long>(g_served_calls.load()),
                         static_cast<long long>(g_fallback_calls.load()));
            std::fclose(f);
          }
        }
      } g_exit_report;

      // TranslationOptions / GenerationOptions::callback behind the C ABI's step callback
      struct CallbackBridge {
        const std::function<bool(GenerationStepResult)>* callback;
        const Vocabulary* vocabulary;
        std::exception_ptr error;  // an exception of the user's callable: rethrown after the call
      };
      int step_trampoline(void* user, int64_t step, int64_t batch_id, int64_t hypothesis_id, int32_t token_id,
                          int32_t has_score, float score, int32_t is_last) {
        auto* bridge = static_cast<CallbackBridge*>(user);
        if (bridge->error) return 1;
        try {
          DecodingStepResult r;
          r.step = static_cast<size_t>(step);
          r.batch_id = static_cast<size_t>(batch_id);
          r.token_id = static_cast<size_t>(token_id);
          r.hypothesis_id = static_cast<size_t>(hypothesis_id);
          if (has_score) r.score = score;
          r.is_last = is_last != 0;
          return (*bridge->callback)(GenerationStepResult(r, *bridge->vocabulary)) ? 1 : 0;
        } catch (...) {  // must not unwind through the C frames
          bridge->error = std::current_exception();
          return 1;
        }
      }

      // the options both front-ends share (sampling: the reference draws from a process-wide
      // generator - here a counter-based one seeded per call, reproducible within a process)
      template <typename Options>
      void set_sampling(ct2x_translation_options& opt, const Options& options) {
        opt.sampling_topk = options.sampling_topk == 0 ? -1 : static_cast<int32_t>(options.sampling_topk);
        opt.sampling_topp = options.sampling_topp;
        opt.sampling_temperature = options.sampling_temperature;
        if (options.sampling_topk != 1) opt.seed = 0x9e3779b9u * (g_sampling_calls.fetch_add(1) + 1);
      }

      bool fused_enabled() {
        static const bool on = [] {
          const char* e = std::getenv("CT2X_LEVEL1_FUSED");
          return e == nullptr || e[0] != '0';
        }();
        return on;
      }

    }  // namespace

    class Mi355EncoderDecoderReplica : public EncoderDecoderReplica {
    public:
      using EncoderDecoderReplica::EncoderDecoderReplica;

    protected:
      std::vector<TranslationResult>
      run_translation(const std::vector<std::vector<std::string>>& source,
                      const std::vector<std::vector<std::string>>& target_prefix,
                      const TranslationOptions& options) override {
        if (!serves(source, target_prefix, options)) {
          ++g_fallback_calls;
          return EncoderDecoderReplica::run_translation(source, target_prefix, options);
        }
        const auto& m = static_cast<const SequenceToSequenceModel&>(*model());
        const Vocabulary& src_vocab = m.get_source_vocabulary();
        const Vocabulary& tgt_vocab = m.get_target_vocabulary();
        const size_t batch = source.size();
        // raw ids: the engine adds the implicit BOS / EOS and truncates to max_input_length the way
        // Vocabulary::to_ids does (ct2x_translate_batch)
        const auto ids = src_vocab.to_ids(source, /*max_length=*/0, /*add_bos=*/false, /*add_eos=*/false);
        size_t stride = 1;
        for (const auto& row : ids) stride = std::max(stride, row.size());
        std::vector<int32_t> flat(batch * stride, 0), lengths(batch);
        for (size_t b = 0; b < batch; ++b) {
          lengths[b] = static_cast<int32_t>(ids[b].size());
          for (size_t i = 0; i < ids[b].size(); ++i) flat[b * stride + i] = static_cast<int32_t>(ids[b][i]);
        }
        ct2x_translation_options opt;
        ct2x_translation_options_default(&opt);
        opt.beam_size = static_cast<int32_t>(options.beam_size);
        opt.patience = options.patience;
        opt.length_penalty = options.length_penalty;
        opt.coverage_penalty = options.coverage_penalty;
        opt.repetition_penalty = options.repetition_penalty;
        opt.no_repeat_ngram_size = static_cast<int32_t>(options.no_repeat_ngram_size);
        opt.max_decoding_length = static_cast<int32_t>(options.max_decoding_length);
        opt.min_decoding_length = static_cast<int32_t>(options.min_decoding_length);
        opt.max_input_length = static_cast<int32_t>(options.max_input_length);
        opt.num_hypotheses = static_cast<int32_t>(options.num_hypotheses);
        opt.return_scores = options.return_scores ? 1 : 0;
        opt.return_end_token = options.return_end_token ? 1 : 0;
        opt.max_batch_size = 0;  // (the Translator above has already cut the batches)
        // disable_unk / suppress_sequences as id sequences (make_logits_processors, src/decoding.cc:1091-1120)
        std::vector<int32_t> sup_ids, sup_lengths;
        for (const auto& seq : tgt_vocab.to_ids(options.suppress_sequences, 0, nullptr, nullptr, false)) {
          sup_lengths.push_back(static_cast<int32_t>(seq.size()));
          for (size_t id : seq) sup_ids.push_back(static_cast<int32_t>(id));
        }
        if (options.disable_unk) {
          sup_lengths.push_back(1);
          sup_ids.push_back(static_cast<int32_t>(tgt_vocab.unk_id()));
        }
        opt.num_suppress_sequences = static_cast<int32_t>(sup_lengths.size());
        opt.suppress_ids = sup_ids.data();
        opt.suppress_lengths = sup_lengths.data();
        set_sampling(opt, options);
        opt.return_alternatives = options.return_alternatives ? 1 : 0;
        opt.min_alternative_expansion_prob = options.min_alternative_expansion_prob;
        opt.prefix_bias_beta = options.prefix_bias_beta;
        // end_token (ResolveEndToken, include/ctranslate2/generation.h:135-162): one id (serves() checked)
        const auto end_ids = std::visit(ResolveEndToken(tgt_vocab), options.end_token);
        if (end_ids[0] != tgt_vocab.eos_id()) opt.end_token_plus_one = static_cast<int32_t>(end_ids[0]) + 1;

        const size_t nh = options.num_hypotheses, max_len = options.max_decoding_length;
        // return_attention || replace_unknowns (src/models/sequence_to_sequence.cc:352): one vector per
        // generated token over the RAW source positions (the engine drops the implicit BOS / EOS)
        const bool want_attention = options.return_attention || options.replace_unknowns;
        std::vector<float> attention;
        if (want_attention) {
          attention.assign(batch * nh * max_len * stride, 0.f);
          opt.attention_stride = static_cast<int32_t>(stride);
          opt.attention_out = attention.data();
        }
        CallbackBridge bridge{&options.callback, &tgt_vocab, nullptr};
        if (options.callback) {
          opt.step_callback = step_trampoline;
          opt.step_callback_data = &bridge;
        }
        // return_logits_vocab (greedy search / sampling: serves() checked)
        const size_t vocab_size = tgt_vocab.size();
        std::vector<float> logits;
        std::vector<int32_t> logits_counts;
        if (options.return_logits_vocab) {
          logits.resize(batch * nh * max_len * vocab_size);
          logits_counts.assign(batch * nh, 0);
          opt.logits_out = logits.data();
          opt.logits_counts_out = logits_counts.data();
          opt.logits_steps = static_cast<int32_t>(max_len);
        }
        // target_prefix (make_target_ids with is_prefix, :168-186): raw ids after the decoder start token
        bool any_prefix = false;
        for (const auto& prefix : target_prefix) any_prefix = any_prefix || !prefix.empty();
        std::vector<int32_t> out_ids(batch * nh * max_len), out_lengths(batch * nh);
        std::vector<float> out_scores(batch * nh);
        int rc;
        if (any_prefix) {
          const auto pids = tgt_vocab.to_ids(target_prefix, 0, nullptr, nullptr);
          size_t pstride = 1;
          for (const auto& row : pids) pstride = std::max(pstride, row.size());
          std::vector<int32_t> pflat(batch * pstride, 0), plengths(batch, 0);
          for (size_t b = 0; b < pids.size() && b < batch; ++b) {
            plengths[b] = static_cast<int32_t>(pids[b].size());
            for (size_t i = 0; i < pids[b].size(); ++i) pflat[b * pstride + i] = static_cast<int32_t>(pids[b][i]);
          }
          rc = ct2x_translate_batch_prefix(_handle->h, flat.data(), lengths.data(), static_cast<int32_t>(batch),
                                           static_cast<int32_t>(stride), pflat.data(), plengths.data(),
                                           static_cast<int32_t>(pstride), &opt, out_ids.data(), out_lengths.data(),
                                           out_scores.data());
        } else {
          rc = ct2x_translate_batch(_handle->h, flat.data(), lengths.data(), static_cast<int32_t>(batch),
                                    static_cast<int32_t>(stride), &opt, out_ids.data(), out_lengths.data(),
                                    out_scores.data());
        }
        if (bridge.error) std::rethrow_exception(bridge.error);
        if (rc != 0) throw std::runtime_error(std::string("MI355X fused engine: ") + ct2x_last_error());
        ++g_served_calls;

        std::vector<TranslationResult> results;
        results.reserve(batch);
        for (size_t b = 0; b < batch; ++b) {
          std::vector<std::vector<size_t>> hyp_ids;
          std::vector<float> scores;
          std::vector<std::vector<std::vector<float>>> att;
          std::vector<std::vector<StorageView>> hyp_logits;
          for (size_t h = 0; h < nh; ++h) {
            const int32_t len = out_lengths[b * nh + h];
            // (fewer hypotheses than asked for: -1 from decode_alternatives, an empty one past the first otherwise)
            if (len < 0 || (!options.return_alternatives && h > 0 && len == 0)) continue;
            const int32_t* row = out_ids.data() + (b * nh + h) * max_len;
            hyp_ids.emplace_back(row, row + len);
            if (options.return_scores) scores.push_back(out_scores[b * nh + h]);
            if (want_attention) {
              // (:395-407: the vector covers the original input, zero past what the encoder saw)
              att.emplace_back();
              for (int32_t t = 0; t < len; ++t) {
                const float* a = attention.data() + ((b * nh + h) * max_len + t) * stride;
                std::vector<float> v(source[b].size(), 0.f);
                std::copy(a, a + std::min<size_t>(v.size(), static_cast<size_t>(lengths[b])), v.begin());
                att.back().push_back(std::move(v));
              }
            }
            if (options.return_logits_vocab) {
              hyp_logits.emplace_back();
              for (int32_t t = 0; t < logits_counts[b * nh + h]; ++t) {
                const float* l = logits.data() + ((b * nh + h) * max_len + t) * vocab_size;
                hyp_logits.back().emplace_back(Shape{static_cast<dim_t>(vocab_size)},
                                               std::vector<float>(l, l + vocab_size));
              }
            }
          }
          auto hypotheses = tgt_vocab.to_tokens(hyp_ids);
          if (options.replace_unknowns) {  // replace_unknown_tokens (:288-302): first maximum
            for (size_t h = 0; h < hypotheses.size(); ++h)
              for (size_t t = 0; t < hypotheses[h].size() && t < att[h].size(); ++t)
                if (hypotheses[h][t] == tgt_vocab.unk_token() && !att[h][t].empty())
                  hypotheses[h][t] = source[b][std::distance(att[h][t].begin(),
                                                             std::max_element(att[h][t].begin(), att[h][t].end()))];
          }
          if (!options.return_attention) att.clear();
          results.emplace_back(std::move(hypotheses), std::move(scores), std::move(att), std::move(hyp_logits));
        }
        return results;
      }

    private:
      std::shared_ptr<FusedHandle> _handle;
      bool _tried = false;

      bool serves(const std::vector<std::vector<std::string>>& source,
                  const std::vector<std::vector<std::string>>& target_prefix,
                  const TranslationOptions& options) {
        const Model& base = *model();
        if (!fused_enabled() || base.device() != Device::CUDA || base.tensor_parallel() || source.empty())
          return false;
        const auto* m = dynamic_cast<const SequenceToSequenceModel*>(&base);
        if (m == nullptr || m->num_source_vocabularies() != 1) return false;
        // (legacy specs carry their head count in the spec NAME, which a loaded model no longer knows)
        if (!base.get_variable_if_exists("num_heads") && !base.get_variable_if_exists("decoder/num_heads"))
          return false;
        bool any_prefix = false;
        for (const auto& prefix : target_prefix) any_prefix = any_prefix || !prefix.empty();
        // (a decoder start token given by the user is the first prefix token: the reference's path)
        if (any_prefix && m->decoder_start_token() == nullptr) return false;
        if (options.use_vmap && m->get_vocabulary_map() != nullptr) return false;
        if (options.return_logits_vocab && options.beam_size != 1) return false;
        // (invalid combinations keep the reference's own error texts)
        if (options.num_hypotheses < 1 || options.beam_size < 1 ||
            (options.num_hypotheses > options.beam_size && !options.return_alternatives && options.sampling_topk == 1) ||
            (options.callback && options.beam_size != 1) || options.max_decoding_length == 0 ||
            (options.prefix_bias_beta > 0 && (options.prefix_bias_beta >= 1 || options.return_alternatives)))
          return false;
        // ONE end token (the engine stops on one id); an unknown token keeps the reference's error
        try {
          const auto& tgt = m->get_target_vocabulary();
          if (std::visit(ResolveEndToken(tgt), options.end_token).size() != 1) return false;
        } catch (...) {
          return false;
        }
        if (!_tried) {
          _tried = true;
          _handle = handle_of(base, [&] {
            return make_handle(base, "TransformerSpec", &m->get_source_vocabulary(), m->get_target_vocabulary());
          });
        }
        return _handle != nullptr;
      }
    };

    // The same for ctranslate2::Generator: DecoderReplica::run_generation
    // (src/models/language_model.cc:150-282) with include_prompt_in_result = false - the prompt is
    // consumed by one sequence-mode pass, ragged prompts continue as hard prefixes - on
    // ct2x_generate_batch, with sampling, the step callback, a custom end token and
    // return_alternatives (round 6) - and with include_prompt_in_result = true, the reference's
    // default: the prompt decoded step by step as a returned hard prefix on the fused engine, the
    // start token put in front here (language_model.cc:262-267), and with a static prompt (forwarded
    // with the rest; its cached state is the reference's optimisation, not made here). Logits and the
    // prompt in the result / a static prompt together with return_alternatives stay on the reference's
    // own path.
    class Mi355DecoderReplica : public DecoderReplica {
    public:
      using DecoderReplica::DecoderReplica;

    protected:
      std::vector<GenerationResult>
      run_generation(const std::vector<std::vector<std::string>>& start_tokens,
                     const GenerationOptions& options) override {
        if (!serves(start_tokens, options)) {
          ++g_fallback_calls;
          return DecoderReplica::run_generation(start_tokens, options);
        }
        const auto& m = static_cast<const LanguageModel&>(*model());
        const Vocabulary& vocab = m.get_vocabulary();
        const size_t batch = start_tokens.size();
        // static_prompt + start_tokens is what the model consumes (language_model.cc:184-206): the
        // static tokens lead every row, ct2x_translation_options.static_prompt_length says how many
        const auto start_ids = vocab.to_ids(start_tokens);
        std::vector<size_t> static_ids;
        for (const auto& token : options.static_prompt) static_ids.push_back(vocab.to_id(token));
        std::vector<std::vector<size_t>> ids(batch, static_ids);
        for (size_t b = 0; b < batch; ++b) ids[b].insert(ids[b].end(), start_ids[b].begin(), start_ids[b].end());
        size_t stride = 1;
        for (const auto& row : ids) stride = std::max(stride, row.size());
        std::vector<int32_t> flat(batch * stride, 0), lengths(batch);
        for (size_t b = 0; b < batch; ++b) {
          lengths[b] = static_cast<int32_t>(ids[b].size());
          for (size_t i = 0; i < ids[b].size(); ++i) flat[b * stride + i] = static_cast<int32_t>(ids[b][i]);
        }
        ct2x_translation_options opt;
        ct2x_translation_options_default(&opt);
        opt.beam_size = static_cast<int32_t>(options.beam_size);
        opt.patience = options.patience;
        opt.length_penalty = options.length_penalty;
        opt.repetition_penalty = options.repetition_penalty;
        opt.no_repeat_ngram_size = static_cast<int32_t>(options.no_repeat_ngram_size);
        opt.max_decoding_length = static_cast<int32_t>(options.max_length);
        opt.min_decoding_length = static_cast<int32_t>(options.min_length);
        opt.num_hypotheses = static_cast<int32_t>(options.num_hypotheses);
        opt.return_scores = options.return_scores ? 1 : 0;
        opt.return_end_token = options.return_end_token ? 1 : 0;
        std::vector<int32_t> sup_ids, sup_lengths;
        for (const auto& seq : vocab.to_ids(options.suppress_sequences, 0, nullptr, nullptr, false)) {
          sup_lengths.push_back(static_cast<int32_t>(seq.size()));
          for (size_t id : seq) sup_ids.push_back(static_cast<int32_t>(id));
        }
        if (options.disable_unk) {
          sup_lengths.push_back(1);
          sup_ids.push_back(static_cast<int32_t>(vocab.unk_id()));
        }
        opt.num_suppress_sequences = static_cast<int32_t>(sup_lengths.size());
        opt.suppress_ids = sup_ids.data();
        opt.suppress_lengths = sup_lengths.data();
        // round 6: sampling, the step callback, a custom end token, return_alternatives
        set_sampling(opt, options);
        opt.return_alternatives = options.return_alternatives ? 1 : 0;
        opt.min_alternative_expansion_prob = options.min_alternative_expansion_prob;
        opt.include_prompt_in_result = options.include_prompt_in_result ? 1 : 0;
        opt.static_prompt_length = static_cast<int32_t>(static_ids.size());
        const auto end_ids = std::visit(ResolveEndToken(vocab), options.end_token);
        if (end_ids[0] != vocab.eos_id()) opt.end_token_plus_one = static_cast<int32_t>(end_ids[0]) + 1;
        CallbackBridge bridge{&options.callback, &vocab, nullptr};
        if (options.callback) {
          opt.step_callback = step_trampoline;
          opt.step_callback_data = &bridge;
        }

        const size_t nh = options.num_hypotheses, max_len = options.max_length;
        std::vector<int32_t> out_ids(batch * nh * max_len), out_lengths(batch * nh);
        std::vector<float> out_scores(batch * nh);
        // return_logits_vocab (greedy search / sampling: serves() checked): one vector per step an
        // entry ran - the forced steps of a longer prompt and the end token's step included
        const size_t vocab_size = vocab.size();
        size_t logits_steps = 0;
        std::vector<float> logits;
        std::vector<int32_t> logits_counts;
        if (options.return_logits_vocab) {
          size_t shortest = ids[0].size(), longest = ids[0].size();
          for (const auto& row : ids) {
            shortest = std::min(shortest, row.size());
            longest = std::max(longest, row.size());
          }
          logits_steps = max_len + (longest - shortest) + 1;
          logits.resize(batch * nh * logits_steps * vocab_size);
          logits_counts.assign(batch * nh, 0);
          opt.logits_out = logits.data();
          opt.logits_counts_out = logits_counts.data();
          opt.logits_steps = static_cast<int32_t>(logits_steps);
        }
        const int rc = ct2x_generate_batch(_handle->h, flat.data(), lengths.data(), static_cast<int32_t>(batch),
                                           static_cast<int32_t>(stride), &opt, out_ids.data(), out_lengths.data(),
                                           out_scores.data());
        if (bridge.error) std::rethrow_exception(bridge.error);
        if (rc != 0) throw std::runtime_error(std::string("MI355X fused engine: ") + ct2x_last_error());
        ++g_served_calls;

        std::vector<GenerationResult> results(batch);
        for (size_t b = 0; b < batch; ++b) {
          for (size_t h = 0; h < nh; ++h) {
            const int32_t len = out_lengths[b * nh + h];
            if (len < 0 || (!options.return_alternatives && h > 0 && len == 0)) continue;  // (fewer hypotheses than asked for)
            const int32_t* row = out_ids.data() + (b * nh + h) * max_len;
            results[b].sequences_ids.emplace_back(row, row + std::min<int32_t>(len, static_cast<int32_t>(max_len)));
            // "Forward the start token to the output if it is not the special BOS token"
            if (options.include_prompt_in_result && start_ids[b][0] != vocab.bos_id())
              results[b].sequences_ids.back().insert(results[b].sequences_ids.back().begin(), start_ids[b][0]);
            if (options.return_scores) results[b].scores.push_back(out_scores[b * nh + h]);
            if (options.return_logits_vocab) {
              results[b].logits.emplace_back();
              for (int32_t t = 0; t < logits_counts[b * nh + h]; ++t) {
                const float* l = logits.data() + ((b * nh + h) * logits_steps + t) * vocab_size;
                results[b].logits.back().emplace_back(Shape{static_cast<dim_t>(vocab_size)},
                                                      std::vector<float>(l, l + vocab_size));
              }
            }
          }
          results[b].sequences = vocab.to_tokens(results[b].sequences_ids);
        }
        return results;
      }

    private:
      std::shared_ptr<FusedHandle> _handle;
      bool _tried = false;

      bool serves(const std::vector<std::vector<std::string>>& start_tokens, const GenerationOptions& options) {
        const Model& base = *model();
        if (!fused_enabled() || base.device() != Device::CUDA || base.tensor_parallel() || start_tokens.empty())
          return false;
        const auto* m = dynamic_cast<const LanguageModel*>(&base);
        if (m == nullptr) return false;
        for (const auto& prompt : start_tokens)
          if (prompt.empty()) return false;
        // (logits under beam search, the prompt in the result or a static prompt together with
        // return_alternatives: the reference's own path)
        if ((options.include_prompt_in_result && options.return_alternatives) ||
            (!options.static_prompt.empty() && options.return_alternatives) ||
            (options.return_logits_vocab && (options.beam_size != 1 || options.return_alternatives)) ||
            options.num_hypotheses < 1 || options.beam_size < 1 || options.max_length == 0 ||
            (options.num_hypotheses > options.beam_size && !options.return_alternatives && options.sampling_topk == 1) ||
            (options.callback && options.beam_size != 1) || (options.return_alternatives && options.sampling_topk != 1))
          return false;
        try {
          if (std::visit(ResolveEndToken(m->get_vocabulary()), options.end_token).size() != 1) return false;
        } catch (...) {
          return false;
        }
        if (!base.get_variable_if_exists("num_heads") && !base.get_variable_if_exists("decoder/num_heads"))
          return false;
        if (!_tried) {
          _tried = true;
          _handle = handle_of(base, [&] { return make_handle(base, "TransformerDecoderSpec", nullptr, m->get_vocabulary()); });
        }
        return _handle != nullptr;
      }
    };

  }
}

// run_translation calls served by the fused engine in this process (tests, tools/level1_speed.py)
extern "C" int64_t ct2x_level15_served_calls(void) {
  return ctranslate2::models::g_served_calls.load();
}

// The reference's translation unit, where it lies, with the replica it instantiates mapped to the
// subclass above (every header it includes has been included already: the mapping only reaches
// the file's own text - their uses there are the make_unique of as_sequence_to_sequence and of
// as_sequence_generator).
#define EncoderDecoderReplica Mi355EncoderDecoderReplica
#define DecoderReplica Mi355DecoderReplica
#include "models/transformer.cc"
#undef DecoderReplica
#undef EncoderDecoderReplica
