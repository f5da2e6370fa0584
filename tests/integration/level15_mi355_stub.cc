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
// every call whose options the fused engine implements with the reference's own semantics; anything
// else - target prefixes, vocabulary maps, return_attention, callbacks, sampling, custom end tokens,
// several source features - falls through to the reference's own op-by-op path unchanged.
// CT2X_LEVEL1_FUSED=0 sends every call there (A/B: tools/level1_speed.py).
#include <atomic>
#include <cstdlib>
#include <cstring>
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
        if (!serves(source, target_prefix, options))
          return EncoderDecoderReplica::run_translation(source, target_prefix, options);
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

        const size_t nh = options.num_hypotheses, max_len = options.max_decoding_length;
        std::vector<int32_t> out_ids(batch * nh * max_len), out_lengths(batch * nh);
        std::vector<float> out_scores(batch * nh);
        if (ct2x_translate_batch(_handle->h, flat.data(), lengths.data(), static_cast<int32_t>(batch),
                                 static_cast<int32_t>(stride), &opt, out_ids.data(), out_lengths.data(),
                                 out_scores.data()) != 0)
          throw std::runtime_error(std::string("MI355X fused engine: ") + ct2x_last_error());
        ++g_served_calls;

        std::vector<TranslationResult> results;
        results.reserve(batch);
        for (size_t b = 0; b < batch; ++b) {
          std::vector<std::vector<size_t>> hyp_ids;
          std::vector<float> scores;
          for (size_t h = 0; h < nh; ++h) {
            const int32_t len = out_lengths[b * nh + h];
            if (len < 0) continue;  // (fewer hypotheses than asked for)
            const int32_t* row = out_ids.data() + (b * nh + h) * max_len;
            hyp_ids.emplace_back(row, row + len);
            if (options.return_scores) scores.push_back(out_scores[b * nh + h]);
          }
          results.emplace_back(tgt_vocab.to_tokens(hyp_ids), std::move(scores),
                               std::vector<std::vector<std::vector<float>>>(),
                               std::vector<std::vector<StorageView>>());
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
        for (const auto& prefix : target_prefix)
          if (!prefix.empty()) return false;
        if (options.use_vmap || options.return_attention || options.replace_unknowns ||
            options.return_logits_vocab || options.return_alternatives || options.callback ||
            options.prefix_bias_beta > 0 || options.sampling_topk != 1 || options.num_hypotheses < 1 ||
            options.beam_size < 1 || options.num_hypotheses > options.beam_size)
          return false;
        // the model's own end token only (ResolveEndToken of the default variant)
        if (options.end_token.index() != 0 || !std::get<std::string>(options.end_token).empty()) return false;
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
    // ct2x_generate_batch. Everything else (the prompt in the result, static prompts, logits,
    // alternatives, callbacks, sampling, custom end tokens) stays on the reference's own path.
    class Mi355DecoderReplica : public DecoderReplica {
    public:
      using DecoderReplica::DecoderReplica;

    protected:
      std::vector<GenerationResult>
      run_generation(const std::vector<std::vector<std::string>>& start_tokens,
                     const GenerationOptions& options) override {
        if (!serves(start_tokens, options)) return DecoderReplica::run_generation(start_tokens, options);
        const auto& m = static_cast<const LanguageModel&>(*model());
        const Vocabulary& vocab = m.get_vocabulary();
        const size_t batch = start_tokens.size();
        const auto ids = vocab.to_ids(start_tokens);
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

        const size_t nh = options.num_hypotheses, max_len = options.max_length;
        std::vector<int32_t> out_ids(batch * nh * max_len), out_lengths(batch * nh);
        std::vector<float> out_scores(batch * nh);
        if (ct2x_generate_batch(_handle->h, flat.data(), lengths.data(), static_cast<int32_t>(batch),
                                static_cast<int32_t>(stride), &opt, out_ids.data(), out_lengths.data(),
                                out_scores.data()) != 0)
          throw std::runtime_error(std::string("MI355X fused engine: ") + ct2x_last_error());
        ++g_served_calls;

        std::vector<GenerationResult> results(batch);
        for (size_t b = 0; b < batch; ++b) {
          for (size_t h = 0; h < nh; ++h) {
            const int32_t len = out_lengths[b * nh + h];
            if (len < 0 || (h > 0 && len == 0)) continue;  // (fewer hypotheses than asked for)
            const int32_t* row = out_ids.data() + (b * nh + h) * max_len;
            results[b].sequences_ids.emplace_back(row, row + std::min<int32_t>(len, static_cast<int32_t>(max_len)));
            if (options.return_scores) results[b].scores.push_back(out_scores[b * nh + h]);
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
        if (options.include_prompt_in_result || !options.static_prompt.empty() || options.return_logits_vocab ||
            options.return_alternatives || options.callback || options.sampling_topk != 1 ||
            options.num_hypotheses < 1 || options.beam_size < 1 || options.num_hypotheses > options.beam_size)
          return false;
        if (options.end_token.index() != 0 || !std::get<std::string>(options.end_token).empty()) return false;
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
