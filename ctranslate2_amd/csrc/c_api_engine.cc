// extern "C" surface of ctranslate2::Translator / Generator on the MI355X engine
// (include/ct2_mi355.h, "engine" section) + the host-only model inspection helpers.

#include "../../include/ct2_mi355.h"

#include <limits>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <cstdlib>
#include <condition_variable>
#include <exception>
#include <functional>
#include <map>
#include <mutex>
#include <numeric>
#include <thread>

#include "decoding.h"
#include "engine.h"
#include "engine_internal.h"
#include "host_math.h"
#include "model.h"
#include "runtime.h"

using namespace ct2x;

namespace {

// Persistent worker threads (one HIP stream each, created once): the reference's replica workers
// (include/ctranslate2/replica_pool.h:305-367) reduced to "run this closure on n threads".
class WorkerPool {
public:
  ~WorkerPool() {
    {
      std::lock_guard<std::mutex> lock(mutex_);
      stop_ = true;
      ++generation_;
    }
    cv_.notify_all();
    for (auto& th : threads_) th.join();
  }

  void run(size_t n, const std::function<void()>& fn) {
    std::unique_lock<std::mutex> lock(mutex_);
    while (threads_.size() < n) {
      const size_t index = threads_.size();
      threads_.emplace_back([this, index] { loop(index); });
    }
    job_ = &fn;
    active_ = n;
    pending_ = n;
    ++generation_;
    cv_.notify_all();
    done_cv_.wait(lock, [this] { return pending_ == 0; });
    job_ = nullptr;
  }

private:
  void loop(size_t index) {
    uint64_t seen = 0;
    while (true) {
      const std::function<void()>* job = nullptr;
      {
        std::unique_lock<std::mutex> lock(mutex_);
        cv_.wait(lock, [&] { return generation_ != seen; });
        seen = generation_;
        if (stop_) return;
        if (index >= active_) continue;
        job = job_;
      }
      (*job)();
      {
        std::lock_guard<std::mutex> lock(mutex_);
        if (--pending_ == 0) done_cv_.notify_all();
      }
    }
  }

  std::vector<std::thread> threads_;
  std::mutex mutex_;
  std::condition_variable cv_, done_cv_;
  const std::function<void()>* job_ = nullptr;
  size_t active_ = 0, pending_ = 0;
  uint64_t generation_ = 0;
  bool stop_ = false;
};

struct Translator {
  std::shared_ptr<ModelFile> model;
  std::unique_ptr<Engine> engine;
  std::shared_ptr<Vocabulary> src_vocab, tgt_vocab;
  bool add_source_bos = false, add_source_eos = false;
  bool has_start_token = true;
  int32_t start_id = 1, eos_id = 2, bos_id = 1;
  std::mutex mutex;
  ct2x_exec_stats last{};
  int64_t last_compactions = 0;  // finished-row removals of the last call (greedy loop)
  // The session of the last PLAIN greedy or beam search translate call (one group, no prefix /
  // processors / attention / callback ...): a call of the same padded shape and device-side options runs its
  // encoder into the same buffers and replays the step graph captured over them - no allocation,
  // no capture, no instantiation (0.4 - 0.5 ms of a 20 ms Transformer-base call). session_key
  // lists everything a captured step depends on. CT2X_SESSION_CACHE=0: every call builds its own.
  std::unique_ptr<DecodeState> session;
  std::vector<int64_t> session_key;
  int64_t session_bytes = 0;  // pool bytes the kept session took when it was built
  int64_t session_hits = 0;
  int64_t cluster_fallbacks = 0;  // groups re-run on the launch-per-op step after a ClusterTimeout
  WorkerPool pool;  // declared last: joined before the engine is destroyed
};

// The session a plain translate call left for the next one of its shape (Translator::session) is
// dropped by every OTHER entry point of the handle: they would otherwise run next to tens of GB
// nobody is going to reuse.
void release_kept_session(Translator* t) {
  if (t == nullptr) return;
  std::lock_guard<std::mutex> lock(t->mutex);
  t->session.reset();
  t->session_key.clear();
}

double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// SequenceToSequenceModel::initialize + load_vocabularies (src/models/sequence_to_sequence.cc:13-77)
// / LanguageModel (src/models/language_model.cc:13-40).
void load_vocabularies(Translator& t, std::shared_ptr<Vocabulary> given_src = nullptr,
                       std::shared_ptr<Vocabulary> given_tgt = nullptr) {
  const ModelFile& m = *t.model;
  const std::string unk = m.config_string("unk_token", "<unk>");
  const std::string bos = m.config_string("bos_token", "<s>");
  const std::string eos = m.config_string("eos_token", "</s>");
  const std::string& dir = m.dir;
  if (t.engine->is_seq2seq()) {
    auto shared = given_tgt ? nullptr : Vocabulary::load(dir, "shared_vocabulary", unk, bos, eos);
    if (given_tgt) {  // a model handed over in memory (ct2x_translator_create_from_memory)
      t.tgt_vocab = given_tgt;
      t.src_vocab = given_src ? given_src : given_tgt;
    } else if (shared) {
      t.src_vocab = t.tgt_vocab = shared;
    } else {
      t.tgt_vocab = Vocabulary::load(dir, "target_vocabulary", unk, bos, eos);
      if (!t.tgt_vocab)
        throw std::runtime_error("Cannot load the target vocabulary from the model directory");
      t.src_vocab = Vocabulary::load(dir, "source_vocabulary", unk, bos, eos);
      if (!t.src_vocab)
        throw std::runtime_error("Cannot load the source vocabulary from the model directory");
    }
    if (m.binary_version < 6 || m.config.kind == Json::kNull) {
      t.add_source_bos = m.flag("with_source_bos", false);
      t.add_source_eos = m.flag("with_source_eos", false);
    }
    t.add_source_bos = m.config_bool("add_source_bos", t.add_source_bos);
    t.add_source_eos = m.config_bool("add_source_eos", t.add_source_eos);
  } else {
    t.src_vocab = t.tgt_vocab = given_tgt ? given_tgt : Vocabulary::load(dir, "vocabulary", unk, bos, eos);
    if (!t.tgt_vocab) throw std::runtime_error("Cannot load the vocabulary from the model directory");
  }
  t.bos_id = static_cast<int32_t>(t.tgt_vocab->bos_id());
  t.eos_id = static_cast<int32_t>(t.tgt_vocab->eos_id());
  // decoder_start_token (sequence_to_sequence.cc:68-73): config value, else BOS when
  // with_target_bos (default), else EOS.
  const Json* start = m.config.get("decoder_start_token");
  if (start && start->kind == Json::kString) {
    t.start_id = static_cast<int32_t>(t.tgt_vocab->to_id(start->str));
  } else if (start && start->kind == Json::kNull && m.config.kind == Json::kObject) {
    t.has_start_token = false;
  } else {
    if (m.flag("user_decoder_start_tokens", false))
      t.has_start_token = false;
    else
      t.start_id = m.flag("with_target_bos", true) ? t.bos_id : t.eos_id;
  }
  if (static_cast<int64_t>(t.tgt_vocab->size()) != t.engine->vocab_size())
    throw std::runtime_error("The target vocabulary has " + std::to_string(t.tgt_vocab->size()) +
                             " tokens but the model output layer has " +
                             std::to_string(t.engine->vocab_size()));
}

DecodingOptions to_decoding_options(const ct2x_translation_options& o) {
  DecodingOptions d;
  d.beam_size = o.beam_size;
  d.max_length = o.max_decoding_length;
  d.min_length = o.min_decoding_length;
  d.num_hypotheses = o.num_hypotheses;
  d.return_scores = o.return_scores != 0;
  d.length_penalty = o.length_penalty;
  d.patience = o.patience;
  d.use_graph = o.use_graph != 0;
  d.include_eos_in_hypotheses = true;  // DecodingOptions default (include/ctranslate2/decoding.h:154)
  // zero-initialised structs: arg-max; -1 = the whole vocabulary (the reference's 0)
  d.sampling_topk = o.sampling_topk == 0 ? 1 : (o.sampling_topk < 0 ? 0 : o.sampling_topk);
  d.sampling_topp = o.sampling_topp == 0.f ? 1.f : o.sampling_topp;
  d.sampling_temperature = o.sampling_temperature == 0.f ? 1.f : o.sampling_temperature;
  d.seed = o.seed;
  d.prefix_bias_beta = o.prefix_bias_beta;
  d.return_alternatives = o.return_alternatives != 0;
  d.min_alternative_expansion_prob = o.min_alternative_expansion_prob;
  d.coverage_penalty = o.coverage_penalty;
  d.return_attention = o.attention_out != nullptr;
  d.return_logits_vocab = o.logits_out != nullptr;
  if (o.step_callback != nullptr) {
    // (the callers below wrap it once more to restore the batch ids of their sub-batches,
    // restore_batch_ids_in_callback, include/ctranslate2/generation.h:120-131)
    auto fn = o.step_callback;
    void* user = o.step_callback_data;
    d.sync_every = 1;  // a streaming consumer: hand tokens over step by step (<= 2 steps late)
    d.callback = [fn, user](const DecodingStepResult& r) {
      return fn(user, r.step, r.batch_id, r.hypothesis_id, r.token_id, r.has_score ? 1 : 0, r.score,
                r.is_last ? 1 : 0) != 0;
    };
  }
  LogitsProcessing& lp = d.logits_processing;
  lp.repetition_penalty = o.repetition_penalty == 0.f ? 1.f : o.repetition_penalty;
  lp.no_repeat_ngram_size = o.no_repeat_ngram_size;
  if (o.num_suppress_sequences < 0 ||
      (o.num_suppress_sequences > 0 && (!o.suppress_ids || !o.suppress_lengths)))
    throw std::invalid_argument("suppress_sequences: ids / lengths missing");
  const int32_t* ids = o.suppress_ids;
  for (int32_t s = 0; s < o.num_suppress_sequences; ++s) {
    const int32_t n = o.suppress_lengths[s];
    if (n < 0) throw std::invalid_argument("suppress_sequences: negative length");
    lp.suppress_sequences.emplace_back(ids, ids + n);
    ids += n;
  }
  return d;
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

void ct2x_translation_options_default(struct ct2x_translation_options* o) {
  // TranslationOptions defaults (include/ctranslate2/translation.h:11-96).
  std::memset(o, 0, sizeof(*o));
  o->beam_size = 2;
  o->max_decoding_length = 256;
  o->min_decoding_length = 1;
  o->num_hypotheses = 1;
  o->return_scores = 0;
  o->return_end_token = 0;
  o->max_batch_size = 0;
  o->max_input_length = 1024;
  o->length_penalty = 1.f;
  o->patience = 1.f;
  o->use_graph = 1;
  o->sampling_topk = 1;
  o->sampling_topp = 1.f;
  o->sampling_temperature = 1.f;
  o->seed = 0;
  o->repetition_penalty = 1.f;
}

void* ct2x_translator_create(const char* model_dir, int device_index, const char* compute_type) {
  try {
    auto t = std::make_unique<Translator>();
    t->model = ModelFile::load(model_dir);
    t->engine = std::make_unique<Engine>(t->model, compute_type ? compute_type : "default",
                                         device_index);
    load_vocabularies(*t);
    return t.release();
  } catch (const std::exception& e) {
    set_last_error(e.what());
    return nullptr;
  }
}

void* ct2x_translator_create_from_memory(const char* spec, uint32_t spec_revision, uint32_t binary_version,
                                         const char* config_json, const struct ct2x_memory_variable* vars,
                                         int64_t num_vars, const char* const* source_tokens,
                                         int64_t num_source_tokens, const char* const* target_tokens,
                                         int64_t num_target_tokens, int device_index,
                                         const char* compute_type) {
  try {
    std::vector<Variable> list;
    list.reserve(static_cast<size_t>(num_vars));
    if (num_vars < 0 || (num_vars > 0 && vars == nullptr))
      throw std::runtime_error("ct2x_translator_create_from_memory: bad variable list");
    for (int64_t i = 0; i < num_vars; ++i) {
      Variable v;
      if (vars[i].name == nullptr) throw std::runtime_error("ct2x_translator_create_from_memory: a variable has no name");
      v.name = vars[i].name;
      v.dtype = vars[i].dtype;
      if (vars[i].rank < 0 || vars[i].rank > 8 || (vars[i].rank > 0 && vars[i].shape == nullptr))
        throw std::runtime_error("Variable " + v.name + " has an invalid rank");
      (void)dtype_size(v.dtype);  // (throws on an unknown data type id)
      uint64_t elements = 1;
      for (int32_t r = 0; r < vars[i].rank; ++r) {
        const int64_t dim = vars[i].shape[r];
        if (dim < 0 || (dim > 0 && elements > (uint64_t(1) << 40) / static_cast<uint64_t>(dim)))
          throw std::runtime_error("Variable " + v.name + " has an invalid shape");
        elements *= static_cast<uint64_t>(dim);
      }
      if (elements * dtype_size(v.dtype) != vars[i].nbytes || (vars[i].nbytes > 0 && vars[i].data == nullptr))
        throw std::runtime_error("Variable " + v.name + " has an invalid payload size");
      v.shape.assign(vars[i].shape, vars[i].shape + vars[i].rank);
      const size_t bytes = static_cast<size_t>(v.size()) * dtype_size(v.dtype);
      const char* src = static_cast<const char*>(vars[i].data);
      v.data.assign(src, src + bytes);
      list.emplace_back(std::move(v));
    }
    auto t = std::make_unique<Translator>();
    t->model = ModelFile::from_memory(spec ? spec : "", spec_revision, binary_version,
                                      config_json ? config_json : "", std::move(list));
    t->engine = std::make_unique<Engine>(t->model, compute_type ? compute_type : "default", device_index);
    const ModelFile& m = *t->model;
    const std::string unk = m.config_string("unk_token", "<unk>");
    const std::string bos = m.config_string("bos_token", "<s>");
    const std::string eos = m.config_string("eos_token", "</s>");
    auto make = [&](const char* const* tokens, int64_t n) -> std::shared_ptr<Vocabulary> {
      if (tokens == nullptr || n <= 0) return nullptr;
      return Vocabulary::from_tokens(std::vector<std::string>(tokens, tokens + n), unk, bos, eos);
    };
    auto tgt = make(target_tokens, num_target_tokens);
    if (!tgt) throw std::runtime_error("ct2x_translator_create_from_memory: the target vocabulary is required");
    load_vocabularies(*t, make(source_tokens, num_source_tokens), tgt);
    return t.release();
  } catch (const std::exception& e) {
    set_last_error(e.what());
    return nullptr;
  }
}

// ---- tensor parallelism (reference ScopedMPISetter + Model::load sharding) --------

int ct2x_tp_unique_id(void* id_out_128_bytes) {
  API_TRY
  rccl_get_unique_id(id_out_128_bytes);
  API_CATCH
}

int ct2x_tp_shard_rows(int64_t n, const int64_t* blocks, int32_t num_blocks, int32_t rank,
                       int32_t world, int64_t* first_out, int64_t* count_out) {
  API_TRY
  CT2X_CHECK(world >= 1 && rank >= 0 && rank < world, "invalid tensor-parallel rank / world size");
  std::vector<int64_t> sizes(blocks, blocks + num_blocks);
  if (sizes.empty()) sizes.push_back(n);
  int64_t total = 0;
  for (int64_t b : sizes) total += b;
  CT2X_CHECK(total == n, "fused blocks do not add up to the weight's rows");
  int64_t base = 0;
  for (size_t i = 0; i < sizes.size(); ++i) {
    CT2X_CHECK(sizes[i] % world == 0, "dimension " + std::to_string(sizes[i]) +
                                          " is not divisible by the tensor-parallel world size");
    first_out[i] = base + rank * (sizes[i] / world);
    count_out[i] = sizes[i] / world;
    base += sizes[i];
  }
  API_CATCH
}

int ct2x_tp_selftest_rccl(int device_index) {
  API_TRY
  // One-rank RCCL communicator on this device: exercises library resolution, the by-value
  // unique id, the datatype / op codes and the stream plumbing of the multi-GPU path on a
  // single-GPU box (an all-reduce over one rank is the identity).
  CT2X_HIP_CHECK(hipSetDevice(device_index));
  char id[kTpUniqueIdBytes];
  rccl_get_unique_id(id);
  auto comm = make_rccl_comm(0, 1, id);
  hipStream_t stream = thread_stream();
  const int n = 1024;
  std::vector<int32_t> hi(n);
  std::vector<float> hf(n);
  for (int i = 0; i < n; ++i) {
    hi[i] = i * 7 - 100;
    hf[i] = 0.25f * i - 3.f;
  }
  auto* di = static_cast<int32_t*>(device_malloc(n * 4));
  auto* df = static_cast<float*>(device_malloc(n * 4));
  CT2X_HIP_CHECK(hipMemcpyAsync(di, hi.data(), n * 4, hipMemcpyHostToDevice, stream));
  CT2X_HIP_CHECK(hipMemcpyAsync(df, hf.data(), n * 4, hipMemcpyHostToDevice, stream));
  comm->all_reduce_sum_i32(di, n, stream);
  comm->all_reduce_max_f32(df, n, stream);
  comm->all_reduce_sum_f32(df, n, stream);
  // GatherAll over one rank = a copy
  auto* dg = static_cast<int32_t*>(device_malloc(n * 4));
  comm->all_gather(di, dg, n * 4, stream);
  std::vector<int32_t> oi(n), og(n);
  std::vector<float> of(n);
  CT2X_HIP_CHECK(hipMemcpyAsync(oi.data(), di, n * 4, hipMemcpyDeviceToHost, stream));
  CT2X_HIP_CHECK(hipMemcpyAsync(of.data(), df, n * 4, hipMemcpyDeviceToHost, stream));
  CT2X_HIP_CHECK(hipMemcpyAsync(og.data(), dg, n * 4, hipMemcpyDeviceToHost, stream));
  CT2X_HIP_CHECK(hipStreamSynchronize(stream));
  CT2X_CHECK(oi == hi && of == hf, "one-rank RCCL all-reduce did not return its input");
  CT2X_CHECK(og == hi, "one-rank RCCL all-gather did not return its input");
  // The collectives are stream work: a hipGraph captures and replays them (what lets a sharded
  // decode step run as one graph like the unsharded one).
  CT2X_CHECK(comm->capturable(), "the RCCL communicator reports itself not capturable");
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  CT2X_HIP_CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
  comm->all_reduce_sum_i32(di, n, stream);
  comm->all_reduce_max_f32(df, n, stream);
  comm->all_gather(di, dg, n * 4, stream);
  CT2X_HIP_CHECK(hipStreamEndCapture(stream, &graph));
  CT2X_HIP_CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  CT2X_HIP_CHECK(hipMemsetAsync(dg, 0, n * 4, stream));
  for (int rep = 0; rep < 3; ++rep) CT2X_HIP_CHECK(hipGraphLaunch(exec, stream));
  CT2X_HIP_CHECK(hipMemcpyAsync(og.data(), dg, n * 4, hipMemcpyDeviceToHost, stream));
  CT2X_HIP_CHECK(hipStreamSynchronize(stream));
  (void)hipGraphExecDestroy(exec);
  (void)hipGraphDestroy(graph);
  device_free(di);
  device_free(df);
  device_free(dg);
  CT2X_CHECK(og == hi, "replayed graph with RCCL collectives did not reproduce the gather");
  API_CATCH
}

int ct2x_tp_selftest_local(int device_index, int world) {
  API_TRY
  // The in-process communicator (ranks = threads on one device): all-reduce (int32 sum, float
  // max) and GatherAll on known data, checked on every rank.
  CT2X_CHECK(world >= 1 && world <= 8, "ct2x_tp_selftest_local: world must be in [1, 8]");
  auto comms = make_local_comms(world);
  const int n = 777;
  std::vector<std::string> errors(world);
  std::vector<std::thread> threads;
  for (int r = 0; r < world; ++r) {
    threads.emplace_back([&, r] {
      try {
        CT2X_HIP_CHECK(hipSetDevice(device_index));
        hipStream_t stream = thread_stream();
        std::vector<int32_t> hi(n);
        std::vector<float> hf(n);
        for (int i = 0; i < n; ++i) {
          hi[i] = (r + 1) * (i - 300);
          hf[i] = static_cast<float>((i * 13 + r * 7) % 101) - 50.f;
        }
        auto* di = static_cast<int32_t*>(device_malloc(n * 4));
        auto* df = static_cast<float*>(device_malloc(n * 4));
        auto* dsend = static_cast<int32_t*>(device_malloc(n * 4));
        auto* dg = static_cast<int32_t*>(device_malloc(static_cast<size_t>(world) * n * 4));
        CT2X_HIP_CHECK(hipMemcpyAsync(di, hi.data(), n * 4, hipMemcpyHostToDevice, stream));
        CT2X_HIP_CHECK(hipMemcpyAsync(dsend, hi.data(), n * 4, hipMemcpyHostToDevice, stream));
        CT2X_HIP_CHECK(hipMemcpyAsync(df, hf.data(), n * 4, hipMemcpyHostToDevice, stream));
        comms[r]->all_reduce_sum_i32(di, n, stream);
        comms[r]->all_reduce_max_f32(df, n, stream);
        comms[r]->all_gather(dsend, dg, n * 4, stream);
        std::vector<int32_t> oi(n), og(static_cast<size_t>(world) * n);
        std::vector<float> of(n);
        CT2X_HIP_CHECK(hipMemcpyAsync(oi.data(), di, n * 4, hipMemcpyDeviceToHost, stream));
        CT2X_HIP_CHECK(hipMemcpyAsync(of.data(), df, n * 4, hipMemcpyDeviceToHost, stream));
        CT2X_HIP_CHECK(hipMemcpyAsync(og.data(), dg, og.size() * 4, hipMemcpyDeviceToHost, stream));
        CT2X_HIP_CHECK(hipStreamSynchronize(stream));
        device_free(di);
        device_free(df);
        device_free(dsend);
        device_free(dg);
        for (int i = 0; i < n; ++i) {
          int32_t want_sum = 0;
          float want_max = -1e30f;
          for (int q = 0; q < world; ++q) {
            want_sum += (q + 1) * (i - 300);
            want_max = std::max(want_max, static_cast<float>((i * 13 + q * 7) % 101) - 50.f);
            if (og[static_cast<size_t>(q) * n + i] != (q + 1) * (i - 300))
              throw Error("all_gather: wrong element");
          }
          if (oi[i] != want_sum) throw Error("all_reduce(int32, sum): wrong element");
          if (of[i] != want_max) throw Error("all_reduce(float, max): wrong element");
        }
      } catch (const std::exception& e) {
        errors[r] = e.what();
      }
    });
  }
  for (auto& th : threads) th.join();
  for (int r = 0; r < world; ++r)
    CT2X_CHECK(errors[r].empty(), "rank " + std::to_string(r) + ": " + errors[r]);
  API_CATCH
}

int ct2x_tp_selftest_multi(int device_index, int rank, int world, const void* unique_id_128_bytes,
                           const char* shm_segment_name, int32_t* mailboxes_out, float* us_per_dense_out) {
  API_TRY
  // The communicator of the tensor-parallel engines across REAL ranks (one process per GPU): every
  // collective the engine calls (ops::ReduceAll / GatherAll, src/ops/nccl_ops_gpu.cu:53-85) on
  // rank-dependent data whose result is known in closed form, then - when the communicator has
  // peer-mapped mailboxes - the two one-shot exchanges of a row-parallel int8 Dense, checked on
  // every rank against what the whole row gives on the host, bit for bit. bench.py calls it before
  // its timed region whenever it runs on more than one GPU, so that the first multi-GPU run says
  // which piece fails instead of producing a number.
  CT2X_CHECK(world >= 1 && world <= kTpMaxWorld && rank >= 0 && rank < world, "ct2x_tp_selftest_multi: rank / world");
  CT2X_CHECK((unique_id_128_bytes != nullptr) != (shm_segment_name != nullptr && shm_segment_name[0] != 0),
             "ct2x_tp_selftest_multi: give the RCCL unique id OR a shared-memory segment name");
  CT2X_HIP_CHECK(hipSetDevice(device_index));
  std::shared_ptr<TpComm> comm = unique_id_128_bytes ? make_rccl_comm(rank, world, unique_id_128_bytes)
                                                     : make_shm_comm(rank, world, shm_segment_name);
  CT2X_CHECK(comm->rank() == rank && comm->world() == world, "communicator reports another rank / world");
  hipStream_t stream = thread_stream();
  const int n = 4096;
  auto vi = [](int i, int r) { return (i * 7 - 100) * (r + 1); };
  auto vf = [](int i, int r) { return 0.25f * static_cast<float>((i * 37 + r * 101) % 1000) - 60.f; };  // exact sums
  std::vector<int32_t> hi(n);
  std::vector<float> hf(n), hm(n);
  for (int i = 0; i < n; ++i) {
    hi[i] = vi(i, rank);
    hf[i] = vf(i, rank);
    hm[i] = vf(i, rank);
  }
  auto* di = static_cast<int32_t*>(device_malloc(n * 4));
  auto* df = static_cast<float*>(device_malloc(n * 4));
  auto* dm = static_cast<float*>(device_malloc(n * 4));
  auto* dg = static_cast<int32_t*>(device_malloc(static_cast<size_t>(n) * 4 * world));
  CT2X_HIP_CHECK(hipMemcpyAsync(di, hi.data(), n * 4, hipMemcpyHostToDevice, stream));
  CT2X_HIP_CHECK(hipMemcpyAsync(df, hf.data(), n * 4, hipMemcpyHostToDevice, stream));
  CT2X_HIP_CHECK(hipMemcpyAsync(dm, hm.data(), n * 4, hipMemcpyHostToDevice, stream));
  comm->all_gather(di, dg, n * 4, stream);   // (before the in-place reduce of di)
  comm->all_reduce_sum_i32(di, n, stream);
  comm->all_reduce_sum_f32(df, n, stream);
  comm->all_reduce_max_f32(dm, n, stream);
  std::vector<int32_t> oi(n), og(static_cast<size_t>(n) * world);
  std::vector<float> of(n), om(n);
  CT2X_HIP_CHECK(hipMemcpyAsync(oi.data(), di, n * 4, hipMemcpyDeviceToHost, stream));
  CT2X_HIP_CHECK(hipMemcpyAsync(of.data(), df, n * 4, hipMemcpyDeviceToHost, stream));
  CT2X_HIP_CHECK(hipMemcpyAsync(om.data(), dm, n * 4, hipMemcpyDeviceToHost, stream));
  CT2X_HIP_CHECK(hipMemcpyAsync(og.data(), dg, og.size() * 4, hipMemcpyDeviceToHost, stream));
  CT2X_HIP_CHECK(hipStreamSynchronize(stream));
  for (int i = 0; i < n; ++i) {
    int64_t si = 0;
    float sf = 0.f, mx = -3.4e38f;
    for (int r = 0; r < world; ++r) {
      si += vi(i, r);
      sf += vf(i, r);  // (quarter-integers below 2^10: every partial sum is exact in any order)
      mx = std::max(mx, vf(i, r));
    }
    CT2X_CHECK(oi[i] == static_cast<int32_t>(si), "all-reduce (int32 sum) differs at element " + std::to_string(i));
    CT2X_CHECK(of[i] == sf, "all-reduce (float sum) differs at element " + std::to_string(i));
    CT2X_CHECK(om[i] == mx, "all-reduce (float max) differs at element " + std::to_string(i));
    for (int r = 0; r < world; ++r)
      CT2X_CHECK(og[static_cast<size_t>(r) * n + i] == vi(i, r),
                 "all-gather differs at rank " + std::to_string(r) + " element " + std::to_string(i));
  }
  for (void* ptr : {static_cast<void*>(di), static_cast<void*>(df), static_cast<void*>(dm), static_cast<void*>(dg)})
    device_free(ptr);
  if (mailboxes_out) *mailboxes_out = comm->peer_box() != nullptr ? 1 : 0;
  if (us_per_dense_out) *us_per_dense_out = 0.f;
  if (comm->peer_box() != nullptr) {
    // one row-parallel int8 Dense at decode size: x [rows, world * depth] split by columns over the
    // ranks, partial_r [rows, n] = (r + 1) * pattern
    const TpPeerBox& pb = *comm->peer_box();
    const int64_t rows = 32, depth = 256, nn = 512;
    std::vector<float> hx(static_cast<size_t>(rows) * world * depth);
    for (size_t i = 0; i < hx.size(); ++i)
      hx[i] = static_cast<float>(static_cast<int>((i * 2654435761u) % 2001) - 1000) / 37.f;
    std::vector<float> h_ws(nn), h_bias(nn);
    for (int64_t j = 0; j < nn; ++j) {
      h_ws[j] = 50.f + static_cast<float>(j % 97);
      h_bias[j] = static_cast<float>(j % 13) * 0.01f;
    }
    std::vector<float> xs(static_cast<size_t>(rows) * depth);
    for (int64_t i = 0; i < rows; ++i)
      for (int64_t j = 0; j < depth; ++j) xs[i * depth + j] = hx[(i * world + rank) * depth + j];
    auto pattern = [](size_t i, int r) { return (r + 1) * (static_cast<int32_t>(i % 1999) - 1000); };
    std::vector<int32_t> part(static_cast<size_t>(rows) * nn);
    for (size_t i = 0; i < part.size(); ++i) part[i] = pattern(i, rank);
    auto* dx = static_cast<float*>(device_malloc(xs.size() * 4));
    auto* dq = static_cast<int8_t*>(device_malloc(xs.size()));
    auto* ds = static_cast<float*>(device_malloc(rows * 4));
    auto* dp = static_cast<int32_t*>(device_malloc(part.size() * 4));
    auto* dy = static_cast<float*>(device_malloc(part.size() * 4));
    auto* dws = static_cast<float*>(device_malloc(nn * 4));
    auto* dbias = static_cast<float*>(device_malloc(nn * 4));
    CT2X_HIP_CHECK(hipMemcpyAsync(dx, xs.data(), xs.size() * 4, hipMemcpyHostToDevice, stream));
    CT2X_HIP_CHECK(hipMemcpyAsync(dp, part.data(), part.size() * 4, hipMemcpyHostToDevice, stream));
    CT2X_HIP_CHECK(hipMemcpyAsync(dws, h_ws.data(), nn * 4, hipMemcpyHostToDevice, stream));
    CT2X_HIP_CHECK(hipMemcpyAsync(dbias, h_bias.data(), nn * 4, hipMemcpyHostToDevice, stream));
    launch_tp_amax_quantize(pb, dx, rows, depth, 3, dq, ds, stream);
    launch_tp_sum_dequant(pb, dp, ds, dws, dbias, nullptr, rows, nn, dy, stream);
    std::vector<float> y(part.size()), scales(rows);
    std::vector<int8_t> q(xs.size());
    CT2X_HIP_CHECK(hipMemcpyAsync(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost, stream));
    CT2X_HIP_CHECK(hipMemcpyAsync(q.data(), dq, q.size(), hipMemcpyDeviceToHost, stream));
    CT2X_HIP_CHECK(hipMemcpyAsync(scales.data(), ds, rows * 4, hipMemcpyDeviceToHost, stream));
    CT2X_HIP_CHECK(hipStreamSynchronize(stream));
    const int iters = 50;
    hipEvent_t e0, e1;
    CT2X_HIP_CHECK(hipEventCreate(&e0));
    CT2X_HIP_CHECK(hipEventCreate(&e1));
    CT2X_HIP_CHECK(hipEventRecord(e0, stream));
    for (int it = 0; it < iters; ++it) {
      launch_tp_amax_quantize(pb, dx, rows, depth, 3, dq, ds, stream);
      launch_tp_sum_dequant(pb, dp, ds, dws, dbias, nullptr, rows, nn, dy, stream);
    }
    CT2X_HIP_CHECK(hipEventRecord(e1, stream));
    CT2X_HIP_CHECK(hipStreamSynchronize(stream));
    float ms = 0.f;
    CT2X_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (us_per_dense_out) *us_per_dense_out = ms * 1e3f / static_cast<float>(iters);
    TpBoxControl ctl;
    CT2X_HIP_CHECK(hipMemcpy(&ctl, pb.box[rank], sizeof(ctl), hipMemcpyDeviceToHost));
    CT2X_CHECK(ctl.status == 0, "a poll of the one-shot exchange gave up (a peer never arrived)");
    // what the whole row gives (ops::Quantize with the u8 shift and rounding = mode 3, then the
    // Dequantize of the summed products: src/cpu/kernels.cc:577-680)
    for (int64_t i = 0; i < rows; ++i) {
      float amax = 0.f;
      for (int64_t j = 0; j < world * depth; ++j) amax = std::max(amax, std::fabs(hx[i * world * depth + j]));
      const float scale = amax != 0.f ? 127.f / amax : 1.f;
      CT2X_CHECK(scales[i] == scale, "one-shot exchange: quantisation scale of row " + std::to_string(i));
      for (int64_t j = 0; j < depth; ++j) {
        const float v = std::nearbyint(std::fma(xs[i * depth + j], scale, 128.f));
        const int8_t want = static_cast<int8_t>((static_cast<int>(v) & 0xff) - 128);
        CT2X_CHECK(q[i * depth + j] == want, "one-shot exchange: quantised value differs");
      }
      for (int64_t j = 0; j < nn; ++j) {
        int64_t sum = 0;
        for (int r = 0; r < world; ++r) sum += pattern(static_cast<size_t>(i * nn + j), r);
        float v = static_cast<float>(static_cast<int32_t>(sum)) * (1.f / scale);
        v = v / h_ws[j];
        v = v + h_bias[j];
        CT2X_CHECK(y[i * nn + j] == v, "one-shot exchange: dequantised sum differs at row " + std::to_string(i));
      }
    }
    for (void* ptr : {static_cast<void*>(dx), static_cast<void*>(dq), static_cast<void*>(ds), static_cast<void*>(dp),
                      static_cast<void*>(dy), static_cast<void*>(dws), static_cast<void*>(dbias)})
      device_free(ptr);
  }
  API_CATCH
}

int ct2x_tp_selftest_oneshot(int device_index, int world, int64_t rows, int64_t depth, int64_t n,
                             int iters, float* us_per_dense_out) {
  API_TRY
  // The one-shot exchanges of a row-parallel int8 Dense (tp_oneshot.hip) on the in-process
  // communicator: `world` rank threads, each with a [rows, depth] slice of one activation and its
  // own int32 partial product [rows, n]. Checked against what ONE device computes from the whole
  // row (global absmax -> scale -> Quantize; sum of the partials -> Dequantize): every rank must
  // hold identical, bit-exact results. Then `iters` exchange pairs are timed per rank (HIP events
  // on the rank's stream; all ranks run concurrently) - the single-device loopback price of the
  // two exchanges, reported as the maximum over the ranks.
  CT2X_CHECK(world == 2, "ct2x_tp_selftest_oneshot: the in-process form has two ranks (see make_local_comms)");
  CT2X_CHECK(tp_oneshot_fits(rows, rows * n), "ct2x_tp_selftest_oneshot: message larger than the mailbox");
  auto comms = make_local_comms(world);
  CT2X_CHECK(comms[0]->peer_box() != nullptr, "one-shot mailboxes are disabled (CT2X_TP_ONESHOT=0)");
  std::vector<std::string> errors(world);
  std::vector<float> us(world, 0.f);
  // host data: x [rows, world * depth], partial_r [rows, n] = (r + 1) * pattern
  std::vector<float> hx(static_cast<size_t>(rows) * world * depth);
  for (size_t i = 0; i < hx.size(); ++i)
    hx[i] = static_cast<float>(static_cast<int>((i * 2654435761u) % 2001) - 1000) / 37.f;
  std::vector<float> h_ws(n), h_bias(n);
  for (int64_t j = 0; j < n; ++j) {
    h_ws[j] = 50.f + static_cast<float>(j % 97);
    h_bias[j] = static_cast<float>(j % 13) * 0.01f;
  }
  std::vector<std::thread> threads;
  std::vector<std::vector<float>> outs(world);
  std::vector<std::vector<int8_t>> qouts(world);
  for (int r = 0; r < world; ++r) {
    threads.emplace_back([&, r] {
      try {
        CT2X_HIP_CHECK(hipSetDevice(device_index));
        hipStream_t stream = thread_stream_with_priority(r);
        const TpPeerBox& pb = *comms[r]->peer_box();
        std::vector<float> xs(static_cast<size_t>(rows) * depth);
        for (int64_t i = 0; i < rows; ++i)
          for (int64_t j = 0; j < depth; ++j)
            xs[i * depth + j] = hx[(i * world + r) * depth + j];
        std::vector<int32_t> part(static_cast<size_t>(rows) * n);
        for (size_t i = 0; i < part.size(); ++i)
          part[i] = (r + 1) * (static_cast<int32_t>(i % 1999) - 1000);
        auto* dx = static_cast<float*>(device_malloc(xs.size() * 4));
        auto* dq = static_cast<int8_t*>(device_malloc(xs.size()));
        auto* ds = static_cast<float*>(device_malloc(rows * 4));
        auto* dp = static_cast<int32_t*>(device_malloc(part.size() * 4));
        auto* dy = static_cast<float*>(device_malloc(part.size() * 4));
        auto* dws = static_cast<float*>(device_malloc(n * 4));
        auto* dbias = static_cast<float*>(device_malloc(n * 4));
        CT2X_HIP_CHECK(hipMemcpyAsync(dx, xs.data(), xs.size() * 4, hipMemcpyHostToDevice, stream));
        CT2X_HIP_CHECK(hipMemcpyAsync(dp, part.data(), part.size() * 4, hipMemcpyHostToDevice, stream));
        CT2X_HIP_CHECK(hipMemcpyAsync(dws, h_ws.data(), n * 4, hipMemcpyHostToDevice, stream));
        CT2X_HIP_CHECK(hipMemcpyAsync(dbias, h_bias.data(), n * 4, hipMemcpyHostToDevice, stream));
        launch_tp_amax_quantize(pb, dx, rows, depth, 3, dq, ds, stream);
        launch_tp_sum_dequant(pb, dp, ds, dws, dbias, nullptr, rows, n, dy, stream);
        outs[r].resize(part.size());
        qouts[r].resize(xs.size());
        CT2X_HIP_CHECK(hipMemcpyAsync(outs[r].data(), dy, part.size() * 4, hipMemcpyDeviceToHost, stream));
        CT2X_HIP_CHECK(hipMemcpyAsync(qouts[r].data(), dq, xs.size(), hipMemcpyDeviceToHost, stream));
        CT2X_HIP_CHECK(hipStreamSynchronize(stream));
        if (iters > 0) {
          hipEvent_t e0, e1;
          CT2X_HIP_CHECK(hipEventCreate(&e0));
          CT2X_HIP_CHECK(hipEventCreate(&e1));
          CT2X_HIP_CHECK(hipEventRecord(e0, stream));
          for (int it = 0; it < iters; ++it) {
            launch_tp_amax_quantize(pb, dx, rows, depth, 3, dq, ds, stream);
            launch_tp_sum_dequant(pb, dp, ds, dws, dbias, nullptr, rows, n, dy, stream);
          }
          CT2X_HIP_CHECK(hipEventRecord(e1, stream));
          CT2X_HIP_CHECK(hipStreamSynchronize(stream));
          float ms = 0.f;
          CT2X_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
          us[r] = ms * 1e3f / static_cast<float>(iters);
          (void)hipEventDestroy(e0);
          (void)hipEventDestroy(e1);
        }
        TpBoxControl ctl;
        CT2X_HIP_CHECK(hipMemcpy(&ctl, pb.box[r], sizeof(ctl), hipMemcpyDeviceToHost));
        if (ctl.status != 0) throw Error("a poll of the one-shot exchange gave up (peer never arrived)");
        for (void* ptr : {static_cast<void*>(dx), static_cast<void*>(dq), static_cast<void*>(ds),
                          static_cast<void*>(dp), static_cast<void*>(dy), static_cast<void*>(dws),
                          static_cast<void*>(dbias)})
          device_free(ptr);
      } catch (const std::exception& e) {
        errors[r] = e.what();
      }
    });
  }
  for (auto& th : threads) th.join();
  for (int r = 0; r < world; ++r)
    CT2X_CHECK(errors[r].empty(), "rank " + std::to_string(r) + ": " + errors[r]);
  // reference: what one device computes from the whole rows (same formulas, host arithmetic)
  for (int64_t i = 0; i < rows; ++i) {
    float amax = 0.f;
    for (int64_t j = 0; j < world * depth; ++j) amax = std::max(amax, std::fabs(hx[i * world * depth + j]));
    const float scale = amax != 0.f ? 127.f / amax : 1.f;
    const float r_a = 1.f / scale;
    for (int r = 0; r < world; ++r) {
      for (int64_t j = 0; j < depth; ++j) {
        // quant_mode 3: u8 shift + round to nearest even (device_utils.h to_s8)
        const float v = std::fmaf(hx[(i * world + r) * depth + j], scale, 128.f);
        const int q = static_cast<int>(std::nearbyintf(v)) - 128;
        if (qouts[r][i * depth + j] != static_cast<int8_t>(q))
          throw Error("one-shot absmax exchange: quantised value differs from the unsharded row");
      }
    }
    for (int64_t j = 0; j < n; ++j) {
      int32_t sum = 0;
      for (int r = 0; r < world; ++r) sum += (r + 1) * (static_cast<int32_t>((i * n + j) % 1999) - 1000);
      float v = static_cast<float>(sum) * r_a;
      v = v / h_ws[j];
      v = v + h_bias[j];
      for (int r = 0; r < world; ++r)
        if (std::memcmp(&outs[r][i * n + j], &v, 4) != 0)
          throw Error("one-shot sum exchange: dequantised value differs from the unsharded product");
    }
  }
  if (us_per_dense_out) {
    float worst = 0.f;
    for (float u : us) worst = std::max(worst, u);
    *us_per_dense_out = worst;
  }
  return 0;
  API_CATCH
}

void* ct2x_translator_create_tp(const char* model_dir, int device_index, const char* compute_type,
                                int rank, int world, const void* unique_id_128_bytes) {
  try {
    CT2X_CHECK(world >= 1 && rank >= 0 && rank < world, "invalid tensor-parallel rank / world size");
    auto t = std::make_unique<Translator>();
    t->model = ModelFile::load(model_dir);
    std::shared_ptr<TpComm> comm;
    if (world > 1) {
      CT2X_CHECK(unique_id_128_bytes != nullptr, "tensor parallelism needs the RCCL unique id");
      CT2X_HIP_CHECK(hipSetDevice(device_index));
      comm = make_rccl_comm(rank, world, unique_id_128_bytes);
    }
    t->engine = std::make_unique<Engine>(t->model, compute_type ? compute_type : "default",
                                         device_index, comm);
    load_vocabularies(*t);
    return t.release();
  } catch (const std::exception& e) {
    set_last_error(e.what());
    return nullptr;
  }
}

void* ct2x_translator_create_tp_shm(const char* model_dir, int device_index, const char* compute_type,
                                    int rank, int world, const char* segment_name) {
  try {
    CT2X_CHECK(world >= 2 && rank >= 0 && rank < world, "invalid tensor-parallel rank / world size");
    CT2X_CHECK(segment_name != nullptr && segment_name[0] != 0, "the shared-memory communicator needs a segment name");
    auto t = std::make_unique<Translator>();
    t->model = ModelFile::load(model_dir);
    CT2X_HIP_CHECK(hipSetDevice(device_index));
    std::shared_ptr<TpComm> comm = make_shm_comm(rank, world, segment_name);
    t->engine = std::make_unique<Engine>(t->model, compute_type ? compute_type : "default",
                                         device_index, comm);
    load_vocabularies(*t);
    return t.release();
  } catch (const std::exception& e) {
    set_last_error(e.what());
    return nullptr;
  }
}

int ct2x_translator_create_tp_local(const char* model_dir, int device_index,
                                    const char* compute_type, int world, void** handles_out) {
  API_TRY
  CT2X_HIP_CHECK(hipSetDevice(device_index));
  auto comms = make_local_comms(world);
  std::vector<std::unique_ptr<Translator>> ts;
  for (int r = 0; r < world; ++r) {
    auto t = std::make_unique<Translator>();
    t->model = ModelFile::load(model_dir);
    t->engine = std::make_unique<Engine>(t->model, compute_type ? compute_type : "default",
                                         device_index, comms[r]);
    load_vocabularies(*t);
    ts.push_back(std::move(t));
  }
  for (int r = 0; r < world; ++r) handles_out[r] = ts[r].release();
  API_CATCH
}

void ct2x_translator_destroy(void* translator) { delete static_cast<Translator*>(translator); }

// decode_alternatives (src/decoding.cc:1122-1290) for a batch of sentences: the target prefix is
// (return_logits_vocab) hypothesis k of a result -> slot `slot` of the caller's buffer
static void write_logits(const ct2x_translation_options& opt, const DecodingResult& r, int64_t slot,
                         int32_t k, int64_t vocab) {
  opt.logits_counts_out[slot] = 0;
  if (k >= static_cast<int32_t>(r.logits_vocab.size())) return;
  const auto& steps = r.logits_vocab[k];
  const int64_t n = std::min<int64_t>(static_cast<int64_t>(steps.size()), opt.logits_steps);
  float* dst = opt.logits_out + slot * opt.logits_steps * vocab;
  for (int64_t s = 0; s < n; ++s) std::copy(steps[s].begin(), steps[s].end(), dst + s * vocab);
  opt.logits_counts_out[slot] = static_cast<int32_t>(n);
}

// consumed by ONE sequence-mode pass of the decoder, the num_hypotheses best next tokens are
// expanded by a one-step beam search, and every (sentence, alternative) pair is then continued
// independently with the requested search. The reference runs this sentence by sentence; here
// sentences with the same prefix length share their launches (the prefix length fixes the shape
// of the sequence-mode pass, hence its summation order - sentences of different prefix lengths
// cannot share it bit for bit). `sources` hold the prepared source ids (empty: skipped).
static void decode_alternatives(Translator& t, const std::vector<std::vector<int32_t>>& sources,
                                const std::vector<std::vector<int32_t>>& prefixes_in,
                                const ct2x_translation_options& opt, const DecodingOptions& dopt,
                                int32_t end_id, std::vector<DecodingResult>& out, double* encode_ms,
                                double* decode_ms) {
  Engine& e = *t.engine;
  const int32_t batch = static_cast<int32_t>(sources.size());
  const int32_t nh = dopt.num_hypotheses, max_length = dopt.max_length;
  // the encoder output of the reference's batch pads every sentence to the longest source
  int64_t s_max = 0;
  for (const auto& s : sources) s_max = std::max<int64_t>(s_max, static_cast<int64_t>(s.size()));
  std::map<size_t, std::vector<int32_t>> by_prefix_length;
  std::vector<std::vector<int32_t>> prefixes(batch);
  for (int32_t b = 0; b < batch; ++b) {
    if (sources[b].empty()) continue;
    prefixes[b] = prefixes_in[b];
    // start_tokens (= start token + prefix) longer than max_length + 1 are cut (:1139-1140)
    if (static_cast<int32_t>(prefixes[b].size()) > max_length) prefixes[b].resize(max_length);
    by_prefix_length[prefixes[b].size()].push_back(b);
  }
  for (const auto& group : by_prefix_length) {
    const int64_t plen = static_cast<int64_t>(group.first);
    const std::vector<int32_t>& members = group.second;
    const int64_t B = static_cast<int64_t>(members.size());
    if (plen == max_length) {  // the prefix is the whole hypothesis (:1176-1177)
      for (int32_t b : members) {
        out[b].hypotheses.assign(nh, prefixes[b]);
        if (dopt.return_scores) out[b].scores.assign(nh, 0.f);
      }
      continue;
    }
    auto make_state = [&](const std::vector<int32_t>& rows_of, int64_t beam, int64_t t_max) {
      const int64_t R = static_cast<int64_t>(rows_of.size());
      std::vector<int32_t> ids(R * s_max, 0), lens(R);
      for (int64_t i = 0; i < R; ++i) {
        const auto& src = sources[rows_of[i]];
        std::copy(src.begin(), src.end(), ids.begin() + i * s_max);
        lens[i] = static_cast<int32_t>(src.size());
      }
      const double t0 = now_ms();
      auto st = e.start(ids.data(), lens.data(), R, s_max, beam, t_max, plen);
      e.synchronize();
      *encode_ms += now_ms() - t0;
      if (plen > 0) {  // decoder inputs "<s> p0 .. p(n-2)" at positions 0 .. n-1 (:1148-1160)
        std::vector<int32_t> pre(static_cast<size_t>(R * beam * plen));
        for (int64_t i = 0; i < R; ++i)
          for (int64_t k = 0; k < beam; ++k) {
            int32_t* row = pre.data() + (i * beam + k) * plen;
            row[0] = t.start_id;
            for (int64_t p = 1; p < plen; ++p) row[p] = prefixes[rows_of[i]][p - 1];
          }
        e.prefill(*st, pre.data(), plen);
      }
      return st;
    };
    // ---- the num_hypotheses best next tokens: BeamSearch(num_hypotheses), one step (:1183-1197)
    const double td0 = now_ms();
    std::vector<int32_t> last_token(B);
    for (int64_t i = 0; i < B; ++i)
      last_token[i] = plen > 0 ? prefixes[members[i]].back() : t.start_id;
    std::vector<DecodingResult> expansion;
    {
      auto st = make_state(members, nh, plen + 2);
      DecodingOptions d1;
      d1.beam_size = nh;
      d1.num_hypotheses = nh;
      d1.max_length = 1;
      d1.min_length = 1;
      d1.length_penalty = 0.f;
      d1.return_scores = true;
      d1.use_graph = false;
      d1.start_step = static_cast<int32_t>(plen);
      d1.logits_processing = dopt.logits_processing;
      expansion = beam_search(e, *st, last_token, end_id, d1, nullptr, nullptr);
      e.synchronize();
    }
    // ---- every alternative that is probable enough becomes a row of the suffix search
    std::vector<int32_t> alt_sentence, alt_token;
    std::vector<float> alt_score;
    for (int64_t i = 0; i < B; ++i) {
      const int32_t b = members[i];
      for (int32_t k = 0; k < nh && k < static_cast<int32_t>(expansion[i].hypotheses.size()); ++k) {
        const float prob = std::exp(expansion[i].scores[k]);
        if (prob < opt.min_alternative_expansion_prob) break;
        std::vector<int32_t> hyp = prefixes[b];
        hyp.push_back(expansion[i].hypotheses[k].back());
        out[b].hypotheses.emplace_back(std::move(hyp));
        if (dopt.return_scores) out[b].scores.push_back(expansion[i].scores[k]);
        alt_sentence.push_back(b);
        alt_token.push_back(expansion[i].hypotheses[k].back());
        alt_score.push_back(expansion[i].scores[k]);
      }
    }
    const int64_t start_step = plen + 1;
    if (start_step == max_length || alt_sentence.empty()) {  // (:1236-1238)
      *decode_ms += now_ms() - td0;
      continue;
    }
    // ---- continue from each alternative independently (:1240-1256)
    DecodingOptions d2 = dopt;
    d2.max_length = std::max<int32_t>(max_length - static_cast<int32_t>(start_step), 0);
    d2.min_length = std::max<int32_t>(dopt.min_length - static_cast<int32_t>(start_step), 0);
    d2.num_hypotheses = 1;
    d2.start_step = static_cast<int32_t>(start_step);
    auto st = make_state(alt_sentence, d2.beam_size, start_step + d2.max_length + 1);
    {
      // the decode step of the last prefix token (the expansion step of the reference, whose
      // state the suffix search inherits)
      std::vector<int32_t> ids(st->rows);
      for (size_t a = 0; a < alt_sentence.size(); ++a) {
        const auto& pfx = prefixes[alt_sentence[a]];
        for (int64_t k = 0; k < d2.beam_size; ++k)
          ids[a * d2.beam_size + k] = plen > 0 ? pfx.back() : t.start_id;
      }
      // (in the reference that step belongs to the expansion search, whose first step runs one row
      // per sentence - the encoder-decoder attention takes the one-row forms, as at every first
      // step of a CPU beam search)
      st->prefix_first_step = static_cast<int32_t>(plen);
      e.set_step(*st, static_cast<int32_t>(plen));
      e.set_ids(*st, ids.data());
      e.step(*st, /*graph=*/false);
      e.advance_step(*st);
    }
    std::vector<DecodingResult> suffix =
        d2.beam_size == 1 ? greedy_search(e, *st, alt_token, end_id, d2, nullptr, nullptr)
                          : beam_search(e, *st, alt_token, end_id, d2, nullptr, nullptr);
    e.synchronize();
    std::map<int32_t, int32_t> next_slot;
    for (size_t a = 0; a < alt_sentence.size(); ++a) {
      const int32_t b = alt_sentence[a];
      const int32_t k = next_slot[b]++;
      auto& hyp = out[b].hypotheses[k];
      if (!suffix[a].hypotheses.empty())
        hyp.insert(hyp.end(), suffix[a].hypotheses[0].begin(), suffix[a].hypotheses[0].end());
      if (dopt.return_scores && !suffix[a].scores.empty()) out[b].scores[k] += suffix[a].scores[0];
    }
    *decode_ms += now_ms() - td0;
  }
}

// decode_alternatives for a decoder-only model (DecoderReplica::run_generation ->
// decode(..., return_alternatives), src/models/language_model.cc:218-245, src/decoding.cc:1122-1290):
// per prompt, the tokens after the batch's common `forward_length` are its start_tokens - all but
// the last are consumed in sequence mode and BEGIN every returned sequence, the num_hypotheses best
// next tokens are expanded by a one-step beam search at start_step = tokens consumed, and every
// alternative is continued independently; max_length / min_length count from step 0 of the
// decoder, i.e. INCLUDING the consumed prompt (the reference's arithmetic: suffix length =
// max_length - start_step). The reference runs prompt by prompt; prompts of one length share their
// launches here (every op is row-wise). Longer prompts of a ragged batch: their whole head is
// consumed by ONE sequence-mode pass here, by two in the reference (the batch's common part, then
// the example's rest) - the same arithmetic in a different summation order.
static void generate_alternatives(Translator& t, const std::vector<std::vector<int32_t>>& prompts,
                                  int32_t forward_length, const ct2x_translation_options& opt,
                                  const DecodingOptions& dopt, int32_t end_id,
                                  std::vector<DecodingResult>& out) {
  Engine& e = *t.engine;
  const int32_t nh = dopt.num_hypotheses, max_length = dopt.max_length;
  struct Plan {
    std::vector<int32_t> head;    // tokens consumed in sequence mode
    std::vector<int32_t> prefix;  // start_tokens[1:]: begins every returned sequence
    int32_t last = 0;             // start_tokens.back(): the expansion's input
  };
  std::vector<Plan> plans(prompts.size());
  std::map<size_t, std::vector<int32_t>> by_head;
  for (size_t b = 0; b < prompts.size(); ++b) {
    std::vector<int32_t> start_tokens(prompts[b].begin() + forward_length, prompts[b].end());
    if (static_cast<int32_t>(start_tokens.size()) > max_length + 1) start_tokens.resize(max_length + 1);  // (:1138-1139)
    Plan& pl = plans[b];
    const int32_t prefix_length = static_cast<int32_t>(start_tokens.size()) - 1;
    pl.prefix.assign(start_tokens.begin() + 1, start_tokens.end());
    pl.last = start_tokens.back();
    pl.head.assign(prompts[b].begin(), prompts[b].begin() + forward_length + prefix_length);
    if (prefix_length > 0 && prefix_length == max_length) {  // (:1176-1177)
      out[b].hypotheses.assign(nh, pl.prefix);
      if (dopt.return_scores) out[b].scores.assign(nh, 0.f);
      continue;
    }
    by_head[pl.head.size()].push_back(static_cast<int32_t>(b));
  }
  for (const auto& group : by_head) {
    const int64_t consumed = static_cast<int64_t>(group.first);
    const std::vector<int32_t>& members = group.second;
    const int64_t B = static_cast<int64_t>(members.size());
    auto make_state = [&](const std::vector<int32_t>& rows_of, int64_t beam, int64_t t_max) {
      const int64_t R = static_cast<int64_t>(rows_of.size());
      auto st = e.start_decoder_only(R, beam, t_max, consumed);
      if (consumed > 0) {
        std::vector<int32_t> pre(static_cast<size_t>(R * beam * consumed));
        for (int64_t i = 0; i < R; ++i)
          for (int64_t k = 0; k < beam; ++k)
            std::copy(plans[rows_of[i]].head.begin(), plans[rows_of[i]].head.end(),
                      pre.begin() + (i * beam + k) * consumed);
        e.prefill(*st, pre.data(), consumed);
      }
      return st;
    };
    // ---- the num_hypotheses best next tokens: BeamSearch(num_hypotheses), one step (:1183-1202)
    std::vector<int32_t> last_token(B);
    for (int64_t i = 0; i < B; ++i) last_token[i] = plans[members[i]].last;
    std::vector<DecodingResult> expansion;
    {
      auto st = make_state(members, nh, consumed + 2);
      DecodingOptions d1;
      d1.beam_size = nh;
      d1.num_hypotheses = nh;
      d1.max_length = 1;
      d1.min_length = 1;
      d1.length_penalty = 0.f;
      d1.return_scores = true;
      d1.use_graph = false;
      d1.start_step = static_cast<int32_t>(consumed);
      d1.logits_processing = dopt.logits_processing;
      expansion = beam_search(e, *st, last_token, end_id, d1, nullptr, nullptr);
      e.synchronize();
    }
    std::vector<int32_t> alt_prompt, alt_token;
    for (int64_t i = 0; i < B; ++i) {
      const int32_t b = members[i];
      for (int32_t k = 0; k < nh && k < static_cast<int32_t>(expansion[i].hypotheses.size()); ++k) {
        const float prob = std::exp(expansion[i].scores[k]);
        if (prob < opt.min_alternative_expansion_prob) break;
        std::vector<int32_t> hyp = plans[b].prefix;
        hyp.push_back(expansion[i].hypotheses[k].back());
        out[b].hypotheses.emplace_back(std::move(hyp));
        if (dopt.return_scores) out[b].scores.push_back(expansion[i].scores[k]);
        alt_prompt.push_back(b);
        alt_token.push_back(expansion[i].hypotheses[k].back());
      }
    }
    const int64_t start_step = consumed + 1;
    if (start_step == max_length || alt_prompt.empty()) continue;  // (:1236-1238)
    // ---- continue from each alternative independently (:1240-1256)
    DecodingOptions d2 = dopt;
    d2.max_length = std::max<int32_t>(max_length - static_cast<int32_t>(start_step), 0);
    d2.min_length = std::max<int32_t>(dopt.min_length - static_cast<int32_t>(start_step), 0);
    d2.num_hypotheses = 1;
    d2.start_step = static_cast<int32_t>(start_step);
    d2.return_prefix = false;
    if (d2.max_length == 0) continue;  // (a search of zero steps returns empty suffixes)
    auto st = make_state(alt_prompt, d2.beam_size, start_step + d2.max_length + 1);
    {
      // the decode step of the last prompt token (the expansion step of the reference, whose state
      // the suffix search inherits)
      std::vector<int32_t> ids(st->rows);
      for (size_t a = 0; a < alt_prompt.size(); ++a)
        for (int64_t k = 0; k < d2.beam_size; ++k) ids[a * d2.beam_size + k] = plans[alt_prompt[a]].last;
      st->prefix_first_step = static_cast<int32_t>(consumed);
      e.set_step(*st, static_cast<int32_t>(consumed));
      e.set_ids(*st, ids.data());
      e.step(*st, /*graph=*/false);
      e.advance_step(*st);
    }
    std::vector<DecodingResult> suffix =
        d2.beam_size == 1 ? greedy_search(e, *st, alt_token, end_id, d2, nullptr, nullptr)
                          : beam_search(e, *st, alt_token, end_id, d2, nullptr, nullptr);
    e.synchronize();
    std::map<int32_t, int32_t> next_slot;
    for (size_t a = 0; a < alt_prompt.size(); ++a) {
      const int32_t b = alt_prompt[a];
      const int32_t k = next_slot[b]++;
      auto& hyp = out[b].hypotheses[k];
      if (!suffix[a].hypotheses.empty())
        hyp.insert(hyp.end(), suffix[a].hypotheses[0].begin(), suffix[a].hypotheses[0].end());
      if (dopt.return_scores && !suffix[a].scores.empty()) out[b].scores[k] += suffix[a].scores[0];
    }
  }
}

static int translate_batch_impl(void* translator, const int32_t* src_ids,
                                const int32_t* src_lengths, int32_t batch, int32_t src_stride,
                                const int32_t* prefix_ids, const int32_t* prefix_lengths,
                                int32_t prefix_stride,
                                const struct ct2x_translation_options* opt_in, int32_t* out_ids,
                                int32_t* out_lengths, float* out_scores) {
  API_TRY
  auto* t = static_cast<Translator*>(translator);
  CT2X_CHECK(t != nullptr, "null translator");
  std::lock_guard<std::mutex> lock(t->mutex);
  CT2X_CHECK(t->engine->is_seq2seq(), "This model cannot be used as a sequence-to-sequence model");
  CT2X_CHECK(t->has_start_token, "models that need user decoder start tokens are not supported");
  ct2x_translation_options opt = *opt_in;
  DecodingOptions dopt = to_decoding_options(opt);
  validate_decoding_options(dopt);
  const int32_t nh = opt.num_hypotheses, max_len = opt.max_decoding_length;
  // TranslationOptions::end_token (sequence_to_sequence.cc:330-334): another token ends decoding
  const int32_t end_id = opt.end_token_plus_one > 0 ? opt.end_token_plus_one - 1 : t->eos_id;
  CT2X_CHECK(end_id >= 0 && end_id < static_cast<int32_t>(t->engine->vocab_size()),
             "end_token is not in the target vocabulary");
  const double t0 = now_ms();
  t->last = ct2x_exec_stats{};
  t->last_compactions = 0;

  // make_source_ids (sequence_to_sequence.cc:140-165) + Vocabulary::to_ids truncation
  // (src/vocabulary.cc:108-146).
  std::vector<std::vector<int32_t>> sources(batch);
  for (int32_t b = 0; b < batch; ++b) {
    auto& s = sources[b];
    if (src_lengths[b] == 0) continue;  // skip_translation: empty source -> empty result
    if (t->add_source_bos) s.push_back(static_cast<int32_t>(t->src_vocab->bos_id()));
    for (int32_t i = 0; i < src_lengths[b]; ++i) {
      const int32_t id = src_ids[static_cast<int64_t>(b) * src_stride + i];
      CT2X_CHECK(id >= 0 && id < static_cast<int32_t>(t->src_vocab->size()),
                 "Invalid source token ID " + std::to_string(id));
      s.push_back(id);
    }
    if (t->add_source_eos) s.push_back(static_cast<int32_t>(t->src_vocab->eos_id()));
    const size_t ml = static_cast<size_t>(opt.max_input_length);
    if (ml > 0 && s.size() > ml) {
      const int32_t eos = static_cast<int32_t>(t->src_vocab->eos_id());
      if (s.back() == eos)
        s[ml - 1] = eos;
      else if (s[s.size() - 2] == eos && ml >= 2) {
        s[ml - 2] = eos;
        s[ml - 1] = s.back();
      }
      s.resize(ml);
    }
  }

  std::vector<DecodingResult> all_results(batch);
  double encode_ms = 0, decode_ms = 0;
  for (int32_t b = 0; b < batch; ++b)
    for (int32_t k = 0; k < nh; ++k) {
      out_lengths[b * nh + k] = 0;
      if (opt.logits_out && opt.logits_counts_out) opt.logits_counts_out[b * nh + k] = 0;
    }
  if (opt.logits_out)
    CT2X_CHECK(opt.logits_counts_out != nullptr && opt.logits_steps > 0,
               "return_logits_vocab: logits_counts_out and logits_steps are required");
  if (opt.return_alternatives) {
    CT2X_CHECK(opt.prefix_bias_beta <= 0.f, "Biased decoding is not compatible with the "
               "return_alternatives mode");
    CT2X_CHECK(opt.attention_out == nullptr, "return_attention is not supported together with "
               "return_alternatives in this backend yet");
    std::vector<std::vector<int32_t>> prefixes(batch);
    if (prefix_ids != nullptr)
      for (int32_t b = 0; b < batch; ++b) {
        const int32_t* p = prefix_ids + static_cast<int64_t>(b) * prefix_stride;
        prefixes[b].assign(p, p + prefix_lengths[b]);
        for (int32_t id : prefixes[b])
          CT2X_CHECK(id >= 0 && id < static_cast<int32_t>(t->engine->vocab_size()),
                     "Invalid target prefix token ID " + std::to_string(id));
      }
    decode_alternatives(*t, sources, prefixes, opt, dopt, end_id, all_results, &encode_ms, &decode_ms);
  } else {
  // rebatch_input (src/batch_reader.cc:175-228): longest first, split by max_batch_size.
  std::vector<int32_t> order;
  for (int32_t b = 0; b < batch; ++b)
    if (!sources[b].empty()) order.push_back(b);
  if (opt.max_batch_size > 0)
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
      return sources[a].size() > sources[b].size();
    });
  // Work list: contiguous groups of the (sorted) examples. max_batch_size bounds a group like
  // rebatch_input does; num_streams additionally splits the work into that many groups so that
  // independent sub-batches decode concurrently on separate streams. Every op on the path is
  // row-wise, so the grouping cannot change a hypothesis (tests/test_translator_gpu.py).
  int32_t streams = opt.num_streams;
  if (streams <= 0) {
    const char* env = std::getenv("CT2X_NUM_STREAMS");
    streams = env ? std::atoi(env) : 1;
  }
  streams = std::max<int32_t>(1, std::min<int32_t>(streams, 16));
  if (t->engine->profiling()) streams = 1;  // event bookkeeping is single-stream
  if (dopt.callback) streams = 1;           // the callback is called from one thread
  const size_t ref_chunk = opt.max_batch_size > 0 ? static_cast<size_t>(opt.max_batch_size)
                                                  : std::max<size_t>(order.size(), 1);
  // Concurrent hipGraph captures from several threads are not reliable on ROCm 7.2 ("operation
  // failed due to a previous error during capture"); captures are serialised (Engine::step) and
  // CT2X_STREAM_GRAPHS=1 lets concurrent sessions replay graphs; by default they launch eagerly.
  if (streams > 1) {
    static const bool stream_graphs = [] {
      const char* e = std::getenv("CT2X_STREAM_GRAPHS");
      return e != nullptr && e[0] == '1';
    }();
    if (!stream_graphs) dopt.use_graph = false;
  }
  // A group = (begin, end, padded source length). The padded length is that of the REFERENCE's
  // batch (rebatch_input) the group is cut from: MKL's one-row attention forms depend on the
  // padded key count, so a sub-batch must keep its parent's padding to stay bit-identical.
  struct Group {
    size_t begin, end;
    int64_t s_max;
  };
  std::vector<Group> groups;
  for (size_t rb = 0; rb < order.size(); rb += ref_chunk) {
    const size_t re = std::min(order.size(), rb + ref_chunk);
    int64_t s_max = 0;
    for (size_t i = rb; i < re; ++i)
      s_max = std::max<int64_t>(s_max, static_cast<int64_t>(sources[order[i]].size()));
    size_t sub = re - rb;
    if (streams > 1) sub = std::max<size_t>(1, (sub + streams - 1) / streams);
    for (size_t begin = rb; begin < re; begin += sub)
      groups.push_back(Group{begin, std::min(re, begin + sub), s_max});
  }

  std::mutex stats_mutex;
  std::atomic<size_t> next_group{0};
  std::exception_ptr first_error;
  auto worker = [&]() {
    try {
      CT2X_HIP_CHECK(hipSetDevice(t->engine->device_index()));
      while (true) {
        const size_t gi = next_group.fetch_add(1);
        if (gi >= groups.size()) break;
        const size_t begin = groups[gi].begin, end = groups[gi].end;
        const int64_t B = static_cast<int64_t>(end - begin);
        const int64_t s_max = groups[gi].s_max;
        // Random sampling with num_hypotheses > 1 decodes every entry num_hypotheses times
        // (greedy_search_samples; the reference repeats the encoded state, repeat_batch
        // src/decoding.cc:753-757 - the encoder is row-wise, so encoding the repeated rows gives
        // the same memory)
        const int64_t samples = (dopt.beam_size == 1 && !(dopt.prefix_bias_beta > 0)) ? dopt.num_hypotheses : 1;
        const int64_t R = B * samples;
        std::vector<int32_t> ids(R * s_max, 0), lens(R);
        for (int64_t i = 0; i < R; ++i) {
          const auto& s = sources[order[begin + i / samples]];
          std::copy(s.begin(), s.end(), ids.begin() + i * s_max);
          lens[i] = static_cast<int32_t>(s.size());
        }
        const double te0 = now_ms();
        const bool biased = dopt.prefix_bias_beta > 0 && prefix_ids != nullptr;
        // session reuse (Translator::session)
        static const bool session_cache = [] {
          const char* e = std::getenv("CT2X_SESSION_CACHE");
          return e == nullptr || e[0] != '0';
        }();
        const bool plain = session_cache && groups.size() == 1 && streams == 1 && samples == 1 &&
                           dopt.sampling_topk == 1 && prefix_ids == nullptr && !(dopt.prefix_bias_beta > 0) &&
                           dopt.coverage_penalty == 0 && !dopt.return_attention &&
                           !dopt.return_logits_vocab && !dopt.callback &&
                           !dopt.logits_processing.active() && dopt.whisper == nullptr &&
                           t->engine->tp_world() == 1 && !t->engine->profiling();
        int32_t patience_bits = 0;
        std::memcpy(&patience_bits, &dopt.patience, sizeof(patience_bits));
        const std::vector<int64_t> key = {
            R, s_max, dopt.beam_size, dopt.num_hypotheses, patience_bits,
            dopt.max_length, dopt.min_length, dopt.return_scores ? 1 : 0, end_id,
            dopt.use_graph ? 1 : 0, dopt.sync_every, dopt.start_step,
            static_cast<int64_t>(reinterpret_cast<uintptr_t>(thread_stream()))};
        // A cluster of the decoder-stack launch that waited in vain (another process's persistent
        // launch on this GPU, a debugger, ...) voids the group's results: the handle then decodes with
        // the launch-per-op step from here on and the group runs again (Engine::check_cluster).
        for (int attempt = 0;; ++attempt) {
          try {
            std::unique_ptr<DecodeState> st;
            int64_t st_bytes = t->session_bytes;
            if (plain && t->session && t->session_key == key) {
              st = std::move(t->session);
              t->engine->reencode(*st, ids.data(), lens.data());
              ++t->session_hits;
            } else {
              t->session.reset();
              const int64_t before = allocated_bytes();
              st = t->engine->start(ids.data(), lens.data(), R, s_max, dopt.beam_size,
                                    dopt.max_length, 0, /*beam_search_state=*/biased);
              st_bytes = allocated_bytes() - before;
              // (sub-batches decoding concurrently on several streams keep the launch-per-op step: the
              // decoder-stack launch runs one session at a time per device, Engine::cluster_gate)
              if (streams > 1) st->cl_xch = nullptr;
            }
            t->engine->synchronize();
            const double te1 = now_ms();
            std::vector<int32_t> start_ids(B, t->start_id);
            // target_prefix (make_target_ids with is_prefix, sequence_to_sequence.cc:168-186): forced
            // after the decoder start token, part of the returned hypothesis (return_prefix).
            std::vector<std::vector<int32_t>> prefix;
            bool any_prefix = false;
            if (prefix_ids != nullptr) {
              prefix.resize(B);
              for (int64_t i = 0; i < B; ++i) {
                const int32_t b = order[begin + i];
                const int32_t* p = prefix_ids + static_cast<int64_t>(b) * prefix_stride;
                prefix[i].assign(p, p + prefix_lengths[b]);
                for (int32_t id : prefix[i])
                  CT2X_CHECK(id >= 0 && id < static_cast<int32_t>(t->engine->vocab_size()),
                             "Invalid target prefix token ID " + std::to_string(id));
                any_prefix = any_prefix || !prefix[i].empty();
              }
            }
            const std::vector<std::vector<int32_t>>* pf = any_prefix ? &prefix : nullptr;
            DecodeStats ds;
            DecodingOptions gopt = dopt;
            if (dopt.callback) {  // the callback sees the caller's batch indices
              gopt.callback = [&dopt, &order, begin](const DecodingStepResult& r) {
                DecodingStepResult m = r;
                m.batch_id = order[begin + static_cast<size_t>(r.batch_id)];
                return dopt.callback(m);
              };
            }
            // make_search_strategy (src/decoding.cc:1075-1087): biased decoding is a beam search
            std::vector<DecodingResult> results =
                (dopt.beam_size == 1 && !(dopt.prefix_bias_beta > 0 && pf))
                    ? greedy_search_samples(*t->engine, *st, start_ids, end_id, gopt, &ds, pf)
                    : beam_search(*t->engine, *st, start_ids, end_id, gopt, &ds, pf);
            t->engine->synchronize();
            const double te2 = now_ms();
            for (int64_t i = 0; i < B; ++i) all_results[order[begin + i]] = std::move(results[i]);
            // A kept session holds its buffers until the next call: only sessions below
            // CT2X_SESSION_CACHE_MB (default 2048) stay - a 256-row NLLB-3.3B session is tens of GB that
            // the reference would have returned to its allocator.
            static const int64_t keep_limit = [] {
              const char* e = std::getenv("CT2X_SESSION_CACHE_MB");
              return (e ? std::atoll(e) : 2048LL) << 20;
            }();
            if (plain && ds.compactions == 0 && st->rows == R * dopt.beam_size && !st->graph_failed &&
                st_bytes <= keep_limit) {
              t->session = std::move(st);  // (an exception on the way here leaves no session behind)
              t->session_key = key;
              t->session_bytes = st_bytes;
            }
            std::lock_guard<std::mutex> lock(stats_mutex);
            encode_ms += te1 - te0;
            decode_ms += te2 - te1;
            t->last.decode_steps = std::max<int64_t>(t->last.decode_steps, ds.steps);
            t->last_compactions += ds.compactions;
            break;
          } catch (const ClusterTimeout&) {
            if (attempt > 0) throw;
            t->engine->disable_cluster();
            t->session.reset();
            ++t->cluster_fallbacks;
          }
        }
      }
    } catch (...) {
      std::lock_guard<std::mutex> lock(stats_mutex);
      if (!first_error) first_error = std::current_exception();
    }
  };
  const size_t n_threads = std::min<size_t>(static_cast<size_t>(streams), groups.size());
  if (n_threads <= 1) {
    worker();
  } else {
    const std::function<void()> job = worker;
    t->pool.run(n_threads, job);
  }
  if (first_error) std::rethrow_exception(first_error);
  }  // (plain decoding)
  for (int32_t b = 0; b < batch; ++b) {
    DecodingResult& r = all_results[b];
    for (int32_t k = 0; k < nh; ++k) {
      if (k >= static_cast<int32_t>(r.hypotheses.size())) {
        // fewer alternatives than num_hypotheses passed min_alternative_expansion_prob
        if (opt.return_alternatives && !sources[b].empty()) out_lengths[b * nh + k] = -1;
        continue;
      }
      auto& hyp = r.hypotheses[k];
      // "Remove EOS token" (sequence_to_sequence.cc:374-384).
      if (!opt.return_end_token)
        while (!hyp.empty() && hyp.back() == end_id) hyp.pop_back();
      const int32_t len = static_cast<int32_t>(hyp.size());
      std::copy(hyp.begin(), hyp.begin() + std::min(len, max_len),
                out_ids + (static_cast<int64_t>(b) * nh + k) * max_len);
      out_lengths[b * nh + k] = len;
      if (opt.return_scores && out_scores && k < static_cast<int32_t>(r.scores.size()))
        out_scores[b * nh + k] = r.scores[k];
      if (opt.attention_out && k < static_cast<int32_t>(r.attention.size())) {
        // one source attention vector per returned token (the end token's goes with it)
        // "Remove attention positions for padding and implicit special tokens"
        // (sequence_to_sequence.cc:399-412): the vector covers the caller's source tokens
        const auto& att = r.attention[k];
        float* dst = opt.attention_out + (static_cast<int64_t>(b) * nh + k) * max_len * opt.attention_stride;
        const size_t first = t->add_source_bos ? 1 : 0;
        const size_t with_specials = sources[b].size();
        const size_t last = with_specials - (t->add_source_eos ? 1 : 0);
        for (int32_t s_ = 0; s_ < len && s_ < max_len && s_ < static_cast<int32_t>(att.size()); ++s_) {
          float* row = dst + static_cast<int64_t>(s_) * opt.attention_stride;
          std::fill(row, row + opt.attention_stride, 0.f);
          for (size_t j = first; j < last && j < att[s_].size() &&
                                 j - first < static_cast<size_t>(opt.attention_stride); ++j)
            row[j - first] = att[s_][j];
        }
      }
      if (opt.logits_out) write_logits(opt, r, (static_cast<int64_t>(b) * nh + k), k, t->engine->vocab_size());
      if (k == 0) t->last.num_tokens += len;  // ExecutionStats: hypothesis 0 tokens
    }
  }
  t->last.num_examples = batch;
  t->last.kernels_per_step = t->engine->kernels_per_step();
  t->last.encode_ms = encode_ms;
  t->last.decode_ms = decode_ms;
  t->last.total_ms = now_ms() - t0;
  API_CATCH
}

int ct2x_translate_batch(void* translator, const int32_t* src_ids, const int32_t* src_lengths,
                         int32_t batch, int32_t src_stride,
                         const struct ct2x_translation_options* opt, int32_t* out_ids,
                         int32_t* out_lengths, float* out_scores) {
  return translate_batch_impl(translator, src_ids, src_lengths, batch, src_stride, nullptr, nullptr,
                              0, opt, out_ids, out_lengths, out_scores);
}

int ct2x_translate_batch_prefix(void* translator, const int32_t* src_ids,
                                const int32_t* src_lengths, int32_t batch, int32_t src_stride,
                                const int32_t* prefix_ids, const int32_t* prefix_lengths,
                                int32_t prefix_stride,
                                const struct ct2x_translation_options* opt, int32_t* out_ids,
                                int32_t* out_lengths, float* out_scores) {
  return translate_batch_impl(translator, src_ids, src_lengths, batch, src_stride, prefix_ids,
                              prefix_lengths, prefix_stride, opt, out_ids, out_lengths, out_scores);
}

int ct2x_generate_batch(void* translator, const int32_t* prompt_ids,
                        const int32_t* prompt_lengths, int32_t batch, int32_t stride,
                        const struct ct2x_translation_options* opt_in, int32_t* out_ids,
                        int32_t* out_lengths, float* out_scores) {
  API_TRY
  auto* t = static_cast<Translator*>(translator);
  release_kept_session(t);  // (the last plain translate call's buffers: not this entry point's to keep)
  CT2X_CHECK(t != nullptr, "null translator");
  std::lock_guard<std::mutex> lock(t->mutex);
  CT2X_CHECK(!t->engine->is_seq2seq(), "This model cannot be used as a sequence generator");
  // DecoderReplica::run_generation with include_prompt_in_result = false
  // (src/models/language_model.cc:150-282): all but the last prompt token are consumed by one
  // sequence-mode pass, decoding starts from the last prompt token at step = prompt_length - 1.
  ct2x_translation_options opt = *opt_in;
  DecodingOptions dopt = to_decoding_options(opt);
  validate_decoding_options(dopt);
  CT2X_CHECK(batch > 0, "No decoder start tokens are set");
  const int32_t beam = dopt.beam_size;
  const int32_t end_id = opt.end_token_plus_one > 0 ? opt.end_token_plus_one - 1 : t->eos_id;
  CT2X_CHECK(end_id >= 0 && end_id < static_cast<int32_t>(t->engine->vocab_size()),
             "end_token is not in the vocabulary");
  // Prompts of different lengths: the shortest prompt minus one token is consumed by the
  // sequence-mode pass for every row; the remaining tokens of a row are its start token and a
  // hard decoding prefix (language_model.cc:218-238 -> decode(..., prefix_ids)).
  int32_t min_len_prompt = prompt_lengths[0], max_len_prompt = prompt_lengths[0];
  for (int32_t b = 0; b < batch; ++b) {
    CT2X_CHECK(prompt_lengths[b] >= 1, "generate_batch: empty prompt");
    min_len_prompt = std::min(min_len_prompt, prompt_lengths[b]);
    max_len_prompt = std::max(max_len_prompt, prompt_lengths[b]);
  }
  // include_prompt_in_result (the reference's default, language_model.cc:218-238 skipped): nothing
  // is forwarded in sequence mode; token 0 starts the decoding, the rest of the prompt is a hard
  // prefix that is part of the result (DecodingOptions::return_prefix stays true).
  const bool with_prompt = opt.include_prompt_in_result != 0;
  CT2X_CHECK(!(with_prompt && opt.return_alternatives),
             "include_prompt_in_result with return_alternatives is not built for the generator in this backend");
  // ... behind a static prompt (language_model.cc:184-206): its tokens - the head of every row - are
  // forwarded in sequence mode and neither returned nor counted
  const int32_t static_len = std::max<int32_t>(0, opt.static_prompt_length);
  CT2X_CHECK(static_len < min_len_prompt, "generate_batch: a prompt needs a start token behind the static prompt");
  // Without the prompt in the result the reference forwards the shortest prompt minus one token -
  // and when that is NO token (a one-token prompt in the batch) it forwards nothing and
  // DecodingOptions::return_prefix keeps its default, true (language_model.cc:224-237: the
  // assignment sits inside `if (forward_length > 0)`): the rest of the longer prompts is then
  // returned and counted like with include_prompt_in_result, only the start token is not put in front.
  const bool returns_prefix = with_prompt || min_len_prompt - static_len - 1 == 0;
  const int32_t forward_length = returns_prefix ? static_len : min_len_prompt - 1;
  const int32_t longest_prefix = max_len_prompt - forward_length - 1;
  // The reference's front-end decodes a batch sorted from the longest to the shortest example
  // (rebatch_input, src/batch_reader.cc:187-199) and returns the results in the caller's order.
  // Results do not depend on the order of a batch - except in ONE configuration: a hidden prefix
  // under beam search with min_length > 0, where apply_min_length disables the end token by the
  // INDEX of the live entry (see beam_search / beam_logprobs_kernel). There the entries are decoded
  // in the reference's order: the same std::sort call with the same comparator on the same input
  // (a batch that is already in non-increasing order is left alone: the reference's own front-end
  // on top of this entry point has sorted it).
  std::vector<int32_t> order(batch);
  std::iota(order.begin(), order.end(), 0);
  std::vector<int32_t> sorted_ids, sorted_lengths;
  {
    static const bool per_entry_min = [] {
      const char* e = std::getenv("CT2X_HIDDEN_PREFIX_MIN_LENGTH_PER_ENTRY");
      return e != nullptr && e[0] == '1';
    }();
    bool non_increasing = true;
    for (int32_t b = 1; b < batch; ++b) non_increasing = non_increasing && prompt_lengths[b] <= prompt_lengths[b - 1];
    if (beam > 1 && longest_prefix > 0 && dopt.min_length > 0 && !per_entry_min && !non_increasing &&
        !opt.return_alternatives && !returns_prefix) {
      std::vector<size_t> example_index(static_cast<size_t>(batch));
      std::iota(example_index.begin(), example_index.end(), size_t(0));
      std::sort(example_index.begin(), example_index.end(), [&](size_t i1, size_t i2) {
        return prompt_lengths[i1] > prompt_lengths[i2];
      });
      sorted_ids.resize(static_cast<size_t>(batch) * stride);
      sorted_lengths.resize(batch);
      for (int32_t b = 0; b < batch; ++b) {
        order[b] = static_cast<int32_t>(example_index[b]);
        sorted_lengths[b] = prompt_lengths[order[b]];
        std::copy(prompt_ids + static_cast<int64_t>(order[b]) * stride,
                  prompt_ids + static_cast<int64_t>(order[b]) * stride + prompt_lengths[order[b]],
                  sorted_ids.begin() + static_cast<int64_t>(b) * stride);
      }
      prompt_ids = sorted_ids.data();
      prompt_lengths = sorted_lengths.data();
    }
  }
  const double t0 = now_ms();
  t->last = ct2x_exec_stats{};
  t->last_compactions = 0;
  if (opt.return_alternatives) {
    CT2X_CHECK(dopt.sampling_topk == 1, "return_alternatives with random sampling is not built for the "
               "generator in this backend");
    CT2X_CHECK(opt.logits_out == nullptr && !dopt.callback,
               "return_logits_vocab / callbacks are not built together with return_alternatives for "
               "the generator in this backend");
    std::vector<std::vector<int32_t>> prompts(batch);
    for (int32_t b = 0; b < batch; ++b) {
      const int32_t* row = prompt_ids + static_cast<int64_t>(b) * stride;
      prompts[b].assign(row, row + prompt_lengths[b]);
      for (int32_t id : prompts[b])
        CT2X_CHECK(id >= 0 && id < static_cast<int32_t>(t->engine->vocab_size()),
                   "Invalid prompt token ID " + std::to_string(id));
    }
    std::vector<DecodingResult> results(batch);
    generate_alternatives(*t, prompts, forward_length, opt, dopt, end_id, results);
    const int32_t nh_alt = dopt.num_hypotheses, max_len_alt = opt.max_decoding_length;
    for (int32_t b = 0; b < batch; ++b)
      for (int32_t k = 0; k < nh_alt; ++k) {
        // -1: fewer alternatives than num_hypotheses passed min_alternative_expansion_prob
        out_lengths[b * nh_alt + k] = -1;
        if (k >= static_cast<int32_t>(results[b].hypotheses.size())) continue;
        auto& hyp = results[b].hypotheses[k];
        if (!opt.return_end_token)
          while (!hyp.empty() && hyp.back() == end_id) hyp.pop_back();
        const int32_t len = static_cast<int32_t>(hyp.size());
        std::copy(hyp.begin(), hyp.begin() + std::min(len, max_len_alt),
                  out_ids + (static_cast<int64_t>(b) * nh_alt + k) * max_len_alt);
        out_lengths[b * nh_alt + k] = len;
        if (opt.return_scores && out_scores && k < static_cast<int32_t>(results[b].scores.size()))
          out_scores[b * nh_alt + k] = results[b].scores[k];
        if (k == 0) t->last.num_tokens += len;
      }
    t->last.num_examples = batch;
    t->last.total_ms = now_ms() - t0;
    return 0;
  }
  const int64_t t_max = static_cast<int64_t>(forward_length) + longest_prefix + dopt.max_length + 1;
  // Beam search: every hypothesis row starts from the same prefilled prompt state; the prompt is
  // forwarded once per row (beam x the prefill work instead of a cache replication pass).
  // Random sampling with num_hypotheses > 1 (greedy_search_samples): num_hypotheses rows per prompt.
  const int32_t nh = dopt.num_hypotheses;
  const int32_t mult = beam > 1 ? beam : nh;  // rows per prompt
  auto st = beam > 1 ? t->engine->start_decoder_only(batch, beam, t_max, forward_length)
                     : t->engine->start_decoder_only(static_cast<int64_t>(batch) * nh, 1, t_max, forward_length);
  const int64_t rows = static_cast<int64_t>(batch) * mult;
  std::vector<int32_t> prefill_ids(static_cast<size_t>(rows) * std::max(forward_length, 1));
  std::vector<int32_t> start_ids(batch);
  std::vector<std::vector<int32_t>> prefix(batch);
  for (int32_t b = 0; b < batch; ++b) {
    const int32_t* row = prompt_ids + static_cast<int64_t>(b) * stride;
    for (int32_t k = 0; k < mult; ++k)
      for (int32_t i = 0; i < forward_length; ++i)
        prefill_ids[(static_cast<size_t>(b) * mult + k) * forward_length + i] = row[i];
    start_ids[b] = row[forward_length];
    prefix[b].assign(row + forward_length + 1, row + prompt_lengths[b]);
  }
  const double te0 = now_ms();
  if (forward_length > 0) t->engine->prefill(*st, prefill_ids.data(), forward_length);
  t->engine->synchronize();
  const double te1 = now_ms();
  dopt.start_step = forward_length;
  dopt.return_prefix = returns_prefix;
  DecodeStats ds;
  std::vector<DecodingResult> results =
      beam > 1 ? beam_search(*t->engine, *st, start_ids, end_id, dopt, &ds,
                             longest_prefix > 0 ? &prefix : nullptr)
               : greedy_search_samples(*t->engine, *st, start_ids, end_id, dopt, &ds,
                                       longest_prefix > 0 ? &prefix : nullptr);
  t->engine->synchronize();
  const double te2 = now_ms();
  const int32_t max_len = opt.max_decoding_length;
  // outputs [batch, num_hypotheses, max_decoding_length] / [batch, num_hypotheses], as ct2x_translate_batch
  for (int32_t sb = 0; sb < batch; ++sb)
    for (int32_t k = 0; k < nh; ++k) {
      const int32_t b = order[sb];  // (the caller's index of the entry decoded as entry sb)
      out_lengths[b * nh + k] = 0;
      if (k >= static_cast<int32_t>(results[sb].hypotheses.size())) continue;
      auto& hyp = results[sb].hypotheses[k];
      if (!opt.return_end_token)
        while (!hyp.empty() && hyp.back() == end_id) hyp.pop_back();
      const int32_t len = static_cast<int32_t>(hyp.size());
      std::copy(hyp.begin(), hyp.begin() + std::min(len, max_len),
                out_ids + (static_cast<int64_t>(b) * nh + k) * max_len);
      out_lengths[b * nh + k] = len;
      if (opt.return_scores && out_scores && k < static_cast<int32_t>(results[sb].scores.size()))
        out_scores[b * nh + k] = results[sb].scores[k];
      if (opt.logits_out) {
        CT2X_CHECK(opt.logits_counts_out != nullptr && opt.logits_steps > 0,
                   "return_logits_vocab: logits_counts_out and logits_steps are required");
        write_logits(opt, results[sb], (static_cast<int64_t>(b) * nh + k), k, t->engine->vocab_size());
      }
      if (k == 0) t->last.num_tokens += len;
    }
  t->last.num_examples = batch;
  t->last_compactions += ds.compactions;
  t->last.decode_steps = ds.steps;
  t->last.kernels_per_step = t->engine->kernels_per_step();
  t->last.encode_ms = te1 - te0;  // prompt prefill
  t->last.decode_ms = te2 - te1;
  t->last.total_ms = now_ms() - t0;
  API_CATCH
}

namespace {

// Vocabulary::to_ids truncation (src/vocabulary.cc:130-141): keep </s> (and an optional language
// code after it) in the last positions.
void truncate_ids(std::vector<int32_t>& ids, size_t max_length, int32_t eos) {
  if (max_length == 0 || ids.size() <= max_length) return;
  if (ids.back() == eos)
    ids[max_length - 1] = eos;
  else if (ids.size() >= 2 && ids[ids.size() - 2] == eos && max_length >= 2) {
    ids[max_length - 2] = eos;
    ids[max_length - 1] = ids.back();
  }
  ids.resize(max_length);
}

}  // namespace

int ct2x_score_batch(void* translator, const int32_t* src_ids, const int32_t* src_lengths,
                     int32_t batch, int32_t src_stride, const int32_t* tgt_ids,
                     const int32_t* tgt_lengths, int32_t tgt_stride, int32_t max_input_length,
                     int32_t* out_ids, float* out_scores, int32_t out_stride,
                     int32_t* out_lengths) {
  API_TRY
  auto* t = static_cast<Translator*>(translator);
  release_kept_session(t);  // (the last plain translate call's buffers: not this entry point's to keep)
  CT2X_CHECK(t != nullptr, "null translator");
  std::lock_guard<std::mutex> lock(t->mutex);
  CT2X_CHECK(t->engine->is_seq2seq(), "This model cannot be used as a sequence-to-sequence model");
  CT2X_CHECK(t->has_start_token, "models that need user decoder start tokens are not supported");
  CT2X_CHECK(max_input_length >= 0, "max_input_length must be >= 0");
  const int32_t src_eos = static_cast<int32_t>(t->src_vocab->eos_id());
  // make_source_ids / make_target_ids (sequence_to_sequence.cc:143-186): the full target is
  // "<s> a b c </s>", one more than max_input_length since the decoder input is "<s> a b c".
  std::vector<std::vector<int32_t>> sources(batch), targets(batch);
  for (int32_t b = 0; b < batch; ++b) {
    auto& s = sources[b];
    if (src_lengths[b] > 0 || t->add_source_bos || t->add_source_eos) {
      if (t->add_source_bos) s.push_back(static_cast<int32_t>(t->src_vocab->bos_id()));
      for (int32_t i = 0; i < src_lengths[b]; ++i) {
        const int32_t id = src_ids[static_cast<int64_t>(b) * src_stride + i];
        CT2X_CHECK(id >= 0 && id < static_cast<int32_t>(t->src_vocab->size()),
                   "Invalid source token ID " + std::to_string(id));
        s.push_back(id);
      }
      if (t->add_source_eos) s.push_back(src_eos);
      truncate_ids(s, static_cast<size_t>(max_input_length), src_eos);
    }
    auto& y = targets[b];
    y.push_back(t->start_id);
    for (int32_t i = 0; i < tgt_lengths[b]; ++i) {
      const int32_t id = tgt_ids[static_cast<int64_t>(b) * tgt_stride + i];
      CT2X_CHECK(id >= 0 && id < static_cast<int32_t>(t->engine->vocab_size()),
                 "Invalid target token ID " + std::to_string(id));
      y.push_back(id);
    }
    y.push_back(t->eos_id);
    truncate_ids(y, max_input_length > 0 ? static_cast<size_t>(max_input_length) + 1 : 0, t->eos_id);
    CT2X_CHECK(static_cast<int32_t>(y.size()) - 1 <= out_stride, "ct2x_score_batch: out_stride is too small");
  }
  std::vector<int32_t> order;
  for (int32_t b = 0; b < batch; ++b) {
    const int32_t n = static_cast<int32_t>(targets[b].size()) - 1;
    out_lengths[b] = n;
    for (int32_t i = 0; i < n; ++i) {
      out_ids[static_cast<int64_t>(b) * out_stride + i] = targets[b][i + 1];
      out_scores[static_cast<int64_t>(b) * out_stride + i] = 0.f;  // skip_scoring: empty source
    }
    if (!sources[b].empty()) order.push_back(b);
  }
  if (order.empty()) return 0;
  const int64_t B = static_cast<int64_t>(order.size());
  int64_t s_max = 0, steps = 0;
  for (int32_t b : order) {
    s_max = std::max<int64_t>(s_max, static_cast<int64_t>(sources[b].size()));
    steps = std::max<int64_t>(steps, static_cast<int64_t>(targets[b].size()) - 1);
  }
  std::vector<int32_t> ids(B * s_max, 0), lens(B);
  for (int64_t i = 0; i < B; ++i) {
    const auto& s = sources[order[i]];
    std::copy(s.begin(), s.end(), ids.begin() + i * s_max);
    lens[i] = static_cast<int32_t>(s.size());
  }
  auto st = t->engine->start(ids.data(), lens.data(), B, s_max, 1, steps);
  // decoder inputs "<s> a b c", outputs "a b c </s>"; rows past their end repeat </s> (ignored)
  std::vector<int32_t> inputs(steps * B, t->eos_id), outputs(steps * B, t->eos_id);
  for (int64_t i = 0; i < B; ++i) {
    const auto& y = targets[order[i]];
    for (size_t s = 0; s + 1 < y.size(); ++s) {
      inputs[s * B + i] = y[s];
      outputs[s * B + i] = y[s + 1];
    }
  }
  t->engine->score_targets(*st, inputs.data(), outputs.data(), steps);
  std::vector<int32_t> h_ids(steps * B);
  std::vector<float> h_scores(steps * B);
  t->engine->read_history(*st, 0, steps, h_ids.data(), h_scores.data());
  for (int64_t i = 0; i < B; ++i) {
    const int32_t b = order[i];
    for (int32_t s = 0; s < out_lengths[b]; ++s)
      out_scores[static_cast<int64_t>(b) * out_stride + s] = h_scores[s * B + i];
  }
  API_CATCH
}

// Generator::forward_batch (include/ctranslate2/generator.h:31-35; SequenceGeneratorReplica::
// forward, src/models/language_model.cc:73-100): one sequence-mode pass, logits of every position.
int ct2x_forward_batch(void* translator, const int32_t* ids, const int32_t* lengths, int32_t batch,
                       int32_t stride, int32_t return_log_probs, float* out) {
  API_TRY
  auto* t = static_cast<Translator*>(translator);
  release_kept_session(t);  // (the last plain translate call's buffers: not this entry point's to keep)
  CT2X_CHECK(t != nullptr, "null translator");
  std::lock_guard<std::mutex> lock(t->mutex);
  Engine& e = *t->engine;
  CT2X_CHECK(!e.is_seq2seq(), "This model cannot be used as a sequence generator");
  CT2X_CHECK(batch > 0 && ids && lengths && out, "ct2x_forward_batch: empty input");
  int64_t T = 0;
  for (int32_t b = 0; b < batch; ++b) {
    CT2X_CHECK(lengths[b] >= 0 && lengths[b] <= stride, "ct2x_forward_batch: bad length");
    T = std::max<int64_t>(T, lengths[b]);
  }
  CT2X_CHECK(T > 0, "ct2x_forward_batch: every sequence is empty");
  const int64_t V = e.vocab_size(), rows = static_cast<int64_t>(batch) * T;
  std::vector<int32_t> padded(rows, 0);  // make_sequence_inputs pads with 0
  for (int32_t b = 0; b < batch; ++b)
    for (int32_t i = 0; i < lengths[b]; ++i) {
      const int32_t id = ids[static_cast<int64_t>(b) * stride + i];
      CT2X_CHECK(id >= 0 && id < V, "Invalid token ID " + std::to_string(id));
      padded[b * T + i] = id;
    }
  auto st = e.start_decoder_only(batch, 1, T, T);
  float* d_logits = static_cast<float*>(device_malloc(sizeof(float) * rows * V));
  st->owned.push_back(d_logits);
  e.prefill(*st, padded.data(), T, d_logits);
  if (return_log_probs) launch_softmax(d_logits, nullptr, rows, V, /*log=*/true, d_logits, st->stream);
  CT2X_HIP_CHECK(hipMemcpyAsync(out, d_logits, sizeof(float) * rows * V, hipMemcpyDeviceToHost, st->stream));
  CT2X_HIP_CHECK(hipStreamSynchronize(st->stream));
  API_CATCH
}

// Generator::score_batch (include/ctranslate2/generator.h:21-25; DecoderReplica::run_scoring ->
// score_sequences, src/models/language_model.cc:110-135, src/scoring.cc:9-60): the log-probability
// of tokens[1:] given the tokens before them; sequences of fewer than 2 tokens are skipped
// (out_lengths 0). out_scores [batch, stride].
int ct2x_generator_score_batch(void* translator, const int32_t* ids, const int32_t* lengths,
                               int32_t batch, int32_t stride, int32_t max_input_length,
                               float* out_scores, int32_t* out_lengths) {
  API_TRY
  auto* t = static_cast<Translator*>(translator);
  release_kept_session(t);  // (the last plain translate call's buffers: not this entry point's to keep)
  CT2X_CHECK(t != nullptr, "null translator");
  std::lock_guard<std::mutex> lock(t->mutex);
  Engine& e = *t->engine;
  CT2X_CHECK(!e.is_seq2seq(), "This model cannot be used as a sequence generator");
  CT2X_CHECK(max_input_length >= 0, "max_input_length must be >= 0");
  const int64_t V = e.vocab_size();
  std::vector<std::vector<int32_t>> seqs(batch);
  std::vector<int32_t> order;
  int64_t T = 0;
  for (int32_t b = 0; b < batch; ++b) {
    out_lengths[b] = 0;
    if (lengths[b] < 2) continue;  // skip_scoring
    auto& s = seqs[b];
    for (int32_t i = 0; i < lengths[b]; ++i) {
      const int32_t id = ids[static_cast<int64_t>(b) * stride + i];
      CT2X_CHECK(id >= 0 && id < V, "Invalid token ID " + std::to_string(id));
      s.push_back(id);
    }
    truncate_ids(s, static_cast<size_t>(max_input_length), t->eos_id);
    if (s.size() < 2) continue;
    order.push_back(b);
    out_lengths[b] = static_cast<int32_t>(s.size()) - 1;
    T = std::max<int64_t>(T, static_cast<int64_t>(s.size()) - 1);
  }
  if (order.empty()) return 0;
  const int64_t B = static_cast<int64_t>(order.size()), rows = B * T;
  std::vector<int32_t> inputs(rows, 0), targets(rows, 0);
  for (int64_t i = 0; i < B; ++i) {
    const auto& s = seqs[order[i]];
    for (size_t k = 0; k + 1 < s.size(); ++k) {
      inputs[i * T + k] = s[k];
      targets[i * T + k] = s[k + 1];
    }
  }
  auto st = e.start_decoder_only(B, 1, T, T);
  float* d_logits = static_cast<float*>(device_malloc(sizeof(float) * rows * V));
  st->owned.push_back(d_logits);
  int32_t* d_targets = static_cast<int32_t*>(device_malloc(sizeof(int32_t) * rows));
  st->owned.push_back(d_targets);
  float* d_scores = static_cast<float*>(device_malloc(sizeof(float) * rows));
  st->owned.push_back(d_scores);
  CT2X_HIP_CHECK(hipMemcpyAsync(d_targets, targets.data(), sizeof(int32_t) * rows, hipMemcpyHostToDevice, st->stream));
  e.prefill(*st, inputs.data(), T, d_logits);
  launch_score_tokens(d_logits, rows, V, d_targets, d_scores, st->stream);
  std::vector<float> h(rows);
  CT2X_HIP_CHECK(hipMemcpyAsync(h.data(), d_scores, sizeof(float) * rows, hipMemcpyDeviceToHost, st->stream));
  CT2X_HIP_CHECK(hipStreamSynchronize(st->stream));
  for (int64_t i = 0; i < B; ++i) {
    const int32_t b = order[i];
    for (int32_t k = 0; k < out_lengths[b] && k < stride; ++k)
      out_scores[static_cast<int64_t>(b) * stride + k] = h[i * T + k];
  }
  API_CATCH
}

int ct2x_step_logits(void* translator, const int32_t* src_ids, const int32_t* src_lengths,
                     int32_t batch, int32_t src_stride, const int32_t* tgt_in, int32_t T,
                     float* logits_out) {
  API_TRY
  auto* t = static_cast<Translator*>(translator);
  release_kept_session(t);  // (the last plain translate call's buffers: not this entry point's to keep)
  CT2X_CHECK(t != nullptr, "null translator");
  std::lock_guard<std::mutex> lock(t->mutex);
  Engine& e = *t->engine;
  std::unique_ptr<DecodeState> st;
  if (e.is_seq2seq())
    st = e.start(src_ids, src_lengths, batch, src_stride, 1, T);
  else
    st = e.start_decoder_only(batch, 1, T);
  const int64_t V = e.vocab_size();
  std::vector<int32_t> ids(batch);
  e.set_step(*st, 0);
  for (int32_t s = 0; s < T; ++s) {
    for (int32_t b = 0; b < batch; ++b) ids[b] = tgt_in[static_cast<int64_t>(b) * T + s];
    e.set_ids(*st, ids.data());
    e.step(*st, /*graph=*/false);
    e.copy_logits_to_host(*st, logits_out + static_cast<int64_t>(s) * batch * V);
    e.advance_step(*st);
  }
  e.synchronize();
  API_CATCH
}

const char* ct2x_host_math_backend(void) {
  try {
    return host_math_backend();
  } catch (const std::exception& e) {
    set_last_error(e.what());
    return nullptr;
  }
}

int ct2x_host_sin_cos(int64_t n, const float* x, float* sin_out, float* cos_out) {
  API_TRY
  CT2X_CHECK(n >= 0 && (n == 0 || x != nullptr), "ct2x_host_sin_cos: bad arguments");
  if (sin_out) host_vsin(n, x, sin_out);
  if (cos_out) host_vcos(n, x, cos_out);
  return 0;
  API_CATCH
}

int ct2x_host_sinusoidal_table(int64_t positions, int64_t depth, float* out) {
  API_TRY
  CT2X_CHECK(positions > 0 && depth >= 2 && depth % 2 == 0 && out != nullptr,
             "ct2x_host_sinusoidal_table: positions > 0, even depth, output buffer");
  const std::vector<float> t = make_sinusoidal_table(positions, depth);
  std::memcpy(out, t.data(), t.size() * sizeof(float));
  API_CATCH
}

int ct2x_host_rotary_tables(int64_t dim, int interleave, int scaling_type, float scaling_factor,
                            float base, float low_freq_factor, float high_freq_factor,
                            int64_t original_max_position_embeddings, int64_t positions,
                            float* sin_out, float* cos_out) {
  API_TRY
  CT2X_CHECK(positions > 0 && dim >= 2 && dim % 2 == 0 && sin_out != nullptr && cos_out != nullptr,
             "ct2x_host_rotary_tables: positions > 0, even dim, output buffers");
  CT2X_CHECK(scaling_type == -1 || scaling_type == 0 || scaling_type == 2,
             "ct2x_host_rotary_tables: scaling type None (-1), Linear (0) or Llama3 (2)");
  std::vector<float> s, c;
  make_rotary_tables(dim, interleave != 0, scaling_type, scaling_factor, base, low_freq_factor,
                     high_freq_factor, original_max_position_embeddings, positions, s, c);
  std::memcpy(sin_out, s.data(), s.size() * sizeof(float));
  std::memcpy(cos_out, c.data(), c.size() * sizeof(float));
  API_CATCH
}

// ---- debugging taps (tests only) -------------------------------------------------------------
namespace {
std::vector<ct2x::Engine::Tap> g_taps;
std::mutex g_taps_mutex;
}  // namespace

int ct2x_debug_step_taps(void* translator, const int32_t* src_ids, const int32_t* src_lengths,
                         int32_t batch, int32_t src_stride, const int32_t* tgt_in, int32_t T,
                         float* logits_out) {
  API_TRY
  auto* t = static_cast<Translator*>(translator);
  release_kept_session(t);  // (the last plain translate call's buffers: not this entry point's to keep)
  CT2X_CHECK(t != nullptr, "null translator");
  std::lock_guard<std::mutex> lock(t->mutex);
  std::lock_guard<std::mutex> tlock(g_taps_mutex);
  Engine& e = *t->engine;
  g_taps.clear();
  struct Guard {
    Engine& e;
    ~Guard() {
      e.set_tap_sink(nullptr);
      e.set_tap_step(-1);
    }
  } guard{e};
  e.set_tap_sink(&g_taps);
  e.set_tap_step(-1);
  std::unique_ptr<DecodeState> st;
  if (e.is_seq2seq())
    st = e.start(src_ids, src_lengths, batch, src_stride, 1, T);
  else
    st = e.start_decoder_only(batch, 1, T);
  const int64_t V = e.vocab_size();
  std::vector<int32_t> ids(batch);
  e.set_step(*st, 0);
  for (int32_t s = 0; s < T; ++s) {
    for (int32_t b = 0; b < batch; ++b) ids[b] = tgt_in[static_cast<int64_t>(b) * T + s];
    e.set_ids(*st, ids.data());
    e.set_tap_step(s);
    e.step(*st, /*graph=*/false);
    if (logits_out) e.copy_logits_to_host(*st, logits_out + static_cast<int64_t>(s) * batch * V);
    e.advance_step(*st);
  }
  e.synchronize();
  API_CATCH
}

int ct2x_debug_tap_count() {
  std::lock_guard<std::mutex> tlock(g_taps_mutex);
  return static_cast<int>(g_taps.size());
}

int ct2x_debug_tap_info(int index, char* name, int name_size, int32_t* dtype, int64_t* rows,
                        int64_t* cols) {
  std::lock_guard<std::mutex> tlock(g_taps_mutex);
  if (index < 0 || index >= static_cast<int>(g_taps.size())) return -1;
  const auto& t = g_taps[index];
  if (name && name_size > 0) {
    std::strncpy(name, t.name.c_str(), name_size - 1);
    name[name_size - 1] = 0;
  }
  if (dtype) *dtype = t.dtype;
  if (rows) *rows = t.rows;
  if (cols) *cols = t.cols;
  return 0;
}

int ct2x_debug_tap_data(int index, void* out) {
  std::lock_guard<std::mutex> tlock(g_taps_mutex);
  if (index < 0 || index >= static_cast<int>(g_taps.size())) return -1;
  std::memcpy(out, g_taps[index].data.data(), g_taps[index].data.size());
  return 0;
}

int ct2x_debug_set_beam_ancestry(int enabled) {
  g_debug_beam_ancestry.store(enabled);
  return 0;
}

int ct2x_debug_set_position_table(void* translator, const float* table, int64_t positions,
                                  int64_t depth) {
  API_TRY
  auto* t = static_cast<Translator*>(translator);
  CT2X_CHECK(t != nullptr, "null translator");
  std::lock_guard<std::mutex> lock(t->mutex);
  t->engine->debug_set_position_table(table, positions, depth);
  API_CATCH
}

int ct2x_model_info(void* translator, int32_t* is_seq2seq, int32_t* vocab_size, int32_t* d_model,
                    int32_t* num_layers_dec, int32_t* num_heads, int32_t* bos_id, int32_t* eos_id,
                    char* compute_type, int ct_size) {
  API_TRY
  auto* t = static_cast<Translator*>(translator);
  CT2X_CHECK(t != nullptr, "null translator");
  if (is_seq2seq) *is_seq2seq = t->engine->is_seq2seq();
  if (vocab_size) *vocab_size = static_cast<int32_t>(t->engine->vocab_size());
  if (d_model) *d_model = static_cast<int32_t>(t->engine->d_model());
  if (num_layers_dec) *num_layers_dec = static_cast<int32_t>(t->engine->num_decoder_layers());
  if (num_heads) *num_heads = static_cast<int32_t>(t->engine->num_heads());
  if (bos_id) *bos_id = t->bos_id;
  if (eos_id) *eos_id = t->eos_id;
  if (compute_type && ct_size > 0) {
    const std::string s = compute_type_name(t->engine->compute_type());
    std::strncpy(compute_type, s.c_str(), ct_size - 1);
    compute_type[ct_size - 1] = 0;
  }
  API_CATCH
}

int ct2x_tp_info(void* translator, int32_t* rank, int32_t* world, int32_t* has_mailboxes,
                 char* backend, int backend_size) {
  API_TRY
  auto* t = static_cast<Translator*>(translator);
  CT2X_CHECK(t != nullptr, "null translator");
  if (rank) *rank = t->engine->tp_rank();
  if (world) *world = t->engine->tp_world();
  if (has_mailboxes) *has_mailboxes = t->engine->tp_has_mailboxes() ? 1 : 0;
  if (backend && backend_size > 0) {
    std::strncpy(backend, t->engine->tp_backend(), backend_size - 1);
    backend[backend_size - 1] = 0;
  }
  API_CATCH
}

int ct2x_vocab_size(void* translator, int which) {
  auto* t = static_cast<Translator*>(translator);
  if (!t) return -1;
  return static_cast<int>((which == 0 ? t->src_vocab : t->tgt_vocab)->size());
}

const char* ct2x_vocab_token(void* translator, int which, int32_t id) {
  auto* t = static_cast<Translator*>(translator);
  try {
    return (which == 0 ? t->src_vocab : t->tgt_vocab)->to_token(static_cast<size_t>(id)).c_str();
  } catch (const std::exception& e) {
    set_last_error(e.what());
    return nullptr;
  }
}

int32_t ct2x_vocab_id(void* translator, int which, const char* token) {
  auto* t = static_cast<Translator*>(translator);
  return static_cast<int32_t>((which == 0 ? t->src_vocab : t->tgt_vocab)->to_id(token));
}

int ct2x_last_stats(void* translator, struct ct2x_exec_stats* stats) {
  API_TRY
  auto* t = static_cast<Translator*>(translator);
  CT2X_CHECK(t != nullptr && stats != nullptr, "null argument");
  *stats = t->last;
  API_CATCH
}

int64_t ct2x_last_compactions(void* translator) {
  auto* t = static_cast<Translator*>(translator);
  return t != nullptr ? t->last_compactions : -1;
}

int64_t ct2x_cluster_fallbacks(void* translator) {
  auto* t = static_cast<Translator*>(translator);
  return t != nullptr ? t->cluster_fallbacks : -1;
}

int ct2x_debug_inject_cluster_timeout(void* translator) {
  API_TRY
  auto* t = static_cast<Translator*>(translator);
  CT2X_CHECK(t != nullptr && t->engine, "ct2x_debug_inject_cluster_timeout: null translator");
  t->engine->debug_inject_cluster_timeout();
  API_CATCH
}

int64_t ct2x_session_hits(void* translator) {
  auto* t = static_cast<Translator*>(translator);
  return t != nullptr ? t->session_hits : -1;
}

int ct2x_release_session(void* translator) {
  API_TRY
  auto* t = static_cast<Translator*>(translator);
  CT2X_CHECK(t != nullptr, "null translator");
  std::lock_guard<std::mutex> lock(t->mutex);
  t->session.reset();
  t->session_key.clear();
  API_CATCH
}

int ct2x_profile_begin(void* translator) {
  API_TRY
  auto* t = static_cast<Translator*>(translator);
  CT2X_CHECK(t != nullptr, "null translator");
  std::lock_guard<std::mutex> lock(t->mutex);
  t->engine->profile_begin();
  API_CATCH
}

int ct2x_profile_end(void* translator, char* json_out, int json_size) {
  API_TRY
  auto* t = static_cast<Translator*>(translator);
  CT2X_CHECK(t != nullptr, "null translator");
  std::lock_guard<std::mutex> lock(t->mutex);
  const auto stats = t->engine->profile_end();
  std::string js = "{";
  bool first = true;
  char buf[256];
  for (const auto& kv : stats) {
    if (!first) js += ",";
    first = false;
    std::snprintf(buf, sizeof(buf), "{\"launches\":%lld,\"total_ms\":%.6f,\"bytes\":%.1f,\"ops\":%.1f}",
                  static_cast<long long>(kv.second.launches), kv.second.total_ms, kv.second.bytes,
                  kv.second.ops);
    js += "\"" + kv.first + "\":" + buf;
  }
  js += "}";
  CT2X_CHECK(json_out != nullptr && static_cast<int>(js.size()) < json_size,
             "ct2x_profile_end: output buffer too small");
  std::memcpy(json_out, js.c_str(), js.size() + 1);
  API_CATCH
}

// ---- host-only ---------------------------------------------------------------------------

int ct2x_inspect_model(const char* model_dir, char* spec_name, int spec_size,
                       int32_t* binary_version, int32_t* revision, int32_t* num_aliases) {
  try {
    auto m = ModelFile::load(model_dir);
    if (spec_name && spec_size > 0) {
      std::strncpy(spec_name, m->spec.c_str(), spec_size - 1);
      spec_name[spec_size - 1] = 0;
    }
    if (binary_version) *binary_version = static_cast<int32_t>(m->binary_version);
    if (revision) *revision = static_cast<int32_t>(m->revision);
    if (num_aliases) *num_aliases = static_cast<int32_t>(m->aliases.size());
    return static_cast<int>(m->variables.size());
  } catch (const std::exception& e) {
    set_last_error(e.what());
    return -1;
  }
}

int ct2x_inspect_variable(const char* model_dir, int32_t index, char* name, int name_size,
                          int32_t* dtype, int32_t* rank, int64_t* dims, int64_t* nbytes) {
  API_TRY
  static thread_local std::string cached_dir;
  static thread_local std::shared_ptr<ModelFile> cached;
  if (!cached || cached_dir != model_dir) {
    cached = ModelFile::load(model_dir);
    cached_dir = model_dir;
  }
  CT2X_CHECK(index >= 0 && index < static_cast<int32_t>(cached->variables.size()),
             "variable index out of range");
  const Variable& v = cached->variables[index];
  if (name && name_size > 0) {
    std::strncpy(name, v.name.c_str(), name_size - 1);
    name[name_size - 1] = 0;
  }
  if (dtype) *dtype = v.dtype;
  if (rank) *rank = static_cast<int32_t>(v.shape.size());
  if (dims)
    for (size_t i = 0; i < v.shape.size() && i < 8; ++i) dims[i] = v.shape[i];
  if (nbytes) *nbytes = static_cast<int64_t>(v.data.size());
  API_CATCH
}

int ct2x_host_beam_step_select(int32_t beam_size, int32_t step, int32_t is_last_step,
                               int32_t end_id, int32_t prefix_length, const int32_t* word_ids,
                               int32_t* finished, int32_t* next) {
  try {
    if (beam_size < 1 || !word_ids || !finished || !next)
      throw std::invalid_argument("ct2x_host_beam_step_select: bad arguments");
    BeamStepInput in;
    in.beam_size = beam_size;
    in.step = step;
    in.is_last_step = is_last_step != 0;
    in.end_id = end_id;
    in.word_ids = word_ids;
    in.prefix_length = prefix_length;
    const BeamStepOutput out = beam_step_select(in);
    for (int32_t k = 0; k < beam_size; ++k) {
      finished[k] = k < static_cast<int32_t>(out.finished.size()) ? out.finished[k] : -1;
      next[k] = out.next[k];
    }
    return static_cast<int>(out.finished.size());
  } catch (const std::exception& e) {
    set_last_error(e.what());
    return -1;
  }
}

float ct2x_host_finalize_score(float score, float length, float length_penalty) {
  return finalize_score(score, length, length_penalty);
}

// negative_dtw (src/dtw.cc:40-82): host work on a small [text tokens, audio frames] matrix. One
// flat cost row pair and a byte trace instead of the reference's vector-of-vectors; same
// recurrence, same column-major fill order, same tie rule, same backtrace (the first row / column
// of the trace are forced to "left" / "up", src/dtw.cc:11-14).
int64_t ct2x_host_negative_dtw(const float* x, int64_t n, int64_t m, int64_t* path_i, int64_t* path_j) {
  try {
    if (!x || !path_i || !path_j || n < 1 || m < 1)
      throw std::invalid_argument("ct2x_host_negative_dtw: bad arguments");
    const float inf = std::numeric_limits<float>::infinity();
    const int64_t w = m + 1;
    std::vector<float> cost(static_cast<size_t>((n + 1) * w), inf);
    std::vector<int8_t> trace(static_cast<size_t>((n + 1) * w), -1);
    cost[0] = 0.f;
    for (int64_t j = 1; j <= m; ++j) {
      for (int64_t i = 1; i <= n; ++i) {
        const float c0 = cost[(i - 1) * w + (j - 1)], c1 = cost[(i - 1) * w + j], c2 = cost[i * w + (j - 1)];
        float c;
        int8_t t;
        if (c0 < c1 && c0 < c2) { c = c0; t = 0; }
        else if (c1 < c0 && c1 < c2) { c = c1; t = 1; }
        else { c = c2; t = 2; }
        cost[i * w + j] = -x[(i - 1) * m + (j - 1)] + c;
        trace[i * w + j] = t;
      }
    }
    for (int64_t k = 0; k <= m; ++k) trace[k] = 2;
    for (int64_t k = 0; k <= n; ++k) trace[k * w] = 1;
    int64_t i = n, j = m, len = 0;
    while (i > 0 || j > 0) {
      path_i[len] = i - 1;
      path_j[len] = j - 1;
      ++len;
      const int8_t t = trace[i * w + j];
      if (t == 0) { --i; --j; }
      else if (t == 1) --i;
      else if (t == 2) --j;
      else throw std::runtime_error("ct2x_host_negative_dtw: unexpected trace entry");
    }
    std::reverse(path_i, path_i + len);
    std::reverse(path_j, path_j + len);
    return len;
  } catch (const std::exception& e) {
    set_last_error(e.what());
    return -1;
  }
}

}  // extern "C"
