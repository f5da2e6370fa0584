// MI355X-native execution engine for Transformer encoder/decoder models: the device side of
// ctranslate2::Translator / Generator. Where the reference composes ~50 generic ops over
// freshly allocated StorageViews per step (src/layers/transformer.cc:621-871,
// src/layers/attention.cc:442-615, src/layers/common.cc:339-442), this engine
//   * keeps every weight and activation buffer resident in HBM for the whole call,
//   * runs each decoder sub-layer as a short chain of fused gfx950 kernels
//     (norm+quantize -> int8 MFMA GEMM+dequant+bias(+act)(+residual) -> fused attention),
//   * appends to a PRE-ALLOCATED KV cache in place and reorders beams by one gather launch,
//   * reads the decoding step from device memory so a whole step is one replayable hipGraph,
//   * samples greedy tokens on the device so consecutive steps need no host round trip.
// Results are the reference's: same formulas, same op order (see the citations in engine.cc).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "kernels.h"
#include "model.h"
#include "runtime.h"
#include "tp_comm.h"

namespace ct2x {

// kFloat16 / kBFloat16: Dense weights in the 16-bit type through the fp16 / bf16 MFMA GEMM
// (gemm_h.hip), fp32 activations everywhere else.
enum class ComputeType { kFloat32, kInt8Float32, kFloat16, kBFloat16 };

std::string compute_type_name(ComputeType t);

// generate_sinusoidal_position_encoding (src/layers/common.cc:204-229) on the host:
// [positions, depth], sin half then cos half, through the host math backend (host_math.h).
std::vector<float> make_sinusoidal_table(int64_t positions, int64_t depth);
// Host side of the rotary tables (RotaryEmbeddings::initialize): s, c [positions, dim].
// su_long / su_short (Su scaling): per-frequency divisors [dim / 2]; max_position_embeddings:
// with original_max_position_embeddings, the sqrt(1 + log(scale) / log(original)) factor applied
// to both tables for every scaling type but Llama3 (attention_layer.cc:331-343).
void make_rotary_tables(int64_t dim, bool interleave, int scaling_type, float scaling_factor,
                        float base, float low_freq_factor, float high_freq_factor,
                        int64_t original_max_position_embeddings, int64_t positions,
                        std::vector<float>& s, std::vector<float>& c,
                        const std::vector<float>* su_long = nullptr,
                        const std::vector<float>* su_short = nullptr,
                        int64_t max_position_embeddings = 0);

struct DenseW {
  // Tensor parallel: the K dimension is split over the ranks (attention-out / FFN-out); the
  // partial products are summed across ranks before bias, residual (see Engine::dense).
  bool row_parallel = false;
  bool quantized = false;
  const int8_t* wq = nullptr;     // [n, k] int8
  // fragment-major copy for the streaming decode GEMM (gemm_s8_stream.hip; models whose weights
  // are not cache resident), packed for the single or for the gated-pair plan
  const int8_t* wq_stream = nullptr;
  bool wq_stream_dual = false;
  const float* wscale = nullptr;  // [n]
  const float* wf = nullptr;      // [n, k] fp32 (float models)
  const void* wh = nullptr;       // [n, k] float16 / bfloat16 (compute types float16 / bfloat16)
  bool wh_bf16 = false;
  const float* bias = nullptr;    // [n] or nullptr
  int64_t n = 0, k = 0;
  bool present = false;
  // AWQ int4 weight (awq.hip): 1 = GEMM layout [k, n / 8], 2 = GEMV layout [n, k / 8]
  int awq = 0;
  const int32_t* aq_w = nullptr;
  const void* aq_s = nullptr;      // float16 scales (bfloat16 when aq_bf16)
  bool aq_bf16 = false;            // compute type bfloat16: the bf16 arm of the AWQ kernels (round 6)
  const int32_t* aq_z = nullptr;
  int64_t aq_group = 0, aq_zw = 0, aq_sw = 0;
};

struct NormW {
  bool present = false;
  const float* gamma = nullptr;
  const float* beta = nullptr;  // nullptr => RMSNorm
  float eps = 1e-5f;
  bool use_residual = false;
};

struct AttnW {
  NormW ln;
  DenseW lin[3];  // self: [qkv, out]; cross: [q, kv, out]
  int n_lin = 0;
  int64_t heads = 0, heads_kv = 0, dhead = 0;
  float queries_scale = 1.f;
  bool rotary = false;
  bool rotary_interleave = true;
  int64_t rotary_dim = 0;  // 0 => dhead
  float rotary_base = 10000.f;
  int rotary_scaling_type = -1;  // -1 none, 0 linear, 1 su, 2 llama3
  float rotary_scaling_factor = 1.f;
  float rotary_low_freq_factor = 1.f, rotary_high_freq_factor = 4.f;
  int64_t original_max_position_embeddings = 0, max_position_embeddings = 0;
  // Su scaling (rotary_scaling_type 1, Phi-3): per-frequency divisors [rotary_dim / 2]
  std::vector<float> rotary_long_factor, rotary_short_factor;
  // layer variants: `ln` is the layer's input norm, the block's output carries no residual
  bool no_residual = false;
};

struct FfnW {
  NormW ln;
  DenseW l0, l0_noact, l1;
  int activation = 0;
  bool no_residual = false;  // layer variants (see LayerW)
};

struct EmbW {
  const void* table = nullptr;
  bool is_int8 = false;
  const float* scale = nullptr;  // per-row scales of an int8 table
  int64_t vocab = 0, depth = 0;
};

struct LayerW {
  AttnW self;
  bool has_cross = false;
  AttnW cross;
  FfnW ffn;
  // TransformerDecoderLayer::operator() arms (src/layers/transformer.cc:196-366):
  //   0  standard: every sub-layer owns its norm and adds its input
  //   1  parallel residual (shared_layer_norm, or input_layer_norm [+ post_attention_layer_norm]):
  //      out = FFN(norm'(x)) + x + Attention(norm(x))
  //   2  pre + post norms (input / post_attention / pre_feedforward / post_feedforward):
  //      h = post_attention_norm(Attention(input_norm(x))) + x;
  //      out = (post_feedforward_norm(FFN(pre_feedforward_norm(h))) + h) * layer_scalar
  // self.ln / ffn.ln hold the norm each sub-layer applies to its input; the sub-layers run with
  // no_residual and launch_step / prefill add the rest.
  int variant = 0;
  NormW post_attn_ln, post_ff_ln;
  float layer_scalar = 1.f;
};

struct StackW {  // encoder or decoder
  EmbW emb;
  bool scale_embeddings = false;
  float embeddings_scale = 1.f;
  const float* position_table = nullptr;  // learned positions [num_positions, d] or nullptr
  int64_t num_positions = 0;
  bool sinusoidal_positions = false;
  NormW layernorm_embedding;
  NormW output_norm;
  std::vector<LayerW> layers;
  bool pre_norm = true;
  int64_t num_heads = 8;
  DenseW projection;  // decoder only
  int64_t d_model = 0;
  // Whisper encoder front end (layers::WhisperEncoder, src/layers/whisper.cc:9-24): two kernel-3
  // convolutions stored as Dense weights [out, in * 3] over im2col rows
  bool has_conv = false;
  DenseW conv1, conv2;
  int64_t n_mels = 0;
};

// Per-call device state of one decoding session over `rows` = batch * beam hypotheses.
struct DecodeState;

// RepetitionPenalty, NoRepeatNgram, SuppressTokens / SuppressSequences (src/decoding_utils.cc).
struct LogitsProcessing {
  float repetition_penalty = 1.f;                          // 1 = off
  int32_t no_repeat_ngram_size = 0;                        // 0 = off
  std::vector<std::vector<int32_t>> suppress_sequences;    // length 1 = token always disabled
  bool active() const {
    return repetition_penalty != 1.f || no_repeat_ngram_size > 0 || !suppress_sequences.empty();
  }
};

// Whisper's logits processors (src/models/whisper.cc:202-231, 741-861; decoding_utils.cc:172-194)
struct WhisperProcessing {
  std::vector<int32_t> begin_ids;  // SuppressTokensBegin (suppress_blank)
  bool timestamp_rules = false;    // ApplyTimestampRules
  int32_t eot_id = 0, no_timestamps_id = 0, timestamp_begin = 0, timestamp_end = 0,
          max_initial_timestamp = 0;
  int32_t no_speech_id = -1;       // >= 0: GetNoSpeechProbs at the first decoding step
};

class Engine {
public:
  // `tp`: tensor-parallel communicator (nullptr or world 1 = the whole model on this device).
  Engine(std::shared_ptr<ModelFile> model, const std::string& compute_type, int device_index,
         std::shared_ptr<TpComm> tp = nullptr);
  ~Engine();

  const ModelFile& model() const { return *model_; }
  ComputeType compute_type() const { return compute_type_; }
  // float16 / bfloat16: no CPU counterpart exists, the exact-order kernels are not required
  bool half_compute() const {
    return compute_type_ == ComputeType::kFloat16 || compute_type_ == ComputeType::kBFloat16;
  }
  // AttnFullParams::half_kind: which 16-bit matrix pipe sequence-mode attention may use
  int half_kind() const {
    return compute_type_ == ComputeType::kFloat16 ? 1 : compute_type_ == ComputeType::kBFloat16 ? 2 : 0;
  }
  bool is_seq2seq() const { return has_encoder_; }
  bool is_whisper() const { return enc_.has_conv; }
  int64_t n_mels() const { return enc_.n_mels; }
  int64_t max_source_positions() const { return enc_.num_positions; }
  int64_t max_target_positions() const { return dec_.num_positions; }
  int64_t vocab_size() const { return dec_.projection.n; }
  int64_t d_model() const { return dec_.d_model; }
  int64_t num_decoder_layers() const { return static_cast<int64_t>(dec_.layers.size()); }
  int64_t num_heads() const { return dec_.num_heads; }
  int device_index() const { return device_; }
  int tp_world() const { return tp_ ? tp_->world() : 1; }
  int tp_rank() const { return tp_ ? tp_->rank() : 0; }
  const char* tp_backend() const { return tp_ ? tp_->backend() : "none"; }
  bool tp_has_mailboxes() const { return tp_ && tp_->peer_box() != nullptr; }
  hipStream_t stream() const { return stream_; }
  hipStream_t cur_stream() const;  // stream of the session driven by the calling thread
  int64_t weight_bytes() const { return weight_bytes_; }

  // ---- session API used by decoding.cc -------------------------------------------------
  // Encodes `batch` padded source sequences (ids [batch, s_max] on host, lengths [batch]) and
  // prepares a decoding state for batch*beam rows and up to t_max decoded positions.
  std::unique_ptr<DecodeState> start(const int32_t* src_ids, const int32_t* src_lengths,
                                     int64_t batch, int64_t s_max, int64_t beam, int64_t t_max,
                                     int64_t prefill_len = 0, bool beam_search_state = false,
                                     const float* features = nullptr, int64_t frames = 0);
  // Session reuse (c_api_engine.cc keeps the last plain greedy session of a Translator): runs the
  // encoder again, for new sentences of the SAME padded shape (batch, s_max), into the buffers of a
  // finished session - allocations, tables and the step graphs captured over those buffers stay.
  void reencode(DecodeState& state, const int32_t* src_ids, const int32_t* src_lengths);
  // `count` consecutive greedy steps. A session that is being reused replays them from ONE graph
  // (the step counter lives in device memory, so the captured sequence serves any `count` steps
  // in a row); every other session - and every other count - takes them one by one (step()).
  void step_group(DecodeState& state, int count, bool graph);
  // Whisper: `features` [batch, n_mels, frames] (host) go through the convolutional front end and
  // the encoder (WhisperEncoder::operator(), src/layers/whisper.cc:26-62); the memory has
  // whisper_memory_length(frames) positions.
  std::unique_ptr<DecodeState> start_features(const float* features, int64_t batch, int64_t frames,
                                              int64_t beam, int64_t t_max, int64_t prefill_len = 0,
                                              bool beam_search_state = false);
  static int64_t whisper_memory_length(int64_t frames) { return frames <= 0 ? 0 : (frames - 1) / 2 + 1; }
  // Copies the encoder output of a state made by start() / start_features() to the host:
  // [batch, s_max, d_model]. Valid until the first prefill / step of that state.
  void read_encoder_output(DecodeState& state, float* out);
  // Decoder-only models: no encoder; the prompt is consumed by prefill().
  std::unique_ptr<DecodeState> start_decoder_only(int64_t batch, int64_t beam, int64_t t_max,
                                                  int64_t prefill_len = 0);
  // Consumes `time` prompt tokens per row (h_ids [rows, time]) in one sequence-mode pass.
  // logits_all (optional): device [rows * time, vocabulary] - the logits of every position
  // capture (optional): raw encoder-decoder attention scores of selected heads (Whisper::align)
  struct ScoreCapture {
    std::vector<std::vector<int32_t>> heads;  // per decoder layer: the heads to record
    std::vector<const int32_t*> d_heads;      // the same lists on the device
    int64_t n_sel = 0;                        // total number of (layer, head) pairs
    float* scores = nullptr;                  // device [batch, n_sel, time, s_max], layer order
  };
  void prefill(DecodeState& state, const int32_t* h_ids, int64_t time, float* logits_all = nullptr,
               const ScoreCapture* capture = nullptr);

  // One decoder step over state.rows rows: reads token ids from state.d_ids and the step index
  // from state.d_step (device memory), leaves logits in state.logits [rows, vocab].
  // With `graph`, the launch sequence is captured once and replayed.
  void step(DecodeState& state, bool graph);

  // Greedy / random sampling (BestSampler / RandomSampler, src/sampling.cc, + DisableTokens
  // min-length masking, src/decoding.cc:384-408, + optional LogSoftMax for scores, :874-880) is
  // part of step() when state.fuse_greedy is set: next ids into state.d_ids, (id, score) appended
  // to the device history, step counter advanced by the sampler itself.
  // Teacher-forced callers advance the step counter themselves:
  void advance_step(DecodeState& state);

  // Beam search (src/decoding.cc:425-720) is device resident: with state.fuse_beam set (by
  // decoding.cc) every step() ends with log_softmax + beam score, top-(2*beam) over the flattened
  // [batch, beam*vocab] candidates, the reference's candidate selection, the cache reorder
  // (Decoder::update_state, src/layers/decoder.cc:33-56) and the step advance - no host round
  // trip. The candidates of every step are recorded; the host reads them in groups and replays
  // the hypothesis bookkeeping.
  // Copies `count` steps of recorded candidates starting at step `first` (relative to the first
  // decoding step) into pinned slot `slot`: word ids, cumulative scores, origin beams, each
  // [count][batch, 2*beam].
  void read_beam_history_async(DecodeState& state, int64_t first, int64_t count, int slot);
  void wait_beam_history(DecodeState& state, int slot, const int32_t** h_words,
                         const float** h_scores, const int32_t** h_origins);
  // Hard target prefix of beam search, one entry per BATCH element (empty = none).
  // extends_min_length: the prefix is hidden (generation): EOS stays disabled prefix_length steps
  // longer for that entry.
  void set_beam_prefix(DecodeState& state, const std::vector<std::vector<int32_t>>& prefix,
                       int32_t first_step, bool extends_min_length = false,
                       float bias_beta = 0.f /* > 0: the prefix biases instead of constraining */);
  // Initial cumulative scores (initialize_beam_scores, src/decoding.cc:84-93).
  void set_beam_scores(DecodeState& state, const float* h_scores);

  // Copies `count` history steps starting at `first` (ids and scores, [step][rows]) to host.
  void read_history(DecodeState& state, int64_t first, int64_t count, int32_t* h_ids,
                    float* h_scores);
  // Asynchronous form: enqueues the D2H copy into pinned slot `slot` (0/1) behind the steps
  // launched so far and records an event; wait_history() blocks on that event only, so the
  // next group of steps can already be queued on the device while the host examines this one.
  void read_history_async(DecodeState& state, int64_t first, int64_t count, int slot);
  void wait_history(DecodeState& state, int slot, const int32_t** h_ids, const float** h_scores);
  void copy_logits_to_host(DecodeState& state, float* h_logits);
  void set_ids(DecodeState& state, const int32_t* h_ids);
  // Finished-sentence removal of the greedy loop (the reference gathers the alive rows out of every
  // state tensor whenever a sentence finishes: src/decoding.cc:916-956, Decoder::update_state
  // src/layers/decoder.cc:33-56): keeps rows `keep` (ascending) of a beam-1 session - self-attention
  // caches, memory K/V and lengths, next ids, the [step][row] histories, the hard prefix - as rows
  // 0 .. keep.size() - 1; the step graph is captured again for the new row count on its next use.
  // The device must be idle on this session (no step in flight). compactable(): whether the session
  // holds nothing that this does not move.
  bool compactable(const DecodeState& state) const;
  void compact_rows(DecodeState& state, const std::vector<int32_t>& keep, int64_t steps_done);
  // The same for a beam-search session: keeps the batch ENTRIES `keep` (ascending) with all their
  // beams (the reference drops a batch entry once it is finished, src/decoding.cc:599-694). Only for
  // sessions that follow the ancestry table: the self-attention caches do not move at all - the
  // table's columns of the kept rows are packed, and the rows write their next positions into slots
  // no one has written yet.
  bool beam_compactable(const DecodeState& state) const;
  void compact_beam_entries(DecodeState& state, const std::vector<int32_t>& keep, int64_t positions_written);
  // Installs per-row forced tokens for the steps first_step, first_step + 1, ... (empty = none).
  void set_prefix(DecodeState& state, const std::vector<std::vector<int32_t>>& prefix,
                  int32_t first_step, bool extends_min_length = true);
  void set_step(DecodeState& state, int32_t step);
  // return_attention / coverage_penalty: from now on every decode step records, per row, the
  // encoder-decoder attention of the alignment layer / heads (decoder attributes alignment_layer,
  // alignment_heads; src/layers/transformer.cc:516-528) at att_hist[step]. read_attention copies
  // steps [first, first + count) to the host: out [count][rows][s_max].
  void enable_attention(DecodeState& state);
  void read_attention(DecodeState& state, int64_t first, int64_t count, float* out);
  // return_logits_vocab (greedy search / random sampling): every step keeps the logits it samples
  // from (after the logits processors and the minimum-length mask) in a ring of kLogitsRing steps;
  // read_logits copies steps [first, first + count) to the host: out [count][rows][vocab_size()].
  // The host must not fall more than kLogitsRing steps behind the device (the greedy loop keeps at
  // most two groups of sync_every steps in flight).
  static constexpr int32_t kLogitsRing = 32;
  void enable_logits_history(DecodeState& state);
  void read_logits(DecodeState& state, int64_t first, int64_t count, float* out);
  void synchronize();
  // Throws ClusterTimeout if a cluster of the decoder-stack launch waited in vain for a member
  // during this session (something else held the compute units): its results are void. The
  // translate entry points catch it, switch the handle to the launch-per-op step
  // (disable_cluster) and run the call again.
  void check_cluster(DecodeState& state);
  void disable_cluster() { cluster_ready_ = false; }
  bool cluster_enabled() const { return cluster_ready_; }
  // Tests: the next check_cluster of a session that ran the launch reports a starved cluster.
  void debug_inject_cluster_timeout() { inject_cluster_timeout_ = true; }
  // The decoder-stack launch keeps its workgroups resident and waiting for each other: two such
  // launches of different sessions on one device could each hold part of the compute units and
  // wait forever (the bounded polls turn that into an error, not a hang). A session therefore
  // uses the launch only while it holds the device's gate (ClusterLease below): one holder per
  // device across the threads of this process (an atomic flag) AND across processes (a
  // non-blocking flock on a file named after the device's PCI bus id). The gate is only ever
  // TRIED: a session that does not get it decodes with the launch-per-op step - same tokens -
  // instead of waiting, so several Translator handles or replica processes on one GPU (a
  // supported deployment: include/ctranslate2/replica_pool.h:339-359) overlap as before, and a
  // callback that re-enters translate cannot deadlock.
  static bool cluster_gate_try_acquire(int device_index);
  static void cluster_gate_release(int device_index);
  // Destroys the session's captured step graphs (the next step captures again).
  void drop_graphs(DecodeState& state);
  static int64_t cluster_max_rows();
  // Teacher-forced scoring (score_sequences, src/scoring.cc): runs `steps` decoder steps feeding
  // h_inputs[s][row] and leaves log P(h_outputs[s][row]) of every step in the score history
  // (read_history). Both arrays are [steps][rows]; nothing is read back between the steps.
  void score_targets(DecodeState& state, const int32_t* h_inputs, const int32_t* h_outputs,
                     int64_t steps);

  // Logits processors of the decoding loop (make_logits_processors, src/decoding.cc:1091-1120):
  // installs them for every following step of `state`. Greedy steps apply them on the device
  // from the device-resident id history, beam steps from the device-resident tokens of the
  // active beams (both inside the step's graph).
  void set_logits_processing(DecodeState& state, const LogitsProcessing& processing);

  // Whisper's processors for every following step of `state` (after set_logits_processing).
  void set_whisper_processing(DecodeState& state, const WhisperProcessing& processing);
  // GetNoSpeechProbs result of the first decoding step: out [rows]
  void read_no_speech_probs(DecodeState& state, float* out);
  // Whisper::align's device part (src/models/whisper.cc:425-553): the decoder in sequence mode over
  // h_ids [batch, time] (state made by start_features with beam 1, prefill_len >= time). Leaves
  //  * token_probs [batch, time]: softmax over the first `text_classes` logits at every position,
  //    probability of h_targets [batch, time];
  //  * weights [batch, time, s_max]: the raw encoder-decoder attention scores of the (layer, head)
  //    pairs, softmax over the first frames[b] memory positions, normalised over the time axis
  //    (per batch entry: its own length lengths[b] and frames[b]), median-filtered over the memory
  //    axis and averaged over the pairs. Entries outside [lengths[b], frames[b]] are unspecified.
  void whisper_align_forward(DecodeState& state, const int32_t* h_ids, const int32_t* h_targets,
                             int64_t time, const std::vector<int32_t>& lengths,
                             const std::vector<int32_t>& frames,
                             const std::vector<std::pair<int32_t, int32_t>>& alignment_heads,
                             int64_t text_classes, int64_t median_filter_width, float* h_token_probs,
                             float* h_weights);

  // Number of kernel launches recorded for one decoder step (reported by bench.py).
  int64_t kernels_per_step() const { return kernels_per_step_; }

  // Per-kernel timing with HIP events on the engine's stream (eager launches only: profiling
  // disables graph replay). Stats are keyed by a label that includes the GEMM shape.
  struct KernelStat {
    int64_t launches = 0;
    double total_ms = 0;
    double bytes = 0;  // algorithmic bytes, summed over launches
    double ops = 0;    // algorithmic integer/float ops, summed over launches
  };
  void profile_begin();
  std::map<std::string, KernelStat> profile_end();
  bool profiling() const { return profiling_; }

  // ---- debugging taps (tests only; eager steps) -------------------------------------------
  // With a sink installed, every producer on the path (embedding, norm + Quantize, Quantize,
  // Dense, attention) copies its output to the host right after its launch, named
  // "<scope>.<producer><index>": the parity tests line these up against the same points of the
  // reference's layers (oracle/ref_capi.cc, ct2ref_tap_forward) and report the FIRST tensor that
  // differs. dtype: 0 = fp32, 1 = int8.
  struct Tap {
    std::string name;
    int dtype = 0;
    int64_t rows = 0, cols = 0;
    std::vector<char> data;
  };
  void set_tap_sink(std::vector<Tap>* sink) { taps_ = sink; }
  void set_tap_step(int step) { tap_step_ = step; }
  // Replaces the sinusoidal position table (debug: lets a test inject the reference's table,
  // which comes from MKL's closed vsSin / vsCos, to show that nothing else differs).
  void debug_set_position_table(const float* host, int64_t positions, int64_t depth);

private:
  friend struct DecodeState;
  struct QuantIn {  // an activation matrix already quantised for an int8 Dense
    const int8_t* q = nullptr;
    const float* s = nullptr;
  };

  // Pre-norm Dense at decode rows with the norm + Quantize as the GEMM's prologue (one launch);
  // false when the shape / layer does not qualify (the caller then runs norm_quant + dense).
  // row_amax_out (optional): the launch also max-combines the row absmax of y into it.
  bool dense_norm(const NormW& ln, const DenseW& w, const float* x, int64_t m, int act,
                  const float* residual, float* y, float* row_amax_out = nullptr);
  // int8 Dense that quantises its fp32 input while loading it (row absmax given by the producer)
  // Fills the leader fields of `ep` from the session being enqueued (false: no session / more than
  // 16 rows / taps: the caller keeps the separate launch).
  bool lead_fill(GemmS8Epilogue& ep, int64_t m);
  void reset_lead_flags(DecodeState& st);
  void dense_qol(const DenseW& w, const float* x, const float* row_amax, int64_t m, int act,
                 const float* residual, float* y, int amax_parts = 1);
  bool stream_qol_applicable(const DenseW& w, int64_t m) const;
  bool stream_lead_quant_applicable(const DenseW& w, int64_t m) const;
  void load_weights();
  bool has_awq_ = false;
  void dense_awq(const DenseW& w, const float* x, int64_t m, int act, const float* residual, float* y,
                 DecodeState* ws);
  // weight prefetch by spare workgroups of the row-wide launches of a decode step (engine.cc)
  void prefetch_weight(DecodeState& st, const DenseW& w, int site);
  bool weight_prefetch_ = false;
  int prefetch_wgs_ = 192;
  static thread_local DecodeState* pf_state_;  // session whose launch_step is running on this thread
  // session whose vocabulary projection carries the fused greedy arg-max (set around that Dense)
  static thread_local const DecodeState* argmax_state_;
  bool fused_argmax_applicable(const DecodeState& state) const;
  int64_t prefetch_bytes_[4] = {0, 0, 0, 0};
  void pack_stream_weights();
  void pack_stream(DenseW& w, bool dual);
  // Cluster-resident decoder stack (decoder_cluster.hip): every decoder layer of a greedy step in
  // ONE launch. pack_cluster_weights() re-packs the six Dense weights of every layer member-major
  // when the model qualifies (int8, pre-norm, encoder-decoder, Transformer-base geometry); a
  // session's launch takes the layer table (weights + that session's caches) as kernel arguments.
  void pack_cluster_weights();
  bool cluster_ready_ = false;
  bool inject_cluster_timeout_ = false;
  std::vector<ClusterLayerArgs> cluster_layers_;  // weights part; caches filled per session
  bool cluster_applicable(const DecodeState& state) const;
  // How a Dense is split over the tensor-parallel ranks (Model::load, src/models/model.cc:
  // 662-743): kColumn = output rows; `blocks` lists the sizes of the fused sub-matrices that are
  // each split separately and re-concatenated (Q | K | V of linear_0, K | V of the cross
  // attention linear_1); kRow = input columns.
  enum class Shard { kNone, kColumn, kRow };
  DenseW load_dense(const std::string& scope, bool required, Shard shard = Shard::kNone,
                    const std::vector<int64_t>& blocks = {});
  NormW load_norm(const std::string& scope);
  AttnW load_attention(const std::string& scope, bool self_attention, int64_t num_heads);
  FfnW load_ffn(const std::string& scope, int activation);
  void load_stack(const std::string& scope, bool decoder, StackW& out);
  const void* upload(const void* host, size_t bytes);
  const float* upload_floats(const std::vector<float>& v);

  // y[m, n] = act(x . W^T + b) (+ residual). Exactly one of `x` (fp32, quantised on the fly for
  // int8 weights) or `qin` must be given.
  void dense(const DenseW& w, const float* x, const QuantIn* qin, int64_t m, int act,
             const float* residual, float* y, DecodeState* ws);
  void norm(const NormW& w, const float* x, int64_t rows, int64_t depth, float* y);
  void norm_quant(const NormW& w, const float* x, int64_t rows, int64_t depth, float* y,
                  int8_t* q, float* s);
  void quant(const float* x, int64_t rows, int64_t depth, int8_t* q, float* s);

  void encode(DecodeState& st, const int32_t* src_ids, const int32_t* src_lengths);
  void encode_features(DecodeState& st, const float* features, int64_t frames);
  // encoder layers + output norm + the cross-attention K/V projections, on x (scratch y)
  void encode_layers(DecodeState& st, float* x, float* y);
  bool padding_removal_pays(int64_t padded_rows, int64_t valid_rows) const;  // (Padder, engine_session.cc)
  // amax_buf: zeroed [rows] scratch for the producer-side absmax of the gated pair (decode steps)
  void run_ffn(const FfnW& ffn, bool pre_norm, int64_t rows, int64_t d, float*& x, float*& y,
               DecodeState& st, float* amax_buf = nullptr, bool qin_ready = false);
  void finish_variant_layer(const LayerW& layer, int64_t rows, int64_t d, float*& x, float*& y,
                            DecodeState& st);
  bool dense_out_next_norm(const DenseW& w, const QuantIn& cq, int64_t rows, const float* residual,
                           float* y, const NormW* next_ln, const DenseW* next_dense,
                           DecodeState& st);
  void launch_step(DecodeState& st);
  const float* sinusoidal_table(int64_t positions, int64_t depth);
  void rotary_tables(const AttnW& a, int64_t positions, const float** sin, const float** cos);
  void begin_kernel(const char* kind, double bytes = 0, double ops = 0, int64_t m = 0,
                    int64_t n = 0, int64_t k = 0);
  void end_kernel();

  std::shared_ptr<ModelFile> model_;
  std::shared_ptr<TpComm> tp_;  // nullptr: no tensor parallelism
  bool tp_fast_ = false;  // CT2X_TP_FAST=1: the reference's single fp32 all-reduce (not bit-exact)
  bool tp_graph_ = true;  // CT2X_TP_GRAPH=0: eager launches under tensor parallelism
  ComputeType compute_type_;
  int device_ = 0;
  hipStream_t stream_ = nullptr;
  bool round_before_cast_ = true;  // weights converted at load time
  int quant_mode_ = 1;             // activation quantisation flags (device_utils.h)
  int64_t alignment_layer_ = 0, alignment_heads_ = 1;  // return_attention (decoder attributes)
  bool has_encoder_ = false;
  StackW enc_, dec_;
  std::vector<void*> device_blocks_;
  int64_t weight_bytes_ = 0;
  std::atomic<int64_t> launch_counter_{0};
  std::mutex table_mutex_;
  std::atomic<bool> profiling_{false};
  int32_t* profile_scratch_ = nullptr;  // target of the calibration no-op (profile_begin)
  struct PendingKernel {
    std::string label;
    hipEvent_t start, stop;
    double bytes, ops;
  };
  std::vector<PendingKernel> pending_;
  std::vector<hipEvent_t> event_pool_;
  std::atomic<int64_t> kernels_per_step_{0};
  std::vector<Tap>* taps_ = nullptr;
  std::string tap_scope_;
  int tap_step_ = -1;
  int tap_index_ = 0;
  void tap_scope(const std::string& scope) {
    if (!taps_) return;
    tap_scope_ = (tap_step_ >= 0 ? "s" + std::to_string(tap_step_) + "." : std::string()) + scope;
    tap_index_ = 0;
  }
  void tap(const char* producer, const void* dev, int dtype, int64_t rows, int64_t cols,
           bool indexed = true);

  // Lazily grown tables.
  float* sin_pos_table_ = nullptr;
  int64_t sin_pos_positions_ = 0, sin_pos_depth_ = 0;
  float* rope_sin_ = nullptr;
  float* rope_cos_ = nullptr;
  int64_t rope_positions_ = 0;
  std::string rope_key_;  // rotary configuration the cached tables were built for
};

struct DecodeState {
  Engine* engine = nullptr;
  hipStream_t stream = nullptr;  // all work of this session is enqueued here
  // weight prefetch (Engine::prefetch_weight): ranges the next row-wide launch carries along
  PrefetchArgs pf_pending;
  bool pf_active = false;  // inside Engine::launch_step (not prefill / encode)
  int64_t batch = 0, beam = 1, rows = 0, t_max = 0, s_max = 0, prefill_max = 0;
  // device buffers (owned; released in the destructor through the caching allocator)
  std::vector<void*> owned;
  float* x0 = nullptr;       // residual stream [rows, d]
  float* x1 = nullptr;
  float* x2 = nullptr;       // third residual-stream buffer (parallel-residual layers only)
  float* qkv = nullptr;      // [rows, q_dim + 2 * kv_dim]
  float* ctx = nullptr;      // [rows, d]
  float* inner = nullptr;    // [rows, ffn]
  float* inner2 = nullptr;   // [rows, ffn] (GLU gate)
  float* logits = nullptr;   // [rows, vocab]
  float* logprobs = nullptr; // [rows, vocab] (beam search)
  float* row_amax = nullptr; // [2 * layers, rows]: producer-side row absmax (zeroed every step)
  float* attn_amax_parts = nullptr;  // [rows, heads_kv]: partial maxima of the d_head 128 attention
  int32_t* tp_acc = nullptr; // tensor parallel: int32 / fp32 partial products [rows, d]
  float* tp_amax = nullptr;  // tensor parallel: per-row absmax [rows]
  int8_t* qbuf = nullptr;    // [rows, max(d, ffn)]
  uint16_t* hbuf = nullptr;  // [rows, max(d, ffn)] float16 activations (AWQ Dense only)
  float* splitk_ws = nullptr;  // split-K partial sums of the 16-bit GEMM (decode rows > 128)
  size_t splitk_floats = 0;
  // tickets of the decode-row form that sums its K slices inside the launch (one per 64-column
  // tile; zero at session start, left zero by every launch)
  unsigned* splitk_tickets = nullptr;
  static constexpr size_t kSplitkTickets = 2048;
  float* awq_ws = nullptr;   // split-K slices of the AWQ GEMM (kAwqGemmWorkspaceFloats)
  float* qscale = nullptr;   // [rows]
  // Leader forms of the streaming Dense (gemm_s8_stream.hip AMODE 3 / 4, round 6): one tag word per
  // row, written by the launch's leader workgroups; zeroed whenever the host writes d_step (a tag is
  // (step + 1) * 1024 + lead_seq: unique only while the step counter moves forward). lead_seq: the
  // leader launches of the step being enqueued so far (host side, reset by launch_step).
  unsigned* lead_flag = nullptr;  // [16]
  int lead_seq = 0;
  int32_t* d_ids = nullptr;  // [rows] next input ids
  int32_t* d_step = nullptr; // [1]
  int32_t* mem_lengths = nullptr;  // [batch]
  // self-attention caches, ping-pong for beam reorder: [2][layers] each [rows, heads_kv, t_max, dh]
  std::vector<float*> self_k[2], self_v[2];
  // Beam search without moving the caches (AttnDecodeParams::ancestry): ONE buffer set and two
  // ping-pong tables [t_max + 1][rows] of cache rows, updated by the beam selection kernel.
  bool ancestry = false;
  int32_t* anc[2] = {nullptr, nullptr};
  int cur = 0;
  const float** d_cache_src = nullptr;  // device arrays of the 2*layers cache base pointers
  float** d_cache_dst = nullptr;
  const float** d_cache_src_alt = nullptr;
  float** d_cache_dst_alt = nullptr;
  // cross-attention K/V: per layer [batch, s_max, 2 * kv_dim] (GEMM output layout)
  std::vector<float*> cross_kv;
  // 16-bit compute types, d_head 64: the same memory K/V in the compute type for the decode
  // steps' encoder-decoder attention (attention_cross_h.hip); empty otherwise
  std::vector<uint16_t*> cross_kv16;
  // history of sampled ids / scores: [t_max][rows]
  int32_t* hist_ids = nullptr;
  float* hist_scores = nullptr;
  // beam search
  float* beam_scores = nullptr;   // [rows] cumulative log prob
  float* topk_scores = nullptr;   // [batch, 2*beam]
  int32_t* topk_ids = nullptr;    // [batch, 2*beam]
  int32_t* d_parents = nullptr;   // [rows]
  void* topk_ws = nullptr;        // scratch of the two-launch TopK
  size_t topk_ws_bytes = 0;
  // Device-resident beam step (set by decoding.cc): candidates of every step, [t_max + 1][batch,
  // 2*beam], and the batch-level hard prefix.
  bool fuse_beam = false;
  int32_t* hist_cand_word = nullptr;
  float* hist_cand_score = nullptr;
  int32_t* hist_cand_origin = nullptr;
  int32_t* d_bprefix = nullptr;      // [batch, bprefix_stride]
  int32_t* d_bprefix_len = nullptr;  // [batch]
  int32_t bprefix_stride = 0;
  int32_t* d_src = nullptr;          // encoder input ids [batch * s_max] (kept for reencode)
  // Padder (src/padder.cc:8-74; TransformerEncoder::operator(), src/layers/transformer.cc:440-470):
  // the encoder's row-wise layers run on the valid source positions only. pad_rows = their count
  // (0: the padded layout throughout); the two index maps of the reference's Padder on the device.
  int64_t pad_rows = 0;
  int32_t* d_padded_to_flat = nullptr;  // [pad_rows] padded row of every valid position
  int32_t* d_flat_to_padded = nullptr;  // [batch * s_max] valid row a padded position shows (its own, or
                                        // the sentence's last valid one behind its end)
  std::vector<int32_t> h_pad_maps;      // host copies the uploads read from
  float* pad_qkv = nullptr;             // the re-padded fused projection when the feed-forward's inner buffer is too narrow for it
  int32_t* anc_init = nullptr;       // the identity the ancestry tables start from (kept for reencode)
  int64_t graph_kernels = 0;         // launches of the captured step (kernels_per_step of a replay)
  float* att_hist = nullptr;         // return_attention: [t_max + 1][rows][s_max] (enable_attention)
  float* logit_hist = nullptr;       // return_logits_vocab: [kLogitsRing][rows][vocab] (enable_logits_history)
  bool beam_mode = false;            // the state holds the beam search's buffers (beam > 1, or forced)
  float bias_beta = 0.f;             // biased decoding (DecodingOptions::prefix_bias_beta)
  uint8_t* d_diverged = nullptr;     // [rows] beams that left their prefix
  int32_t* pin_cand_word[2] = {nullptr, nullptr};
  float* pin_cand_score[2] = {nullptr, nullptr};
  int32_t* pin_cand_origin[2] = {nullptr, nullptr};
  int64_t pin_cand_capacity = 0;
  hipEvent_t pin_cand_event[2] = {nullptr, nullptr};
  // pinned host staging for the pipelined history read-back (greedy loop)
  // pinned mirror of hist_ids written by the fused greedy sampler itself (GreedyFuse::host_hist_ids):
  // while a step group ran on that path (hist_on_host) the loop reads the ids from it and no copy
  // is enqueued; pin_first = the group's first step per slot
  int32_t* host_hist_ids = nullptr;
  bool hist_on_host = false;
  int64_t pin_first[2] = {0, 0};
  bool pin_from_host[2] = {false, false};
  int32_t* pin_ids[2] = {nullptr, nullptr};
  float* pin_scores[2] = {nullptr, nullptr};
  int64_t pin_capacity = 0;
  hipEvent_t pin_event[2] = {nullptr, nullptr};
  // Greedy sampling fused at the end of the step (and of its graph): set by decoding.cc.
  bool fuse_greedy = false;
  int32_t g_eos = -1, g_min_length = 0;
  bool g_scores = false;
  // Hard decoding prefix of the fused sampler (ragged prompts): ids [rows, prefix_stride], lengths
  int32_t* d_done = nullptr;   // rows-finished counter of the fused sampler (advances d_step)
  // greedy arg-max fused into the vocabulary projection (GreedyFuse, kernels.h): per-row packed
  // maxima and the workgroup ticket, both zero between launches
  unsigned long long* gf_slots = nullptr;
  unsigned int* gf_ticket = nullptr;
  int32_t* d_prefix = nullptr;
  int32_t* d_prefix_len = nullptr;
  int32_t prefix_stride = 0, prefix_first_step = 0;
  bool prefix_extends_min = true;  // return_prefix == false (generation)
  // logits processors (set_logits_processing)
  bool lp_active = false;
  float lp_penalty = 1.f;
  int32_t lp_ngram = 0, lp_n_sup = 0;
  int32_t* d_sup_ids = nullptr;
  int32_t* d_sup_off = nullptr;
  int32_t* alive_seq[2] = {nullptr, nullptr};  // beam search: tokens of the active beams,
                                               // [rows, t_max], ping-pong with the cache parity
  // Whisper's processors (set_whisper_processing)
  bool wh_active = false;          // SuppressTokensBegin and / or ApplyTimestampRules
  WhisperRulesParams wh;           // static part: ids, flags
  int32_t wh_no_speech_id = -1;    // >= 0: GetNoSpeechProbs at the first decoding step
  float* d_no_speech = nullptr;    // [rows]
  float* enc_out = nullptr;        // encoder output (read_encoder_output)
  // RandomSampler settings of the fused sampler (topk == 1: BestSampler / arg-max)
  int32_t g_topk = 1;
  float g_topp = 1.f, g_temperature = 1.f;
  uint32_t g_seed = 0;
  // graph
  hipGraph_t graph = nullptr;
  hipGraphExec_t graph_exec[2] = {nullptr, nullptr};  // one per cache parity
  bool graph_failed = false;  // capture refused (tensor parallel + collective library): eager
  bool graph_ready[2] = {false, false};
  // a whole group of greedy steps as ONE graph (Engine::step_group; reused sessions only)
  // cluster-resident decoder stack: hand-over lines, arrival counters (zero at session start,
  // monotonic afterwards), the "a member waited in vain" flag
  float* cl_xch = nullptr;
  unsigned* cl_counters = nullptr;
  size_t cl_xch_floats = 0, cl_counter_words = 0;
  // projection tail of the one-row launch (ClusterProjArgs): the clusters' quantised output rows
  // as tagged slots, the session's launch count (the tag)
  unsigned long long* cl_proj_q = nullptr;
  unsigned* cl_proj_seq = nullptr;
  unsigned long long* cl_proj_board = nullptr;  // [rows][rows * 8] row maxima of the workgroups (ClusterProjArgs::board)
  // beam search, round 6: per-row candidates of the fused LogSoftMax + TopK ([rows, 2 * beam]) and the
  // ticket of the step advance that rides in beam_select_kernel (nullptr: the separate launches)
  float* bt_part_v = nullptr;
  int32_t* bt_part_i = nullptr;
  unsigned* bt_ticket = nullptr;
  // Only a decoding loop that holds the device's gate (ClusterLease) runs the launch; every other
  // caller of Engine::step (scoring, step_logits, the alternatives' pre-step) and a session that
  // did not get the gate take the launch-per-op step.
  bool cl_active = false;
  int graph_cluster = -1;  // what the captured graphs contain: -1 nothing captured, 0 / 1 = cl_active then
  int* cl_error = nullptr;
  int cl_t_pad = 0;
  unsigned long long* cl_stamps = nullptr;  // CT2X_CLUSTER_STAMPS=1: timeline of workgroup 0, last step
  // 16-bit encoder-decoder attention with the memory split over workgroups (attention_cross_h.hip):
  // per-chunk records and the (entry, head) tickets (zero at session start, left zero by a launch)
  float* xh_scratch = nullptr;
  unsigned* xh_tickets = nullptr;
  hipGraphExec_t group_exec = nullptr;
  int group_steps = 0;
  int group_parity = 0;
  int64_t reuses = 0;  // times Engine::reencode restarted this session

  ~DecodeState();
};

// Thrown by Engine::check_cluster (see there).
class ClusterTimeout : public Error {
public:
  using Error::Error;
};

// Scope of a decoding loop's use of the decoder-stack launch: tries the device's gate and marks
// the session accordingly; releases on destruction (Engine::cluster_gate_try_acquire).
class ClusterLease {
public:
  ClusterLease(Engine& engine, DecodeState& st, bool allowed) : st_(st), device_(engine.device_index()) {
    held_ = allowed && st.cl_xch != nullptr && Engine::cluster_gate_try_acquire(device_);
    st.cl_active = held_;
  }
  ~ClusterLease() {
    st_.cl_active = false;
    if (held_) Engine::cluster_gate_release(device_);
  }
  ClusterLease(const ClusterLease&) = delete;
  ClusterLease& operator=(const ClusterLease&) = delete;
  bool held() const { return held_; }

private:
  DecodeState& st_;
  int device_;
  bool held_ = false;
};

}  // namespace ct2x
