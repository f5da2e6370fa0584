// Sessions of the engine (engine.h): the per-call device state, the encoder passes (token ids or
// Whisper features), the feed-forward block, layer variants and the sequence-mode decoder pass.
#include "engine.h"

#include <fcntl.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include "host_math.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>

#include "engine_internal.h"

#include "runtime.h"

namespace ct2x {

// ---------------------------------------------------------------------------------------------
// Session
// ---------------------------------------------------------------------------------------------

// tests only (ct2x_debug_set_beam_ancestry): 0 = beam search moves the caches every step
std::atomic<int> g_debug_beam_ancestry{-1};

DecodeState::~DecodeState() {
  if (engine) {
    (void)hipSetDevice(engine->device_index());
    (void)hipStreamSynchronize(stream ? stream : engine->stream());
  }
  for (int i = 0; i < 2; ++i) {
    if (graph_exec[i]) (void)hipGraphExecDestroy(graph_exec[i]);
    if (i == 0 && group_exec) (void)hipGraphExecDestroy(group_exec);
    if (i == 0) pinned_free(host_hist_ids);
    pinned_free(pin_ids[i]);
    pinned_free(pin_scores[i]);
    if (pin_event[i]) (void)hipEventDestroy(pin_event[i]);
    pinned_free(pin_cand_word[i]);
    pinned_free(pin_cand_score[i]);
    pinned_free(pin_cand_origin[i]);
    if (pin_cand_event[i]) (void)hipEventDestroy(pin_cand_event[i]);
  }
  if (graph) (void)hipGraphDestroy(graph);

  for (void* p : owned) {
    try { device_free(p); } catch (...) {}
  }
}


std::unique_ptr<DecodeState> Engine::start_decoder_only(int64_t batch, int64_t beam,
                                                        int64_t t_max, int64_t prefill_len) {
  return start(nullptr, nullptr, batch, 0, beam, t_max, prefill_len);
}

std::unique_ptr<DecodeState> Engine::start_features(const float* features, int64_t batch,
                                                    int64_t frames, int64_t beam, int64_t t_max,
                                                    int64_t prefill_len, bool beam_search_state) {
  CT2X_CHECK(is_whisper(), "this model has no convolutional front end (not a Whisper model)");
  CT2X_CHECK(features != nullptr && batch > 0, "start_features: no features");
  // WhisperEncoder::operator() (src/layers/whisper.cc:33-50): at most max_input_time() =
  // 2 * positions frames
  if (frames < 1 || whisper_memory_length(frames) > enc_.num_positions)
    throw std::invalid_argument("Invalid input features shape: expected at most " +
                                std::to_string(2 * enc_.num_positions) + " frames, but got " +
                                std::to_string(frames));
  return start(nullptr, nullptr, batch, whisper_memory_length(frames), beam, t_max, prefill_len,
               beam_search_state, features, frames);
}

std::unique_ptr<DecodeState> Engine::start(const int32_t* src_ids, const int32_t* src_lengths,
                                           int64_t batch, int64_t s_max, int64_t beam,
                                           int64_t t_max, int64_t prefill_len, bool beam_search_state,
                                           const float* features, int64_t frames) {
  CT2X_CHECK(is_whisper() == (features != nullptr),
             "Whisper models take input features, every other model token ids");
  CT2X_HIP_CHECK(hipSetDevice(device_));
  auto st = std::make_unique<DecodeState>();
  st->engine = this;
  st->stream = thread_stream();  // the calling (worker) thread's stream: sessions run concurrently
  if (tp_ && tp_->peer_box() && std::string(tp_->backend()) == "in-process")
    st->stream = thread_stream_with_priority(tp_->rank());  // (ranks on one GPU: distinct queues)
  tls_stream = st->stream;
  set_active_stream(tls_stream);
  st->batch = batch;
  st->beam = beam;
  // a beam search of width 1 (biased decoding with beam_size 1) still needs the search's buffers
  const bool beam_mode = beam > 1 || beam_search_state;
  st->beam_mode = beam_mode;
  st->rows = batch * beam;
  st->t_max = t_max;
  st->s_max = s_max;
  const int64_t rows = st->rows, d = dec_.d_model, V = dec_.projection.n;
  CT2X_CHECK(t_max + 1 <= 16384 && s_max + 1 <= 16384,
             "sequences longer than 16383 positions are not supported by this backend yet");

  int64_t ffn = 0, qkv_dim = 0;
  for (const auto& l : dec_.layers) {
    ffn = std::max(ffn, l.ffn.l0.n);
    qkv_dim = std::max(qkv_dim, l.self.lin[0].n);
  }
  // Workspace sized for the larger of the decode rows and the encoder rows.
  // (a beam-search prefill forwards the prompt once per hypothesis row)
  // (the first convolution of a Whisper encoder runs over `frames` rows per batch entry)
  const int64_t enc_rows = std::max(has_encoder_ ? batch * std::max(s_max, frames) : 0,
                                    rows * prefill_len);
  st->prefill_max = prefill_len;
  int64_t enc_ffn = 0, enc_qkv = 0;
  for (const auto& l : enc_.layers) {
    enc_ffn = std::max(enc_ffn, l.ffn.l0.n);
    enc_qkv = std::max(enc_qkv, l.self.lin[0].n);
  }
  const int64_t wrows = std::max(rows, enc_rows);
  const int64_t wffn = std::max(ffn, enc_ffn);
  const int64_t wqkv = std::max(qkv_dim, enc_qkv);
  st->x0 = alloc<float>(*st, wrows * d);
  st->x1 = alloc<float>(*st, wrows * d);
  for (const auto& l : dec_.layers)
    if (l.variant == 1 && !st->x2) st->x2 = alloc<float>(*st, wrows * d);
  st->qkv = alloc<float>(*st, wrows * wqkv);
  st->ctx = alloc<float>(*st, wrows * d);
  st->inner = alloc<float>(*st, wrows * wffn);
  st->inner2 = alloc<float>(*st, wrows * wffn);
  st->row_amax = alloc<float>(*st, 2 * static_cast<int64_t>(dec_.layers.size()) * rows);
  {
    int64_t max_kv = 0;
    for (const auto& l : dec_.layers)
      if (l.self.dhead == 128) max_kv = std::max(max_kv, l.self.heads_kv);
    if (max_kv > 0) st->attn_amax_parts = alloc<float>(*st, rows * max_kv);
  }
  if (tp_) {
    st->tp_acc = alloc<int32_t>(*st, wrows * d);
    st->tp_amax = alloc<float>(*st, wrows);
  }
  st->qbuf = alloc<int8_t>(*st, wrows * std::max(d, wffn));
  if (has_awq_) {
    st->hbuf = alloc<uint16_t>(*st, wrows * std::max(d, wffn));
    // split-K slices of the GEMM-layout AWQ Dense: owned by the session, so that a captured step
    // neither allocates nor replays a pointer the allocator has handed to someone else
    st->awq_ws = alloc<float>(*st, static_cast<int64_t>(kAwqGemmWorkspaceFloats));
  }
  st->qscale = alloc<float>(*st, wrows);
  st->lead_flag = alloc<unsigned>(*st, 16);
  if (compute_type_ == ComputeType::kFloat16 || compute_type_ == ComputeType::kBFloat16) {
    // split-K partials of the 16-bit GEMM at a few hundred decode rows (beam search on large
    // batches): session owned, so a captured step never allocates
    st->splitk_floats = static_cast<size_t>(16) * rows * std::max(std::max(d, wffn), wqkv);
    // (the vocabulary projection of a few rows: K slices of 512 over workgroups, gemm_h_rows_kernel)
    if (rows <= 64)
      st->splitk_floats = std::max(st->splitk_floats,
                                   static_cast<size_t>((d + 511) / 512) * rows * static_cast<size_t>(V));
    st->splitk_ws = alloc<float>(*st, static_cast<int64_t>(st->splitk_floats));
    st->splitk_tickets = alloc<unsigned>(*st, static_cast<int64_t>(DecodeState::kSplitkTickets));
    CT2X_HIP_CHECK(hipMemsetAsync(st->splitk_tickets, 0, sizeof(unsigned) * DecodeState::kSplitkTickets, cur_stream()));
  }
  st->logits = alloc<float>(*st, rows * V);
  st->d_ids = alloc<int32_t>(*st, rows);
  st->d_step = alloc<int32_t>(*st, 1);
  st->d_done = alloc<int32_t>(*st, 1);
  CT2X_HIP_CHECK(hipMemsetAsync(st->d_done, 0, sizeof(int32_t), cur_stream()));
  st->gf_slots = alloc<unsigned long long>(*st, rows * GreedyFuse::kSubSlots);
  st->gf_ticket = alloc<unsigned int>(*st, 1);
  CT2X_HIP_CHECK(hipMemsetAsync(st->gf_slots, 0, sizeof(unsigned long long) * rows * GreedyFuse::kSubSlots,
                                cur_stream()));
  CT2X_HIP_CHECK(hipMemsetAsync(st->gf_ticket, 0, sizeof(unsigned int), cur_stream()));
  st->hist_ids = alloc<int32_t>(*st, (t_max + 1) * rows);
  st->host_hist_ids = static_cast<int32_t*>(pinned_malloc(sizeof(int32_t) * (t_max + 1) * rows));
  st->hist_scores = alloc<float>(*st, (t_max + 1) * rows);
  CT2X_HIP_CHECK(hipMemsetAsync(st->d_step, 0, sizeof(int32_t), cur_stream()));
  reset_lead_flags(*st);
  st->logprobs = alloc<float>(*st, rows * V);  // log-probabilities (beam search) / exp scratch
  if (beam_mode) {
    st->beam_scores = alloc<float>(*st, rows);
    st->topk_scores = alloc<float>(*st, batch * 2 * beam);
    st->topk_ids = alloc<int32_t>(*st, batch * 2 * beam);
    st->d_parents = alloc<int32_t>(*st, rows);
    CT2X_CHECK(2 * beam <= 64, "beam_size must be <= 32");
    // CT2X_BEAM_FUSED_TOPK=0: LogSoftMax, TopK (two launches) and the step advance as launches of their own (A/B)
    static const bool fused_topk = [] {
      const char* e = std::getenv("CT2X_BEAM_FUSED_TOPK");
      return e == nullptr || e[0] != '0';
    }();
    if (fused_topk && beam_logprobs_topk_applicable(V, !half_compute(), static_cast<int32_t>(beam),
                                                    static_cast<int32_t>(2 * beam))) {
      st->bt_part_v = alloc<float>(*st, rows * 2 * beam);
      st->bt_part_i = alloc<int32_t>(*st, rows * 2 * beam);
      st->bt_ticket = alloc<unsigned>(*st, 1);
      CT2X_HIP_CHECK(hipMemsetAsync(st->bt_ticket, 0, sizeof(unsigned), cur_stream()));
    }
    st->topk_ws_bytes = topk_workspace_bytes(batch, beam * V, 2 * beam);
    if (st->topk_ws_bytes) st->topk_ws = alloc<char>(*st, static_cast<int64_t>(st->topk_ws_bytes));
    const int64_t cand = (t_max + 1) * batch * 2 * beam;
    st->hist_cand_word = alloc<int32_t>(*st, cand);
    st->hist_cand_score = alloc<float>(*st, cand);
    st->hist_cand_origin = alloc<int32_t>(*st, cand);
  }

  const int n_layers = static_cast<int>(dec_.layers.size());
  // Beam search reads the self-attention caches through an ancestry table instead of reordering
  // them every step when every layer's decode attention runs the exact kernel, which follows the
  // table (CT2X_BEAM_ANCESTRY=0: the one-launch gather between two buffer sets).
  if (beam_mode) {
    static const bool allowed = [] {
      const char* e = tuning_env("CT2X_BEAM_ANCESTRY");
      return e == nullptr || e[0] != '0';
    }();
    bool ok = allowed && g_debug_beam_ancestry.load() != 0;
    const float* aligned = reinterpret_cast<const float*>(static_cast<uintptr_t>(256));
    for (int l = 0; ok && l < n_layers; ++l) {
      const AttnW& a = dec_.layers[l].self;
      AttnDecodeParams p;
      p.q = p.k_new = p.v_new = aligned;
      p.k_cache = p.v_cache = const_cast<float*>(aligned);
      p.kv_row_stride = a.heads_kv * t_max * a.dhead;
      p.kv_head_stride = t_max * a.dhead;
      p.kv_time_stride = a.dhead;
      p.rows = rows;
      p.heads = a.heads;
      p.heads_kv = a.heads_kv;
      p.dhead = a.dhead;
      p.t_max = t_max;
      if (a.rotary) {
        p.rope_sin = p.rope_cos = aligned;
        p.rot_dim = a.rotary_dim == 0 ? a.dhead : a.rotary_dim;
      }
      p.ancestry = reinterpret_cast<const int32_t*>(aligned);
      p.ancestry_stride = rows;
      p.allow_inexact = half_compute();
      // (16-bit compute types: the split-mode kernel follows the table too, attention_decode2.hip)
      ok = attention_exact_decode_applicable(p, false) ||
           (p.allow_inexact && attention_decode2_applicable(p, false));
    }
    st->ancestry = ok;
    if (ok) {
      std::vector<int32_t> identity(static_cast<size_t>((t_max + 1) * rows));
      for (int64_t t = 0; t <= t_max; ++t)
        for (int64_t r = 0; r < rows; ++r) identity[t * rows + r] = static_cast<int32_t>(r);
      for (int i = 0; i < 2; ++i) {
        st->anc[i] = alloc<int32_t>(*st, (t_max + 1) * rows);
        CT2X_HIP_CHECK(hipMemcpy(st->anc[i], identity.data(), identity.size() * sizeof(int32_t),
                                 hipMemcpyHostToDevice));
      }
      // (a pristine copy: a reused session starts from the identity again, Engine::reencode)
      st->anc_init = alloc<int32_t>(*st, (t_max + 1) * rows);
      CT2X_HIP_CHECK(hipMemcpy(st->anc_init, identity.data(), identity.size() * sizeof(int32_t),
                               hipMemcpyHostToDevice));
    }
  }
  const int n_sets = (beam_mode && !st->ancestry) ? 2 : 1;
  for (int s = 0; s < n_sets; ++s) {
    for (int l = 0; l < n_layers; ++l) {
      const AttnW& a = dec_.layers[l].self;
      const int64_t sz = rows * a.heads_kv * t_max * a.dhead;
      st->self_k[s].push_back(alloc<float>(*st, sz));
      st->self_v[s].push_back(alloc<float>(*st, sz));
    }
  }
  if (beam_mode && !st->ancestry) {
    // Device tables of cache base pointers for the one-launch beam gather, both directions.
    std::vector<const float*> src(2 * n_layers), src_alt(2 * n_layers);
    std::vector<float*> dst(2 * n_layers), dst_alt(2 * n_layers);
    for (int l = 0; l < n_layers; ++l) {
      src[2 * l] = st->self_k[0][l];
      src[2 * l + 1] = st->self_v[0][l];
      dst[2 * l] = st->self_k[1][l];
      dst[2 * l + 1] = st->self_v[1][l];
      src_alt[2 * l] = st->self_k[1][l];
      src_alt[2 * l + 1] = st->self_v[1][l];
      dst_alt[2 * l] = st->self_k[0][l];
      dst_alt[2 * l + 1] = st->self_v[0][l];
    }
    st->d_cache_src = alloc<const float*>(*st, 2 * n_layers);
    st->d_cache_dst = alloc<float*>(*st, 2 * n_layers);
    st->d_cache_src_alt = alloc<const float*>(*st, 2 * n_layers);
    st->d_cache_dst_alt = alloc<float*>(*st, 2 * n_layers);
    const size_t bytes = sizeof(void*) * 2 * n_layers;
    CT2X_HIP_CHECK(hipMemcpy(st->d_cache_src, src.data(), bytes, hipMemcpyHostToDevice));
    CT2X_HIP_CHECK(hipMemcpy(st->d_cache_dst, dst.data(), bytes, hipMemcpyHostToDevice));
    CT2X_HIP_CHECK(hipMemcpy(st->d_cache_src_alt, src_alt.data(), bytes, hipMemcpyHostToDevice));
    CT2X_HIP_CHECK(hipMemcpy(st->d_cache_dst_alt, dst_alt.data(), bytes, hipMemcpyHostToDevice));
  }

  if (has_encoder_) {
    st->mem_lengths = alloc<int32_t>(*st, batch);
    for (int l = 0; l < n_layers; ++l) {
      const AttnW& a = dec_.layers[l].cross;
      st->cross_kv.push_back(alloc<float>(*st, batch * s_max * a.lin[1].n));
    }
    {
      // a 16-bit copy of the memory K/V for the decode steps (bytes halved, one workgroup per
      // (entry, head) shares every key / value among the entry's hypothesis rows)
      static const bool disabled = tuning_env("CT2X_NO_CROSS_KV16") != nullptr;  // (A/B runs, TUNING=1)
      bool ok = half_compute() && !tp_ && !disabled;
      for (int l = 0; ok && l < n_layers; ++l) {
        const AttnW& a = dec_.layers[l].cross;
        CrossAttnHParams cp;
        cp.kv = reinterpret_cast<const uint16_t*>(static_cast<uintptr_t>(256));
        cp.batch = batch;
        cp.beam = beam;
        cp.heads = a.heads;
        cp.heads_kv = a.heads_kv;
        cp.dhead = a.dhead;
        cp.s_max = s_max;
        ok = s_max >= 1 && a.lin[1].n == 2 * a.heads_kv * a.dhead && cross_attention_h_applicable(cp);
        // d_head 128 (round 4): only for long memories. At the NLLB-3.3B shape (32 source positions)
        // one 512-thread workgroup per (entry, head) has 32 keys to score and the per-(row, head)
        // kernel over the fp32 memory is faster (measured: 7.2 - 7.5 against 8.1 ms per decode step)
        if (a.dhead == 128 && s_max < 128) ok = false;
      }
      if (ok) {
        for (int l = 0; l < n_layers; ++l)
          st->cross_kv16.push_back(alloc<uint16_t>(*st, batch * s_max * dec_.layers[l].cross.lin[1].n));
        // long memories: several workgroups per (entry, head), merged by the last to arrive
        int64_t floats = 0, heads = 0;
        for (int l = 0; l < n_layers; ++l) {
          const AttnW& a = dec_.layers[l].cross;
          floats = std::max(floats, cross_attention_h_scratch_floats(batch, beam, a.heads, a.dhead, s_max));
          heads = std::max<int64_t>(heads, a.heads);
        }
        if (floats > 0) {
          st->xh_scratch = alloc<float>(*st, floats);
          st->xh_tickets = alloc<unsigned>(*st, batch * heads);
          CT2X_HIP_CHECK(hipMemsetAsync(st->xh_tickets, 0, sizeof(unsigned) * batch * heads, cur_stream()));
        }
      }
    }
    // greedy: a cluster per row up to 32 rows, 2 - 4 independent rows per cluster up to 128 rows
    // (profiles/r05_cluster_many_rows_ab_b.txt: +31 % at 64 rows, +10 % at 96, +6.5 % at 128 against
    // the launch-per-op step; the first, spilling build of the several-rows form lost 9 % at 128)
    const bool cluster_greedy = !beam_mode && beam == 1 &&
                                (rows <= 32 ? decoder_cluster_lds_ok(decoder_cluster_t_pad(t_max, s_max))
                                            : (rows <= cluster_max_rows() && decoder_cluster_beam_lds_ok(decoder_cluster_t_pad(t_max, s_max))));
    // beam search: a cluster per batch entry with its (<= 4) rows, through the ancestry table
    const bool cluster_beam = beam_mode && st->ancestry && beam >= 2 && beam <= 4 && batch <= 32 &&
                              decoder_cluster_beam_lds_ok(decoder_cluster_t_pad(t_max, s_max));
    if (cluster_ready_ && prefill_len == 0 && (cluster_greedy || cluster_beam)) {
      // the decoder stack as one launch per step (decoder_cluster.hip)
      const size_t xch_floats = (cluster_beam || rows > 32) ? decoder_cluster_beam_xch_floats(std::min<int64_t>(rows, 32))
                                                             : decoder_cluster_xch_floats(rows);
      st->cl_xch = alloc<float>(*st, static_cast<int64_t>(xch_floats));
      st->cl_counters = alloc<unsigned>(*st, rows * 16);
      st->cl_xch_floats = xch_floats;
      st->cl_counter_words = static_cast<size_t>(rows) * 16;
      st->cl_error = alloc<int>(*st, 1);
      st->cl_t_pad = decoder_cluster_t_pad(t_max, s_max);
      // (tagged hand-over: a slot's tag must never match by accident - the lines start at tag 0,
      // which no meeting uses, and the row epochs start at 0)
      CT2X_HIP_CHECK(hipMemsetAsync(st->cl_xch, 0, sizeof(float) * xch_floats, cur_stream()));
      CT2X_HIP_CHECK(hipMemsetAsync(st->cl_counters, 0, sizeof(unsigned) * rows * 16, cur_stream()));
      CT2X_HIP_CHECK(hipMemsetAsync(st->cl_error, 0, sizeof(int), cur_stream()));
      if (cluster_greedy && rows <= 32) {
        st->cl_proj_q = alloc<unsigned long long>(*st, rows * kClusterProjRowSlots);
        st->cl_proj_seq = alloc<unsigned>(*st, 1);
        CT2X_HIP_CHECK(hipMemsetAsync(st->cl_proj_q, 0, sizeof(unsigned long long) * rows * kClusterProjRowSlots, cur_stream()));
        CT2X_HIP_CHECK(hipMemsetAsync(st->cl_proj_seq, 0, sizeof(unsigned), cur_stream()));
        st->cl_proj_board = alloc<unsigned long long>(*st, rows * rows * kClusterMembers);
        CT2X_HIP_CHECK(hipMemsetAsync(st->cl_proj_board, 0, sizeof(unsigned long long) * rows * rows * kClusterMembers, cur_stream()));
      }
      if (const char* e = std::getenv("CT2X_CLUSTER_STAMPS"); e != nullptr && e[0] == '1')
        st->cl_stamps = alloc<unsigned long long>(*st, 2 + kClusterStampsPerLayer * n_layers + 8);
    }
    if (features)
      encode_features(*st, features, frames);
    else
      encode(*st, src_ids, src_lengths);
  }
  return st;
}

// largest greedy batch on the decoder-stack launch (CT2X_CLUSTER_MAX_ROWS, A/B runs; <= 128)
int64_t Engine::cluster_max_rows() {
  static const int64_t rows = [] {
    const char* e = std::getenv("CT2X_CLUSTER_MAX_ROWS");
    const int64_t v = e ? std::atoll(e) : 128;
    return std::max<int64_t>(32, std::min<int64_t>(v, 128));
  }();
  return rows;
}

namespace {
struct ClusterGate {
  std::atomic<bool> held{false};
  int fd = -1;           // the lock file, opened on first use (-2: no directory took it)
  std::mutex open_mutex;
};
ClusterGate& cluster_gate_of(int device_index) {
  static ClusterGate gates[64];
  return gates[device_index & 63];
}
}  // namespace

bool Engine::cluster_gate_try_acquire(int device_index) {
  ClusterGate& g = cluster_gate_of(device_index);
  bool expected = false;
  if (!g.held.compare_exchange_strong(expected, true, std::memory_order_acquire)) return false;
  // this process holds the gate; now the other processes on the same GPU
  {
    std::lock_guard<std::mutex> lock(g.open_mutex);
    if (g.fd == -1) {
      char bus[64] = "unknown";
      if (hipDeviceGetPCIBusId(bus, sizeof(bus), device_index) != hipSuccess) (void)hipGetLastError();
      for (char* c = bus; *c; ++c)
        if (*c == ':' || *c == '.' || *c == '/') *c = '_';
      g.fd = -2;
      for (const char* dir : {"/dev/shm", "/tmp"}) {
        const std::string path = std::string(dir) + "/ct2x_cluster_gate_" + bus;
        const int fd = ::open(path.c_str(), O_CREAT | O_RDWR | O_CLOEXEC, 0666);
        if (fd >= 0) {
          (void)::fchmod(fd, 0666);  // (another user's process must be able to open it too)
          g.fd = fd;
          break;
        }
      }
    }
  }
  // (no lock file anywhere: only this process is guarded - a cluster that then starves is voided
  // and the call re-run by the ClusterTimeout path)
  if (g.fd >= 0 && ::flock(g.fd, LOCK_EX | LOCK_NB) != 0) {
    g.held.store(false, std::memory_order_release);
    return false;
  }
  return true;
}

void Engine::cluster_gate_release(int device_index) {
  ClusterGate& g = cluster_gate_of(device_index);
  if (g.fd >= 0) (void)::flock(g.fd, LOCK_UN);
  g.held.store(false, std::memory_order_release);
}

void Engine::check_cluster(DecodeState& st) {
  if (st.cl_error == nullptr) return;
  if (st.cl_stamps != nullptr) {
    // phase timeline of workgroup 0 in the session's last step (wall_clock64: 100 MHz)
    const size_t n = 2 + kClusterStampsPerLayer * dec_.layers.size() + 8;  // (+ the projection tail's)
    std::vector<unsigned long long> h(n);
    CT2X_HIP_CHECK(hipMemcpy(h.data(), st.cl_stamps, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::fprintf(stderr, "decoder_stack timeline (us since launch start):");
    for (size_t i = 0; i < n; ++i) {
      if (i >= 1 && (i - 1) % kClusterStampsPerLayer == 0) std::fprintf(stderr, "\n  ");
      std::fprintf(stderr, " %.2f", static_cast<double>(h[i] - h[0]) * 0.01);
    }
    std::fprintf(stderr, "\n");
  }
  int flag = 0;
  CT2X_HIP_CHECK(hipMemcpyAsync(&flag, st.cl_error, sizeof(int), hipMemcpyDeviceToHost, st.stream));
  CT2X_HIP_CHECK(hipStreamSynchronize(st.stream));
  if (inject_cluster_timeout_ && st.cl_active) {
    inject_cluster_timeout_ = false;
    flag = 1;
  }
  if (flag != 0) {
    CT2X_HIP_CHECK(hipMemsetAsync(st.cl_error, 0, sizeof(int), st.stream));
    throw ClusterTimeout("decoder stack launch: a workgroup cluster waited in vain for one of its members (is "
                         "another persistent launch holding the device?); CT2X_CLUSTER_STACK=0 selects the "
                         "launch-per-op step");
  }
}

void Engine::reencode(DecodeState& st, const int32_t* src_ids, const int32_t* src_lengths) {
  CT2X_CHECK(has_encoder_ && !is_whisper() && st.engine == this && (st.beam == 1) == !st.beam_mode,
             "reencode: a plain greedy / beam search encoder-decoder session of this engine");
  CT2X_HIP_CHECK(hipSetDevice(device_));
  tls_stream = st.stream;
  set_active_stream(tls_stream);
  // what a fresh session starts from: step 0, the first cache set, no fused tail selected (the
  // done counter and the fused sampler's slots are zero between launches by construction)
  CT2X_HIP_CHECK(hipMemsetAsync(st.d_step, 0, sizeof(int32_t), cur_stream()));
  reset_lead_flags(st);
  st.cur = 0;
  st.fuse_greedy = st.fuse_beam = false;
  ++st.reuses;
  if (st.ancestry) {
    const size_t bytes = static_cast<size_t>((st.t_max + 1) * st.rows) * sizeof(int32_t);
    for (int i = 0; i < 2; ++i)
      CT2X_HIP_CHECK(hipMemcpyAsync(st.anc[i], st.anc_init, bytes, hipMemcpyDeviceToDevice, cur_stream()));
  }
  encode(st, src_ids, src_lengths);
}

// TransformerEncoder::operator() (src/layers/transformer.cc:427-471) on padded rows followed by
// the cross-attention K/V projection of every decoder layer (process_cross_attention,
// src/layers/attention.cc:371-440), which the reference runs lazily at decoding step 0.
// Padded positions are computed but never observed: every op on this path is row-wise and the
// attention kernels mask keys beyond lengths[b].
void Engine::encode(DecodeState& st, const int32_t* src_ids, const int32_t* src_lengths) {
  const int64_t B = st.batch, S = st.s_max, d = enc_.d_model, rows = B * S;
  if (st.d_src == nullptr) st.d_src = alloc<int32_t>(st, rows);
  int32_t* d_src = st.d_src;
  CT2X_HIP_CHECK(hipMemcpyAsync(d_src, src_ids, sizeof(int32_t) * rows, hipMemcpyHostToDevice,
                                cur_stream()));
  CT2X_HIP_CHECK(hipMemcpyAsync(st.mem_lengths, src_lengths, sizeof(int32_t) * B,
                                hipMemcpyHostToDevice, cur_stream()));
  const float* pos = enc_.position_table;
  if (enc_.sinusoidal_positions) pos = sinusoidal_table(S, d);
  if (enc_.position_table && S > enc_.num_positions)
    throw std::runtime_error("No position encodings are defined for positions >= " +
                             std::to_string(enc_.num_positions) + ", but got position " +
                             std::to_string(S - 1));
  float* x = st.x0;
  float* y = st.x1;
  launch_embed(enc_.emb.table, enc_.emb.is_int8, enc_.emb.scale, d_src, rows, d,
               enc_.embeddings_scale, enc_.scale_embeddings, pos, S, 0, nullptr, x, cur_stream());
  tap_scope("enc");
  tap("emb", x, 0, rows, d);
  if (enc_.layernorm_embedding.present) norm(enc_.layernorm_embedding, x, rows, d, x);
  // Padder::remove_padding (src/layers/transformer.cc:443-450, src/padder.cc:8-66): the layers run
  // on the valid positions only; attention re-pads its projections and un-pads its context
  // (src/layers/attention.cc:390-403, 590-596), the encoder output is padded again at the end - a
  // padded position then shows its sentence's LAST valid row, as Padder::add_padding's gather does.
  st.pad_rows = 0;
  int64_t total = 0;
  for (int64_t b = 0; b < B; ++b) total += std::max<int64_t>(0, std::min<int64_t>(src_lengths[b], S));
  bool all_positive = true;
  for (int64_t b = 0; b < B; ++b) all_positive = all_positive && src_lengths[b] >= 1;
  if (all_positive && !enc_.layers.empty() && padding_removal_pays(rows, total) && !tp_ && !taps_) {
    // (the re-padded projection borrows the feed-forward's inner buffer where a row of it fits)
    int64_t widest = 0;
    bool fits = true;
    for (const LayerW& layer : enc_.layers) {
      widest = std::max(widest, layer.self.lin[0].n);
      fits = fits && layer.self.lin[0].n <= layer.ffn.l0.n;
    }
    if (!fits && st.pad_qkv == nullptr) st.pad_qkv = alloc<float>(st, rows * widest);
    st.h_pad_maps.resize(static_cast<size_t>(total + rows));
    int32_t* p2f = st.h_pad_maps.data();
    int32_t* f2p = p2f + total;
    int64_t flat = 0;
    for (int64_t b = 0; b < B; ++b) {
      const int64_t len = std::min<int64_t>(src_lengths[b], S);
      for (int64_t t = 0; t < len; ++t) {
        p2f[flat + t] = static_cast<int32_t>(b * S + t);
        f2p[b * S + t] = static_cast<int32_t>(flat + t);
      }
      for (int64_t t = len; t < S; ++t) f2p[b * S + t] = static_cast<int32_t>(flat + len - 1);
      flat += len;
    }
    if (st.d_padded_to_flat == nullptr) {
      st.d_padded_to_flat = alloc<int32_t>(st, rows);
      st.d_flat_to_padded = alloc<int32_t>(st, rows);
    }
    CT2X_HIP_CHECK(hipMemcpyAsync(st.d_padded_to_flat, p2f, sizeof(int32_t) * total, hipMemcpyHostToDevice,
                                  cur_stream()));
    CT2X_HIP_CHECK(hipMemcpyAsync(st.d_flat_to_padded, f2p, sizeof(int32_t) * rows, hipMemcpyHostToDevice,
                                  cur_stream()));
    begin_kernel("padder_remove", double(total) * d * 8, 0, total, 0, d);
    launch_gather_rows(x, st.d_padded_to_flat, total, d * sizeof(float), y, cur_stream());
    end_kernel();
    std::swap(x, y);
    st.pad_rows = total;
  }
  encode_layers(st, x, y);
}

// Whether the reference's padding removal is applied (it always is when a length differs,
// src/layers/transformer.cc:446-450): here only where it returns more than its four gathers per
// layer cost - from a few thousand rows up with an eighth of them padding (below that an encoder
// layer is launch bound and costs the same for any row count; measured: DESIGN 5.00000).
// CT2X_PADDER=1 / 0: always (when a length differs) / never. Same results for every valid position.
bool Engine::padding_removal_pays(int64_t padded_rows, int64_t valid_rows) const {
  static const int mode = [] {
    const char* e = std::getenv("CT2X_PADDER");
    return e == nullptr ? -1 : (e[0] == '0' ? 0 : 1);
  }();
  if (valid_rows >= padded_rows || valid_rows < 1) return false;
  if (mode >= 0) return mode == 1;
  return padded_rows >= 4096 && (padded_rows - valid_rows) * 8 >= padded_rows;
}

// WhisperEncoder::operator() (src/layers/whisper.cc:26-62): conv1 -> conv2 -> Transpose ->
// position embeddings -> encoder layers -> layer_norm. Each convolution is im2col + Dense with
// the bias + GELU epilogue (the reference's CPU path is the same GEMM, conv1d_cpu.cc:139-207, with
// the operands swapped); the GEMM output [batch * time, channels] is already the transposed
// layout. int8 weights: the im2col rows are quantised per row, round-before-cast, no u8 shift
// (conv1d_cpu.cc:176-178, ops::Quantize::operator() INT8 arm, src/ops/quantize.cc:35-43).
void Engine::encode_features(DecodeState& st, const float* features, int64_t frames) {
  const int64_t B = st.batch, S = st.s_max, d = enc_.d_model, C = enc_.n_mels;
  CT2X_CHECK(S == whisper_memory_length(frames), "encode_features: state / frames mismatch");
  std::vector<int32_t> lens(static_cast<size_t>(B), static_cast<int32_t>(S));
  CT2X_HIP_CHECK(hipMemcpyAsync(st.mem_lengths, lens.data(), sizeof(int32_t) * B,
                                hipMemcpyHostToDevice, cur_stream()));
  float* d_feat = alloc<float>(st, B * C * frames);
  float* col1 = alloc<float>(st, B * frames * C * 3);
  float* h1 = alloc<float>(st, B * frames * enc_.conv1.n);
  float* col2 = alloc<float>(st, B * S * enc_.conv2.k);
  // int32 accumulators of an int8 convolution (the larger of the two outputs)
  float* conv_acc = enc_.conv1.quantized ? alloc<float>(st, B * frames * enc_.conv1.n) : nullptr;
  CT2X_HIP_CHECK(hipMemcpyAsync(d_feat, features, sizeof(float) * B * C * frames,
                                hipMemcpyHostToDevice, cur_stream()));
  CT2X_HIP_CHECK(hipStreamSynchronize(cur_stream()));  // `lens` and the caller's features may go away
  auto conv = [&](const DenseW& w, const float* cols, int64_t rows, float* y) {
    const int kGelu = 3;  // ops::ActivationType::GELU
    if (w.quantized) {
      begin_kernel("quantize_s8", double(rows) * w.k * 5 + rows * 4, 0, rows, 0, w.k);
      launch_quantize_s8(cols, rows, w.k, /*round before cast, no u8 shift*/ 1, st.qbuf, st.qscale, cur_stream());
      end_kernel();
      // exact int32 product, then the reference's convolution form of Dequantize (weight scale as
      // the reciprocal factor, input scale as the divisor: conv_dequantize_kernel), bias, GELU
      int32_t* acc = reinterpret_cast<int32_t*>(conv_acc);
      GemmS8Epilogue ep;
      ep.mode = 0;
      ep.c_i32 = acc;
      ep.ldy = w.n;
      begin_kernel("gemm_s8s8s32", double(rows) * w.k + double(w.n) * w.k + 4.0 * rows * w.n,
                   2.0 * rows * w.n * w.k, rows, w.n, w.k);
      launch_gemm_s8(st.qbuf, w.wq, rows, w.n, w.k, ep, cur_stream());
      end_kernel();
      begin_kernel("conv_dequantize", double(rows) * w.n * 8, 0, rows, w.n, 0);
      launch_conv_dequantize(acc, st.qscale, w.wscale, w.bias, rows, w.n, kGelu, y, cur_stream());
      end_kernel();
    } else {
      dense(w, cols, nullptr, rows, kGelu, nullptr, y, &st);
    }
  };
  tap_scope("enc.conv");
  begin_kernel("im2col", double(B) * frames * C * 16, 0, B * frames, 0, C * 3);
  launch_im2col_k3(d_feat, B, C, frames, frames, 1, 1, /*time_major=*/false, col1, cur_stream());
  end_kernel();
  conv(enc_.conv1, col1, B * frames, h1);
  begin_kernel("im2col", double(B) * S * enc_.conv2.k * 8, 0, B * S, 0, enc_.conv2.k);
  launch_im2col_k3(h1, B, enc_.conv1.n, frames, S, 2, 1, /*time_major=*/true, col2, cur_stream());
  end_kernel();
  float* x = st.x0;
  float* y = st.x1;
  conv(enc_.conv2, col2, B * S, x);
  CT2X_CHECK(enc_.position_table != nullptr, "Whisper encoder without position encodings");
  begin_kernel("add_positions", double(B) * S * d * 12, 0, B * S, 0, d);
  launch_add_positions(x, enc_.position_table, B, S, d, cur_stream());
  end_kernel();
  tap("emb", x, 0, B * S, d);
  encode_layers(st, x, y);
}

void Engine::encode_layers(DecodeState& st, float* x, float* y) {
  const int64_t B = st.batch, S = st.s_max, d = enc_.d_model, padded_rows = B * S;
  // (Padder: the row-wise layers see the valid positions only, Engine::encode)
  const bool unpadded = st.pad_rows > 0;
  const int64_t rows = unpadded ? st.pad_rows : padded_rows;
  int enc_layer_index = 0;
  for (const LayerW& layer : enc_.layers) {
    const AttnW& a = layer.self;
    tap_scope("enc.l" + std::to_string(enc_layer_index) + ".self");
    // --- self attention (MultiHeadAttention::operator(), src/layers/attention.cc:442-615)
    QuantIn qin;
    const float* attn_in = x;
    const bool q8 = a.lin[0].quantized;
    if (a.ln.present && enc_.pre_norm) {
      if (q8) {
        norm_quant(a.ln, x, rows, d, nullptr, st.qbuf, st.qscale);
        qin.q = st.qbuf;
        qin.s = st.qscale;
      } else {
        norm(a.ln, x, rows, d, st.ctx);
        attn_in = st.ctx;
      }
    }
    dense(a.lin[0], attn_in, qin.q ? &qin : nullptr, rows, -1, nullptr, st.qkv, &st);
    const int64_t qdim = a.heads * a.dhead, kvdim = a.heads_kv * a.dhead;
    const int64_t ld = qdim + 2 * kvdim;
    // Padder::add_padding on the fused projection (src/layers/attention.cc:390-403): one gather
    // into the padded layout the attention kernels index (the feed-forward's inner buffer is free here)
    const float* qkv_in = st.qkv;
    if (unpadded) {
      begin_kernel("padder_add", double(padded_rows) * ld * 8, 0, padded_rows, 0, ld);
      float* padded = st.pad_qkv ? st.pad_qkv : st.inner;
      launch_gather_rows(st.qkv, st.d_flat_to_padded, padded_rows, ld * sizeof(float), padded, cur_stream());
      end_kernel();
      qkv_in = padded;
    }
    AttnFullParams f;
    f.q = qkv_in;
    f.k = qkv_in + qdim;
    f.v = qkv_in + qdim + kvdim;
    f.q_batch_stride = S * ld;
    f.q_time_stride = ld;
    f.k_batch_stride = S * ld;
    f.k_time_stride = ld;
    f.k_head_stride = a.dhead;
    f.v_batch_stride = S * ld;
    f.v_time_stride = ld;
    f.v_head_stride = a.dhead;
    f.lengths = st.mem_lengths;
    f.batch = B;
    f.tq = S;
    f.tk = S;
    f.heads = a.heads;
    f.heads_kv = a.heads_kv;
    f.dhead = a.dhead;
    f.causal = false;
    f.allow_inexact = half_compute();
    f.half_kind = half_kind();
    f.scale = a.queries_scale;
    f.out = st.ctx;
    begin_kernel("attention_encoder", 0, 0, padded_rows, a.heads, a.dhead);
    launch_attention_full(f, cur_stream());
    end_kernel();
    const float* ctx_in = st.ctx;
    if (unpadded) {  // Padder::remove_padding on the context (attention.cc:590-596); st.qkv is free
      begin_kernel("padder_remove", double(rows) * qdim * 8, 0, rows, 0, qdim);
      launch_gather_rows(st.ctx, st.d_padded_to_flat, rows, qdim * sizeof(float), st.qkv, cur_stream());
      end_kernel();
      ctx_in = st.qkv;
    }
    tap("ctx", ctx_in, 0, rows, a.heads * a.dhead);
    dense(a.lin[1], ctx_in, nullptr, rows, -1, a.ln.present ? x : nullptr, y, &st);
    if (a.ln.present && !enc_.pre_norm) norm(a.ln, y, rows, d, y);
    std::swap(x, y);

    // --- feed forward (FeedForwardNetwork::operator(), src/layers/transformer.cc:21-51)
    tap_scope("enc.l" + std::to_string(enc_layer_index) + ".ffn");
    run_ffn(layer.ffn, enc_.pre_norm, rows, d, x, y, st);
    ++enc_layer_index;
  }
  if (enc_.output_norm.present) norm(enc_.output_norm, x, rows, d, x);
  if (unpadded) {  // Padder::add_padding on the output (src/layers/transformer.cc:469-470)
    begin_kernel("padder_add", double(padded_rows) * d * 8, 0, padded_rows, 0, d);
    launch_gather_rows(x, st.d_flat_to_padded, padded_rows, d * sizeof(float), y, cur_stream());
    end_kernel();
    std::swap(x, y);
  }
  st.enc_out = x;
  tap_scope("enc.out");
  tap("x", x, 0, padded_rows, d);

  // Cross-attention memory projections: linear_1(memory) -> [B*S, 2*kv_dim], kept in the GEMM
  // output layout (the attention kernel takes strides; no split_heads transpose).
  bool quantized_once = false;
  tap_scope("enc.kv");
  for (size_t l = 0; l < dec_.layers.size(); ++l) {
    const AttnW& a = dec_.layers[l].cross;
    QuantIn mq;
    if (a.lin[1].quantized) {
      if (!quantized_once) {  // the same memory feeds every layer: quantise it once
        quant(x, padded_rows, d, st.qbuf, st.qscale);
        quantized_once = true;
      }
      mq.q = st.qbuf;
      mq.s = st.qscale;
    }
    dense(a.lin[1], x, mq.q ? &mq : nullptr, padded_rows, -1, nullptr, st.cross_kv[l], &st);
    if (!st.cross_kv16.empty()) {
      const int type = compute_type_ == ComputeType::kBFloat16 ? kBF16 : kF16;  // DataType ids
      begin_kernel("convert", double(padded_rows) * a.lin[1].n * 6, 0, padded_rows, a.lin[1].n, 0);
      launch_convert(st.cross_kv[l], kF32, st.cross_kv16[l], type, padded_rows * a.lin[1].n, cur_stream());
      end_kernel();
    }
  }
  check_last_launch("encode");
}

// FeedForwardNetwork::operator() (src/layers/transformer.cc:21-51): pre-norm -> Dense+act
// [-> x Dense-noact (GLU)] -> Dense + residual -> post-norm. x is the sub-layer input, the result
// lands in y and the pointers are swapped so that x is the output on return.
void Engine::run_ffn(const FfnW& ffn, bool pre_norm, int64_t rows, int64_t d, float*& x, float*& y,
                     DecodeState& st, float* amax_buf, bool qin_ready) {
  QuantIn fq;
  const float* ffn_in = x;
  // (layer variants: the caller adds what has to be added)
  const float* resid = (ffn.ln.present && !ffn.no_residual) ? x : nullptr;
  if (qin_ready) {
    // the previous block's row-owning output projection already normalised and quantised x
    fq.q = st.qbuf;
    fq.s = st.qscale;
  } else if (ffn.ln.present && pre_norm && !ffn.l0_noact.present) {
    // The inner activation's Quantize has no launch of its own when the producer can leave the row
    // absmax and the consumer quantises while it loads (6 launches less per Transformer-base step)
    const bool qol = amax_buf != nullptr && !tp_ && !taps_ && gemm_s8_qol_enabled() &&
                     ffn.l1.quantized && !ffn.l1.row_parallel &&
                     gemm_s8_qol_applicable(rows, ffn.l1.n, ffn.l1.k);
    if (dense_norm(ffn.ln, ffn.l0, x, rows, ffn.activation, nullptr, st.inner,
                   qol ? amax_buf : nullptr)) {
      if (qol) dense_qol(ffn.l1, st.inner, amax_buf, rows, -1, resid, y);
      else dense(ffn.l1, st.inner, nullptr, rows, -1, resid, y, &st);
      std::swap(x, y);
      return;
    }
  }
  // Gated pair at <= 16 rows of a model with streamed weights: the RMSNorm + Quantize runs as the
  // prologue of the pair's launch (gemm_s8_stream.hip AMODE 2)
  const bool glu_streamable =
      !qin_ready && ffn.ln.present && pre_norm && ffn.l0_noact.present && ffn.l0.quantized &&
      ffn.l0_noact.quantized && !tp_ && !taps_ && gemm_s8_norm_fusion_enabled() &&
      ffn.l0.wq_stream && ffn.l0.wq_stream_dual && ffn.l0_noact.wq_stream &&
      ffn.l0_noact.wq_stream_dual && ffn.l0_noact.n == ffn.l0.n;
  // (round 6: by sixteen leader workgroups of the pair's launch, CT2X_STREAM_PROLOGUE bit 2)
  const bool glu_lead = glu_streamable && (gemm_s8_stream_prologue_mask() & 4) && pf_state_ == &st &&
                        st.lead_flag != nullptr && rows <= 16 && st.rows <= 16 &&
                        gemm_s8_stream_lead_norm_applicable(rows, ffn.l0.n, ffn.l0.k, true, ffn.ln.beta != nullptr);
  const bool glu_norm =
      glu_lead || (glu_streamable && (gemm_s8_stream_prologue_mask() & 1) &&
                   gemm_s8_stream_norm_applicable(rows, ffn.l0.n, ffn.l0.k, true, ffn.ln.beta != nullptr));
  if (qin_ready || glu_norm) {
  } else if (ffn.ln.present && pre_norm) {
    if (ffn.l0.quantized) {
      prefetch_weight(st, ffn.l0, 2);  // FFN input weight(s), next to the norm launch
      if (ffn.l0_noact.present) prefetch_weight(st, ffn.l0_noact, 2);
      norm_quant(ffn.ln, x, rows, d, nullptr, st.qbuf, st.qscale);
      fq.q = st.qbuf;
      fq.s = st.qscale;
    } else {
      norm(ffn.ln, x, rows, d, st.ctx);
      ffn_in = st.ctx;
    }
  } else if (ffn.l0.quantized && ffn.l0_noact.present) {
    quant(x, rows, d, st.qbuf, st.qscale);
    fq.q = st.qbuf;
    fq.s = st.qscale;
  }
  if (ffn.l0_noact.present && (fq.q || glu_norm) && ffn.l0.quantized && ffn.l0_noact.quantized &&
      rows <= 32 && ffn.l0.k % 16 == 0 && ffn.l0_noact.n == ffn.l0.n) {
    // Decode rows: gate and up projection in ONE launch, y = act(gate) * up (no fp32
    // intermediates, no ops::Mul launch).
    GemmS8Epilogue ep;
    ep.mode = 1;
    ep.a_scale = fq.s;
    if (glu_norm) {
      ep.norm_x = x;
      ep.norm_gamma = ffn.ln.gamma;
      ep.norm_beta = ffn.ln.beta;
      ep.norm_eps = ffn.ln.eps;
      ep.norm_rms_residual = ffn.ln.use_residual;
      ep.quant_mode = quant_mode_;
      if (glu_lead) lead_fill(ep, rows);
    }
    ep.w_scale = ffn.l0.wscale;
    ep.bias = ffn.l0.bias;
    ep.act = ffn.activation;
    ep.b2 = ffn.l0_noact.wq;
    ep.w_scale2 = ffn.l0_noact.wscale;
    ep.bias2 = ffn.l0_noact.bias;
    ep.y = st.inner;
    ep.ldy = ffn.l0.n;
    // The gated pair also leaves the row absmax of its output: the Quantize that feeds linear_1
    // becomes an element-wise launch over (rows x column chunks) instead of one workgroup per row.
    const bool split_quant = amax_buf != nullptr && !tp_ && ffn.l1.quantized && !ffn.l1.row_parallel;
    if (split_quant) ep.row_amax_out = amax_buf;
    const double M = double(rows), N = double(ffn.l0.n), K = double(ffn.l0.k);
    begin_kernel(glu_norm ? "norm_gemm_s8_glu" : "gemm_s8_glu",
                 (glu_norm ? 4 : 1) * M * K + 2 * N * K + 8 * N + 4 * M + 4 * M * N, 4 * M * N * K, rows,
                 ffn.l0.n, ffn.l0.k);
    const int8_t* w0 = ffn.l0.wq;
    if (ffn.l0.wq_stream && ffn.l0.wq_stream_dual && ffn.l0_noact.wq_stream &&
        ffn.l0_noact.wq_stream_dual && gemm_s8_stream_applicable(rows, ffn.l0.n, ffn.l0.k, true)) {
      w0 = ffn.l0.wq_stream;
      ep.b2 = ffn.l0_noact.wq_stream;
      ep.packed_b = true;
    }
    launch_gemm_s8(fq.q, w0, rows, ffn.l0.n, ffn.l0.k, ep, cur_stream());
    end_kernel();
    if (split_quant && stream_lead_quant_applicable(ffn.l1, rows)) {
      // ... and the element-wise Quantize is the work of the output projection's leader workgroups
      dense_qol(ffn.l1, st.inner, amax_buf, rows, -1, resid, y);
      if (ffn.ln.present && !pre_norm) norm(ffn.ln, y, rows, d, y);
      std::swap(x, y);
      return;
    }
    if (split_quant) {
      prefetch_weight(st, ffn.l1, 3);  // FFN output weight, next to the Quantize launch
      begin_kernel("quantize_cols", double(rows) * ffn.l0.n * 5, 0, rows, 0, ffn.l0.n);
      launch_quantize_cols(st.inner, amax_buf, rows, ffn.l0.n, quant_mode_, st.qbuf, st.qscale,
                           cur_stream(), st.pf_pending.wgs > 0 ? &st.pf_pending : nullptr);
      st.pf_pending = PrefetchArgs();
      end_kernel();
      QuantIn iq;
      iq.q = st.qbuf;
      iq.s = st.qscale;
      dense(ffn.l1, nullptr, &iq, rows, -1, resid, y, &st);
      if (ffn.ln.present && !pre_norm) norm(ffn.ln, y, rows, d, y);
      std::swap(x, y);
      return;
    }
  } else {
  dense(ffn.l0, ffn_in, fq.q ? &fq : nullptr, rows, ffn.activation, nullptr, st.inner, &st);
  if (ffn.l0_noact.present) {
    dense(ffn.l0_noact, ffn_in, fq.q ? &fq : nullptr, rows, -1, nullptr, st.inner2, &st);
    begin_kernel("mul");
    launch_mul(st.inner2, st.inner, rows * ffn.l0.n, st.inner, cur_stream());
    end_kernel();
  }
  }
  dense(ffn.l1, st.inner, nullptr, rows, -1, resid, y, &st);
  if (ffn.ln.present && !pre_norm) norm(ffn.ln, y, rows, d, y);
  std::swap(x, y);
}

// The part of a layer variant (LayerW::variant) that follows its self-attention block: on entry
// x = the layer input, y = Attention(norm(x)) without residual. On return x = the layer output.
void Engine::finish_variant_layer(const LayerW& layer, int64_t rows, int64_t d, float*& x, float*& y,
                                  DecodeState& st) {
  auto add = [&](const float* a, const float* b, float* out) {
    begin_kernel("add", double(rows) * d * 12, 0, rows, 0, d);
    launch_add(a, b, rows * d, out, cur_stream());
    end_kernel();
  };
  if (layer.variant == 2) {
    // post_attention_layer_norm(context) + input (transformer.cc:239-240)
    norm(layer.post_attn_ln, y, rows, d, y);
    add(y, x, y);
    std::swap(x, y);  // x = h, y free
    run_ffn(layer.ffn, true, rows, d, x, y, st);  // x = FFN(pre_ff_norm(h)), y = h
    norm(layer.post_ff_ln, x, rows, d, x);
    add(x, y, x);  // ops::Add()(output, context, output) (:278)
    if (layer.layer_scalar != 1.f) {
      begin_kernel("mul_scalar", double(rows) * d * 8, 0, rows, 0, d);
      launch_binary_scalar(/*mul*/ 2, layer.layer_scalar, x, x, rows * d, /*float32*/ 0, cur_stream());
      end_kernel();
    }
  } else {
    // parallel residual (transformer.cc:287-317): FFN(norm'(input)) + input + attention
    CT2X_CHECK(st.x2 != nullptr, "parallel-residual layer without its third buffer");
    float* in = x;
    float* z = st.x2;
    if (in == st.x2 || y == st.x2) z = (in != st.x0 && y != st.x0) ? st.x0 : st.x1;
    float* fx = in;
    float* fy = z;
    run_ffn(layer.ffn, true, rows, d, fx, fy, st);  // fx = z = FFN(norm'(in)), fy = in
    add(fx, in, fx);  // ops::Add()(output, input, output)
    add(fx, y, fx);   // ops::Add()(output, attn, output)
    x = fx;
    y = in;
  }
}

// Sequence-mode forward of a decoder-only model over the prompt (TransformerDecoder::decode with
// lengths, src/layers/transformer.cc:673-696; DecoderReplica::run_generation prefill,
// src/models/language_model.cc:217-238): fills the self-attention caches for positions
// [0, time) of every row and leaves the step counter at `time`. No logits are produced.
void Engine::prefill(DecodeState& st, const int32_t* h_ids, int64_t time, float* logits_all,
                     const ScoreCapture* capture) {
  CT2X_HIP_CHECK(hipSetDevice(device_));
  tls_stream = st.stream;
  set_active_stream(tls_stream);
  // (with beam > 1 the caller passes the prompt of a sentence once per hypothesis row)
  CT2X_CHECK(time <= st.t_max && time <= st.prefill_max, "prompt longer than the decoding state");
  if (time == 0) return;
  const int64_t B = st.rows, d = dec_.d_model, rows = B * time;
  int32_t* d_prompt = nullptr;
  {
    void* p = device_malloc(sizeof(int32_t) * rows);
    st.owned.push_back(p);
    d_prompt = static_cast<int32_t*>(p);
  }
  CT2X_HIP_CHECK(hipMemcpyAsync(d_prompt, h_ids, sizeof(int32_t) * rows, hipMemcpyHostToDevice,
                                cur_stream()));
  const float* pos = dec_.position_table;
  if (dec_.sinusoidal_positions) pos = sinusoidal_table(st.t_max, d);
  float* x = st.x0;
  float* y = st.x1;
  begin_kernel("embed");
  launch_embed(dec_.emb.table, dec_.emb.is_int8, dec_.emb.scale, d_prompt, rows, d,
               dec_.embeddings_scale, dec_.scale_embeddings, pos, time, 0, nullptr, x,
               cur_stream());
  end_kernel();
  if (dec_.layernorm_embedding.present) norm(dec_.layernorm_embedding, x, rows, d, x);

  for (size_t l = 0; l < dec_.layers.size(); ++l) {
    const LayerW& layer = dec_.layers[l];
    const AttnW& a = layer.self;
    QuantIn qin;
    const float* attn_in = x;
    if (a.ln.present && dec_.pre_norm) {
      if (a.lin[0].quantized) {
        norm_quant(a.ln, x, rows, d, nullptr, st.qbuf, st.qscale);
        qin.q = st.qbuf;
        qin.s = st.qscale;
      } else {
        norm(a.ln, x, rows, d, st.ctx);
        attn_in = st.ctx;
      }
    }
    dense(a.lin[0], attn_in, qin.q ? &qin : nullptr, rows, -1, nullptr, st.qkv, &st);
    const float *rs = nullptr, *rc = nullptr;
    int64_t rot_dim = 0;
    if (a.rotary) {
      rotary_tables(a, st.t_max + 1, &rs, &rc);
      rot_dim = a.rotary_dim == 0 ? a.dhead : a.rotary_dim;
    }
    float* q_rot = st.inner;  // [rows, heads*dhead]; dead before the FFN reuses the buffer
    begin_kernel("qkv_to_cache");
    launch_qkv_to_cache(st.qkv, B, time, a.heads, a.heads_kv, a.dhead, st.t_max, 0, rs, rc, rot_dim,
                        a.rotary_interleave, q_rot, st.self_k[st.ancestry ? 0 : st.cur][l],
                        st.self_v[st.ancestry ? 0 : st.cur][l],
                        cur_stream());
    end_kernel();
    AttnFullParams f;
    f.q = q_rot;
    f.q_batch_stride = time * a.heads * a.dhead;
    f.q_time_stride = a.heads * a.dhead;
    f.k = st.self_k[st.ancestry ? 0 : st.cur][l];
    f.v = st.self_v[st.ancestry ? 0 : st.cur][l];
    f.k_batch_stride = f.v_batch_stride = a.heads_kv * st.t_max * a.dhead;
    f.k_head_stride = f.v_head_stride = st.t_max * a.dhead;
    f.k_time_stride = f.v_time_stride = a.dhead;
    f.lengths = nullptr;
    f.batch = B;
    f.tq = time;
    f.tk = time;
    f.heads = a.heads;
    f.heads_kv = a.heads_kv;
    f.dhead = a.dhead;
    f.causal = true;
    f.allow_inexact = half_compute();
    f.half_kind = half_kind();
    f.q_offset = 0;
    f.scale = a.queries_scale;
    f.out = st.ctx;
    begin_kernel("attention_prefill", 0, 0, rows, a.heads, a.dhead);
    launch_attention_full(f, cur_stream());
    end_kernel();
    dense(a.lin[1], st.ctx, nullptr, rows, -1, (a.ln.present && !a.no_residual) ? x : nullptr, y, &st);
    if (a.ln.present && !dec_.pre_norm) norm(a.ln, y, rows, d, y);
    if (layer.variant != 0) {
      finish_variant_layer(layer, rows, d, x, y, st);
      continue;
    }
    std::swap(x, y);
    if (layer.has_cross) {
      // encoder-decoder attention in sequence mode (the target prefix of decode_alternatives,
      // src/decoding.cc:1148-1160; WhisperDecoder::forward_prompt): queries of every prefix
      // position against the memory keys projected by encode()
      const AttnW& c = layer.cross;
      QuantIn cq;
      const float* cin = x;
      if (c.ln.present && dec_.pre_norm) {
        if (c.lin[0].quantized) {
          norm_quant(c.ln, x, rows, d, nullptr, st.qbuf, st.qscale);
          cq.q = st.qbuf;
          cq.s = st.qscale;
        } else {
          norm(c.ln, x, rows, d, st.ctx);
          cin = st.ctx;
        }
      }
      dense(c.lin[0], cin, cq.q ? &cq : nullptr, rows, -1, nullptr, st.qkv, &st);
      const int64_t qdim = c.heads * c.dhead, kvdim = c.heads_kv * c.dhead;
      if (capture && !capture->heads[l].empty()) {
        // the scores before their softmax (return_normalized_attention() == false,
        // include/ctranslate2/layers/whisper.h), selected heads only (transformer.cc:811-814)
        CT2X_CHECK(st.beam == 1 && c.heads_kv == c.heads, "attention scores: beam 1, no GQA");
        int64_t offset = 0;
        for (size_t k = 0; k < l; ++k) offset += static_cast<int64_t>(capture->heads[k].size());
        const int64_t n_here = static_cast<int64_t>(capture->heads[l].size());
        begin_kernel("attention_scores", 0, 0, rows, n_here, c.dhead);
        for (int64_t b = 0; b < st.batch; ++b)  // [batch, n_sel, time, s_max]: this layer's slice
          launch_attention_scores(st.qkv + b * time * qdim, 0, qdim,
                                  st.cross_kv[l] + b * st.s_max * 2 * kvdim, 0, 2 * kvdim,
                                  capture->d_heads[l], 1, n_here, time, st.s_max, c.dhead,
                                  c.queries_scale,
                                  capture->scores + (b * capture->n_sel + offset) * time * st.s_max,
                                  cur_stream());
        end_kernel();
      }
      for (int64_t j = 0; j < st.beam; ++j) {  // hypothesis row j of every sentence: the same memory
        AttnFullParams f;
        f.q = st.qkv + j * time * qdim;
        f.q_batch_stride = st.beam * time * qdim;
        f.q_time_stride = qdim;
        f.k = st.cross_kv[l];
        f.v = st.cross_kv[l] + kvdim;
        f.k_batch_stride = f.v_batch_stride = st.s_max * 2 * kvdim;
        f.k_time_stride = f.v_time_stride = 2 * kvdim;
        f.k_head_stride = f.v_head_stride = c.dhead;
        f.lengths = st.mem_lengths;
        f.batch = st.batch;
        f.tq = time;
        f.tk = st.s_max;
        f.heads = c.heads;
        f.heads_kv = c.heads_kv;
        f.dhead = c.dhead;
        f.causal = false;
        f.allow_inexact = half_compute();
        f.half_kind = half_kind();
        f.scale = c.queries_scale;
        f.out = st.ctx + j * time * qdim;
        f.out_batch_stride = st.beam * time * qdim;
        f.out_time_stride = qdim;
        begin_kernel("attention_cross_prefill", 0, 0, st.batch * time, c.heads, c.dhead);
        launch_attention_full(f, cur_stream());
        end_kernel();
      }
      dense(c.lin[2], st.ctx, nullptr, rows, -1, c.ln.present ? x : nullptr, y, &st);
      if (c.ln.present && !dec_.pre_norm) norm(c.ln, y, rows, d, y);
      std::swap(x, y);
    }
    run_ffn(layer.ffn, dec_.pre_norm, rows, d, x, y, st);
  }
  if (logits_all) {
    // every position's logits (DecoderReplica::forward / score_sequences: the decoder in sequence
    // mode with a logits output, src/models/language_model.cc:279-300, src/scoring.cc:9-60)
    const float* h = x;
    if (dec_.output_norm.present) {
      norm(dec_.output_norm, x, rows, d, y);
      h = y;
    }
    dense(dec_.projection, h, nullptr, rows, -1, nullptr, logits_all, &st);
  }
  const int32_t step = static_cast<int32_t>(time);
  CT2X_HIP_CHECK(hipMemcpyAsync(st.d_step, &step, sizeof(int32_t), hipMemcpyHostToDevice,
                                cur_stream()));
  reset_lead_flags(st);
  CT2X_HIP_CHECK(hipStreamSynchronize(cur_stream()));
  check_last_launch("prefill");
}

}  // namespace ct2x
