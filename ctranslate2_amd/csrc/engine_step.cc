// One decoding step of the engine (engine.h) and everything the decoding loops drive between
// steps: histories, prefixes, logits processors, attention read-back, Whisper's align pass.
#include "engine.h"
#include "host_math.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>

#include "engine_internal.h"

#include <type_traits>
#include "runtime.h"

namespace ct2x {

// One capture at a time per process: concurrent captures from several worker threads fail on
// ROCm 7.2 ("operation failed due to a previous error during capture"); replays are free.
static std::mutex& graph_capture_mutex() {
  static std::mutex m;
  return m;
}


// TransformerDecoder::decode for one step (src/layers/transformer.cc:621-871) with
// TransformerDecoderLayer::operator() (standard pre/post-norm arm, :334-365).
void Engine::launch_step(DecodeState& st) {
  const int64_t rows = st.rows, d = dec_.d_model;
  const float* pos = dec_.position_table;
  if (dec_.sinusoidal_positions) pos = sinusoidal_table(st.t_max, d);
  float* x = st.x0;
  float* y = st.x1;
  // producer-side absmax slots (2 per layer) are max-combined into: zero them for this step
  // (only models whose layers take that path pay for the memset node: gated-FFN decoders)
  st.pf_active = true;
  st.lead_seq = 0;  // (tags of the leader launches: their index inside this step)
  st.pf_pending = PrefetchArgs();
  pf_state_ = &st;
  // (a launch that throws must not leave the worker thread pointing at a dead session)
  struct PfGuard {
    DecodeState& s;
    ~PfGuard() {
      s.pf_active = false;
      s.pf_pending = PrefetchArgs();
      Engine::pf_state_ = nullptr;
    }
  } pf_guard{st};
  const bool split_quant = !tp_ && rows <= 32 && !dec_.layers.empty() &&
                           dec_.layers[0].ffn.l0_noact.present;
  // ... and so do the feed-forward blocks whose inner Quantize is folded into the consuming GEMM
  // (run_ffn / dense_qol). The slots are cleared by the step's first launch, not by a memset node.
  const bool ffn_amax = !tp_ && !taps_ && rows <= 32 && !dec_.layers.empty() && gemm_s8_qol_enabled();
  const bool zero_amax = split_quant || ffn_amax;
  const bool cluster = cluster_applicable(st);
  bool cluster_out_quant = false;
  bool cluster_proj = false;  // the vocabulary projection + greedy arg-max ran as the launch's tail
  if (cluster) {
    // embedding + every decoder layer: ONE launch (decoder_cluster.hip)
    ClusterStackArgs ca;
    ca.n_layers = static_cast<int>(dec_.layers.size());
    for (int l = 0; l < ca.n_layers; ++l) {
      ca.layers[l] = cluster_layers_[l];
      ca.layers[l].self_k = st.self_k[0][l];
      ca.layers[l].self_v = st.self_v[0][l];
      ca.layers[l].cross_kv = st.cross_kv[l];
    }
    ca.rows = static_cast<int>(rows);
    if (st.beam_mode) {
      ca.beam = static_cast<int>(st.beam);
      ca.rows_per_cluster = ca.beam;
      ca.ancestry = st.anc[st.cur];
      ca.one_row_step = st.prefix_first_step;  // (the first step of the CPU beam search: one row per entry)
    } else if (rows > 32) {
      ca.rows_per_cluster = static_cast<int>((rows + 31) / 32);  // 2 - 4 independent rows per cluster
    }
    ca.t_max = static_cast<int>(st.t_max);
    ca.s_max = static_cast<int>(st.s_max);
    ca.t_pad = st.cl_t_pad;
    ca.quant_mode = quant_mode_;
    ca.act = dec_.layers[0].ffn.activation;
    ca.emb_table = dec_.emb.table;
    ca.emb_int8 = dec_.emb.is_int8 ? 1 : 0;
    ca.emb_scale = dec_.emb.scale;
    ca.emb_apply_mul = dec_.scale_embeddings ? 1 : 0;
    ca.emb_mul = dec_.embeddings_scale;
    ca.pos_table = pos;
    ca.ids = st.d_ids;
    ca.step_ptr = st.d_step;
    ca.mem_lengths = st.mem_lengths;
    ca.x_out = x;
    ca.xch = st.cl_xch;
    ca.counters = st.cl_counters;
    ca.error = st.cl_error;
    ca.stamps = st.cl_stamps;
    // the output norm + Quantize ride in the launch's tail where the projection would otherwise
    // take them from their own launch (dense_norm's prologue form stops below 8192 columns)
    cluster_out_quant = dec_.output_norm.present && dec_.projection.quantized && !dec_.projection.awq &&
                        !gemm_s8_norm_applicable(rows, dec_.projection.n, dec_.projection.k);
    if (cluster_out_quant) {
      ca.out_norm.gamma = dec_.output_norm.gamma;
      ca.out_norm.beta = dec_.output_norm.beta;
      ca.out_norm.eps = dec_.output_norm.eps;
      ca.out_norm.rms_use_residual = dec_.output_norm.use_residual ? 1 : 0;
      ca.q_out = st.qbuf;
      ca.qscale_out = st.qscale;
      // ... and so do the projection itself and the greedy sampler where the fused arg-max form
      // applies (fused_argmax_applicable) and the launch's workgroups can hold their share of the
      // weight rows: the step is then ONE launch
      const int proj_cols = (!st.beam_mode && rows <= 32 && st.cl_proj_q != nullptr && fused_argmax_applicable(st) &&
                             (reinterpret_cast<uintptr_t>(dec_.projection.wq) & 15) == 0)
                                ? decoder_cluster_proj_cols(rows, dec_.projection.n, dec_.projection.k)
                                : 0;
      if (proj_cols > 0) {
        cluster_proj = true;
        const DenseW& w = dec_.projection;
        ca.q_out = nullptr;
        ca.qscale_out = nullptr;
        ca.proj.w = w.wq;
        ca.proj.wscale = w.wscale;
        ca.proj.bias = w.bias;
        ca.proj.vocab = static_cast<int>(w.n);
        ca.proj.cols_per_wg = proj_cols;
        ca.proj.rows_q = st.cl_proj_q;
        ca.proj.seq = st.cl_proj_seq;
        static const bool board = [] {  // CT2X_CLUSTER_PROJ_BOARD=0: the slots / ticket form of the finish (A/B runs)
          const char* e = std::getenv("CT2X_CLUSTER_PROJ_BOARD");
          return e == nullptr || e[0] != '0';
        }();
        ca.proj.board = (board && w.n <= 65536) ? st.cl_proj_board : nullptr;
        ca.proj.gf.slots = st.gf_slots;
        ca.proj.gf.ticket = st.gf_ticket;
        ca.proj.gf.step_ptr = st.d_step;
        ca.proj.gf.disable_id = st.g_eos;
        ca.proj.gf.min_length = st.g_min_length;
        ca.proj.gf.prefix_ids = st.d_prefix;
        ca.proj.gf.prefix_lens = st.d_prefix_len;
        ca.proj.gf.prefix_stride = st.prefix_stride;
        ca.proj.gf.prefix_first_step = st.prefix_first_step;
        ca.proj.gf.prefix_extends_min = st.prefix_extends_min ? 1 : 0;
        ca.proj.gf.next_ids = st.d_ids;
        ca.proj.gf.hist_ids = st.hist_ids;
        ca.proj.gf.hist_scores = st.hist_scores;
        ca.proj.gf.advance_step_ptr = st.d_step;
        ca.proj.gf.host_hist_ids = st.host_hist_ids;
      }
    }
    {
      double wbytes = 0;
      for (const LayerW& layer : dec_.layers)
        wbytes += double(layer.self.lin[0].n + layer.self.lin[1].n + layer.cross.lin[0].n + layer.cross.lin[2].n) * d +
                  2.0 * double(layer.ffn.l0.n) * d;
      if (cluster_proj) wbytes += double(dec_.projection.n) * (d + 4 + (dec_.projection.bias ? 4 : 0));
      begin_kernel("decoder_stack", wbytes + double(rows) * d * 8, 2.0 * rows * wbytes, rows, d, d);
    }
    launch_decoder_cluster(ca, cur_stream());
    end_kernel();
  } else {
  begin_kernel("embed");
  launch_embed(dec_.emb.table, dec_.emb.is_int8, dec_.emb.scale, st.d_ids, rows, d,
               dec_.embeddings_scale, dec_.scale_embeddings, pos, 1, 0, st.d_step, x, cur_stream(),
               zero_amax ? st.row_amax : nullptr,
               zero_amax ? 2 * static_cast<int64_t>(dec_.layers.size()) * rows : 0);
  end_kernel();
  tap_scope("dec");
  tap("emb", x, 0, rows, d);
  if (dec_.layernorm_embedding.present) norm(dec_.layernorm_embedding, x, rows, d, x);

  // set when the previous block's row-owning output projection already produced LayerNorm(x) +
  // Quantize for the block that follows (st.qbuf / st.qscale)
  bool qin_ready = false;
  for (size_t l = 0; l < dec_.layers.size(); ++l) {
    const LayerW& layer = dec_.layers[l];
    // ---- masked self-attention -----------------------------------------------------------
    {
      const AttnW& a = layer.self;
      tap_scope("dec.l" + std::to_string(l) + ".self");
      QuantIn qin;
      const float* attn_in = x;
      // the sub-layer's input is added to its output unless a layer variant does its own sums
      const float* resid = (a.ln.present && !a.no_residual) ? x : nullptr;
      const bool fused_in = !qin_ready && dec_.pre_norm &&
                            dense_norm(a.ln, a.lin[0], x, rows, -1, nullptr, st.qkv);
      if (qin_ready) {
        qin.q = st.qbuf;
        qin.s = st.qscale;
        qin_ready = false;
      } else if (!fused_in && a.ln.present && dec_.pre_norm) {
        if (a.lin[0].quantized) {
          prefetch_weight(st, a.lin[0], 0);  // QKV weight, next to the norm launch
          norm_quant(a.ln, x, rows, d, nullptr, st.qbuf, st.qscale);
          qin.q = st.qbuf;
          qin.s = st.qscale;
        } else {
          norm(a.ln, x, rows, d, st.ctx);
          attn_in = st.ctx;
        }
      }
      if (!fused_in)
        dense(a.lin[0], attn_in, qin.q ? &qin : nullptr, rows, -1, nullptr, st.qkv, &st);
      const int64_t qdim = a.heads * a.dhead, kvdim = a.heads_kv * a.dhead;
      const int64_t ld = qdim + 2 * kvdim;
      AttnDecodeParams p;
      p.q = st.qkv;
      p.q_row_stride = ld;
      p.k_new = st.qkv + qdim;
      p.v_new = st.qkv + qdim + kvdim;
      p.new_row_stride = ld;
      p.k_cache = st.self_k[st.ancestry ? 0 : st.cur][l];
      p.v_cache = st.self_v[st.ancestry ? 0 : st.cur][l];
      if (st.ancestry) {
        p.ancestry = st.anc[st.cur];
        p.ancestry_stride = rows;
      }
      p.kv_row_stride = a.heads_kv * st.t_max * a.dhead;
      p.kv_head_stride = st.t_max * a.dhead;
      p.kv_time_stride = a.dhead;
      p.step_ptr = st.d_step;
      p.rows = rows;
      p.heads = a.heads;
      p.heads_kv = a.heads_kv;
      p.dhead = a.dhead;
      p.kv_group = 1;
      p.t_max = st.t_max;
      p.scale = a.queries_scale;
      p.allow_inexact = half_compute();
      if (a.rotary) {
        rotary_tables(a, st.t_max + 1, &p.rope_sin, &p.rope_cos);
        p.rot_dim = a.rotary_dim == 0 ? a.dhead : a.rotary_dim;
        p.rope_interleave = a.rotary_interleave;
      }
      p.out = st.ctx;
      if (a.lin[1].quantized && !tp_ && attention_decode_can_fuse_quantize(p)) {
        // attention + ops::Quantize of the context in one launch (kFusedQ)
        begin_kernel("attention_self_decode_q", 0, 0, rows, a.heads, a.dhead);
        launch_attention_decode_quantized(p, quant_mode_, st.qbuf, st.qscale, cur_stream());
        end_kernel();
        if (taps_) {
          launch_attention_decode(p, cur_stream());  // same context in fp32 (idempotent append)
          tap("f", st.ctx, 0, rows, qdim);
          tap("q", st.qbuf, 1, rows, qdim);
          tap("qs", st.qscale, 0, rows, 1);
        }
        QuantIn cq;
        cq.q = st.qbuf;
        cq.s = st.qscale;
        const NormW* next_ln = layer.has_cross ? &layer.cross.ln : &layer.ffn.ln;
        const DenseW* next_dense = layer.has_cross ? &layer.cross.lin[0] : &layer.ffn.l0;
        if (a.ln.present && layer.variant == 0 && (layer.has_cross || !layer.ffn.l0_noact.present) &&
            dense_out_next_norm(a.lin[1], cq, rows, x, y, next_ln, next_dense, st))
          qin_ready = true;
        else
          dense(a.lin[1], nullptr, &cq, rows, -1, resid, y, &st);
      } else if (split_quant && a.lin[1].quantized && !a.lin[1].row_parallel &&
                 attention_decode_emits_row_amax(p)) {
        // long caches / few rows: one workgroup per (row, head) + producer-side absmax, then an
        // element-wise Quantize over (rows x column chunks) workgroups
        float* amax = st.row_amax + (2 * l) * rows;
        int amax_parts = 1;
        if (st.attn_amax_parts != nullptr && p.dhead == 128) {
          // (the d_head 128 exact kernel: one plain store per (row, kv head) instead of atomics)
          amax = st.attn_amax_parts;
          amax_parts = static_cast<int>(a.heads_kv);
          p.row_amax_parts = amax_parts;
        }
        p.row_amax_out = amax;
        begin_kernel("attention_self_decode", 0, 0, rows, a.heads, a.dhead);
        launch_attention_decode(p, cur_stream());
        end_kernel();
        if (stream_qol_applicable(a.lin[1], rows)) {
          // ... and the element-wise Quantize is the prologue of the streaming out projection
          dense_qol(a.lin[1], st.ctx, amax, rows, -1, resid, y, amax_parts);
        } else {
        prefetch_weight(st, a.lin[1], 1);  // attention output weight, next to the Quantize launch
        begin_kernel("quantize_cols", double(rows) * qdim * 5, 0, rows, 0, qdim);
        launch_quantize_cols(st.ctx, amax, rows, qdim, quant_mode_, st.qbuf, st.qscale, cur_stream(),
                             st.pf_pending.wgs > 0 ? &st.pf_pending : nullptr, amax_parts);
        st.pf_pending = PrefetchArgs();
        end_kernel();
        QuantIn cq;
        cq.q = st.qbuf;
        cq.s = st.qscale;
        dense(a.lin[1], nullptr, &cq, rows, -1, resid, y, &st);
        }
      } else {
        begin_kernel("attention_self_decode", 0, 0, rows, a.heads, a.dhead);
        launch_attention_decode(p, cur_stream());
        end_kernel();
        dense(a.lin[1], st.ctx, nullptr, rows, -1, resid, y, &st);
      }
      if (a.ln.present && !dec_.pre_norm) norm(a.ln, y, rows, d, y);
      if (layer.variant != 0) {
        // parallel residual / pre + post norms: the rest of the layer (no cross attention)
        tap_scope("dec.l" + std::to_string(l) + ".ffn");
        finish_variant_layer(layer, rows, d, x, y, st);
        continue;
      }
      std::swap(x, y);
    }
    // ---- encoder-decoder attention -------------------------------------------------------
    if (layer.has_cross) {
      const AttnW& a = layer.cross;
      tap_scope("dec.l" + std::to_string(l) + ".cross");
      QuantIn qin;
      const float* attn_in = x;
      const bool fused_in = !qin_ready && dec_.pre_norm &&
                            dense_norm(a.ln, a.lin[0], x, rows, -1, nullptr, st.qkv);
      if (qin_ready) {
        qin.q = st.qbuf;
        qin.s = st.qscale;
        qin_ready = false;
      } else if (!fused_in && a.ln.present && dec_.pre_norm) {
        if (a.lin[0].quantized) {
          norm_quant(a.ln, x, rows, d, nullptr, st.qbuf, st.qscale);
          qin.q = st.qbuf;
          qin.s = st.qscale;
        } else {
          norm(a.ln, x, rows, d, st.ctx);
          attn_in = st.ctx;
        }
      }
      if (!fused_in)
        dense(a.lin[0], attn_in, qin.q ? &qin : nullptr, rows, -1, nullptr, st.qkv, &st);
      const int64_t kvdim = a.heads_kv * a.dhead;
      if (st.att_hist && static_cast<int64_t>(l) == alignment_layer_) {
        begin_kernel("attention_probs");
        launch_attention_probs(st.qkv, a.heads * a.dhead, st.cross_kv[l], st.s_max * 2 * kvdim,
                               2 * kvdim, st.mem_lengths, rows, static_cast<int32_t>(st.beam),
                               static_cast<int32_t>(a.heads), static_cast<int32_t>(a.heads_kv),
                               static_cast<int32_t>(a.dhead), static_cast<int32_t>(alignment_heads_),
                               static_cast<int32_t>(st.s_max), a.queries_scale, st.d_step,
                               st.att_hist, cur_stream());
        end_kernel();
      }
      AttnDecodeParams p;
      p.q = st.qkv;
      p.q_row_stride = a.heads * a.dhead;
      p.k_cache = st.cross_kv[l];
      p.v_cache = st.cross_kv[l] + kvdim;
      p.kv_row_stride = st.s_max * 2 * kvdim;
      p.kv_head_stride = a.dhead;
      p.kv_time_stride = 2 * kvdim;
      p.lengths = st.mem_lengths;
      p.rows = rows;
      p.heads = a.heads;
      p.heads_kv = a.heads_kv;
      p.dhead = a.dhead;
      p.kv_group = st.beam;
      p.t_max = st.s_max;
      p.scale = a.queries_scale;
      p.allow_inexact = half_compute();
      p.out = st.ctx;
      if (st.beam > 1) {  // first beam-search step: the CPU reference has not expanded the batch yet
        p.one_row_step_ptr = st.d_step;
        p.one_row_step = st.prefix_first_step;
      }
      const double attn_bytes =
          double(st.batch) * st.s_max * 2 * kvdim * 4 + double(rows) * a.heads * a.dhead * 8;
      const double attn_ops = double(rows) * a.heads * st.s_max * a.dhead * 4;
      if (a.lin[2].quantized && !tp_ && attention_decode_can_fuse_quantize(p)) {
        begin_kernel("attention_cross_decode_q", attn_bytes, attn_ops, rows, a.heads, a.dhead);
        launch_attention_decode_quantized(p, quant_mode_, st.qbuf, st.qscale, cur_stream());
        end_kernel();
        if (taps_) {
          launch_attention_decode(p, cur_stream());
          tap("f", st.ctx, 0, rows, a.heads * a.dhead);
          tap("q", st.qbuf, 1, rows, a.heads * a.dhead);
          tap("qs", st.qscale, 0, rows, 1);
        }
        QuantIn cq;
        cq.q = st.qbuf;
        cq.s = st.qscale;
        if (a.ln.present && !layer.ffn.l0_noact.present &&
            dense_out_next_norm(a.lin[2], cq, rows, x, y, &layer.ffn.ln, &layer.ffn.l0, st))
          qin_ready = true;
        else
          dense(a.lin[2], nullptr, &cq, rows, -1, a.ln.present ? x : nullptr, y, &st);
      } else if (!st.cross_kv16.empty()) {
        // 16-bit memory K/V, one workgroup per (batch entry, head) for all its hypothesis rows
        CrossAttnHParams cp;
        cp.q = st.qkv;
        cp.q_row_stride = a.heads * a.dhead;
        cp.kv = st.cross_kv16[l];
        cp.bf16 = compute_type_ == ComputeType::kBFloat16;
        cp.lengths = st.mem_lengths;
        cp.batch = st.batch;
        cp.beam = st.beam;
        cp.heads = a.heads;
        cp.heads_kv = a.heads_kv;
        cp.dhead = a.dhead;
        cp.s_max = st.s_max;
        cp.scale = a.queries_scale;
        cp.out = st.ctx;
        cp.out_row_stride = a.heads * a.dhead;
        cp.split_scratch = st.xh_scratch;
        cp.split_tickets = st.xh_tickets;
        begin_kernel("attention_cross_decode_h", attn_bytes / 2, attn_ops, rows, a.heads, a.dhead);
        launch_cross_attention_h(cp, cur_stream());
        end_kernel();
        dense(a.lin[2], st.ctx, nullptr, rows, -1, a.ln.present ? x : nullptr, y, &st);
      } else {
        begin_kernel("attention_cross_decode", attn_bytes, attn_ops, rows, a.heads, a.dhead);
        launch_attention_decode(p, cur_stream());
        end_kernel();
        dense(a.lin[2], st.ctx, nullptr, rows, -1, a.ln.present ? x : nullptr, y, &st);
      }
      if (a.ln.present && !dec_.pre_norm) norm(a.ln, y, rows, d, y);
      std::swap(x, y);
    }
    // ---- feed forward ----------------------------------------------------------------------
    tap_scope("dec.l" + std::to_string(l) + ".ffn");
    run_ffn(layer.ffn, dec_.pre_norm, rows, d, x, y, st,
            (split_quant || ffn_amax) ? st.row_amax + (2 * l + 1) * rows : nullptr, qin_ready);
    qin_ready = false;
  }

  }  // !cluster

  st.pf_active = false;
  st.pf_pending = PrefetchArgs();
  pf_state_ = nullptr;
  // Final norm + vocabulary projection (transformer.cc:842-855).
  tap_scope("dec.out");
  QuantIn pq;
  const float* proj_in = x;
  bool fused_argmax = false;  // the vocabulary projection sampled the step itself (GreedyFuse)
  if (cluster_proj) {
    fused_argmax = true;  // (projected and sampled in the decoder-stack launch's tail)
  } else if (cluster_out_quant || !dense_norm(dec_.output_norm, dec_.projection, x, rows, -1, nullptr, st.logits)) {
    if (cluster_out_quant) {
      pq.q = st.qbuf;
      pq.s = st.qscale;
    } else if (dec_.output_norm.present) {
      if (dec_.projection.quantized) {
        norm_quant(dec_.output_norm, x, rows, d, nullptr, st.qbuf, st.qscale);
        pq.q = st.qbuf;
        pq.s = st.qscale;
      } else {
        norm(dec_.output_norm, x, rows, d, st.ctx);
        proj_in = st.ctx;
      }
    }
    const bool ride = fused_argmax_applicable(st);
    {
      struct ArgmaxScope {  // (cleared on every way out: the pointer must not outlive the launch)
        explicit ArgmaxScope(const DecodeState* s) { Engine::argmax_state_ = s; }
        ~ArgmaxScope() { Engine::argmax_state_ = nullptr; }
      } scope(ride ? &st : nullptr);
      dense(dec_.projection, proj_in, pq.q ? &pq : nullptr, rows, -1, nullptr, st.logits, &st);
    }
    fused_argmax = ride;
  }

  if (st.wh_no_speech_id >= 0 && (st.fuse_greedy || st.fuse_beam)) {
    // GetNoSpeechProbs (apply_first, src/models/whisper.cc:202-231): the untouched logits of step 0
    begin_kernel("no_speech_prob");
    launch_no_speech_prob(st.logits, st.rows, dec_.projection.n, st.wh_no_speech_id, st.d_step,
                          st.prefix_first_step, st.d_no_speech, cur_stream());
    end_kernel();
  }
  if (st.fuse_greedy) {
    if (st.lp_active) {
      // history of this decoding session: hist_ids rows [first_step, *d_step)
      begin_kernel("process_logits");
      launch_process_logits(st.logits, st.rows, dec_.projection.n,
                            st.hist_ids + static_cast<int64_t>(st.prefix_first_step) * st.rows,
                            st.rows, 1, st.d_step, st.prefix_first_step, 0,
                            static_cast<int32_t>(st.t_max), st.lp_penalty, st.lp_ngram,
                            st.d_sup_ids, st.d_sup_off, st.lp_n_sup, cur_stream());
      end_kernel();
    }
    if (st.wh_active) {
      WhisperRulesParams wp = st.wh;
      wp.logits = st.logits;
      wp.rows = st.rows;
      wp.vocab = dec_.projection.n;
      wp.seq = st.hist_ids + static_cast<int64_t>(st.prefix_first_step) * st.rows;
      wp.t_stride = st.rows;
      wp.row_stride = 1;
      wp.step_ptr = st.d_step;
      wp.first_step = st.prefix_first_step;
      wp.prefix_len = st.d_prefix_len;
      wp.prefix_div = 1;
      begin_kernel("whisper_rules");
      launch_whisper_rules(wp, cur_stream());
      end_kernel();
    }
    if (st.logit_hist) {
      begin_kernel("record_logits");
      launch_record_logits(st.logits, st.rows, dec_.projection.n, vocab_size(), st.g_eos,
                           st.g_min_length, st.d_prefix ? st.d_prefix_len : nullptr,
                           st.prefix_extends_min, st.d_step, kLogitsRing, st.logit_hist, cur_stream());
      end_kernel();
    }
    st.hist_on_host = fused_argmax && st.host_hist_ids != nullptr;
    if (fused_argmax) {
      // (sampled by the last workgroup of the projection)
    } else if (st.g_topk == 1) {
      begin_kernel("greedy_sample");
      launch_greedy_sample(st.logits, st.rows, dec_.projection.n, st.g_eos, st.g_min_length,
                           st.g_scores, st.d_step, st.d_ids, st.hist_ids, st.hist_scores,
                           cur_stream(), st.d_prefix, st.d_prefix_len, st.prefix_stride,
                           st.prefix_first_step, st.prefix_extends_min, st.d_step, st.d_done,
                           st.logprobs);
      end_kernel();
    } else {
      begin_kernel("random_sample");
      launch_random_sample(st.logits, st.rows, dec_.projection.n, st.g_eos, st.g_min_length,
                           st.g_scores, st.g_topk, st.g_topp, st.g_temperature, st.g_seed, st.d_step,
                           st.d_ids, st.hist_ids, st.hist_scores, cur_stream(), st.d_prefix,
                           st.d_prefix_len, st.prefix_stride, st.prefix_first_step,
                           st.prefix_extends_min, st.d_step, st.d_done);
      end_kernel();
    }
    // the sampler's last row advanced d_step (decode_ops.hip: row_finished)
  }

  if (st.fuse_beam) {
    // The rest of a beam-search step (src/decoding.cc:507-700) on the device.
    const int64_t V = dec_.projection.n;
    const int par = st.cur;
    if (st.lp_active) {
      begin_kernel("process_logits");
      launch_process_logits(st.logits, st.rows, V, st.alive_seq[par], 1, st.t_max, st.d_step,
                            st.prefix_first_step, 0, static_cast<int32_t>(st.t_max), st.lp_penalty,
                            st.lp_ngram, st.d_sup_ids, st.d_sup_off, st.lp_n_sup, cur_stream());
      end_kernel();
    }
    if (st.wh_active) {
      WhisperRulesParams wp = st.wh;
      wp.logits = st.logits;
      wp.rows = st.rows;
      wp.vocab = V;
      wp.seq = st.alive_seq[par];
      wp.t_stride = 1;
      wp.row_stride = st.t_max;
      wp.step_ptr = st.d_step;
      wp.first_step = st.prefix_first_step;
      wp.prefix_len = st.d_bprefix_len;
      wp.prefix_div = static_cast<int32_t>(st.beam);
      begin_kernel("whisper_rules");
      launch_whisper_rules(wp, cur_stream());
      end_kernel();
    }
    // Round 6: where the LogSoftMax keeps the row in registers its 2 * beam best leave as candidates
    // (no row-wide write), their merge and the step advance ride in beam_select: 6 launches -> 3
    // after the projection. (Biased decoding rewrites rows on another path: the separate launches.)
    const float bias_beta = st.d_bprefix ? st.bias_beta : 0.f;
    const bool fused_topk = st.bt_part_v != nullptr && !(bias_beta > 0.f);
    const int32_t cand = static_cast<int32_t>(2 * st.beam);
    begin_kernel("beam_logprobs", double(st.rows) * V * (fused_topk ? 4 : 12), 0, st.rows, 0, V);
    launch_beam_logprobs(st.logits, st.beam_scores, st.rows, V, st.g_eos, st.g_min_length,
                         st.d_step, st.logprobs, cur_stream(),
                         st.prefix_extends_min ? st.d_bprefix_len : nullptr,
                         static_cast<int32_t>(st.beam), !half_compute(), st.d_bprefix, st.d_bprefix_len,
                         st.bprefix_stride, st.prefix_first_step, bias_beta, st.d_diverged,
                         fused_topk ? st.bt_part_v : nullptr, fused_topk ? st.bt_part_i : nullptr,
                         fused_topk ? cand : 0);
    end_kernel();
    if (!fused_topk) {
      begin_kernel("beam_topk", double(st.rows) * V * 4, 0, st.batch, 2 * st.beam, st.beam * V);
      launch_topk(st.logprobs, st.batch, st.beam * V, 2 * st.beam, st.topk_scores, st.topk_ids,
                  cur_stream(), st.topk_ws, st.topk_ws_bytes);
      end_kernel();
    }
    begin_kernel("beam_select");
    launch_beam_select(st.topk_scores, st.topk_ids, st.batch, static_cast<int32_t>(st.beam), V,
                       st.g_eos, st.d_bprefix, st.d_bprefix_len, st.bprefix_stride, st.d_step,
                       st.prefix_first_step, st.hist_cand_word, st.hist_cand_score,
                       st.hist_cand_origin, st.d_ids, st.d_parents, st.beam_scores,
                       (st.lp_active || st.wh_active) ? st.alive_seq[par] : nullptr,
                       (st.lp_active || st.wh_active) ? st.alive_seq[par ^ 1] : nullptr,
                       static_cast<int32_t>(st.t_max), cur_stream(),
                       st.ancestry ? st.anc[par] : nullptr, st.ancestry ? st.anc[par ^ 1] : nullptr,
                       (st.d_bprefix && st.bias_beta > 0.f) ? st.d_diverged : nullptr,
                       fused_topk ? st.bt_part_v : nullptr, fused_topk ? st.bt_part_i : nullptr,
                       fused_topk ? static_cast<int32_t>(st.beam) * cand : 0,
                       fused_topk ? st.d_step : nullptr, fused_topk ? st.bt_ticket : nullptr);
    end_kernel();
    // Positions [0, step] are live; the gather copies the positions below the advanced counter.
    if (!fused_topk) {
      begin_kernel("advance_step");
      launch_advance_step(st.d_step, cur_stream());
      end_kernel();
    }
    if (!st.ancestry) {
      const AttnW& a0 = dec_.layers.front().self;
      const int n_tensors = static_cast<int>(2 * dec_.layers.size());
      begin_kernel("cache_gather");
      launch_cache_gather(par == 0 ? st.d_cache_src : st.d_cache_src_alt,
                          par == 0 ? st.d_cache_dst : st.d_cache_dst_alt, n_tensors, st.d_parents,
                          st.rows, a0.heads_kv, st.t_max, a0.dhead, st.d_step, 0, cur_stream());
      end_kernel();
    }
  }
}

void Engine::step(DecodeState& st, bool graph) {
  tls_stream = st.stream;
  set_active_stream(tls_stream);
  CT2X_HIP_CHECK(hipSetDevice(device_));
  // Make sure lazily built tables exist before any capture (no allocation inside a capture).
  if (dec_.sinusoidal_positions) (void)sinusoidal_table(st.t_max, dec_.d_model);
  if (dec_.layers.front().self.rotary) {
    const float *s, *c;
    rotary_tables(dec_.layers.front().self, st.t_max + 1, &s, &c);
  }
  // Under tensor parallelism the step is captured when the communicator's collectives are plain
  // stream work (RCCL); the in-process test double rendezvous on the host and runs eagerly.
  // (with mailboxes, a step whose exchanges all fit the one-shot form contains no collective call at
  // all - that also makes the in-process communicator capturable)
  bool tp_stream_only = tp_ && tp_->capturable();
  if (tp_ && !tp_stream_only && tp_->peer_box() && !tp_fast_) {
    int64_t widest = dec_.d_model;
    tp_stream_only = tp_oneshot_fits(st.rows, st.rows * widest);
  }
  const bool tp_eager = tp_ && !(tp_graph_ && tp_stream_only);
  if (!graph || profiling_ || tp_eager || st.graph_failed) {
    const int64_t before = tls_launches;
    launch_step(st);
    kernels_per_step_ = tls_launches - before;
    check_last_launch("decoder step");
    if (st.fuse_beam) st.cur ^= 1;  // the step reordered the caches into the other buffer set
    return;
  }
  // (a kept session may hold graphs of the other step form: it got the cluster gate in one call
  // and not in the next)
  if (st.graph_cluster != static_cast<int>(st.cl_active)) {
    if (st.graph_cluster >= 0) drop_graphs(st);
    st.graph_cluster = static_cast<int>(st.cl_active);
  }
  const int parity = st.cur;
  if (!st.graph_ready[parity]) {
    // One capture at a time per process: concurrent captures from several worker threads fail on
    // ROCm 7.2 ("operation failed due to a previous error during capture"); replays are free.
    std::lock_guard<std::mutex> capture_lock(graph_capture_mutex());
    hipGraph_t g = nullptr;
    const int64_t before = tls_launches;
    CT2X_HIP_CHECK(hipStreamBeginCapture(cur_stream(), hipStreamCaptureModeThreadLocal));
    try {
      launch_step(st);
    } catch (...) {
      hipGraph_t dead = nullptr;
      (void)hipStreamEndCapture(cur_stream(), &dead);
      if (dead) (void)hipGraphDestroy(dead);
      if (!tp_) throw;
      // tensor parallel: a collective that refuses capture throws from inside launch_step; the
      // capture recorded, it did not execute - this session runs eagerly from here on
      (void)hipGetLastError();
      st.graph_failed = true;
      step(st, false);
      return;
    }
    if (tp_) {
      // a collective library that refuses capture must not take the session down: this state
      // falls back to eager launches (the step has not run yet - a capture records, not executes)
      hipError_t e = hipStreamEndCapture(cur_stream(), &g);
      if (e == hipSuccess) e = hipGraphInstantiate(&st.graph_exec[parity], g, nullptr, nullptr, 0);
      if (g) (void)hipGraphDestroy(g);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        st.graph_failed = true;
        step(st, false);
        return;
      }
      kernels_per_step_ = tls_launches - before;
      st.graph_ready[parity] = true;
    } else {
    CT2X_HIP_CHECK(hipStreamEndCapture(cur_stream(), &g));
    kernels_per_step_ = tls_launches - before;
    st.graph_kernels = kernels_per_step_;
    CT2X_HIP_CHECK(hipGraphInstantiate(&st.graph_exec[parity], g, nullptr, nullptr, 0));
    CT2X_HIP_CHECK(hipGraphDestroy(g));
    st.graph_ready[parity] = true;
    }
  }
  CT2X_HIP_CHECK(hipGraphLaunch(st.graph_exec[parity], cur_stream()));
  if (st.graph_kernels > 0) kernels_per_step_ = st.graph_kernels;  // (a reused session replays at once)
  if (st.fuse_beam) st.cur ^= 1;
}

void Engine::drop_graphs(DecodeState& st) {
  for (int i = 0; i < 2; ++i) {
    if (st.graph_exec[i]) (void)hipGraphExecDestroy(st.graph_exec[i]);
    st.graph_exec[i] = nullptr;
    st.graph_ready[i] = false;
  }
  if (st.group_exec) (void)hipGraphExecDestroy(st.group_exec);
  st.group_exec = nullptr;
  st.group_steps = 0;
}

void Engine::step_group(DecodeState& st, int count, bool graph) {
  static const int group_graph = [] {
    const char* e = std::getenv("CT2X_GROUP_GRAPH");
    return e ? std::atoi(e) : 1;
  }();
  if (graph && st.graph_cluster >= 0 && st.graph_cluster != static_cast<int>(st.cl_active)) {
    drop_graphs(st);
    st.graph_cluster = -1;
  }
  // (beam search alternates between two cache / table sets: an even count ends on the parity it
  // started from, and the graph serves the groups that start on the parity it was captured at)
  const bool eligible = graph && group_graph != 0 && count > 1 && (st.fuse_greedy != st.fuse_beam) &&
                        (!st.fuse_beam || count % 2 == 0) &&
                        !profiling_ && !tp_ && !taps_ && !st.graph_failed &&
                        (st.reuses >= 1 || group_graph == 2) && st.graph_ready[st.cur] &&
                        (st.group_exec == nullptr ||
                         (st.group_steps == count && st.group_parity == st.cur));
  if (!eligible) {
    for (int i = 0; i < count; ++i) step(st, graph);
    return;
  }
  tls_stream = st.stream;
  set_active_stream(tls_stream);
  CT2X_HIP_CHECK(hipSetDevice(device_));
  if (st.group_exec == nullptr) {
    std::lock_guard<std::mutex> capture_lock(graph_capture_mutex());
    hipGraph_t g = nullptr;
    CT2X_HIP_CHECK(hipStreamBeginCapture(cur_stream(), hipStreamCaptureModeThreadLocal));
    const int parity = st.cur;
    try {
      for (int i = 0; i < count; ++i) {
        launch_step(st);
        if (st.fuse_beam) st.cur ^= 1;
      }
    } catch (...) {
      hipGraph_t dead = nullptr;
      (void)hipStreamEndCapture(cur_stream(), &dead);
      if (dead) (void)hipGraphDestroy(dead);
      st.cur = parity;
      throw;
    }
    st.cur = parity;  // (count is even under beam search)
    CT2X_HIP_CHECK(hipStreamEndCapture(cur_stream(), &g));
    CT2X_HIP_CHECK(hipGraphInstantiate(&st.group_exec, g, nullptr, nullptr, 0));
    CT2X_HIP_CHECK(hipGraphDestroy(g));
    st.group_steps = count;
    st.group_parity = parity;
  }
  CT2X_HIP_CHECK(hipGraphLaunch(st.group_exec, cur_stream()));
  if (st.graph_kernels > 0) kernels_per_step_ = st.graph_kernels;
}

void Engine::advance_step(DecodeState& st) {
  tls_stream = st.stream;
  set_active_stream(tls_stream);
  launch_advance_step(st.d_step, cur_stream());
}

void Engine::read_beam_history_async(DecodeState& st, int64_t first, int64_t count, int slot) {
  tls_stream = st.stream;
  set_active_stream(tls_stream);
  const int64_t per_step = st.batch * 2 * st.beam;
  const int64_t need = count * per_step;
  if (st.pin_cand_capacity < need) {
    for (int i = 0; i < 2; ++i) {
      pinned_free(st.pin_cand_word[i]);
      pinned_free(st.pin_cand_score[i]);
      pinned_free(st.pin_cand_origin[i]);
      st.pin_cand_word[i] = static_cast<int32_t*>(pinned_malloc(need * sizeof(int32_t)));
      st.pin_cand_score[i] = static_cast<float*>(pinned_malloc(need * sizeof(float)));
      st.pin_cand_origin[i] = static_cast<int32_t*>(pinned_malloc(need * sizeof(int32_t)));
      if (!st.pin_cand_event[i])
        CT2X_HIP_CHECK(hipEventCreateWithFlags(&st.pin_cand_event[i], hipEventDisableTiming));
    }
    st.pin_cand_capacity = need;
  }
  const int64_t off = first * per_step;
  CT2X_HIP_CHECK(hipMemcpyAsync(st.pin_cand_word[slot], st.hist_cand_word + off,
                                sizeof(int32_t) * need, hipMemcpyDeviceToHost, cur_stream()));
  CT2X_HIP_CHECK(hipMemcpyAsync(st.pin_cand_score[slot], st.hist_cand_score + off,
                                sizeof(float) * need, hipMemcpyDeviceToHost, cur_stream()));
  CT2X_HIP_CHECK(hipMemcpyAsync(st.pin_cand_origin[slot], st.hist_cand_origin + off,
                                sizeof(int32_t) * need, hipMemcpyDeviceToHost, cur_stream()));
  CT2X_HIP_CHECK(hipEventRecord(st.pin_cand_event[slot], cur_stream()));
}

void Engine::wait_beam_history(DecodeState& st, int slot, const int32_t** h_words,
                               const float** h_scores, const int32_t** h_origins) {
  CT2X_HIP_CHECK(hipEventSynchronize(st.pin_cand_event[slot]));
  *h_words = st.pin_cand_word[slot];
  *h_scores = st.pin_cand_score[slot];
  *h_origins = st.pin_cand_origin[slot];
}

void Engine::set_beam_prefix(DecodeState& st, const std::vector<std::vector<int32_t>>& prefix,
                             int32_t first_step, bool extends_min_length, float bias_beta) {
  tls_stream = st.stream;
  set_active_stream(tls_stream);
  CT2X_HIP_CHECK(hipSetDevice(device_));
  st.prefix_first_step = first_step;
  st.prefix_extends_min = extends_min_length;
  st.bias_beta = bias_beta;
  if (bias_beta > 0.f) {  // biased decoding: no beam has left its prefix yet
    if (!st.d_diverged) {
      st.d_diverged = static_cast<uint8_t*>(device_malloc(static_cast<size_t>(st.rows)));
      st.owned.push_back(st.d_diverged);
    }
    CT2X_HIP_CHECK(hipMemsetAsync(st.d_diverged, 0, static_cast<size_t>(st.rows), cur_stream()));
  }
  st.d_bprefix = st.d_bprefix_len = nullptr;
  st.bprefix_stride = 0;
  size_t longest = 0;
  for (const auto& p : prefix) longest = std::max(longest, p.size());
  if (longest == 0) return;
  CT2X_CHECK(static_cast<int64_t>(prefix.size()) == st.batch, "set_beam_prefix: one prefix per batch entry");
  std::vector<int32_t> ids(static_cast<size_t>(st.batch) * longest, 0), lens(st.batch, 0);
  for (int64_t b = 0; b < st.batch; ++b) {
    lens[b] = static_cast<int32_t>(prefix[b].size());
    std::copy(prefix[b].begin(), prefix[b].end(), ids.begin() + b * longest);
  }
  auto upload_i32 = [&](const std::vector<int32_t>& v) {
    void* p = device_malloc(v.size() * sizeof(int32_t));
    st.owned.push_back(p);
    CT2X_HIP_CHECK(hipMemcpyAsync(p, v.data(), v.size() * sizeof(int32_t), hipMemcpyHostToDevice,
                                  cur_stream()));
    return static_cast<int32_t*>(p);
  };
  st.d_bprefix = upload_i32(ids);
  st.d_bprefix_len = upload_i32(lens);
  st.bprefix_stride = static_cast<int32_t>(longest);
  CT2X_HIP_CHECK(hipStreamSynchronize(cur_stream()));  // the host vectors go out of scope
}

void Engine::set_beam_scores(DecodeState& st, const float* h_scores) {
  tls_stream = st.stream;
  set_active_stream(tls_stream);
  CT2X_HIP_CHECK(hipMemcpyAsync(st.beam_scores, h_scores, sizeof(float) * st.rows,
                                hipMemcpyHostToDevice, cur_stream()));
  CT2X_HIP_CHECK(hipStreamSynchronize(cur_stream()));
}

void Engine::read_history(DecodeState& st, int64_t first, int64_t count, int32_t* h_ids,
                          float* h_scores) {
  tls_stream = st.stream;
  set_active_stream(tls_stream);
  if (count <= 0) return;
  CT2X_HIP_CHECK(hipMemcpyAsync(h_ids, st.hist_ids + first * st.rows,
                                sizeof(int32_t) * count * st.rows, hipMemcpyDeviceToHost,
                                cur_stream()));
  if (h_scores)
    CT2X_HIP_CHECK(hipMemcpyAsync(h_scores, st.hist_scores + first * st.rows,
                                  sizeof(float) * count * st.rows, hipMemcpyDeviceToHost,
                                  cur_stream()));
  CT2X_HIP_CHECK(hipStreamSynchronize(cur_stream()));
}

void Engine::read_history_async(DecodeState& st, int64_t first, int64_t count, int slot) {
  tls_stream = st.stream;
  set_active_stream(tls_stream);
  const int64_t need = count * st.rows;
  if (st.pin_capacity < need) {
    for (int i = 0; i < 2; ++i) {
      pinned_free(st.pin_ids[i]);
      pinned_free(st.pin_scores[i]);
      st.pin_ids[i] = static_cast<int32_t*>(pinned_malloc(need * sizeof(int32_t)));
      st.pin_scores[i] = static_cast<float*>(pinned_malloc(need * sizeof(float)));
      if (!st.pin_event[i]) CT2X_HIP_CHECK(hipEventCreateWithFlags(&st.pin_event[i], hipEventDisableTiming));
    }
    st.pin_capacity = need;
  }
  // (the fused greedy sampler stored the ids of these steps in the pinned mirror itself: nothing to
  // copy - the event alone says the group has run; scores only travel when somebody asked for them)
  st.pin_from_host[slot] = st.hist_on_host;
  st.pin_first[slot] = first;
  if (!st.hist_on_host)
    CT2X_HIP_CHECK(hipMemcpyAsync(st.pin_ids[slot], st.hist_ids + first * st.rows,
                                  sizeof(int32_t) * need, hipMemcpyDeviceToHost, cur_stream()));
  if (st.g_scores || !st.fuse_greedy)
    CT2X_HIP_CHECK(hipMemcpyAsync(st.pin_scores[slot], st.hist_scores + first * st.rows,
                                  sizeof(float) * need, hipMemcpyDeviceToHost, cur_stream()));
  CT2X_HIP_CHECK(hipEventRecord(st.pin_event[slot], cur_stream()));
}

void Engine::wait_history(DecodeState& st, int slot, const int32_t** h_ids,
                          const float** h_scores) {
  CT2X_HIP_CHECK(hipEventSynchronize(st.pin_event[slot]));
  *h_ids = st.pin_from_host[slot] ? st.host_hist_ids + st.pin_first[slot] * st.rows : st.pin_ids[slot];
  *h_scores = st.pin_scores[slot];
}

void Engine::copy_logits_to_host(DecodeState& st, float* h_logits) {
  tls_stream = st.stream;
  set_active_stream(tls_stream);
  CT2X_HIP_CHECK(hipMemcpyAsync(h_logits, st.logits, sizeof(float) * st.rows * dec_.projection.n,
                                hipMemcpyDeviceToHost, cur_stream()));
  CT2X_HIP_CHECK(hipStreamSynchronize(cur_stream()));
}

void Engine::set_ids(DecodeState& st, const int32_t* h_ids) {
  tls_stream = st.stream;
  set_active_stream(tls_stream);
  CT2X_HIP_CHECK(hipMemcpyAsync(st.d_ids, h_ids, sizeof(int32_t) * st.rows,
                                hipMemcpyHostToDevice, cur_stream()));
  CT2X_HIP_CHECK(hipStreamSynchronize(cur_stream()));  // h_ids may be a temporary
}

// The greedy sampler can ride in the vocabulary projection when the step needs nothing but each
// row's arg-max (src/decoding.cc:874-880: no scores -> no LogSoftMax) and nothing else reads or
// rewrites the logits: no logits processors, no Whisper rules / no-speech probe, no logits
// history, no taps. CT2X_FUSED_ARGMAX=0 restores the sampler launch (same ids).
bool Engine::fused_argmax_applicable(const DecodeState& st) const {
  static const bool enabled = [] {
    const char* e = std::getenv("CT2X_FUSED_ARGMAX");
    return e == nullptr || e[0] != '0';
  }();
  const DenseW& w = dec_.projection;
  return enabled && st.fuse_greedy && !st.fuse_beam && st.g_topk == 1 && !st.g_scores &&
         !st.lp_active && !st.wh_active && st.wh_no_speech_id < 0 && st.logit_hist == nullptr &&
         !taps_ && !tp_ && w.quantized && !w.awq && !w.row_parallel && st.gf_slots != nullptr &&
         !(w.wq_stream && !w.wq_stream_dual && gemm_s8_stream_applicable(st.rows, w.n, w.k, false)) &&
         gemm_s8_argmax_applicable(st.rows, w.n, w.k) &&
         (reinterpret_cast<uintptr_t>(w.wq) & 15) == 0;
}

bool Engine::cluster_applicable(const DecodeState& st) const {
  static const bool beam_enabled = [] {  // CT2X_CLUSTER_BEAM=0: beam search keeps the launch-per-op step
    const char* e = std::getenv("CT2X_CLUSTER_BEAM");
    return e == nullptr || e[0] != '0';
  }();
  if (!cluster_ready_ || st.cl_xch == nullptr || !st.cl_active || taps_ || tp_ || st.att_hist != nullptr ||
      st.mem_lengths == nullptr || st.rows < 1)
    return false;
  if (st.beam_mode)  // a cluster per batch entry, the self-attention through the ancestry table
    return beam_enabled && st.ancestry && st.beam >= 2 && st.beam <= 4 && st.batch <= 32 &&
           st.rows == st.batch * st.beam;
  return st.beam == 1 && !st.ancestry && st.cur == 0 && st.rows <= cluster_max_rows();
}

bool Engine::compactable(const DecodeState& st) const {
  return !tp_ && !taps_ && !profiling_ && st.beam == 1 && !st.beam_mode && !st.ancestry &&
         st.att_hist == nullptr && st.logit_hist == nullptr && !st.wh_active &&
         st.wh_no_speech_id < 0 && st.rows > 1;
}

void Engine::compact_rows(DecodeState& st, const std::vector<int32_t>& keep, int64_t steps_done) {
  CT2X_CHECK(compactable(st), "compact_rows: this session holds state that is not moved");
  const int64_t n_new = static_cast<int64_t>(keep.size()), n_old = st.rows;
  CT2X_CHECK(n_new >= 1 && n_new < n_old, "compact_rows: nothing to drop");
  tls_stream = st.stream;
  set_active_stream(tls_stream);
  CT2X_HIP_CHECK(hipSetDevice(device_));
  auto fresh = [&](size_t bytes) {
    void* p = device_malloc(bytes);
    st.owned.push_back(p);  // (the buffers it replaces stay owned until the session ends)
    return p;
  };
  int32_t* d_keep = static_cast<int32_t*>(fresh(sizeof(int32_t) * n_new));
  CT2X_HIP_CHECK(hipMemcpyAsync(d_keep, keep.data(), sizeof(int32_t) * n_new, hipMemcpyHostToDevice,
                                cur_stream()));
  // (the buffers a gather replaces go back to the pool once the stream has drained - a session that
  // halves its rows several times must not keep every generation of its caches until it ends)
  std::vector<void*> superseded;
  auto rows_of = [&](auto*& buf, int64_t row_elems) {
    if (buf == nullptr || row_elems == 0) return;
    using T = std::remove_reference_t<decltype(*buf)>;
    const size_t row_bytes = static_cast<size_t>(row_elems) * sizeof(T);
    T* out = static_cast<T*>(fresh(row_bytes * n_new));
    launch_gather_rows(buf, d_keep, n_new, static_cast<int64_t>(row_bytes), out, cur_stream());
    superseded.push_back(const_cast<void*>(static_cast<const void*>(buf)));
    buf = out;
  };
  for (size_t l = 0; l < dec_.layers.size(); ++l) {
    const AttnW& a = dec_.layers[l].self;
    const int64_t row = a.heads_kv * st.t_max * a.dhead;
    rows_of(st.self_k[0][l], row);
    rows_of(st.self_v[0][l], row);
    if (l < st.cross_kv.size()) {
      const int64_t kv = 2 * dec_.layers[l].cross.heads_kv * dec_.layers[l].cross.dhead;
      rows_of(st.cross_kv[l], st.s_max * kv);
      if (l < st.cross_kv16.size()) rows_of(st.cross_kv16[l], st.s_max * kv);
    }
  }
  rows_of(st.mem_lengths, 1);
  rows_of(st.d_ids, 1);
  if (st.d_prefix != nullptr) {
    rows_of(st.d_prefix, st.prefix_stride);
    rows_of(st.d_prefix_len, 1);
  }
  // histories [step][row]: the steps behind us in the new row numbering (the device-side logits
  // processors read them), the steps ahead are written in it
  const int64_t past = std::min<int64_t>(steps_done + st.prefix_first_step, st.t_max + 1);
  {
    int32_t* ids = static_cast<int32_t*>(fresh(sizeof(int32_t) * (st.t_max + 1) * n_new));
    float* scores = static_cast<float*>(fresh(sizeof(float) * (st.t_max + 1) * n_new));
    launch_gather_cols(st.hist_ids, d_keep, past, n_old, n_new, ids, cur_stream());
    launch_gather_cols(st.hist_scores, d_keep, past, n_old, n_new, scores, cur_stream());
    superseded.push_back(st.hist_ids);
    superseded.push_back(st.hist_scores);
    st.hist_ids = ids;
    st.hist_scores = scores;
  }
  // The several-rows form of the decoder-stack launch (> 32 rows) indexes the hand-over lines and
  // the arrival counters per CLUSTER, the one-row form per ROW: a session that crosses 32 rows
  // starts the other layout from clean lines (tag 0 = "never written", epoch 0).
  if (st.cl_xch != nullptr && n_old > 32 && n_new <= 32) {
    CT2X_HIP_CHECK(hipMemsetAsync(st.cl_xch, 0, sizeof(float) * st.cl_xch_floats, cur_stream()));
    CT2X_HIP_CHECK(hipMemsetAsync(st.cl_counters, 0, sizeof(unsigned) * st.cl_counter_words, cur_stream()));
  }
  check_last_launch("compact_rows");
  CT2X_HIP_CHECK(hipStreamSynchronize(cur_stream()));  // (`keep` may be a temporary)
  superseded.push_back(d_keep);
  for (void* old : superseded) {
    auto it = std::find(st.owned.begin(), st.owned.end(), old);
    if (it == st.owned.end()) continue;  // (not a buffer of its own: part of a larger allocation)
    st.owned.erase(it);
    device_free(old);
  }
  st.rows = n_new;
  st.batch = n_new;
  for (int i = 0; i < 2; ++i) {
    if (st.graph_exec[i]) (void)hipGraphExecDestroy(st.graph_exec[i]);
    st.graph_exec[i] = nullptr;
    st.graph_ready[i] = false;
  }
  if (st.group_exec) (void)hipGraphExecDestroy(st.group_exec);
  st.group_exec = nullptr;
  st.group_steps = 0;
}

bool Engine::beam_compactable(const DecodeState& st) const {
  return !tp_ && !taps_ && !profiling_ && st.beam_mode && st.ancestry && st.anc[0] != nullptr &&
         st.att_hist == nullptr && !st.wh_active && st.wh_no_speech_id < 0 && st.batch > 1;
}

void Engine::compact_beam_entries(DecodeState& st, const std::vector<int32_t>& keep,
                                  int64_t positions_written) {
  CT2X_CHECK(beam_compactable(st), "compact_beam_entries: this session holds state that is not moved");
  const int64_t b_new = static_cast<int64_t>(keep.size()), b_old = st.batch, beam = st.beam;
  CT2X_CHECK(b_new >= 1 && b_new < b_old, "compact_beam_entries: nothing to drop");
  const int64_t r_new = b_new * beam, r_old = st.rows;
  tls_stream = st.stream;
  set_active_stream(tls_stream);
  CT2X_HIP_CHECK(hipSetDevice(device_));
  auto fresh = [&](size_t bytes) {
    void* p = device_malloc(bytes);
    st.owned.push_back(p);
    return p;
  };
  std::vector<int32_t> keep_rows(static_cast<size_t>(r_new));
  for (int64_t e = 0; e < b_new; ++e)
    for (int64_t k = 0; k < beam; ++k) keep_rows[e * beam + k] = static_cast<int32_t>(keep[e] * beam + k);
  int32_t* d_keep = static_cast<int32_t*>(fresh(sizeof(int32_t) * b_new));
  int32_t* d_keep_rows = static_cast<int32_t*>(fresh(sizeof(int32_t) * r_new));
  CT2X_HIP_CHECK(hipMemcpyAsync(d_keep, keep.data(), sizeof(int32_t) * b_new, hipMemcpyHostToDevice,
                                cur_stream()));
  CT2X_HIP_CHECK(hipMemcpyAsync(d_keep_rows, keep_rows.data(), sizeof(int32_t) * r_new,
                                hipMemcpyHostToDevice, cur_stream()));
  auto gather = [&](auto*& buf, int64_t row_elems, const int32_t* idx, int64_t n) {
    if (buf == nullptr || row_elems == 0) return;
    using T = std::remove_reference_t<decltype(*buf)>;
    const size_t row_bytes = static_cast<size_t>(row_elems) * sizeof(T);
    T* out = static_cast<T*>(fresh(row_bytes * n));
    launch_gather_rows(buf, idx, n, static_cast<int64_t>(row_bytes), out, cur_stream());
    buf = out;
  };
  for (size_t l = 0; l < st.cross_kv.size(); ++l) {
    const int64_t kv = 2 * dec_.layers[l].cross.heads_kv * dec_.layers[l].cross.dhead;
    gather(st.cross_kv[l], st.s_max * kv, d_keep, b_new);
    if (l < st.cross_kv16.size()) gather(st.cross_kv16[l], st.s_max * kv, d_keep, b_new);
  }
  gather(st.mem_lengths, 1, d_keep, b_new);
  if (st.d_bprefix != nullptr) {
    gather(st.d_bprefix, st.bprefix_stride, d_keep, b_new);
    gather(st.d_bprefix_len, 1, d_keep, b_new);
  }
  gather(st.d_ids, 1, d_keep_rows, r_new);
  gather(st.beam_scores, 1, d_keep_rows, r_new);
  gather(st.d_parents, 1, d_keep_rows, r_new);
  gather(st.d_diverged, 1, d_keep_rows, r_new);
  gather(st.alive_seq[0], st.t_max, d_keep_rows, r_new);
  gather(st.alive_seq[1], st.t_max, d_keep_rows, r_new);
  // ancestry tables [t_max + 1][rows] -> [t_max + 1][rows']: for the positions written so far the
  // entries (cache rows) stay what they are - the caches are not touched; the positions ahead are the
  // identity of the NEW numbering (a row appends to its own cache row, attention_exact.hip, and
  // beam_select_kernel copies only written positions: the slots (row', t >= written) are unwritten)
  const int64_t written = std::min<int64_t>(std::max<int64_t>(positions_written, 0), st.t_max + 1);
  std::vector<int32_t> identity(static_cast<size_t>((st.t_max + 1 - written) * r_new));
  for (size_t e = 0; e < identity.size(); ++e) identity[e] = static_cast<int32_t>(e % r_new);
  for (int i = 0; i < 2; ++i) {
    int32_t* t = static_cast<int32_t*>(fresh(sizeof(int32_t) * (st.t_max + 1) * r_new));
    launch_gather_cols(st.anc[i], d_keep_rows, written, r_old, r_new, t, cur_stream());
    if (!identity.empty())
      CT2X_HIP_CHECK(hipMemcpyAsync(t + written * r_new, identity.data(), identity.size() * sizeof(int32_t),
                                    hipMemcpyHostToDevice, cur_stream()));
    st.anc[i] = t;
  }
  check_last_launch("compact_beam_entries");
  CT2X_HIP_CHECK(hipStreamSynchronize(cur_stream()));
  st.batch = b_new;
  st.rows = r_new;
  for (int i = 0; i < 2; ++i) {
    if (st.graph_exec[i]) (void)hipGraphExecDestroy(st.graph_exec[i]);
    st.graph_exec[i] = nullptr;
    st.graph_ready[i] = false;
  }
  if (st.group_exec) (void)hipGraphExecDestroy(st.group_exec);
  st.group_exec = nullptr;
  st.group_steps = 0;
}

void Engine::set_prefix(DecodeState& st, const std::vector<std::vector<int32_t>>& prefix,
                        int32_t first_step, bool extends_min_length) {
  st.prefix_extends_min = extends_min_length;
  tls_stream = st.stream;
  set_active_stream(tls_stream);
  CT2X_HIP_CHECK(hipSetDevice(device_));
  size_t longest = 0;
  for (const auto& p : prefix) longest = std::max(longest, p.size());
  st.d_prefix = nullptr;
  st.d_prefix_len = nullptr;
  st.prefix_stride = 0;
  st.prefix_first_step = first_step;
  if (longest == 0) return;
  CT2X_CHECK(static_cast<int64_t>(prefix.size()) == st.rows, "set_prefix: one prefix per row");
  std::vector<int32_t> ids(static_cast<size_t>(st.rows) * longest, 0), lens(st.rows, 0);
  for (int64_t r = 0; r < st.rows; ++r) {
    lens[r] = static_cast<int32_t>(prefix[r].size());
    std::copy(prefix[r].begin(), prefix[r].end(), ids.begin() + r * longest);
  }
  auto upload_i32 = [&](const std::vector<int32_t>& v) {
    void* p = device_malloc(v.size() * sizeof(int32_t));
    st.owned.push_back(p);
    CT2X_HIP_CHECK(hipMemcpyAsync(p, v.data(), v.size() * sizeof(int32_t), hipMemcpyHostToDevice,
                                  cur_stream()));
    return static_cast<int32_t*>(p);
  };
  st.d_prefix = upload_i32(ids);
  st.d_prefix_len = upload_i32(lens);
  st.prefix_stride = static_cast<int32_t>(longest);
  CT2X_HIP_CHECK(hipStreamSynchronize(cur_stream()));  // the host vectors go out of scope
}

void Engine::set_logits_processing(DecodeState& st, const LogitsProcessing& lp) {
  tls_stream = st.stream;
  set_active_stream(tls_stream);
  CT2X_HIP_CHECK(hipSetDevice(device_));
  st.lp_active = lp.active();
  st.lp_penalty = lp.repetition_penalty;
  st.lp_ngram = lp.no_repeat_ngram_size;
  st.lp_n_sup = 0;
  st.d_sup_ids = st.d_sup_off = nullptr;
  if (!st.lp_active) return;
  CT2X_CHECK(lp.repetition_penalty > 0.f, "repetition_penalty must be > 0");
  CT2X_CHECK(lp.no_repeat_ngram_size >= 0, "no_repeat_ngram_size must be >= 0");
  // the gather buffer of the penalty lives in LDS: one float per generated position
  CT2X_CHECK(lp.repetition_penalty == 1.f || st.t_max <= 16000,
             "repetition_penalty supports at most 16000 decoding positions");
  const int64_t vocab = dec_.projection.n;
  std::vector<int32_t> ids, off{0};
  for (const auto& s : lp.suppress_sequences) {
    if (s.empty()) continue;  // SuppressSequences ctor, src/decoding_utils.cc:109-111
    for (int32_t id : s) {
      CT2X_CHECK(id >= 0 && id < vocab, "suppressed token id out of the vocabulary");
      ids.push_back(id);
    }
    off.push_back(static_cast<int32_t>(ids.size()));
  }
  auto upload_i32 = [&](const std::vector<int32_t>& v) {
    void* p = device_malloc(v.size() * sizeof(int32_t));
    st.owned.push_back(p);
    CT2X_HIP_CHECK(hipMemcpyAsync(p, v.data(), v.size() * sizeof(int32_t), hipMemcpyHostToDevice,
                                  cur_stream()));
    return static_cast<int32_t*>(p);
  };
  st.lp_n_sup = static_cast<int32_t>(off.size()) - 1;
  if (st.lp_n_sup > 0) {
    st.d_sup_ids = upload_i32(ids);
    st.d_sup_off = upload_i32(off);
  }
  if (st.beam_mode && st.alive_seq[0] == nullptr) {
    for (int i = 0; i < 2; ++i) {
      st.alive_seq[i] = static_cast<int32_t*>(device_malloc(sizeof(int32_t) * st.rows * st.t_max));
      st.owned.push_back(st.alive_seq[i]);
    }
  }
  CT2X_HIP_CHECK(hipStreamSynchronize(cur_stream()));  // the host vectors go out of scope
}

void Engine::enable_attention(DecodeState& st) {
  CT2X_CHECK(has_encoder_, "return_attention / coverage_penalty need an encoder-decoder model");
  tls_stream = st.stream;
  set_active_stream(tls_stream);
  CT2X_HIP_CHECK(hipSetDevice(device_));
  if (st.att_hist) return;
  const size_t n = static_cast<size_t>(st.t_max + 1) * st.rows * st.s_max;
  st.att_hist = static_cast<float*>(device_malloc(n * sizeof(float)));
  st.owned.push_back(st.att_hist);
  CT2X_HIP_CHECK(hipMemsetAsync(st.att_hist, 0, n * sizeof(float), cur_stream()));
  st.graph_ready[0] = st.graph_ready[1] = false;  // (called before the first step: nothing captured yet)
}

void Engine::read_attention(DecodeState& st, int64_t first, int64_t count, float* out) {
  tls_stream = st.stream;
  set_active_stream(tls_stream);
  CT2X_CHECK(st.att_hist != nullptr && first >= 0 && first + count <= st.t_max + 1,
             "read_attention: no attention history for these steps");
  const size_t per_step = static_cast<size_t>(st.rows) * st.s_max;
  CT2X_HIP_CHECK(hipMemcpyAsync(out, st.att_hist + first * per_step, sizeof(float) * count * per_step,
                                hipMemcpyDeviceToHost, cur_stream()));
  CT2X_HIP_CHECK(hipStreamSynchronize(cur_stream()));
}

void Engine::enable_logits_history(DecodeState& st) {
  tls_stream = st.stream;
  set_active_stream(tls_stream);
  CT2X_HIP_CHECK(hipSetDevice(device_));
  if (st.logit_hist) return;
  const size_t n = static_cast<size_t>(kLogitsRing) * st.rows * vocab_size();
  st.logit_hist = static_cast<float*>(device_malloc(n * sizeof(float)));
  st.owned.push_back(st.logit_hist);
  st.graph_ready[0] = st.graph_ready[1] = false;  // (called before the first step: nothing captured yet)
}

void Engine::read_logits(DecodeState& st, int64_t first, int64_t count, float* out) {
  tls_stream = st.stream;
  set_active_stream(tls_stream);
  CT2X_CHECK(st.logit_hist != nullptr && first >= 0 && count <= kLogitsRing,
             "read_logits: no logits history for these steps");
  const size_t per_step = static_cast<size_t>(st.rows) * vocab_size();
  for (int64_t s = 0; s < count; ++s)
    CT2X_HIP_CHECK(hipMemcpyAsync(out + s * per_step,
                                  st.logit_hist + ((first + s) % kLogitsRing) * per_step,
                                  sizeof(float) * per_step, hipMemcpyDeviceToHost, cur_stream()));
  CT2X_HIP_CHECK(hipStreamSynchronize(cur_stream()));
}

void Engine::set_step(DecodeState& st, int32_t step) {
  tls_stream = st.stream;
  set_active_stream(tls_stream);
  CT2X_HIP_CHECK(hipMemcpyAsync(st.d_step, &step, sizeof(int32_t), hipMemcpyHostToDevice,
                                cur_stream()));
  reset_lead_flags(st);
  CT2X_HIP_CHECK(hipStreamSynchronize(cur_stream()));
}

void Engine::synchronize() { CT2X_HIP_CHECK(hipStreamSynchronize(cur_stream())); }

void Engine::read_encoder_output(DecodeState& st, float* out) {
  tls_stream = st.stream;
  set_active_stream(tls_stream);
  CT2X_CHECK(st.enc_out != nullptr, "read_encoder_output: the state holds no encoder output");
  CT2X_HIP_CHECK(hipMemcpyAsync(out, st.enc_out, sizeof(float) * st.batch * st.s_max * enc_.d_model,
                                hipMemcpyDeviceToHost, cur_stream()));
  CT2X_HIP_CHECK(hipStreamSynchronize(cur_stream()));
}

void Engine::set_whisper_processing(DecodeState& st, const WhisperProcessing& wp) {
  tls_stream = st.stream;
  set_active_stream(tls_stream);
  CT2X_HIP_CHECK(hipSetDevice(device_));
  const int64_t vocab = dec_.projection.n;
  st.wh = WhisperRulesParams();
  st.wh_active = !wp.begin_ids.empty() || wp.timestamp_rules;
  st.wh_no_speech_id = wp.no_speech_id;
  if (wp.no_speech_id >= 0) {
    CT2X_CHECK(wp.no_speech_id < vocab, "no-speech token id out of the vocabulary");
    if (!st.d_no_speech) st.d_no_speech = alloc<float>(st, st.rows);
  }
  if (!wp.begin_ids.empty()) {
    for (int32_t id : wp.begin_ids)
      CT2X_CHECK(id >= 0 && id < vocab, "suppressed token id out of the vocabulary");
    int32_t* d = alloc<int32_t>(st, static_cast<int64_t>(wp.begin_ids.size()));
    CT2X_HIP_CHECK(hipMemcpyAsync(d, wp.begin_ids.data(), sizeof(int32_t) * wp.begin_ids.size(),
                                  hipMemcpyHostToDevice, cur_stream()));
    st.wh.begin_ids = d;
    st.wh.n_begin = static_cast<int32_t>(wp.begin_ids.size());
  }
  st.wh.timestamp_rules = wp.timestamp_rules;
  if (wp.timestamp_rules) {
    CT2X_CHECK(wp.eot_id >= 0 && wp.eot_id < vocab && wp.no_timestamps_id >= 0 &&
                   wp.no_timestamps_id < vocab && wp.timestamp_begin >= 0 &&
                   wp.timestamp_begin <= wp.timestamp_end && wp.timestamp_end < vocab,
               "timestamp rules: token ids out of the vocabulary");
    st.wh.eot_id = wp.eot_id;
    st.wh.no_timestamps_id = wp.no_timestamps_id;
    st.wh.timestamp_begin = wp.timestamp_begin;
    st.wh.timestamp_end = wp.timestamp_end;
    st.wh.max_initial_timestamp = wp.max_initial_timestamp;
  }
  // beam search: the rules read the tokens of the active beams (kept by beam_select_kernel)
  if (st.wh_active && st.beam_mode && st.alive_seq[0] == nullptr) {
    for (int i = 0; i < 2; ++i) st.alive_seq[i] = alloc<int32_t>(st, st.rows * st.t_max);
  }
  CT2X_HIP_CHECK(hipStreamSynchronize(cur_stream()));  // the host vector may go out of scope
}

void Engine::whisper_align_forward(DecodeState& st, const int32_t* h_ids, const int32_t* h_targets,
                                   int64_t time, const std::vector<int32_t>& lengths,
                                   const std::vector<int32_t>& frames,
                                   const std::vector<std::pair<int32_t, int32_t>>& alignment_heads,
                                   int64_t text_classes, int64_t median_filter_width,
                                   float* h_token_probs, float* h_weights) {
  tls_stream = st.stream;
  set_active_stream(tls_stream);
  CT2X_HIP_CHECK(hipSetDevice(device_));
  const int64_t B = st.batch, S = st.s_max, V = dec_.projection.n;
  CT2X_CHECK(st.beam == 1 && static_cast<int64_t>(lengths.size()) == B &&
                 static_cast<int64_t>(frames.size()) == B, "whisper_align_forward: batch mismatch");
  const int64_t n_layers = static_cast<int64_t>(dec_.layers.size());
  ScoreCapture cap;
  cap.heads.resize(n_layers);
  cap.d_heads.resize(n_layers, nullptr);
  for (const auto& lh : alignment_heads) {
    CT2X_CHECK(lh.first >= 0 && lh.first < n_layers && lh.second >= 0 && lh.second < dec_.num_heads,
               "alignment_heads: (layer, head) out of range");
    cap.heads[lh.first].push_back(lh.second);
  }
  cap.n_sel = static_cast<int64_t>(alignment_heads.size());
  CT2X_CHECK(cap.n_sel > 0, "alignment_heads is empty");
  for (int64_t l = 0; l < n_layers; ++l)
    if (!cap.heads[l].empty()) {
      int32_t* d = alloc<int32_t>(st, static_cast<int64_t>(cap.heads[l].size()));
      CT2X_HIP_CHECK(hipMemcpyAsync(d, cap.heads[l].data(), sizeof(int32_t) * cap.heads[l].size(),
                                    hipMemcpyHostToDevice, cur_stream()));
      cap.d_heads[l] = d;
    }
  cap.scores = alloc<float>(st, B * cap.n_sel * time * S);
  float* logits_all = alloc<float>(st, B * time * V);
  int32_t* d_targets = alloc<int32_t>(st, B * time);
  float* d_probs = alloc<float>(st, B * time);
  CT2X_HIP_CHECK(hipMemcpyAsync(d_targets, h_targets, sizeof(int32_t) * B * time,
                                hipMemcpyHostToDevice, cur_stream()));
  prefill(st, h_ids, time, logits_all, &cap);
  // "Get the probabilities of the text tokens" (whisper.cc:497-506)
  launch_token_probs(logits_all, B * time, V, text_classes, d_targets, d_probs, cur_stream());
  CT2X_HIP_CHECK(hipMemcpyAsync(h_token_probs, d_probs, sizeof(float) * B * time,
                                hipMemcpyDeviceToHost, cur_stream()));
  // compute_alignments (whisper.cc:388-423) per batch entry. With the same number of frames for
  // every entry the reference normalises over the batch's padded token axis (rows past an entry's
  // length repeat its last row); with different numbers it cuts every entry to its own length.
  bool variable_frames = false;
  for (int64_t b = 0; b < B; ++b) variable_frames = variable_frames || frames[b] != frames[0];
  float* probs = alloc<float>(st, cap.n_sel * time * S);
  float* normed = alloc<float>(st, cap.n_sel * time * S);
  float* mean = alloc<float>(st, time * S);
  for (int64_t b = 0; b < B; ++b) {
    const int64_t F = frames[b], len = lengths[b];
    if (F <= 0 || len <= 0) continue;
    CT2X_CHECK(F <= S && len <= time, "whisper_align_forward: frames / length out of range");
    const int64_t rows_out = variable_frames ? len : time;
    launch_align_softmax(cap.scores + b * cap.n_sel * time * S, cap.n_sel, time, S, len, rows_out, F,
                         probs, cur_stream());
    launch_layer_norm_axis(probs, cap.n_sel, rows_out, F, F, normed, cur_stream());
    // MedianFilter passes the input through when the axis is not longer than width / 2
    // (src/ops/median_filter_cpu.cc:24-28)
    const float* filtered = normed;
    if (F > median_filter_width / 2) {
      launch_median_filter(normed, cap.n_sel * rows_out, F, median_filter_width, probs, cur_stream());
      filtered = probs;
    }
    launch_mean(filtered, 1, cap.n_sel, rows_out * F, false, mean, cur_stream());
    CT2X_HIP_CHECK(hipMemcpyAsync(h_weights + b * time * S, mean, sizeof(float) * rows_out * F,
                                  hipMemcpyDeviceToHost, cur_stream()));
    CT2X_HIP_CHECK(hipStreamSynchronize(cur_stream()));
  }
  CT2X_HIP_CHECK(hipStreamSynchronize(cur_stream()));
  check_last_launch("whisper_align_forward");
}

void Engine::read_no_speech_probs(DecodeState& st, float* out) {
  tls_stream = st.stream;
  set_active_stream(tls_stream);
  CT2X_CHECK(st.d_no_speech != nullptr, "read_no_speech_probs: GetNoSpeechProbs was not installed");
  CT2X_HIP_CHECK(hipMemcpyAsync(out, st.d_no_speech, sizeof(float) * st.rows, hipMemcpyDeviceToHost,
                                cur_stream()));
  CT2X_HIP_CHECK(hipStreamSynchronize(cur_stream()));
}

void Engine::score_targets(DecodeState& st, const int32_t* h_inputs, const int32_t* h_outputs,
                           int64_t steps) {
  tls_stream = st.stream;
  set_active_stream(tls_stream);
  CT2X_HIP_CHECK(hipSetDevice(device_));
  CT2X_CHECK(steps <= st.t_max, "score_targets: more steps than the decoding state holds");
  if (steps == 0) return;
  const size_t bytes = sizeof(int32_t) * static_cast<size_t>(steps) * st.rows;
  int32_t* d_inputs = static_cast<int32_t*>(device_malloc(bytes));
  st.owned.push_back(d_inputs);
  CT2X_HIP_CHECK(hipMemcpyAsync(d_inputs, h_inputs, bytes, hipMemcpyHostToDevice, cur_stream()));
  // the id history doubles as the table of target ids, the score history receives the result
  CT2X_HIP_CHECK(hipMemcpyAsync(st.hist_ids, h_outputs, bytes, hipMemcpyHostToDevice, cur_stream()));
  CT2X_HIP_CHECK(hipStreamSynchronize(cur_stream()));  // the host arrays may go away
  st.fuse_greedy = false;
  st.fuse_beam = false;
  set_step(st, 0);
  for (int64_t s = 0; s < steps; ++s) {
    CT2X_HIP_CHECK(hipMemcpyAsync(st.d_ids, d_inputs + s * st.rows, sizeof(int32_t) * st.rows,
                                  hipMemcpyDeviceToDevice, cur_stream()));
    step(st, /*graph=*/false);
    launch_score_tokens(st.logits, st.rows, dec_.projection.n, st.hist_ids + s * st.rows,
                        st.hist_scores + s * st.rows, cur_stream());
    launch_advance_step(st.d_step, cur_stream());
  }
  check_last_launch("score_targets");
}

}  // namespace ct2x
