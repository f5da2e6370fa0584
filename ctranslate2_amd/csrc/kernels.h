// Host-callable launchers for the gfx950 kernels. Everything is asynchronous on `stream`.
// Pointers are device pointers. These are the functions behind the C ABI (include/ct2_mi355.h)
// and the building blocks of the decode engine (engine.cc).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ct2x {

// ---- quantisation / normalisation (rowops.hip) ---------------------------------------

void launch_quantize_s8(const float* x, int64_t rows, int64_t depth, int quant_mode,
                        int8_t* q, float* scales, hipStream_t stream);

void launch_dequantize_rows(const int8_t* x, const float* scale, int64_t rows, int64_t depth,
                            float* y, hipStream_t stream);

void launch_dequantize_gemm_output(const int32_t* c, const float* a_scale, const float* b_scale,
                                   const float* bias, int64_t m, int64_t n, int act, float* y,
                                   hipStream_t stream, const float* residual = nullptr);

// Tensor-parallel helpers (rowops.hip): per-row absmax of a K slice, Quantize with a given
// (rank-reduced) absmax, the float row-parallel epilogue, and the rank-order combine used by the
// in-process communicator.
// Second half of a Quantize whose row absmax was produced by the previous launch (absmax in
// amax[rows]): element-wise over (rows x column chunks) workgroups instead of one workgroup per
// row - at 16 decode rows the row-per-workgroup kernels leave 240 CUs idle.
struct PrefetchRange {
  const void* p = nullptr;
  int64_t unit16 = 0, take16 = 0, chunks = 0;
};
struct PrefetchArgs {
  PrefetchRange r[2];
  int wgs = 0;  // spare workgroups of 256 threads appended to the launch (0 = none)
};

void launch_quantize_cols(const float* x, const float* amax, int64_t rows, int64_t depth,
                          int quant_mode, int8_t* q, float* scales, hipStream_t stream,
                          const PrefetchArgs* pf = nullptr, int amax_parts = 1);
void launch_row_amax(const float* x, int64_t rows, int64_t depth, float* amax, hipStream_t stream);
void launch_quantize_with_amax(const float* x, const float* amax, int64_t rows, int64_t depth,
                               int quant_mode, int8_t* q, float* scales, hipStream_t stream);
void launch_bias_residual(const float* x, const float* bias, const float* residual, int64_t m,
                          int64_t n, float* y, hipStream_t stream);
void launch_reduce_ranks(const void* const* src, int world, int64_t n, bool is_int32, bool is_max,
                         void* out, hipStream_t stream);

// beta == nullptr -> RMSNorm. y and/or q may be nullptr (q==nullptr: plain norm).
void launch_norm(const float* x, const float* gamma, const float* beta, int64_t rows,
                 int64_t depth, float eps, bool rms_use_residual, int quant_mode,
                 float* y, int8_t* q, float* scales, hipStream_t stream,
                 const PrefetchArgs* pf = nullptr, bool fast_stats = false);
// fast_stats: tree-reduced statistics instead of the CPU path's ordered chain (NOT bit-identical
// to the reference's CPU build; the engine asks for it under the 16-bit compute types only).

void launch_softmax(const float* x, const int32_t* lengths, int64_t rows, int64_t depth,
                    bool log, float* y, hipStream_t stream);

// log_softmax(x[row,:]) + row_bias[row] (beam scores), the prologue of beam-search TopK
// (src/decoding.cc:545-556). row_bias may be nullptr.
void launch_log_softmax_bias(const float* x, const float* row_bias, int64_t rows, int64_t depth,
                             float* y, hipStream_t stream);

void launch_rotary(const float* x, const float* sin, const float* cos, int64_t batch,
                   int64_t time, int64_t depth, int64_t ndims, bool interleave, float* y,
                   hipStream_t stream);

// Top-k per row over `depth` columns (values desc; ties -> lowest index first).
// Rows of >= 8192 columns take a two-launch form (per-chunk top-k, then merge) when the caller
// provides topk_workspace_bytes() bytes of scratch; without it they fall back to one workgroup
// per row.
size_t topk_workspace_bytes(int64_t rows, int64_t depth, int64_t k);
void launch_topk(const float* x, int64_t rows, int64_t depth, int64_t k, float* values,
                 int32_t* indices, hipStream_t stream, void* workspace = nullptr,
                 size_t workspace_bytes = 0);

void launch_gather_rows(const void* data, const int32_t* idx, int64_t n_idx, int64_t row_bytes,
                        void* out, hipStream_t stream);
// out[t, i] = in[t, idx[i]], t < steps, 4-byte elements ([step][row] histories)
void launch_gather_cols(const void* in, const int32_t* idx, int64_t steps, int64_t n_old, int64_t n_new,
                        void* out, hipStream_t stream);

// out[r,:] = table[ids[r],:] (dequantised with 1/scale[id] when table is int8) * emb_scale
//            + pos[(pos_base + r % time) ...]   (see engine.cc for the position convention)
// Mirrors Embeddings::operator() + Mul + PositionEncoder (src/layers/common.cc:64-81,151-174,
// src/layers/transformer.cc:636-648). `step_ptr` (device int) overrides pos_index when non-null.
void launch_embed(const void* table, bool table_is_int8, const float* table_scale,
                  const int32_t* ids, int64_t rows, int64_t depth, float emb_scale,
                  bool apply_scale, const float* pos_table, int64_t time, int64_t pos_index,
                  const int32_t* step_ptr, float* out, hipStream_t stream,
                  float* zero_buf = nullptr, int64_t zero_count = 0);

// y = a + b (residual adds outside a GEMM epilogue), y = a * b (GLU gate).
void launch_add(const float* a, const float* b, int64_t n, float* y, hipStream_t stream);
void launch_mul(const float* a, const float* b, int64_t n, float* y, hipStream_t stream);

// ---- int8 GEMM (gemm_s8.hip) ---------------------------------------------------------

// Greedy sampler fused into the vocabulary projection (gemm_s8.hip, ARGMAX form): a greedy step
// that returns no scores only needs the arg-max of every logits row (src/decoding.cc:874-880), so
// the projection's workgroups max-combine packed (value, column) keys per row and the LAST
// workgroup to finish plays greedy_sample_kernel's tail (prefix forcing, id to the next step,
// history, step counter): the logits never reach memory and the sampler launch disappears.
struct GreedyFuse {
  // Same-address atomics from 500 workgroups serialise: a row's maximum is spread over kSubSlots
  // slots (workgroup index modulo kSubSlots) that the finishing workgroup folds.
  static constexpr int kSubSlots = 16;
  unsigned long long* slots = nullptr;  // [m, kSubSlots] packed maxima, zero between launches; nullptr = off
  unsigned int* ticket = nullptr;       // workgroup arrival counter, zero between launches
  const int32_t* step_ptr = nullptr;
  int32_t disable_id = -1;              // DisableTokens below the minimum length (the end token)
  int32_t min_length = 0;
  const int32_t* prefix_ids = nullptr;  // SamplePrefix (decode_ops.hip)
  const int32_t* prefix_lens = nullptr;
  int32_t prefix_stride = 0;
  int32_t prefix_first_step = 0;
  int32_t prefix_extends_min = 0;
  int32_t* next_ids = nullptr;
  int32_t* hist_ids = nullptr;
  float* hist_scores = nullptr;
  int32_t* advance_step_ptr = nullptr;  // nullptr: the caller advances the step itself
  // pinned host copy of hist_ids (same layout) or nullptr: the finishing workgroup stores the ids
  // there as well, so the decoding loop reads them without a copy kernel after every group of steps
  int32_t* host_hist_ids = nullptr;
};
// shapes the fused form serves (the skinny kernel's vocabulary-projection instantiations)
bool gemm_s8_argmax_applicable(int64_t m, int64_t n, int64_t k);

struct GemmS8Epilogue {
  // mode 0: raw int32 store to c_i32.
  // mode 1: y = act(acc * (1/a_scale[i]) / w_scale[j] + bias[j]) + residual[i,j]  (fp32 store)
  int mode = 0;
  int32_t* c_i32 = nullptr;
  const float* a_scale = nullptr;
  const float* w_scale = nullptr;
  const float* bias = nullptr;
  const float* residual = nullptr;
  float* y = nullptr;
  int act = -1;
  int64_t ldy = 0;  // leading dimension of y / residual / c_i32 (>= n)
  // Gated pair (ffn_glu): when b2 != nullptr the launch also streams a second weight [n, k] and
  // stores y = act(acc * .. / w_scale + bias) * (acc2 * .. / w_scale2 + bias2). M <= 32 only.
  const int8_t* b2 = nullptr;
  const float* w_scale2 = nullptr;
  const float* bias2 = nullptr;
  // Norm prologue (pre-norm Dense at decode rows): when norm_x != nullptr the launch ignores `a`
  // and a_scale and computes them itself: LayerNorm (norm_beta != nullptr) or RMSNorm of
  // norm_x [m, k] followed by Quantize (quant_mode = device_utils.h flags). M <= 32, K <= 1024.
  const float* norm_x = nullptr;
  const float* norm_gamma = nullptr;
  const float* norm_beta = nullptr;
  float norm_eps = 1e-5f;
  bool norm_rms_residual = false;
  int quant_mode = 0;
  // Skinny kernel: per-row absmax of the stored y, max-combined into row_amax_out[m] (must be
  // zero beforehand) - the first half of the following ops::Quantize, see launch_quantize_cols.
  float* row_amax_out = nullptr;
  // Quantise-on-load (gemm_s8_qol_applicable): when a_f32 != nullptr the launch ignores `a` and
  // a_scale and quantises the fp32 activation a_f32 [m, k] while it fetches its A fragments, with
  // the row scale 127 / a_amax[row] (1 for a zero row) - ops::Quantize given the absmax a previous
  // launch left through row_amax_out. quant_mode as for the norm prologue.
  const float* a_f32 = nullptr;
  const float* a_amax = nullptr;
  // Streaming form only: a_amax is [m, a_amax_parts] partial maxima (plain stores of the producer's
  // workgroups, attention_exact128.hip) which the consumer folds; 0 / 1 = one value per row.
  int a_amax_parts = 0;
  // 256 x 256 kernel: divide with the IEEE sequence even where exact_div applies (A/B timing:
  // CT2X_EXACT_DIV=0; the results are the same bits either way).
  bool plain_division = false;
  // Streaming form (gemm_s8_stream.hip): `b` (and b2) hold the weight in the fragment-major
  // layout pack_stream_weight() produced for this (n, k) - every 1 KB request of a wave is one
  // contiguous block and a wave's requests of a slab follow each other in memory.
  bool packed_b = false;
  // Streaming form, leader prologues (round 6, gemm_s8_stream.hip AMODE 3 / 4): lead_flag != nullptr
  // turns the norm_x (RMSNorm + Quantize) or a_f32 / a_amax (ops::Quantize) prologue into work done
  // ONCE by the launch's workgroups 0..15 (one row each), which leave the int8 rows in lead_q
  // [16, k], the row scales in lead_scale [16] and a tag per row in lead_flag [16]. tag =
  // (*lead_step + 1) * 1024 + lead_tag (lead_tag in 1..1023: the launch's index inside the step;
  // lead_step: the device-side step counter, or nullptr): unique per launch as long as lead_flag is
  // zeroed whenever the host rewinds the step counter.
  int8_t* lead_q = nullptr;
  float* lead_scale = nullptr;
  unsigned* lead_flag = nullptr;
  const int32_t* lead_step = nullptr;
  unsigned lead_tag = 0;
  int lead_headstart = 6;  // x 512 clocks before the other workgroups ask for their weight ring
  // gf.slots != nullptr: no logits store, fused greedy arg-max (gemm_s8_argmax_applicable shapes)
  GreedyFuse gf;
};

bool gemm_s8_norm_applicable(int64_t m, int64_t n, int64_t k);
bool gemm_s8_qol_applicable(int64_t m, int64_t n, int64_t k);
bool gemm_s8_qol_enabled();  // CT2X_QUANT_ON_LOAD=0 restores the Quantize launch (bit-identical, same speed)
bool gemm_s8_norm_fusion_enabled();  // CT2X_NORM_FUSION=1 (off by default: measured slower)

// C[m,n] = A[m,k] . B[n,k]^T   (lda = ldb = k). Chooses a tile shape from (m, n, k).
void launch_gemm_s8(const int8_t* a, const int8_t* b, int64_t m, int64_t n, int64_t k,
                    const GemmS8Epilogue& ep, hipStream_t stream);

// 256x256-tile DMA-staged kernel for MFMA-bound shapes (gemm_s8_big.hip); launch_gemm_s8 picks
// it when gemm_s8_big_applicable (env CT2X_NO_BIG_GEMM=1 disables, for A/B comparisons).
bool gemm_s8_big_applicable(int64_t m, int64_t n, int64_t k);
// Round-4 form of the same kernel (gemm_s8_big3.hip: 32x32x32 MFMA, swapped operands for 16-byte
// stores, persistent tile stream); launch_gemm_s8_big forwards to it.
void launch_gemm_s8_big3(const int8_t* a, const int8_t* b, int M, int N, int K,
                         const GemmS8Epilogue& ep, hipStream_t stream);
// Tests: the reduced division of the exponential activations' epilogue against `/` on n pairs
// (gemm_s8_big3.hip div_noscale); adds the number of differing results to *d_mismatches.
void launch_fast_div_check(unsigned long long n, unsigned seed, unsigned long long* d_mismatches,
                           hipStream_t stream);
// Two-workgroups-per-CU form (gemm_s8_big5.hip: 128 x 256 tiles, four waves, three-stage ring).
void launch_gemm_s8_big5(const int8_t* a, const int8_t* b, int M, int N, int K,
                         const GemmS8Epilogue& ep, hipStream_t stream);
void launch_gemm_s8_big(const int8_t* a, const int8_t* b, int M, int N, int K,
                        const GemmS8Epilogue& ep, hipStream_t stream);

// Persistent weight-streaming form for decode rows of models whose weights are not cache resident
// (gemm_s8_stream.hip); launch_gemm_s8 picks it when gemm_s8_stream_applicable
// (CT2X_STREAM_GEMM=0 disables, for A/B comparisons; bit-identical).
bool gemm_s8_stream_applicable(int64_t m, int64_t n, int64_t k, bool dual);
// Prologue forms of the streaming kernel (round 4): GemmS8Epilogue::norm_x (RMSNorm + Quantize) and
// ::a_f32 / ::a_amax (quantise on load) at K = 4096. The engine takes them only where
// gemm_s8_stream_prologue_mask() says so (CT2X_STREAM_PROLOGUE: bit 0 norm, bit 1 quantise on load;
// default 0 = measured not faster, see gemm_s8_stream.hip).
bool gemm_s8_stream_norm_applicable(int64_t m, int64_t n, int64_t k, bool dual, bool has_beta);
bool gemm_s8_stream_qol_applicable(int64_t m, int64_t n, int64_t k);
int gemm_s8_stream_prologue_mask();
int gemm_s8_stream_set_prologue_mask(int mask);
bool gemm_s8_stream_lead_norm_applicable(int64_t m, int64_t n, int64_t k, bool dual, bool has_beta);
bool gemm_s8_stream_lead_quant_applicable(int64_t m, int64_t n, int64_t k);
void launch_gemm_s8_stream(const int8_t* a, const int8_t* b, int M, int N, int K,
                           const GemmS8Epilogue& ep, hipStream_t stream);
// Fragment-major copy of a weight for the streaming form (GemmS8Epilogue::packed_b): dst [n * k]
// bytes. `dual` = the weight is one of a gated pair (the pair's plan differs from the single one).
bool gemm_s8_stream_pack_applicable(int64_t n, int64_t k, bool dual);
void launch_pack_stream_weight(const int8_t* src, int64_t n, int64_t k, bool dual, int8_t* dst,
                               hipStream_t stream);
// ---- cluster-resident decoder stack (decoder_cluster.hip) -------------------------------------
// All decoder layers of one greedy step as ONE launch: a batch row is owned by a cluster of eight
// workgroups (one per attention head) that meet six times per layer through memory-side atomics.
constexpr int kClusterMembers = 8;
struct ClusterDenseW {
  const int8_t* w = nullptr;      // launch_pack_cluster_weight layout
  const float* wscale = nullptr;  // [n], the Dense's own order
  const float* bias = nullptr;    // [n] or nullptr
};
struct ClusterNormW {
  const float* gamma = nullptr;
  const float* beta = nullptr;    // nullptr: RMSNorm
  float eps = 1e-5f;
  int rms_use_residual = 0;
};
struct ClusterLayerArgs {
  ClusterNormW ln_self, ln_cross, ln_ffn;
  ClusterDenseW qkv, self_out, cross_q, cross_out, ffn0, ffn1;
  float* self_k = nullptr;         // [rows, heads, t_max, d_head]
  float* self_v = nullptr;
  const float* cross_kv = nullptr; // [batch, s_max, 2 * d_model] (K | V)
  float self_scale = 1.f, cross_scale = 1.f;
};
constexpr int kClusterMaxLayers = 8;
// The vocabulary projection + the greedy arg-max as the TAIL of the one-row launch (round 6): the
// launch's workgroups (rows * 8, all resident) each take cols_per_wg columns of the projection -
// their weight rows travel into LDS by DMA under the last layer's last meeting -, the quantised
// output rows of all clusters are exchanged through tagged slots, and the GreedyFuse machinery of
// gemm_s8.hip finishes the step: the step is ONE launch.
constexpr int kClusterProjCols = 128;      // columns of a pass: eight waves x one 16-column MFMA tile
constexpr int kClusterProjMaxPasses = 4;
constexpr int kClusterProjRowSlots = 512 / 4 + 1;  // a row's 128 x 4 int8 + its scale, 8-byte tagged slots
struct ClusterProjArgs {
  const int8_t* w = nullptr;       // [vocab, 512] int8, the Dense's own row-major layout; nullptr = off
  const float* wscale = nullptr;   // [vocab]
  const float* bias = nullptr;     // [vocab] or nullptr
  int vocab = 0;
  int cols_per_wg = 0;             // a multiple of kClusterProjCols; rows * 8 * cols_per_wg >= vocab
  unsigned long long* rows_q = nullptr;  // [rows, kClusterProjRowSlots], zero at session start
  unsigned* seq = nullptr;         // launches of the session so far (the slots' tag), zero at session start
  // Vocabularies of at most 65536 entries: the row maxima of the workgroups meet on a BOARD of tagged
  // 8-byte slots [rows][workgroups] = {order of the value : 32 | 0xFFFF - column : 16 | tag : 16} -
  // plain stores, and workgroup r (< rows) polls row r's slots and finishes that row: one store-to-
  // load trip instead of a returning atomic + the ticket + the last workgroup's exchanges.
  // nullptr: the GreedyFuse slots / ticket form (gemm_s8.hip's).
  unsigned long long* board = nullptr;  // [rows * rows * 8], zero at session start
  GreedyFuse gf;
};
struct ClusterStackArgs {
  // by value: kernel arguments are read with scalar loads the compiler knows to be invariant (a
  // table in device memory costs a vector load + a full drain of the memory pipe per field)
  ClusterLayerArgs layers[kClusterMaxLayers];
  int n_layers = 0, rows = 0, t_max = 0, s_max = 0, t_pad = 0, quant_mode = 0, act = 0;
  // beam search (beam > 1): a cluster owns a batch ENTRY with its `beam` rows (<= 4); the
  // self-attention follows the ancestry table [t_max + 1][rows]; one_row_step = the search's first
  // step (the reference has not expanded the batch there: one-row attention forms)
  int beam = 1, one_row_step = -1;
  const int32_t* ancestry = nullptr;
  // > 0: the several-rows launch - a cluster owns rows_per_cluster (<= 4) rows: the hypotheses of a
  // batch entry (beam search, = beam) or that many independent rows of a greedy batch of more than
  // 32 rows (ceil(rows / rows_per_cluster) <= 32 clusters)
  int rows_per_cluster = 0;
  const void* emb_table = nullptr;
  int emb_int8 = 0, emb_apply_mul = 0;
  const float* emb_scale = nullptr;
  float emb_mul = 1.f;
  const float* pos_table = nullptr;     // [positions, d_model] or nullptr
  const int32_t* ids = nullptr;         // [rows] input ids of the step
  const int32_t* step_ptr = nullptr;
  const int32_t* mem_lengths = nullptr; // [rows]
  float* x_out = nullptr;               // [rows, d_model]: the stack's output (before the output norm)
  // q_out != nullptr: the launch also runs the decoder's output norm + ops::Quantize on that row
  // (the input of the vocabulary projection): q_out [rows, d_model] int8, qscale_out [rows]
  ClusterNormW out_norm;
  int8_t* q_out = nullptr;
  float* qscale_out = nullptr;
  ClusterProjArgs proj;                 // one-row launch, tagged hand-over only
  float* xch = nullptr;                 // decoder_cluster_xch_floats(rows): the hand-over lines
  unsigned* counters = nullptr;         // [rows * 16], zero at session start, never reset
  int* error = nullptr;                 // set when a member waited in vain (results void)
  // timeline of workgroup 0 (CT2X_CLUSTER_STAMPS=1): 1 + 14 per layer + 1 wall_clock64 values
  unsigned long long* stamps = nullptr;
};
constexpr int kClusterStampsPerLayer = 28;  // 14 phase stamps + 8 inside the self sub-layer + 6 poll counts (decoder_cluster.hip)
// d_model 512, 8 heads of 64, feed-forward 2048 (Transformer-base); everything else keeps the
// launch-per-op path
bool decoder_cluster_geometry_ok(int64_t d_model, int64_t heads, int64_t dhead, int64_t ffn);
size_t decoder_cluster_xch_floats(int64_t rows);
int decoder_cluster_t_pad(int64_t t_max, int64_t s_max);
bool decoder_cluster_lds_ok(int t_pad);  // the score row of the longest cache fits next to the tiles
// the projection tail: cols_per_wg for (rows, vocab, k), 0 = not served (k != 512, too many passes,
// the counter form of the hand-over)
int decoder_cluster_proj_cols(int64_t rows, int64_t vocab, int64_t k);
bool decoder_cluster_beam_lds_ok(int t_pad);
size_t decoder_cluster_beam_xch_floats(int64_t clusters);
// member-major, chunk-major copy of a Dense weight [n, k]; sections = 3 for the fused Q | K | V
void launch_pack_cluster_weight(const int8_t* w, int64_t n, int64_t k, int sections, int8_t* packed,
                                hipStream_t stream);
void launch_decoder_cluster(const ClusterStackArgs& a, hipStream_t stream);

// Warm the memory-side cache with (a leading fraction of every wave's stream of) a packed weight:
// up to two ranges carried by spare workgroups of a row-wide launch (launch_norm,
// launch_quantize_cols) while HBM is otherwise idle. Piece c of a range is the (c % take16)-th
// 16-byte piece of unit c / take16; units are unit16 pieces apart.
// Fills one range for `bytes` of the fragment-major weight [n, k] (false: no streaming form).
bool weight_prefetch_range(const int8_t* packed, int64_t n, int64_t k, bool dual, int64_t bytes,
                           PrefetchRange* out);

// ---- AWQ int4 weights (awq.hip; ops::DequantizeAwq / GemmAwq / GemvAwq, src/ops/awq/*) ----------
// GEMM layout: qweight [in_c, out_c / 8], scales [in_c / G, out_c] f16, qzeros [in_c / G, out_c / 8].
// bf16 = true (round 6): x / scales / 16-bit outputs are bfloat16 (the arm the reference lacks).
void launch_awq_dequantize_gemm(const int32_t* qweight, const void* scales_f16, const int32_t* qzeros,
                                int64_t in_c, int64_t out_c, int64_t group, void* out_f16,
                                hipStream_t stream, bool bf16 = false);
// Split-K workspace of launch_awq_gemm (floats): pass one to keep the launch allocation free.
constexpr size_t kAwqGemmWorkspaceFloats = size_t(1) << 20;
void launch_awq_gemm(const void* x_f16, const int32_t* qweight, const void* scales_f16,
                     const int32_t* qzeros, int64_t m, int64_t in_c, int64_t out_c, int64_t group,
                     void* y_f16, float* y_f32, hipStream_t stream, float* slices_ws = nullptr,
                     bool bf16 = false);
// GEMV layout: qweight [out_c, in_c / 8], scales [out_c, sf_w] f16, qzeros [out_c, zeros_w].
void launch_awq_dequantize_gemv(const int32_t* qweight, const void* scales_f16, const int32_t* qzeros,
                                int64_t out_c, int64_t in_c, int64_t group, int64_t zeros_w,
                                int64_t sf_w, void* out_f16, hipStream_t stream, bool bf16 = false);
void launch_awq_gemv(const void* x_f16, const int32_t* qweight, const void* scales_f16,
                     const int32_t* qzeros, int64_t m, int64_t in_c, int64_t out_c, int64_t group,
                     int64_t zeros_w, int64_t sf_w, void* y_f16, float* y_f32, hipStream_t stream,
                     bool bf16 = false);

// ---- fp32 GEMM (gemm_f32.hip) --------------------------------------------------------

// C[b] = alpha * A[b] . op(B[b]) (+ bias[j]) -> act -> (+ residual). batch==1 for plain GEMM.
void launch_gemm_f32(const float* a, const float* b, int64_t batch, int64_t m, int64_t n,
                     int64_t k, bool trans_b, float alpha, int64_t stride_a, int64_t stride_b,
                     int64_t stride_c, const float* bias, const float* residual, int act,
                     float* c, hipStream_t stream);

// ---- float16 / bfloat16 weight GEMM (gemm_h.hip) -------------------------------------

// splitk_ws (optional, caller owned, splitk_floats floats): lets launches with too few 128 x 128 tiles
// for the chip split K over gridDim.z and reduce in a second launch (deterministic order).
// tickets (optional, n_tickets counters, zero before the first launch and left zero by every launch):
// decode rows (m <= 64) sum their K slices inside the launch instead (the last workgroup of a
// column tile to arrive adds them in the same order).
// y = act(a[m,k] (fp32, rounded to the weight type at the matrix pipe) . w[n,k]^T + bias) + residual,
// fp32 accumulation on v_mfma_f32_16x16x32_{f16,bf16}. k % 8 == 0.
void launch_gemm_h(const float* a, const void* w, bool bf16, int64_t m, int64_t n, int64_t k,
                   const float* bias, const float* residual, int act, float* y,
                   hipStream_t stream, float* splitk_ws = nullptr, size_t splitk_floats = 0,
                   unsigned* tickets = nullptr, size_t n_tickets = 0);

// ---- attention (attention.hip) -------------------------------------------------------

struct AttnDecodeParams {
  // Queries: q + row*q_row_stride + head*dhead.
  const float* q = nullptr;
  int64_t q_row_stride = 0;
  // New K/V of this step (self-attention only; nullptr for cross-attention):
  // k_new + row*new_row_stride + head_kv*dhead. They are appended to the cache at position
  // `pos` by the wave of the first query head of each KV head.
  const float* k_new = nullptr;
  const float* v_new = nullptr;
  int64_t new_row_stride = 0;
  // Caches: k_cache + kvrow*kv_row_stride + head_kv*kv_head_stride + t*kv_time_stride.
  float* k_cache = nullptr;
  float* v_cache = nullptr;
  int64_t kv_row_stride = 0, kv_head_stride = 0, kv_time_stride = 0;
  // Number of valid cached positions (excluding the new one): lengths[kvrow] if given, else
  // *step_ptr if given, else `length`.
  const int32_t* lengths = nullptr;
  const int32_t* step_ptr = nullptr;
  int64_t length = 0;
  int64_t rows = 0, heads = 0, heads_kv = 0, dhead = 0;
  int64_t kv_group = 1;  // rows sharing one cache row (beam over encoder memory)
  int64_t t_max = 0;     // bound on positions (LDS sizing)
  float scale = 1.f;
  // Optional RoPE applied to q and k_new at position `pos` (non-interleaved or interleaved),
  // tables [positions, rot_dim].
  const float* rope_sin = nullptr;
  const float* rope_cos = nullptr;
  int64_t rot_dim = 0;
  bool rope_interleave = false;
  float* out = nullptr;  // [rows, heads*dhead]
  // attention_decode2 split mode only: per-row absmax of the context, max-combined into
  // row_amax_out[rows] (zero beforehand); attention_decode_emits_row_amax tells whether it will.
  float* row_amax_out = nullptr;
  // attention_exact128 only: > 0 = row_amax_out is [rows, row_amax_parts] and workgroup (row, kv
  // head) STORES its partial maximum at [row, kv head] (no atomics, nothing to zero): 1024 device
  // scope atomicMax on the one cache line of a 16-row vector cost this launch 12 us
  // (profiles/r03_attn128_probe_g.txt); the consumer (launch_quantize_cols) folds the parts.
  int row_amax_parts = 0;
  // attention_exact only: while *one_row_step_ptr == one_row_step the rows of a kv group are
  // computed with the one-query-row forms of the reference's MatMul (first step of a CPU beam
  // search: the batch is not expanded yet, src/decoding.cc:453-454).
  const int32_t* one_row_step_ptr = nullptr;
  int32_t one_row_step = 0;
  // attention_exact only - beam search without moving the caches (the reference reorders every
  // cache tensor with ops::Gather after each step, src/layers/decoder.cc:33-56): position t of
  // this row's history lives in cache row ancestry[t * ancestry_stride + row]; the new key / value
  // is appended to the row's own cache row. nullptr: the row's own cache row holds everything.
  const int32_t* ancestry = nullptr;
  int64_t ancestry_stride = 0;
  // 16-bit compute types have no CPU counterpart to reproduce bit for bit: the exact-order kernels
  // are skipped in favour of the fast ones
  bool allow_inexact = false;
};

bool attention_decode_emits_row_amax(const AttnDecodeParams& p);

void launch_attention_decode(const AttnDecodeParams& p, hipStream_t stream);

// Same attention with ops::Quantize of the combined-heads context fused in (one workgroup per
// row): writes q_out [rows, heads*dhead] int8 + q_scales [rows] instead of p.out.
bool attention_decode_can_fuse_quantize(const AttnDecodeParams& p);
void launch_attention_decode_quantized(const AttnDecodeParams& p, int quant_mode, int8_t* q_out,
                                       float* q_scales, hipStream_t stream);

// Row-owning Dense(out) + residual + next norm + Quantize in one launch (rowblock.hip).
bool rowgemv_ln_applicable(int64_t rows, int64_t n, int64_t k);
void launch_rowgemv_ln(const int8_t* qa, const float* a_scale, const int8_t* w, const float* w_scale,
                       const float* bias, const float* residual, float* y, int64_t rows, int64_t n,
                       int64_t k, const float* gamma, const float* beta, float eps,
                       bool rms_use_residual, int quant_mode, int8_t* q_out, float* q_scale_out,
                       hipStream_t stream);

// Attention that rounds exactly like the reference CPU path (attention_exact.hip: MKL's sgemm
// summation trees, the AVX2 exp polynomial and sum order of cpu::softmax); d_head 64, no RoPE.
// launch_attention_decode / _quantized / _full route to it whenever it applies
// (CT2X_FAST_ATTENTION=1 switches it off for comparison runs).
bool attention_exact_decode_applicable(const AttnDecodeParams& p, bool fused_quantize);
void launch_attention_exact_decode(const AttnDecodeParams& p, int quant_mode, int8_t* q_out,
                                   float* q_scales, hipStream_t stream);

// d_head 128 (attention_exact128.hip; Llama-3 geometry): self-attention decode with MKL's one-row
// forms on 16-lane DPP rows, one workgroup per (row, KV head), optional producer-side row absmax.
bool attention_exact128_decode_applicable(const AttnDecodeParams& p);
void launch_attention_exact128_decode(const AttnDecodeParams& p, hipStream_t stream);

// Second-generation decode kernel (attention_decode2.hip: lane = key for QK^T, lane = channel for
// PV, GQA group per wave); launch_attention_decode[_quantized] route to it when applicable.
bool attention_decode2_applicable(const AttnDecodeParams& p, bool fused_quantize);
void launch_attention_decode2(const AttnDecodeParams& p, int quant_mode, int8_t* q_out,
                              float* q_scales, hipStream_t stream);

// Full (prefill / encoder) attention: q [B, Tq, H, dh] (row stride given), k/v with strides,
// lengths[b] valid keys, causal => key t visible to query i iff t <= i + q_offset.
struct AttnFullParams {
  const float* q = nullptr;
  const float* k = nullptr;
  const float* v = nullptr;
  int64_t q_batch_stride = 0, q_time_stride = 0;   // head stride = dhead
  int64_t k_batch_stride = 0, k_time_stride = 0, k_head_stride = 0;
  int64_t v_batch_stride = 0, v_time_stride = 0, v_head_stride = 0;
  const int32_t* lengths = nullptr;  // [B] valid key count (nullptr = tk)
  int64_t batch = 0, tq = 0, tk = 0, heads = 0, heads_kv = 0, dhead = 0;
  bool causal = false;
  int64_t q_offset = 0;
  float scale = 1.f;
  float* out = nullptr;  // [B, Tq, H*dh]
  int64_t out_batch_stride = 0, out_time_stride = 0;
  // 16-bit compute types: no CPU order to reproduce - skip the exact-order kernels (which
  // materialise the scores) for the tiled fp32-MFMA one
  bool allow_inexact = false;
  // 1 = float16, 2 = bfloat16 compute: the tiled kernel runs on the 16-bit matrix pipe with every
  // fp32 operand split into hi + lo (attention_prefill.hip); 0 = the fp32 pipe
  int half_kind = 0;
};

void launch_attention_full(const AttnFullParams& p, hipStream_t stream);

// ops::MatMul with MKL's summation order per output element (one thread per output; the unfused
// Level-1 front-end calls MatMul / SoftMax / MatMul as separate ops).
bool matmul_exact_applicable(int64_t m, int64_t n, int64_t k, bool trans_b);
void launch_matmul_exact(const float* a, const float* b, int64_t batch, int64_t m, int64_t n,
                         int64_t k, bool trans_b, float alpha, float* c, hipStream_t stream);
bool attention_exact_full_applicable(const AttnFullParams& p);
// Any head size / length, on the fp32 matrix cores (attention_seq_exact.hip): bit for bit the
// reference's MatMul -> SoftMax -> MatMul with MKL's several-row forms (alpha placement, k partition).
bool mkl_sequence_packs_b(int64_t m, int64_t n, int64_t k, float alpha);
bool attention_seq_exact_applicable(const AttnFullParams& p);
void launch_attention_seq_exact(const AttnFullParams& p, hipStream_t stream);
void launch_attention_exact_full(const AttnFullParams& p, hipStream_t stream);

// Tiled form for d_head 64 / 128 (attention_prefill.hip: 64-query tiles, K/V tiles shared through
// LDS, both products on v_mfma_f32_16x16x4_f32, online softmax); launch_attention_full routes to it.
bool attention_prefill_applicable(const AttnFullParams& p);
void launch_attention_prefill(const AttnFullParams& p, hipStream_t stream);

// Prefill helper: for every (row = (b, t), head) of a fused QKV projection [R, q_dim + 2*kv_dim]
// (R = batch * time): q -> q_out [R, q_dim] (RoPE applied at position pos0 + t when tables are
// given), k -> cache_k[b, hk, pos0 + t, :] (RoPE applied), v -> cache_v[b, hk, pos0 + t, :].
// Replaces split_heads + Rotary + the cache Concat of sequence-mode decoding
// (src/layers/attention.cc:514-546).
void launch_qkv_to_cache(const float* qkv, int64_t batch, int64_t time, int64_t heads,
                         int64_t heads_kv, int64_t dhead, int64_t t_max, int64_t pos0,
                         const float* rope_sin, const float* rope_cos, int64_t rot_dim,
                         bool interleave, float* q_out, float* cache_k, float* cache_v,
                         hipStream_t stream);

// Beam reorder of pre-allocated caches: for each tensor t in [0, n_tensors):
//   dst[t][r, :, 0:len, :] = src[t][parent[r], :, 0:len, :]
// `tensors_src/dst` are device arrays of base pointers.
void launch_cache_gather(const float* const* tensors_src, float* const* tensors_dst,
                         int n_tensors, const int32_t* parent, int64_t rows, int64_t heads,
                         int64_t t_max, int64_t dhead, const int32_t* step_ptr, int64_t length,
                         hipStream_t stream);

}  // namespace ct2x

// ---- decode-loop helpers (decode_ops.hip) ---------------------------------------------
namespace ct2x {

// Device-resident greedy step: DisableTokens(min length) + [LogSoftMax] + TopK(k=1) + id feed
// for the next step + history append, one launch (src/decoding.cc:852-880, src/sampling.cc:25-32).
// step = *step_ptr. The disabled token (if step < min_length) is treated as -FLT_MAX.
void launch_greedy_sample(const float* logits, int64_t rows, int64_t vocab, int32_t disable_id,
                          int32_t min_length, bool with_scores, const int32_t* step_ptr,
                          int32_t* next_ids, int32_t* hist_ids, float* hist_scores,
                          hipStream_t stream, const int32_t* prefix_ids = nullptr,
                          const int32_t* prefix_lens = nullptr, int32_t prefix_stride = 0,
                          int32_t prefix_first_step = 0, bool prefix_extends_min = true,
                          int32_t* advance_step_ptr = nullptr, int32_t* done_counter = nullptr,
                          float* exp_scratch = nullptr /* [rows, vocab], needed with_scores */);

// return_logits_vocab: hist[*step_ptr % ring][row][0 .. out_vocab) = the row's logits with the
// end token masked (lowest()) below the row's minimum length (what the samplers mask on the fly)
void launch_record_logits(const float* logits, int64_t rows, int64_t vocab, int64_t out_vocab,
                          int32_t disable_id, int32_t min_length, const int32_t* prefix_lens,
                          bool prefix_extends_min, const int32_t* step_ptr, int32_t ring, float* hist,
                          hipStream_t stream);

// *step_ptr += 1
void launch_advance_step(int32_t* step_ptr, hipStream_t stream);

// RandomSampler (src/sampling.cc:41-101), one sample per row: top-k restriction (1 < topk <=
// max_sampling_topk()), temperature, top-p mask, softmax, multinomial draw from a counter-based
// hash of (seed, step, row). Same outputs as launch_greedy_sample.
void launch_random_sample(const float* logits, int64_t rows, int64_t vocab, int32_t disable_id,
                          int32_t min_length, bool with_scores, int topk, float topp,
                          float temperature, uint32_t seed, const int32_t* step_ptr,
                          int32_t* next_ids, int32_t* hist_ids, float* hist_scores,
                          hipStream_t stream, const int32_t* prefix_ids = nullptr,
                          const int32_t* prefix_lens = nullptr, int32_t prefix_stride = 0,
                          int32_t prefix_first_step = 0, bool prefix_extends_min = true,
                          int32_t* advance_step_ptr = nullptr, int32_t* done_counter = nullptr,
                          const float* uniforms = nullptr /* tests: [rows] draws in [0, 1) */);
int max_sampling_topk();

// ops::TopPMask / ops::Multinomial as standalone ops (the Level-1 stubs; the engine fuses both
// into the samplers above).
void launch_topp_mask(const float* x, const float* probs, int64_t rows, int64_t depth, float p,
                      float mask, float* y, hipStream_t stream);
// primitives::convert between CT2X_F32 / CT2X_F16 / CT2X_BF16 (the values of ct2x_dtype).
void launch_convert(const void* x, int src_dtype, void* y, int dst_dtype, int64_t n,
                    hipStream_t stream);
// Ops outside the hot path (primitives.hip): Whisper's Conv1D / MedianFilter, ALiBi bias.
void launch_conv1d(const float* x, const float* w, int64_t batch, int64_t in_ch, int64_t len,
                   int64_t out_ch, int64_t kernel, int64_t stride, int64_t padding, int64_t dilation,
                   int64_t groups, float* y, hipStream_t stream);
void launch_median_filter(const float* x, int64_t rows, int64_t depth, int64_t width, float* y,
                          hipStream_t stream);
void launch_alibi_add(const float* x, const float* alibi, int64_t batch, int64_t heads,
                      int64_t queries, int64_t keys, int64_t alibi_keys, int64_t offset, float* y,
                      hipStream_t stream);
void launch_gumbel_noise(const float* x, float* y, int64_t n, uint32_t seed, uint32_t call,
                         hipStream_t stream);
void launch_multinomial(const float* weights, int64_t rows, int64_t classes, int64_t samples,
                        uint32_t seed, uint32_t call, int32_t* out, hipStream_t stream);

// RepetitionPenalty / NoRepeatNgram / SuppressSequences (src/decoding_utils.cc:37-153) in place on
// the step's logits, one launch. seq[t * t_stride + row * row_stride] = token t of the row's
// generated history; its length is *step_ptr - first_step (device resident) or `length`.
// max_length bounds the history (sizes the gather buffer, <= 16000).
void launch_process_logits(float* logits, int64_t rows, int64_t vocab, const int32_t* seq,
                           int64_t t_stride, int64_t row_stride, const int32_t* step_ptr,
                           int32_t first_step, int32_t length, int32_t max_length, float penalty,
                           int32_t ngram, const int32_t* sup_ids, const int32_t* sup_off,
                           int32_t n_sup, hipStream_t stream);

// out[row] = log_softmax(logits[row, :])[targets[row]]  (score_sequences, src/scoring.cc:42-48)
void launch_score_tokens(const float* logits, int64_t rows, int64_t vocab, const int32_t* targets,
                         float* out, hipStream_t stream);

// One beam-search step after TopK on the device (src/decoding.cc:559-700): unflatten, hard
// prefix, record the candidates (hist_*[step - first_step][batch, 2*beam]) and install the next
// step's ids / cumulative scores / parent rows. 2 * beam <= 64. seq_src/seq_dst (optional):
// the tokens generated by each active beam, [batch*beam, seq_stride], ping-pong.
void launch_beam_select(const float* topk_scores, const int32_t* topk_ids, int64_t batch,
                        int32_t beam, int64_t vocab, int32_t eos, const int32_t* prefix_ids,
                        const int32_t* prefix_lens, int32_t prefix_stride, const int32_t* step_ptr,
                        int32_t first_step, int32_t* hist_word, float* hist_score,
                        int32_t* hist_origin, int32_t* next_ids, int32_t* parents,
                        float* beam_scores, const int32_t* seq_src, int32_t* seq_dst,
                        int32_t seq_stride, hipStream_t stream,
                        const int32_t* ancestry_src = nullptr, int32_t* ancestry_dst = nullptr,
                        uint8_t* diverged = nullptr /* biased decoding: [batch * beam] flags */,
                        // round 6: merge the rows' candidates of launch_beam_logprobs here
                        // ([batch, n_part] with n_part = beam * 2 * beam <= 64) instead of reading
                        // topk_scores / topk_ids; advance *advance_step by one when the last
                        // workgroup finishes (ticket: one zeroed word, left zero)
                        const float* part_v = nullptr, const int32_t* part_i = nullptr,
                        int32_t n_part = 0, int32_t* advance_step = nullptr, unsigned* ticket = nullptr);

// return_attention: out[*step_ptr][row][0 .. s_max) = mean over the first align_heads heads of
// softmax(q . k * scale) against the projected memory keys (decode_ops.hip).
void launch_attention_probs(const float* q, int64_t q_row_stride, const float* k, int64_t k_batch_stride,
                            int64_t k_time_stride, const int32_t* lengths, int64_t rows,
                            int32_t kv_group, int32_t heads, int32_t heads_kv, int32_t dhead,
                            int32_t align_heads, int32_t s_max, float scale, const int32_t* step_ptr,
                            float* out, hipStream_t stream);

// Beam-search scores: y[row, :] = log_softmax(x[row, :] with x[disable_id] = -FLT_MAX while
// step < min_length) + beam_scores[row]   (src/decoding.cc:512-556).
// extra_min_per_batch (optional, [rows / beam]): added to min_length for the rows of that batch
// entry (hidden hard prefix, apply_min_length with return_prefix = false).
void launch_beam_logprobs(const float* logits, const float* beam_scores, int64_t rows,
                          int64_t vocab, int32_t disable_id, int32_t min_length,
                          const int32_t* step_ptr, float* y, hipStream_t stream,
                          const int32_t* extra_min_per_batch = nullptr, int32_t beam = 1,
                          bool exact = true,
                          // biased decoding (BiasedDecoder, src/decoding.cc:256-323): bias_beta > 0
                          const int32_t* bias_ids = nullptr, const int32_t* bias_lens = nullptr,
                          int32_t bias_stride = 0, int32_t bias_first_step = 0, float bias_beta = 0.f,
                          const uint8_t* diverged = nullptr,
                          // round 6 (beam_logprobs_topk_applicable): the row's `topk` best scores as
                          // candidates part_v / part_i [rows, topk] (flat index (row % beam) * vocab +
                          // word) INSTEAD of the row y (which is then not written)
                          float* part_v = nullptr, int32_t* part_i = nullptr, int32_t topk = 0);
bool beam_logprobs_topk_applicable(int64_t vocab, bool exact, int32_t beam, int32_t topk);

}  // namespace ct2x

// ---- element-wise / layout members of primitives<Device::CUDA> (primitives.hip) -------------
namespace ct2x {

// op: 0 add, 1 sub, 2 mul, 3 min, 4 max. dtype: ct2x_dtype (float32, int8, int16, int32).
void launch_binary(int op, const void* a, const void* b, void* c, int64_t size, int dtype,
                   hipStream_t stream);
// y = x op scalar (add, mul, min, max)
void launch_binary_scalar(int op, double scalar, const void* x, void* y, int64_t size, int dtype,
                          hipStream_t stream);
// kind 0 batch (c[i] = a[i % a_size] op b[i]), 1 depth (a[i / (b_size / a_size)]), 2 block
// (a[(i / block) % a_size]); op add or mul.
void launch_broadcast(int op, int kind, const void* a, const void* b, void* c, int64_t a_size,
                      int64_t b_size, int64_t block, int dtype, hipStream_t stream);
void launch_fill(void* x, double value, int64_t size, int64_t stride, int dtype, hipStream_t stream);
void launch_indexed_fill(void* x, double value, const int32_t* indices, int64_t n, int dtype,
                         hipStream_t stream);
// fn 0..6 = ops::ActivationType, 7 exp, 8 log, 9 sin, 10 cos
void launch_unary(int fn, const float* x, float* y, int64_t size, hipStream_t stream);
// Concat / Split / Slide / Tile as strided block copies; Mean / Sum over the middle axis.
void launch_block_copy(const void* src, void* dst, int64_t outer, int64_t block_bytes,
                       int64_t src_stride_bytes, int64_t dst_stride_bytes, hipStream_t stream);
void launch_mean(const float* x, int64_t outer, int64_t axis, int64_t inner, bool get_sum, float* y,
                 hipStream_t stream);
// b = transpose(a reshaped to dims[rank], perm); rank 2..4; dims / perm are HOST arrays
void launch_transpose(const void* a, const int64_t* dims, const int64_t* perm, int rank, void* b,
                      int dtype, hipStream_t stream);
void launch_prepare_length_mask(const int32_t* lengths, int64_t batch, int64_t heads,
                                int64_t queries, bool mask_future, bool multi_query,
                                int32_t* mask, hipStream_t stream);

// ---- Whisper glue (whisper_ops.hip; src/models/whisper.cc, src/layers/whisper.cc) -------------
// im2col of a kernel-3 convolution: out [batch * out_len, channels * 3] (column = c * 3 + k), from
// x [batch, channels, length] or, time_major, [batch, length, channels].
void launch_im2col_k3(const float* x, int64_t batch, int64_t channels, int64_t length,
                      int64_t out_len, int stride, int padding, bool time_major, float* out,
                      hipStream_t stream);
// x [batch, time, depth] += table [time, depth]
void launch_add_positions(float* x, const float* table, int64_t batch, int64_t time, int64_t depth,
                          hipStream_t stream);
// GetNoSpeechProbs (whisper.cc:202-231): softmax(logits)[no_speech_id] per row, only at the
// first step of the decoding loop (*step_ptr == first_step).
void launch_no_speech_prob(const float* logits, int64_t rows, int64_t vocab, int32_t no_speech_id,
                           const int32_t* step_ptr, int32_t first_step, float* out,
                           hipStream_t stream);
// SuppressTokensBegin (decoding_utils.cc:172-194) + ApplyTimestampRules (whisper.cc:741-861) in
// place on the step's logits. History addressing as in launch_process_logits; prefix_len (optional)
// holds the hard-prefix length of entry row / prefix_div (the rule's sample_begin).
struct WhisperRulesParams {
  float* logits = nullptr;
  int64_t rows = 0, vocab = 0;
  const int32_t* seq = nullptr;
  int64_t t_stride = 0, row_stride = 0;
  const int32_t* step_ptr = nullptr;
  int32_t first_step = 0;
  const int32_t* prefix_len = nullptr;
  int32_t prefix_div = 1;
  const int32_t* begin_ids = nullptr;  // SuppressTokensBegin ids
  int32_t n_begin = 0;
  bool timestamp_rules = false;
  int32_t eot_id = 0, no_timestamps_id = 0, timestamp_begin = 0, timestamp_end = 0,
          max_initial_timestamp = 0;
};
void launch_whisper_rules(const WhisperRulesParams& p, hipStream_t stream);
// out [batch, n_sel, tq, tk] = scale * q[b, t, heads[i]] . k[b, s, heads[i]] (no softmax)
void launch_attention_scores(const float* q, int64_t q_batch_stride, int64_t q_time_stride,
                             const float* k, int64_t k_batch_stride, int64_t k_time_stride,
                             const int32_t* heads, int64_t batch, int64_t n_sel, int64_t tq,
                             int64_t tk, int64_t dhead, float scale, float* out,
                             hipStream_t stream);
// ops::LayerNorm(axis -2, epsilon 0), no gamma / beta, on [outer, axis, inner]; only the first
// inner_used columns are normalised.
void launch_layer_norm_axis(const float* x, int64_t outer, int64_t axis, int64_t inner,
                            int64_t inner_used, float* y, hipStream_t stream);
// softmax over the first `classes` logits of every row, probability of targets[row]
void launch_token_probs(const float* logits, int64_t rows, int64_t vocab, int64_t classes,
                        const int32_t* targets, float* out, hipStream_t stream);
// out [rows, n] = softmax_j(logits[row, ids[j]])
void launch_gather_softmax(const float* logits, int64_t rows, int64_t vocab, const int32_t* ids,
                           int64_t n, float* out, hipStream_t stream);

// scores [n_sel, tq, tk] of one batch entry -> out [n_sel, rows_out, frames]: softmax over the
// first `frames` keys of row min(t, length - 1)
void launch_align_softmax(const float* scores, int64_t n_sel, int64_t tq, int64_t tk, int64_t length,
                          int64_t rows_out, int64_t frames, float* out, hipStream_t stream);

// int32 product of an int8 convolution -> y = act((c * (1 / w_scale[col])) / x_scale[row] + bias)
void launch_conv_dequantize(const int32_t* c, const float* x_scale, const float* w_scale,
                            const float* bias, int64_t rows, int64_t n, int act, float* y,
                            hipStream_t stream);

// Encoder-decoder attention of a decode step over 16-bit memory keys / values (attention_cross_h.hip):
// kv [batch, s_max, 2 * heads_kv * 64] in float16 / bfloat16 (keys, then values, per position - the
// cross-attention K/V projection's layout), q / out fp32 rows (batch * beam hypotheses).
struct CrossAttnHParams {
  const float* q = nullptr;
  int64_t q_row_stride = 0;
  const uint16_t* kv = nullptr;
  bool bf16 = false;
  const int32_t* lengths = nullptr;  // [batch] valid memory positions (nullptr = s_max)
  int64_t batch = 0, beam = 1, heads = 0, heads_kv = 0, dhead = 0, s_max = 0;
  float scale = 1.f;
  float* out = nullptr;
  int64_t out_row_stride = 0;
  // Optional: the memory of a (batch entry, head) split over several workgroups.
  // split_scratch: cross_attention_h_scratch_floats(...) floats; split_tickets: batch * heads
  // counters, zero before the first launch (every launch leaves them zero). Both null = one
  // workgroup per (entry, head). `splits` / `chunk` are filled in by the launcher.
  float* split_scratch = nullptr;
  unsigned* split_tickets = nullptr;
  int splits = 1, chunk = 0;
};
int cross_attention_h_splits(int64_t batch, int64_t beam, int64_t heads, int64_t s_max);
int64_t cross_attention_h_scratch_floats(int64_t batch, int64_t beam, int64_t heads, int64_t dhead,
                                         int64_t s_max);
bool cross_attention_h_applicable(const CrossAttnHParams& p);
#ifdef CT2X_TUNING
void xattn_h_read_stamps(unsigned long long* host16);  // tools/xattn_probe.py
void gemm_h_rows_read_stamps(unsigned long long* host16);  // tools/gemm_rows_probe.py
#endif
void launch_cross_attention_h(const CrossAttnHParams& p, hipStream_t stream);

}  // namespace ct2x
