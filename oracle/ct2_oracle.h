/* TEST INFRASTRUCTURE ONLY -- the parity oracle. Never linked into, imported or called by the
 * product (ctranslate2_amd/): only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may use it, and only as the checker.
 *
 * Plain-C (scalar, IEEE, no fast-math, no FMA contraction) restatement of the reference's CPU
 * algorithm for the quantized decoder hot path. Each function cites the reference file:line it
 * follows. Pinned against the reference's own known-answer vectors (tests/golden/*.json,
 * transcribed from tests/ops_test.cc) and against the reference itself (oracle/_ref) by
 * tests/test_oracle.py.
 */
#ifndef CT2_ORACLE_H_
#define CT2_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ops::Quantize int8 (src/ops/quantize_cpu.cc:12-25 -> src/cpu/kernels.cc:577-651). */
void orc_quantize_s8(const float* x, int64_t rows, int64_t depth, int round_before_cast,
                     int8_t* q, float* scales);

/* primitives<CPU>::gemm<int8,int32> semantic (src/cpu/primitives.cc:901-1073; KAT
 * tests/ops_test.cc:683-708): C[m,n] = A[m,k] . B[n,k]^T exactly. */
void orc_gemm_s8(const int8_t* a, const int8_t* b, int64_t m, int64_t n, int64_t k, int32_t* c);

/* cpu::dequantize_gemm_output (src/cpu/kernels.cc:653-756). act = ops::ActivationType or -1. */
void orc_dequantize_gemm_output(const int32_t* c, const float* a_scale, const float* b_scale,
                                const float* bias, int64_t m, int64_t n, int act, float* y);

/* Dequantize rows x * (1/scale) (src/ops/dequantize_cpu.cc:10-50). */
void orc_dequantize_rows(const int8_t* x, const float* scale, int64_t rows, int64_t depth,
                         float* y);

/* sgemm semantic used by ops::Gemm float / ops::MatMul (src/cpu/primitives.cc:660-790,
 * 1085-1160): C = alpha * A . op(B); then act(x + bias + residual) (src/ops/gemm.cc:11-25). */
void orc_gemm_f32(const float* a, const float* b, int64_t m, int64_t n, int64_t k, int trans_b,
                  float alpha, const float* bias, const float* residual, int act, float* c);

/* cpu::layer_norm (src/cpu/kernels.cc:461-496) / cpu::rms_norm (:543-573). */
void orc_layer_norm(const float* x, const float* gamma, const float* beta, int64_t rows,
                    int64_t depth, float eps, float* y);
void orc_rms_norm(const float* x, const float* gamma, int64_t rows, int64_t depth, float eps,
                  int use_residual, float* y);

/* cpu::softmax (src/cpu/kernels.cc:402-459). lengths may be NULL. */
void orc_softmax(const float* x, const int32_t* lengths, int64_t rows, int64_t depth, int log,
                 float* y);

/* exp256_ps as compiled into the reference's AVX2 kernels (third_party/avx_mathfun.h:290-344). */
float orc_exp_avx2(float x);

/* ops::MatMul on CPU (MKL cblas_sgemm_batch_strided, src/cpu/primitives.cc:1085-1101) with the
 * summation order recovered from MKL 2021.4 on this container; see ct2_oracle.c. a [batch, m, k],
 * b [batch, n, k] (trans_b) or [batch, k, n]. Returns -1 when a shape is outside the validated
 * envelope (that batch entry is left untouched): one query row with trans_b needs k % 4 == 0,
 * 4 <= k <= 256; several rows with trans_b need m >= 3, n >= 5 and k < 24 m; the no-transpose
 * forms need m >= 2 or n in {4, 8} or n >= 16. alpha multiplies the finished sum (pinned for
 * the trans_b and sequence forms; the one-row no-transpose form is only pinned at alpha = 1). */
int orc_matmul(const float* a, const float* b, int64_t batch, int64_t m, int64_t n, int64_t k,
               int trans_b, float alpha, float* c);

/* rotary_kernel (src/ops/rotary_cpu.cc:9-44): x [batch, time, depth], sin/cos [time, ndims]. */
void orc_rotary(const float* x, const float* sin, const float* cos, int64_t batch, int64_t time,
                int64_t depth, int64_t ndims, int interleave, float* y);

/* TopK::compute<CPU> (src/ops/topk_cpu.cc:13-55); ties -> lowest index. */
void orc_topk(const float* x, int64_t rows, int64_t depth, int64_t k, float* values,
              int32_t* indices);

/* RepetitionPenalty / NoRepeatNgram / SuppressSequences (src/decoding_utils.cc:37-153), in place. */
void orc_process_logits(float* logits, int64_t rows, int64_t vocab, const int32_t* seq,
                        int32_t length, float penalty, int32_t ngram, const int32_t* sup_ids,
                        const int32_t* sup_off, int32_t n_sup);

/* One beam-search step after TopK (src/decoding.cc:559-700): unflatten, hard prefix, candidate
 * selection. */
void orc_beam_select(const float* topk_scores, const int32_t* topk_ids, int64_t batch,
                     int32_t beam, int32_t vocab, int32_t eos, const int32_t* prefix_ids,
                     const int32_t* prefix_lens, int32_t prefix_stride, int32_t step,
                     int32_t* cand_word, float* cand_score, int32_t* cand_origin,
                     int32_t* next_ids, float* next_scores, int32_t* parents);

/* ops::TopPMask (src/ops/topp_mask_cpu.cc:13-38): keep while the mass visited before is < p. */
void orc_topp_mask(const float* x, int64_t rows, int64_t depth, float p, float mask, float* y);

/* ops::Gather axis 0 (src/ops/gather_cpu.cc). */
void orc_gather_rows(const void* data, const int32_t* idx, int64_t n_idx, int64_t row_bytes,
                     void* out);

/* Activation functors (src/cpu/kernels.cc:146-207). */
float orc_activation(float v, int act);

/* dot_product_attention for one query per row (src/layers/attention.cc:178-287 with
 * Tq = 1): scores = scale * q.k_t over the first len positions, softmax, out = P.V.
 * q [rows, heads, dhead]; k/v [rows/kv_group, heads_kv, t_max, dhead]; lengths per kv row
 * (NULL -> `length`). */
void orc_attention_decode(const float* q, const float* k, const float* v,
                          const int32_t* lengths, int64_t length, int64_t rows, int64_t heads,
                          int64_t heads_kv, int64_t dhead, int64_t t_max, int64_t kv_group,
                          float scale, float* out);

/* exact_div of the device epilogues (device_utils.h) restated on fmaf, and its bit-for-bit
 * comparison with t / d over n pseudo-random pairs of the guarded domain (returns mismatches). */
float orc_exact_div(float t, float d);
int64_t orc_exact_div_check(int64_t n, uint64_t seed);

/* MKL VML vsSin / vsCos (HA mode, |x| <= 10000) restated from the library's disassembly; returns
 * -1 if an argument was outside that range (its y is 0). */
int orc_vml_sin(const float* x, int64_t n, float* y);
int orc_vml_cos(const float* x, int64_t n, float* y);

#ifdef __cplusplus
}
#endif
#endif
