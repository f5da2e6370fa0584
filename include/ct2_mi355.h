/*
 * ct2_mi355.h -- C ABI of the MI355X (gfx950) backend for CTranslate2's quantized
 * decoder hot path.
 *
 * The reference has no FFI for this path: its device boundary is C++ explicit template
 * specialisation (`primitives<Device::CUDA>` and `ops::X::compute<Device::CUDA, T>`,
 * selected by src/device_dispatch.h:20-32). Each entry point below is what such a
 * specialisation would call; the comment on each one names the reference symbol it
 * stands behind (file:line under /root/reference). INTEGRATION.md shows the C++ stub a
 * maintainer adds on the reference side.
 *
 * Conventions
 *  - Every pointer named d_* is a DEVICE pointer on the current HIP device; `stream`
 *    is a hipStream_t passed as void* (NULL = the library's per-thread stream, the
 *    reference's stream-per-thread contract, src/cuda/utils.cc:75-125).
 *  - All work is enqueued asynchronously on `stream`; nothing synchronises the device
 *    except ct2x_stream_synchronize / the *_to_host helpers.
 *  - Row-major, contiguous tensors; sizes are int64_t (the reference's dim_t).
 *  - Return value: 0 on success, non-zero on error (ct2x_last_error() gives the
 *    message; the C++ stub rethrows it as std::runtime_error / std::invalid_argument
 *    like CUDA_CHECK / THROW_RUNTIME_ERROR do, src/cuda/utils.h:68-96).
 *  - Float tensors are fp32 unless a `dtype` argument says otherwise
 *    (CT2X_F32 / CT2X_F16 / CT2X_BF16 mirror DataType, include/ctranslate2/types.h:16-23).
 *  - No CPU fallback exists anywhere in this library.
 */
#ifndef CT2_MI355_H_
#define CT2_MI355_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CT2X_VERSION 1

/* DataType ids = the reference's enum (include/ctranslate2/types.h:16-23) and the
 * type ids stored in model.bin (python/ctranslate2/specs/model_spec.py:_dtype_to_type_id). */
enum ct2x_dtype {
  CT2X_F32 = 0,
  CT2X_I8 = 1,
  CT2X_I16 = 2,
  CT2X_I32 = 3,
  CT2X_F16 = 4,
  CT2X_BF16 = 5,
};

/* ops::ActivationType (include/ctranslate2/ops/activation.h:9-17); CT2X_ACT_NONE = no
 * activation (a null ActivationType* in the reference). */
enum ct2x_activation {
  CT2X_ACT_NONE = -1,
  CT2X_ACT_RELU = 0,
  CT2X_ACT_GELU_TANH = 1,
  CT2X_ACT_SWISH = 2,
  CT2X_ACT_GELU = 3,
  CT2X_ACT_GELU_SIGMOID = 4,
  CT2X_ACT_TANH = 5,
  CT2X_ACT_SIGMOID = 6,
};

/* ---- runtime ------------------------------------------------------------------- */

const char* ct2x_last_error(void);
int ct2x_version(void);

/* cuda::get_gpu_count / has_gpu (src/cuda/utils.cc:163-175). */
int ct2x_device_count(void);
/* set_device_index<Device::CUDA> / get_device_index (src/devices.cc:78-88). */
int ct2x_set_device(int device_index);
int ct2x_get_device(int* device_index);
/* cuda::get_device_properties(i).name, gcnArchName, CU count, HBM bytes. */
int ct2x_device_info(int device_index, char* name, int name_size, char* arch, int arch_size,
                     int* compute_units, int64_t* hbm_bytes);
/* cuda::get_cuda_stream() (src/cuda/utils.cc:122-125): the calling thread's stream. */
void* ct2x_thread_stream(void);
/* synchronize_stream(Device::CUDA) (src/devices.cc:106-117). */
int ct2x_stream_synchronize(void* stream);
/* synchronize_device(Device::CUDA) (src/devices.cc:95-105): every stream of the current device. */
int ct2x_device_synchronize(void);

/* get_allocator<Device::CUDA>() (src/cuda/allocator.cc:34-185): stream-ordered arena.
 * allocate/free are safe to call in any order on one stream; blocks are recycled
 * without returning to the driver until ct2x_clear_cache (Allocator::clear_cache). */
int ct2x_malloc(void** d_ptr, size_t bytes);
int ct2x_free(void* d_ptr);
int ct2x_clear_cache(void);
int64_t ct2x_allocated_bytes(void);

/* cross_device_primitives<CPU,CUDA>::copy / <CUDA,CPU>::copy (src/cuda/primitives.cu:685-754).
 * The D2H form returns only after the bytes have landed (callers read the host buffer
 * immediately, src/sampling.cc:19-20 -> src/decoding.cc:900). */
int ct2x_copy_to_device(const void* h_src, void* d_dst, size_t bytes, void* stream);
int ct2x_copy_to_host(const void* d_src, void* h_dst, size_t bytes, void* stream);
int ct2x_copy_device(const void* d_src, void* d_dst, size_t bytes, void* stream);

/* HIP events on the library's stream, for benchmark timing (bench.py roofline leg). */
int ct2x_event_create(void** event);
int ct2x_event_destroy(void* event);
int ct2x_event_record(void* event, void* stream);
int ct2x_event_elapsed_ms(void* start, void* stop, float* ms); /* synchronises on stop */

/* ---- hot-path ops -------------------------------------------------------------- */

/* ops::Quantize::quantize<Device::CUDA, T, int8_t> (src/ops/quantize_gpu.cu:87-103;
 * CPU semantics src/cpu/kernels.cc:577-651): per row amax -> scale = amax ? 127/amax : 1,
 * q = int8(round_or_trunc(x*scale)). `round_before_cast` is a flag word: bit 0 = round to
 * nearest even (0 = the legacy truncation of binary_version < 5 models,
 * include/ctranslate2/models/model.h:87-89); bit 1 = reproduce the u8 domain shift of the
 * reference's x86 CPU GEMM backends, q = cvtt(round_func(x*scale + 128)) - 128
 * (shift_to_uint8, src/cpu/kernels.cc:592-604), which the engine uses for CPU token parity. */
int ct2x_quantize_s8(const void* d_x, int dtype, int64_t rows, int64_t depth,
                     int round_before_cast, int8_t* d_q, float* d_scales, void* stream);

/* primitives<Device::CUDA>::gemm<int8_t,int32_t> (src/cuda/primitives.cu:571-597), the
 * call shape Dense uses (src/layers/common.cc:287-293,392): C[m,n] = A[m,k] . B[n,k]^T,
 * alpha=1, beta=0, exact int32. */
int ct2x_gemm_s8s8s32(const int8_t* d_a, const int8_t* d_b, int64_t m, int64_t n, int64_t k,
                      int32_t* d_c, void* stream);

/* ops::Dequantize::dequantize_gemm_output<Device::CUDA, T> (src/ops/dequantize_gpu.cu:31-121)
 * with the CPU formula order (src/cpu/kernels.cc:666-673), chosen for token parity:
 *   y[i,j] = act( float(c[i,j]) * (1/a_scale[i]) / b_scale[j] + bias[j] ). */
int ct2x_dequantize_gemm_output(const int32_t* d_c, const float* d_a_scale,
                                const float* d_b_scale, const float* d_bias, int64_t m,
                                int64_t n, int activation, void* d_y, int dtype, void* stream);

/* The fused form the backend actually runs for layers::Dense on int8 weights
 * (src/layers/common.cc:353-401: Quantize -> Gemm -> Dequantize(+bias,+act) -> Add residual):
 *   y = act( (qa . W^T) * (1/a_scale) / w_scale + bias ) + residual
 * The int32 accumulator never reaches HBM. d_bias / d_residual may be NULL. */
int ct2x_gemm_s8_dequant(const int8_t* d_qa, const float* d_a_scale, const int8_t* d_w,
                         const float* d_w_scale, const float* d_bias, const void* d_residual,
                         int64_t m, int64_t n, int64_t k, int activation, void* d_y, int dtype,
                         void* stream);

/* The same Dense for the vocabulary projection of a greedy step that returns no scores
 * (src/decoding.cc:874-880: GreedySearch takes ops::TopK(1) of the raw logits): the arg-max rides in
 * the launch, the logits are never stored. d_ids_out[m] receives the index of every row's largest
 * value, the LOWEST index among equal values (-0.0 == +0.0), NaN never chosen. d_scratch: at least
 * ct2x_gemm_s8_argmax_scratch_bytes(m) bytes, zero on entry, zero again on return.
 * m <= 32, n >= 8192, k % 16 == 0, k < 4096 (an error otherwise). */
int ct2x_gemm_s8_argmax(const int8_t* d_qa, const float* d_a_scale, const int8_t* d_w,
                        const float* d_w_scale, const float* d_bias, int64_t m, int64_t n, int64_t k,
                        int32_t* d_ids_out, void* d_scratch, void* stream);
int64_t ct2x_gemm_s8_argmax_scratch_bytes(int64_t m);

/* The gated FFN input pair (FeedForwardNetwork::operator(), src/layers/transformer.cc:21-51 with
 * ffn_glu: linear_0 + activation, linear_0_noact, ops::Mul) as ONE launch for decode rows
 * (m <= 32):  y = act( (qa . W0^T) * (1/a_scale) / w0_scale + b0 ) * ( (qa . W1^T) * (1/a_scale)
 * / w1_scale + b1 ). Biases may be NULL. */
int ct2x_gemm_s8_glu(const int8_t* d_qa, const float* d_a_scale, const int8_t* d_w0,
                     const float* d_w0_scale, const float* d_b0, const int8_t* d_w1,
                     const float* d_w1_scale, const float* d_b1, int64_t m, int64_t n, int64_t k,
                     int activation, void* d_y, int dtype, void* stream);

/* ct2x_gemm_s8_glu that also leaves the row absmax of y in d_row_amax_out[m] (zero on entry;
 * max-combined): the producer half of the ops::Quantize that feeds linear_1 of the feed-forward
 * block (src/layers/transformer.cc:21-51), consumed by the engine's element-wise Quantize. */
int ct2x_gemm_s8_glu_amax(const int8_t* d_qa, const float* d_a_scale, const int8_t* d_w0,
                          const float* d_w0_scale, const float* d_b0, const int8_t* d_w1,
                          const float* d_w1_scale, const float* d_b1, int64_t m, int64_t n, int64_t k,
                          int activation, float* d_y, float* d_row_amax_out, void* stream);

/* Fragment-major copy of an int8 weight [n, k] for the persistent streaming decode GEMM (models
 * whose weights are not cache resident; gemm_s8_stream.hip): d_packed [n * k] bytes. `dual` != 0
 * packs for the gated-pair launch (ct2x_gemm_s8_glu_packed). Returns an error when (n, k) has no
 * streaming form (ct2x_gemm_s8_stream_packable() == 0). The packed weight replaces d_w in
 * ct2x_gemm_s8_dequant_packed / ct2x_gemm_s8_glu_packed (m <= 16 rows), same results bit for bit
 * as the row-major entry points (the reference keeps one row-major weight:
 * src/layers/common.cc:266-303; its CPU build packs for MKL the same way, `weight_packed`). */
int ct2x_gemm_s8_stream_packable(int64_t n, int64_t k, int dual);
int ct2x_gemm_s8_stream_pack(const int8_t* d_w, int64_t n, int64_t k, int dual, int8_t* d_packed,
                             void* stream);
int ct2x_gemm_s8_dequant_packed(const int8_t* d_qa, const float* d_a_scale, const int8_t* d_w_packed,
                                const float* d_w_scale, const float* d_bias, const float* d_residual,
                                int64_t m, int64_t n, int64_t k, int activation, float* d_y,
                                void* stream);
int ct2x_gemm_s8_glu_packed(const int8_t* d_qa, const float* d_a_scale, const int8_t* d_w0_packed,
                            const float* d_w0_scale, const float* d_b0, const int8_t* d_w1_packed,
                            const float* d_w1_scale, const float* d_b1, int64_t m, int64_t n,
                            int64_t k, int activation, float* d_y, float* d_row_amax_out,
                            void* stream);

/* The row-wide launches in front of the streamed GEMMs as PROLOGUES of the weight stream (round 4;
 * m <= 16, k = 4096, packed weights of ct2x_gemm_s8_stream_pack; an error for other shapes):
 *  - ct2x_norm_gemm_s8_dequant_packed: layers::LayerNorm in its RMSNorm form (src/layers/common.cc:
 *    464-472, src/cpu/kernels.cc:543-573) -> ops::Quantize -> Gemm -> Dequantize(+bias, activation)
 *    -> Add(residual), bit-identical to ct2x_norm_quantize_s8 + ct2x_gemm_s8_dequant_packed;
 *    d_row_amax_out may be NULL;
 *  - ct2x_norm_gemm_s8_glu_packed: the same prologue in front of the gated pair
 *    (ct2x_gemm_s8_glu_packed);
 *  - ct2x_gemm_s8_dequant_quantizing_packed: ops::Quantize of the fp32 activation d_x folded into
 *    the fragment fetch, its row absmax given as d_row_amax[m, amax_parts] partial maxima left by
 *    the producing launch (src/ops/quantize.cc:21-51 split into its two halves). */
int ct2x_norm_gemm_s8_dequant_packed(const float* d_x, const float* d_gamma, float eps,
                                     int rms_use_residual, int quant_flags, const int8_t* d_w_packed,
                                     const float* d_w_scale, const float* d_bias,
                                     const float* d_residual, int64_t m, int64_t n, int64_t k,
                                     int activation, float* d_y, float* d_row_amax_out, void* stream);
int ct2x_norm_gemm_s8_glu_packed(const float* d_x, const float* d_gamma, float eps,
                                 int rms_use_residual, int quant_flags, const int8_t* d_w0_packed,
                                 const float* d_w0_scale, const float* d_b0, const int8_t* d_w1_packed,
                                 const float* d_w1_scale, const float* d_b1, int64_t m, int64_t n,
                                 int64_t k, int activation, float* d_y, float* d_row_amax_out,
                                 void* stream);
int ct2x_gemm_s8_dequant_quantizing_packed(const float* d_x, const float* d_row_amax, int amax_parts,
                                           int quant_flags, const int8_t* d_w_packed,
                                           const float* d_w_scale, const float* d_bias,
                                           const float* d_residual, int64_t m, int64_t n, int64_t k,
                                           int activation, float* d_y, void* stream);

/* A pre-norm int8 Dense at decode rows as ONE launch: layers::LayerNorm (LayerNorm when d_beta is
 * given, RMSNorm otherwise; src/layers/common.cc:464-472) -> ops::Quantize -> Gemm -> Dequantize
 * (+bias, activation) -> Add(residual); bit-identical to ct2x_norm_quantize_s8 followed by
 * ct2x_gemm_s8_dequant. m <= 128, k in {64, 128, 256, 512}, n < 8192 (an error otherwise).
 * quant_flags as in ct2x_quantize_s8. */
int ct2x_norm_gemm_s8_dequant(const float* d_x, const float* d_gamma, const float* d_beta, float eps,
                              int rms_use_residual, int quant_flags, const int8_t* d_w,
                              const float* d_w_scale, const float* d_bias, const float* d_residual,
                              int64_t m, int64_t n, int64_t k, int activation, float* d_y,
                              void* stream);

/* ct2x_norm_gemm_s8_dequant that also leaves the row absmax of y: d_row_amax_out[m] (zero on
 * entry) is max-combined with max_j |y[i, j]| - the first half of the ops::Quantize
 * (src/ops/quantize_cpu.cc / src/cpu/kernels.cc:577-651) that feeds the next Dense. */
int ct2x_norm_gemm_s8_dequant_amax(const float* d_x, const float* d_gamma, const float* d_beta,
                                   float eps, int rms_use_residual, int quant_flags,
                                   const int8_t* d_w, const float* d_w_scale, const float* d_bias,
                                   const float* d_residual, int64_t m, int64_t n, int64_t k,
                                   int activation, float* d_y, float* d_row_amax_out, void* stream);

/* layers::Dense int8 (src/layers/common.cc:348-401) with its ops::Quantize folded into the GEMM's
 * operand fetch: d_x is the fp32 activation [m, k], d_row_amax[m] its row absmax (as left by
 * ct2x_norm_gemm_s8_dequant_amax); every row is quantised with 127 / amax (1 for a zero row) and
 * quant_flags as in ct2x_quantize_s8 while it is loaded, then Gemm -> Dequantize(+bias, activation)
 * -> Add(residual). Bit-identical to ct2x_quantize_s8 followed by ct2x_gemm_s8_dequant.
 * m <= 128, k % 64 == 0, n <= 4096, m * k <= 262144 (an error otherwise). */
int ct2x_gemm_s8_dequant_quantizing(const float* d_x, const float* d_row_amax, int quant_flags,
                                    const int8_t* d_w, const float* d_w_scale, const float* d_bias,
                                    const float* d_residual, int64_t m, int64_t n, int64_t k,
                                    int activation, float* d_y, void* stream);

/* ---- AWQ int4 weights. The reference implements these for CUDA only and refuses
 * them on ROCm (src/layers/common.cc:402-404, "AWQ unsupported with ROCm"); parity is unpinned
 * (no CPU implementation, no vectors): oracle/awq_oracle.py restates formats and arithmetic.
 * float16 activations / scales / outputs are passed as raw 2-byte data. ---- */

/* ops::DequantizeAwq::dequantize<Device::CUDA,int,float16_t> (src/ops/awq/dequantize_gpu.cu:8-98):
 * GEMM layout - d_qweight [in_c, out_c / 8] int32 (nibble i of a word = output channel
 * {0,2,4,6,1,3,5,7}[i] of its group of eight), d_scales [in_c / group, out_c] f16, d_qzeros
 * [in_c / group, out_c / 8] - to d_out [in_c, out_c] f16, w = (q - z) * s rounded once. */
int ct2x_awq_dequantize(const int32_t* d_qweight, const void* d_scales_f16, const int32_t* d_qzeros,
                        int64_t in_c, int64_t out_c, int64_t group, void* d_out_f16, void* stream);

/* ops::GemmAwq::compute<Device::CUDA,float16_t,int> + the ops::Sum over its split-K slices
 * (src/ops/awq/gemm.cc:9-31, gemm_gpu.cu:470-530): y[m, out_c] = x[m, in_c] . dequantize(w), one
 * fused launch (fp32 accumulation, one rounding of the result). d_y_f16 and / or d_y_f32 receive
 * the result (either may be NULL). in_c % 32 == 0, out_c % 8 == 0, group % 8 == 0. */
int ct2x_awq_gemm(const void* d_x_f16, const int32_t* d_qweight, const void* d_scales_f16,
                  const int32_t* d_qzeros, int64_t m, int64_t in_c, int64_t out_c, int64_t group,
                  void* d_y_f16, float* d_y_f32, void* stream);

/* ops::GemvAwq::compute_gemv / compute_gemv2 (src/ops/awq/gemv_gpu.cu:236-547): GEMV layout -
 * d_qweight [out_c, in_c / 8] (nibble i of word w = input channel 8 w + i), d_scales [out_c, sf_w]
 * f16, d_qzeros [out_c, zeros_w] (group g = nibble g % 8 of word g / 8). in_c % 128 == 0,
 * group % 32 == 0. The dequantised weight alone: ct2x_awq_dequantize_gemv -> [out_c, in_c] f16. */
int ct2x_awq_gemv(const void* d_x_f16, const int32_t* d_qweight, const void* d_scales_f16,
                  const int32_t* d_qzeros, int64_t m, int64_t in_c, int64_t out_c, int64_t group,
                  int64_t zeros_w, int64_t sf_w, void* d_y_f16, float* d_y_f32, void* stream);
int ct2x_awq_dequantize_gemv(const int32_t* d_qweight, const void* d_scales_f16,
                             const int32_t* d_qzeros, int64_t out_c, int64_t in_c, int64_t group,
                             int64_t zeros_w, int64_t sf_w, void* d_out_f16, void* stream);

/* The bfloat16 arm of the four entry points above (round 6; SURVEY 8(f3) asks for "a bf16 variant
 * the reference lacks" - the reference instantiates float16 only, src/ops/awq/gemm_gpu.cu:541, and
 * forces FLOAT16 for AWQ models, src/models/model.cc:754-757): the same formats with bfloat16
 * activations / scales / 16-bit outputs (raw 2-byte data). Arithmetic, defined here and restated in
 * oracle/awq_oracle.py (bf16=True): w = (q - z) * s rounded ONCE to bfloat16 (the product is exact
 * in fp32), products with the bfloat16 activation accumulate in fp32. Same shape preconditions. */
int ct2x_awq_dequantize_bf16(const int32_t* d_qweight, const void* d_scales_bf16, const int32_t* d_qzeros,
                             int64_t in_c, int64_t out_c, int64_t group, void* d_out_bf16, void* stream);
int ct2x_awq_gemm_bf16(const void* d_x_bf16, const int32_t* d_qweight, const void* d_scales_bf16,
                       const int32_t* d_qzeros, int64_t m, int64_t in_c, int64_t out_c, int64_t group,
                       void* d_y_bf16, float* d_y_f32, void* stream);
int ct2x_awq_gemv_bf16(const void* d_x_bf16, const int32_t* d_qweight, const void* d_scales_bf16,
                       const int32_t* d_qzeros, int64_t m, int64_t in_c, int64_t out_c, int64_t group,
                       int64_t zeros_w, int64_t sf_w, void* d_y_bf16, float* d_y_f32, void* stream);
int ct2x_awq_dequantize_gemv_bf16(const int32_t* d_qweight, const void* d_scales_bf16,
                                  const int32_t* d_qzeros, int64_t out_c, int64_t in_c, int64_t group,
                                  int64_t zeros_w, int64_t sf_w, void* d_out_bf16, void* stream);

/* ops::Dequantize::dequantize<Device::CUDA,int8_t,T> (src/ops/dequantize_gpu.cu:17-28):
 * y[r,:] = x[r,:] / scale[r]  (int8 embedding rows, src/layers/common.cc:64-81). */
int ct2x_dequantize_rows(const int8_t* d_x, const float* d_scale, int64_t rows, int64_t depth,
                         void* d_y, int dtype, void* stream);

/* primitives<Device::CUDA>::gemm<float,float> + apply_bias_and_activation
 * (src/cuda/primitives.cu:485-517, src/ops/gemm.cc:11-25): C = act(A.B^T + bias) + residual. */
int ct2x_gemm_f32(const float* d_a, const float* d_b, int64_t m, int64_t n, int64_t k,
                  int trans_b, const float* d_bias, const float* d_residual, int activation,
                  float* d_c, void* stream);

/* primitives<Device::CUDA>::gemm<float16_t / bfloat16_t> + apply_bias_and_activation
 * (src/cuda/primitives.cu:519-569 cublasGemmEx with fp32 accumulation, src/ops/gemm.cc:11-25):
 * C = act(A.W^T + bias) + residual with W [n, k] stored as CT2X_F16 or CT2X_BF16, A [m, k] fp32
 * rounded to that type when it enters the matrix pipe, fp32 accumulation (MFMA), fp32 output.
 * k % 8 == 0. This is what the float16 / bfloat16 compute types of the engine run. */
int ct2x_gemm_f16w(const float* d_a, const void* d_w, int w_dtype, int64_t m, int64_t n, int64_t k,
                   const float* d_bias, const float* d_residual, int activation, float* d_c,
                   void* stream);

/* ops::LayerNorm::compute<Device::CUDA,T> last-axis form (src/ops/layer_norm_gpu.cu:174-213;
 * CPU src/cpu/kernels.cc:461-496): y = (x-mean)*rsqrt(max(E[x^2]-mean^2,0)+eps)*gamma+beta. */
int ct2x_layer_norm(const void* d_x, const float* d_gamma, const float* d_beta, int64_t rows,
                    int64_t depth, float epsilon, void* d_y, int dtype, void* stream);

/* ops::RMSNorm::compute<Device::CUDA,T> (src/ops/rms_norm_gpu.cu:20-63; CPU
 * src/cpu/kernels.cc:543-573): y = x*rsqrt(mean(x^2)+eps)*(gamma or 1+gamma). */
int ct2x_rms_norm(const void* d_x, const float* d_gamma, int64_t rows, int64_t depth,
                  float epsilon, int use_residual, void* d_y, int dtype, void* stream);

/* Fused norm + Quantize (the first two ops of every pre-norm Dense; replaces
 * layer_norm_gpu.cu + quantize_gpu.cu launches). d_beta==NULL selects RMSNorm as
 * layers::LayerNorm::operator() does (src/layers/common.cc:464-472). d_y (the float norm
 * output) may be NULL when only the quantised form is consumed. */
int ct2x_norm_quantize_s8(const float* d_x, const float* d_gamma, const float* d_beta,
                          int64_t rows, int64_t depth, float epsilon, int use_residual,
                          int round_before_cast, float* d_y, int8_t* d_q, float* d_scales,
                          void* stream);

/* ops::SoftMax / LogSoftMax::compute<Device::CUDA,T> (src/ops/softmax_gpu.cu:196-272; CPU
 * src/cpu/kernels.cc:402-459): per-row (log-)softmax over the first lengths[row] columns
 * (all when d_lengths==NULL); masked tail is exact 0. In-place allowed. */
int ct2x_softmax(const void* d_x, const int32_t* d_lengths, int64_t rows, int64_t depth,
                 int log, void* d_y, int dtype, void* stream);

/* ops::Rotary::compute<Device::CUDA,T> (src/ops/rotary_gpu.cu:28-85; CPU
 * src/ops/rotary_cpu.cc:9-62) on x[batch*heads, time, depth]; sin/cos [time, ndims]. */
int ct2x_rotary(const float* d_x, const float* d_sin, const float* d_cos, int64_t batch,
                int64_t time, int64_t depth, int64_t ndims, int interleave, float* d_y,
                void* stream);

/* ops::TopK::compute<Device::CUDA,T,int32_t> (src/ops/topk_gpu.cu:28-54; CPU
 * src/ops/topk_cpu.cc:13-55). Ties resolve to the LOWEST index, like std::max_element. */
int ct2x_topk(const void* d_x, int dtype, int64_t rows, int64_t depth, int64_t k,
              void* d_values, int32_t* d_indices, void* stream);

/* The logits processors of the decoding loop in one launch, in place on d_logits [rows, vocab]:
 * RepetitionPenalty (src/decoding_utils.cc:37-66 + primitives<D>::penalize_previous_tokens,
 * src/cpu/primitives.cc:411-428; CUDA src/cuda/primitives.cu penalize_previous_tokens),
 * NoRepeatNgram (:69-105), SuppressTokens / SuppressSequences (:108-170) followed by
 * DisableTokens::apply (primitives<D>::indexed_fill, :19-35). d_sequences [rows, length] = the
 * tokens generated so far (alive_seq). d_suppress_ids holds num_suppress sequences back to back,
 * d_suppress_offsets [num_suppress + 1] their start offsets. repetition_penalty 1 / ngram 0 /
 * num_suppress 0 switch a processor off. length <= 16000. */
int ct2x_process_logits(float* d_logits, int64_t rows, int64_t vocab, const int32_t* d_sequences,
                        int32_t length, float repetition_penalty, int32_t no_repeat_ngram_size,
                        const int32_t* d_suppress_ids, const int32_t* d_suppress_offsets,
                        int32_t num_suppress, void* stream);

/* primitives::convert (src/cuda/primitives.cu:160-196): element-wise conversion between CT2X_F32,
 * CT2X_F16 and CT2X_BF16 (round to nearest even on the way down). */
int ct2x_convert(const void* d_x, int src_dtype, void* d_y, int dst_dtype, int64_t size, void* stream);

/* Ops outside the decoder hot path that the reference's front-end can still ask Device::CUDA for
 * (direct one-thread-per-output kernels; correctness arms of the Level-1 boundary):
 * ops::Conv1D without bias / activation (src/ops/conv1d_gpu.cu; x [batch, in, len], w [out,
 * in / groups, kernel], y [batch, out, (len + 2 pad - dil (kernel - 1) - 1) / stride + 1]);
 * ops::MedianFilter over the last axis (src/ops/median_filter_cpu.cc:12-61, odd width <= 63);
 * ops::AlibiAdd (src/ops/alibi_add_cpu.cc:8-31; alibi [1, heads, 1, alibi_keys]). */
int ct2x_conv1d(const float* d_x, const float* d_w, int64_t batch, int64_t in_channels, int64_t length,
                int64_t out_channels, int64_t kernel, int64_t stride, int64_t padding,
                int64_t dilation, int64_t groups, float* d_y, void* stream);
int ct2x_median_filter(const float* d_x, int64_t rows, int64_t depth, int64_t width, float* d_y,
                       void* stream);
int ct2x_alibi_add(const float* d_x, const float* d_alibi, int64_t batch, int64_t heads,
                   int64_t queries, int64_t keys, int64_t alibi_keys, int64_t alibi_offset,
                   float* d_y, void* stream);

/* ops::TopPMask (src/ops/topp_mask_cpu.cc:12-39): y = x where the probability mass ranked before
 * the element (by decreasing d_probs) is < p, else mask_value. Rows of any width. */
int ct2x_topp_mask(const float* d_x, const float* d_probs, int64_t rows, int64_t depth, float p,
                   float mask_value, float* d_y, void* stream);
/* ops::Multinomial (src/ops/multinomial_cpu.cc): `samples` draws per row of weights [rows, classes]
 * by inverse CDF; counter-based RNG on (seed, call number, draw). d_out [rows, samples]. */
int ct2x_multinomial(const float* d_weights, int64_t rows, int64_t classes, int64_t samples,
                     uint32_t seed, int32_t* d_out, void* stream);

/* ops::GumbelMax::add_gumbel_noise<Device::CUDA,float> (src/ops/gumbel_max_gpu.cu:12-38):
 * y[i] = x[i] - log(u_i), u_i uniform in (0, 1] (x == y allowed); the front-end's TopK follows
 * (src/ops/gumbel_max.cc:14-30). Counter-based RNG on (seed, call number, element). */
int ct2x_gumbel_noise(const float* d_x, float* d_y, int64_t size, uint32_t seed, void* stream);

/* RandomSampler for one step (src/sampling.cc:41-101): best `sampling_topk` candidates (any k;
 * <= 0: the whole vocabulary) -> temperature -> TopPMask (src/ops/topp_mask_cpu.cc:24-31) ->
 * SoftMax -> Multinomial by inverse CDF (candidates in TopK order for k <= 64; index order for a
 * larger k and the whole vocabulary). d_uniforms [rows] in [0, 1) injects the draws (tests); NULL uses the counter-based
 * hash of (seed, *d_step, row). d_ids [rows], d_scores [rows] = log-probability of the id under the
 * ORIGINAL distribution. *d_step must be 0 for this standalone form. */
int ct2x_random_sample(const float* d_logits, int64_t rows, int64_t vocab, int32_t sampling_topk,
                       float sampling_topp, float temperature, uint32_t seed,
                       const float* d_uniforms, const int32_t* d_step, int32_t* d_ids,
                       float* d_scores, void* stream);

/* One beam-search step after TopK on the device (BeamSearch::search, src/decoding.cc:559-700):
 * unflatten_ids (:95-116), update_sample_with_prefix beam arm (:21-66), the candidate selection
 * (:618-655). d_topk_* [batch, 2*beam]; d_prefix_ids [batch, prefix_stride] / d_prefix_lens
 * [batch] may be NULL; d_step: device int32 = the absolute step, first_step is subtracted.
 * Outputs: the candidates as selected from (d_cand_* [batch, 2*beam], written at step index
 * step - first_step of those arrays) and per active beam the next input id, cumulative score
 * and parent row. beam_size <= 32. */
int ct2x_beam_select(const float* d_topk_scores, const int32_t* d_topk_ids, int64_t batch,
                     int32_t beam_size, int64_t vocab, int32_t end_id,
                     const int32_t* d_prefix_ids, const int32_t* d_prefix_lens,
                     int32_t prefix_stride, const int32_t* d_step, int32_t first_step,
                     int32_t* d_cand_word, float* d_cand_score, int32_t* d_cand_origin,
                     int32_t* d_next_ids, float* d_next_scores, int32_t* d_parents, void* stream);

/* ---- element-wise / layout members of primitives<Device::CUDA> --------------------------------
 * What the reference's UNFUSED front-end calls between the hot ops (include/ctranslate2/
 * primitives.h:10-233; src/cuda/primitives.cu, CPU bodies src/cpu/primitives.cc). The fused engine
 * needs none of them. dtype: CT2X_F32, CT2X_I8, CT2X_I16 or CT2X_I32. */
enum ct2x_binary_op { CT2X_OP_ADD = 0, CT2X_OP_SUB = 1, CT2X_OP_MUL = 2, CT2X_OP_MIN = 3, CT2X_OP_MAX = 4 };
/* primitives::add / sub / mul / min / max (const T* a, const T* b, T* c, size): c = a op b. */
int ct2x_binary(int op, const void* d_a, const void* d_b, void* d_c, int64_t size, int dtype,
                void* stream);
/* primitives::add / mul / min / max (T a, const T* x, T* y, size): y = x op a. */
int ct2x_binary_scalar(int op, double a, const void* d_x, void* d_y, int64_t size, int dtype,
                       void* stream);
/* primitives::add_batch_broadcast / mul_batch_broadcast (kind 0: c[i] = a[i % a_size] op b[i]),
 * add_depth_broadcast (kind 1: a[i / (b_size / a_size)]), add_block_broadcast (kind 2:
 * a[(i / block) % a_size]); op CT2X_OP_ADD or CT2X_OP_MUL (src/cpu/primitives.cc:150-190,283-292). */
int ct2x_broadcast(int op, int kind, const void* d_a, const void* d_b, void* d_c, int64_t a_size,
                   int64_t b_size, int64_t block, int dtype, void* stream);
/* primitives::fill (stride 1) / strided_fill: x[i * stride] = value for i < size. */
int ct2x_fill(void* d_x, double value, int64_t size, int64_t stride, int dtype, void* stream);
/* primitives::indexed_fill: x[indices[i]] = value (DisableTokens::apply, src/decoding_utils.cc:19-35). */
int ct2x_indexed_fill(void* d_x, double value, const int32_t* d_indices, int64_t num_indices,
                      int dtype, void* stream);
/* primitives::relu / gelu / gelu_tanh / gelu_sigmoid / sigmoid / swish / tanh (fn = the
 * ct2x_activation id) and exp / log / sin / cos (fn = 7 / 8 / 9 / 10), float32. */
int ct2x_unary(int fn, const float* d_x, float* d_y, int64_t size, void* stream);
/* primitives::transpose_2d / 3d / 4d: b = transpose(a viewed as dims[rank], perm), rank 2..4;
 * dims and perm are HOST arrays. */
int ct2x_transpose(const void* d_a, const int64_t* dims, const int64_t* perm, int rank, void* d_b,
                   int dtype, void* stream);
/* primitives::prepare_length_mask (src/cpu/primitives.cc:431-448): mask [batch, heads * queries]. */
/* ops::Concat / Split / Slide / Tile (src/ops/concat_split_slide_gpu.cu:66-209,
 * src/ops/tile_gpu.cu:36-70) are "outer x block" strided copies: `outer` blocks of block_bytes,
 * block o read at d_src + o*src_stride_bytes and written at d_dst + o*dst_stride_bytes (a stride
 * of 0 repeats the block: Tile). Any dtype. */
int ct2x_block_copy(const void* d_src, void* d_dst, int64_t outer, int64_t block_bytes,
                    int64_t src_stride_bytes, int64_t dst_stride_bytes, void* stream);
/* ops::Mean / the Sum form (src/ops/mean_gpu.cu:12-63) over the middle axis of
 * [outer, axis, inner] fp32. */
int ct2x_mean(const float* d_x, int64_t outer, int64_t axis, int64_t inner, int get_sum, float* d_y,
              void* stream);

int ct2x_prepare_length_mask(const int32_t* d_lengths, int64_t batch, int64_t heads,
                             int64_t queries, int mask_future, int multi_query, int32_t* d_mask,
                             void* stream);

/* ops::Gather::compute<Device::CUDA,T> axis 0 (src/ops/gather_gpu.cu:53-91):
 * out[i,:] = data[idx[i],:] for rows of row_bytes bytes. */
int ct2x_gather_rows(const void* d_data, const int32_t* d_idx, int64_t n_idx,
                     int64_t row_bytes, void* d_out, void* stream);

/* primitives<Device::CUDA>::gemm_batch_strided<float> as ops::MatMul calls it
 * (src/cuda/primitives.cu:599-683, src/ops/matmul.cc:62-73):
 * C_b[m,n] = alpha * A_b[m,k] . op(B_b). */
int ct2x_matmul_f32(const float* d_a, const float* d_b, int64_t batch, int64_t m, int64_t n,
                    int64_t k, int trans_b, float alpha, float* d_c, void* stream);

/* The fused single-query attention the backend runs in place of
 * MatMul(QK^T, alpha) -> SoftMax(lengths) -> MatMul(PV) (src/layers/attention.cc:216-277)
 * for decode steps. q [rows, heads, dhead]; K/V caches [rows/kv_group, heads_kv, t_max, dhead]
 * (kv_group>1 = beam rows sharing one encoder memory, attention.cc:805-806; heads_kv<heads =
 * GQA without materialising replicate_heads, attention.cc:291-295). Row r attends to the first
 * lengths[r / kv_group] (or `length` when d_lengths==NULL) positions. out [rows, heads*dhead]. */
int ct2x_attention_decode(const float* d_q, const float* d_k_cache, const float* d_v_cache,
                          const int32_t* d_lengths, int64_t length, int64_t rows, int64_t heads,
                          int64_t heads_kv, int64_t dhead, int64_t t_max, int64_t kv_group,
                          float scale, float* d_out, void* stream);

/* Encoder-decoder attention of one decode step over 16-bit memory keys / values (the float16 /
 * bfloat16 compute types' memory cache, src/layers/attention.cc:371-440 + 442-615): q [batch * beam,
 * heads * 64] fp32, kv [batch, s_max, 2 * heads_kv * 64] in kv_dtype (CT2X_F16 / CT2X_BF16; per
 * position the keys of all heads, then the values), lengths [batch] valid positions (NULL = s_max);
 * the beam rows of a batch entry share its memory. out [batch * beam, heads * 64] fp32. d_head 64,
 * beam <= 8. */
int ct2x_cross_attention_h(const float* d_q, const void* d_kv, int kv_dtype, const int32_t* d_lengths,
                           int64_t batch, int64_t beam, int64_t heads, int64_t heads_kv,
                           int64_t s_max, float scale, float* d_out, void* stream);
/* The same with the head width as an argument: d_head 64 or 128 (round 4: the NLLB-3.3B geometry). */
int ct2x_cross_attention_hd(const float* d_q, const void* d_kv, int kv_dtype, const int32_t* d_lengths,
                            int64_t batch, int64_t beam, int64_t heads, int64_t heads_kv,
                            int64_t dhead, int64_t s_max, float scale, float* d_out, void* stream);
/* The same with the memory of a (batch entry, head) cut into chunks that run as separate workgroups
 * (long memories on few (entry, head) units - Whisper: 8 windows x 20 heads over 1500 positions are
 * 160 workgroups on 256 CUs - leave the chip idle and walk the memory serially); the chunks are
 * merged in the same launch by the last one to finish. d_scratch: at least
 * ct2x_cross_attention_split_scratch_bytes(...) bytes of device memory, content irrelevant. Short
 * memories / many units run the one-workgroup form unchanged. */
int64_t ct2x_cross_attention_split_scratch_bytes(int64_t batch, int64_t beam, int64_t heads,
                                                 int64_t dhead, int64_t s_max);
int ct2x_cross_attention_hd_split(const float* d_q, const void* d_kv, int kv_dtype,
                                  const int32_t* d_lengths, int64_t batch, int64_t beam,
                                  int64_t heads, int64_t heads_kv, int64_t dhead, int64_t s_max,
                                  float scale, float* d_out, void* d_scratch, int64_t scratch_bytes,
                                  void* stream);

/* Sequence-mode MultiHeadAttention core (src/layers/attention.cc:178-287 dot_product_attention:
 * MatMul QK^T * queries_scale -> length / causal mask -> SoftMax -> MatMul PV) for Tq > 1:
 * q [B, Tq, H*dh], k / v [B, Tk, Hkv*dh] (heads contiguous per token), lengths [B] valid keys
 * (NULL = Tk), causal: key t visible to query i iff t <= i + q_offset. out [B, Tq, H*dh]. */
int ct2x_attention_full(const float* d_q, const float* d_k, const float* d_v,
                        const int32_t* d_lengths, int64_t batch, int64_t tq, int64_t tk,
                        int64_t heads, int64_t heads_kv, int64_t dhead, int causal,
                        int64_t q_offset, float scale, float* d_out, void* stream);
/* The same op as the engine runs it under compute type float16 (compute_kind 1) / bfloat16 (2):
 * tensors stay fp32 (the engine keeps activations in fp32 under the 16-bit types), the two products
 * run on the 16-bit matrix cores with every operand split into hi + lo 16-bit halves and fp32
 * accumulation (attention_prefill.hip) - no CPU order to reproduce (the reference has no 16-bit CPU
 * path), within 1e-5 / 2e-4 of the float64 statement. d_head 64 / 128 and Tq >= 16 take that form;
 * other shapes run the fp32 kernels of ct2x_attention_full. */
int ct2x_attention_full_half(const float* d_q, const float* d_k, const float* d_v,
                             const int32_t* d_lengths, int64_t batch, int64_t tq, int64_t tk,
                             int64_t heads, int64_t heads_kv, int64_t dhead, int causal,
                             int64_t q_offset, float scale, int compute_kind, float* d_out,
                             void* stream);

/* ---- engine: ctranslate2::Translator / Generator ------------------------------- */

/* models::Model::load + ReplicaPool ctor (src/models/model.cc:561-784,
 * include/ctranslate2/replica_pool.h:33-45). compute_type strings are the reference's
 * ("default","auto","int8","int8_float32","int8_float16","int8_bfloat16","float32",
 * "float16","bfloat16"; src/types.cc:18-66). Returns NULL on error. */
void* ct2x_translator_create(const char* model_dir, int device_index, const char* compute_type);
void ct2x_translator_destroy(void* translator);

/* The same engine for a model that is already in the caller's memory - Model::load(ModelReader&)
 * on a ModelMemoryReader (include/ctranslate2/models/model_reader.h:36-55), and the entry a host
 * application uses that has loaded the model ITSELF: the reference's own Translator on this backend
 * hands its variables over here (INTEGRATION.md "Level 1.5"). Variables in the layout of model.bin
 * (dtype ids of include/ctranslate2/types.h:16-23; HOST pointers, copied), the text of config.json
 * (or NULL), the vocabularies as token lists in id order (source NULL = shared with the target). */
struct ct2x_memory_variable {
  const char* name;
  int32_t dtype;
  int32_t rank;
  const int64_t* shape;
  const void* data;
  /* bytes readable at `data`: must equal product(shape) x sizeof(dtype) - a mismatch, a negative
   * rank or dimension or an unknown dtype is refused before anything is copied (round 6; the
   * reader of model.bin checks the same against the byte count the file states,
   * src/models/model.cc:594-661). The binary version handed to ct2x_translator_create_from_memory
   * is checked like Model::load's (<= 6). */
  uint64_t nbytes;
};
void* ct2x_translator_create_from_memory(const char* spec, uint32_t spec_revision, uint32_t binary_version,
                                         const char* config_json, const struct ct2x_memory_variable* vars,
                                         int64_t num_vars, const char* const* source_tokens,
                                         int64_t num_source_tokens, const char* const* target_tokens,
                                         int64_t num_target_tokens, int device_index,
                                         const char* compute_type);

/* TranslationOptions subset on the hot path (include/ctranslate2/translation.h:11-96). */
struct ct2x_translation_options {
  int32_t beam_size;            /* 1 = greedy */
  int32_t max_decoding_length;
  int32_t min_decoding_length;
  int32_t num_hypotheses;
  int32_t return_scores;
  int32_t return_end_token;
  int32_t max_batch_size;       /* 0 = whole batch */
  int32_t max_input_length;     /* 0 = no truncation */
  float length_penalty;
  float patience;
  int32_t use_graph;            /* replay the decode step from a hipGraph */
  int32_t num_streams;          /* >1: split the batch into independent sub-batches decoded
                                   concurrently on separate HIP streams of the SAME GPU (the
                                   reference's several replicas per device, replica_pool.h:243-260);
                                   0 = choose automatically */
  /* RandomSampler (src/sampling.cc:41-101; python: sampling_topk / sampling_topp /
   * sampling_temperature), greedy search (beam_size 1) only. sampling_topk: 1 = arg-max
   * (BestSampler), k >= 2 = sample among the best k, -1 = sample from the whole vocabulary (the
   * reference's sampling_topk = 0; 0 here means arg-max so that a zero-initialised struct is
   * deterministic); sampling_topp < 1 keeps the smallest
   * prefix of those whose mass reaches it; seed drives a counter-based generator
   * (seed, step, row) -> reproducible for a seed. */
  int32_t sampling_topk;
  float sampling_topp;
  float sampling_temperature;
  uint32_t seed;
  /* Logits processors (TranslationOptions::repetition_penalty / no_repeat_ngram_size /
   * disable_unk / suppress_sequences, include/ctranslate2/translation.h:25-31 ->
   * make_logits_processors, src/decoding.cc:1091-1120; src/decoding_utils.cc:37-153), greedy,
   * sampling and beam search. Sequences are given as ids: `suppress_ids` holds
   * `num_suppress_sequences` sequences back to back, `suppress_lengths` their lengths; a
   * sequence of ONE id is disabled at every step (pass the <unk> id for disable_unk). */
  float repetition_penalty;        /* 1 (or 0 in a zero-initialised struct) = off */
  int32_t no_repeat_ngram_size;    /* 0 = off */
  int32_t num_suppress_sequences;
  int32_t end_token_plus_one;      /* TranslationOptions::end_token as a target id, + 1;
                                      0 = the model's end token */
  const int32_t* suppress_ids;
  const int32_t* suppress_lengths;
  /* The remaining decoding options of TranslationOptions (include/ctranslate2/translation.h:33-71):
   *  return_alternatives: the num_hypotheses best FIRST tokens after the target prefix, each
   *    continued independently (decode_alternatives, src/decoding.cc:1122-1290); alternatives whose
   *    probability is below min_alternative_expansion_prob are dropped - the missing hypotheses of a
   *    sentence get out_lengths -1; ct2x_generate_batch honours it as well (round 5: the same
   *    function behind DecoderReplica::run_generation, src/models/language_model.cc:241-245 - the
   *    step arithmetic then counts the consumed prompt: a sequence has max_decoding_length -
   *    prompt length + 1 tokens; greedy / beam search only);
   *  prefix_bias_beta in (0, 1): the target prefix is a BIAS instead of a constraint (BiasedDecoder,
   *    src/decoding.cc:256-343; always a beam search, beam_size 1 included);
   *  attention_out (return_attention): HOST buffer [batch, num_hypotheses, max_decoding_length,
   *    attention_stride] receiving, per generated token, the encoder-decoder attention of the
   *    decoder's alignment layer and heads over the source positions (zero beyond them);
   *  coverage_penalty: GNMT coverage term of the final score (src/decoding.cc:176-207). */
  int32_t return_alternatives;
  float min_alternative_expansion_prob;
  float prefix_bias_beta;
  float coverage_penalty;
  int32_t attention_stride;
  /* ct2x_generate_batch - GenerationOptions::static_prompt (include/ctranslate2/generation.h:66-70;
   * src/models/language_model.cc:184-206): the first static_prompt_length tokens of every prompt row
   * are the static prompt - consumed by one sequence-mode pass, never returned, not counted by max /
   * min_decoding_length; with include_prompt_in_result the decoding starts from the token behind
   * them. Without it the static prompt is the head of what the sequence-mode pass consumes anyway;
   * the length still matters in one corner the reference has: when the shortest prompt behind the
   * static one is a single token nothing more is forwarded and the rest of the longer prompts is
   * RETURNED (return_prefix keeps its default, language_model.cc:224-237). The reference's cache of
   * the static prompt's state is an optimisation this entry does not make. */
  int32_t static_prompt_length;
  float* attention_out;
  /* TranslationOptions / GenerationOptions::callback (include/ctranslate2/translation.h:85,
   * generation.h:77; invoked by GreedySearch::search, src/decoding.cc:925-939): called for every
   * generated token of every batch entry, in step order per entry, with the step (counted from the
   * first decoding step), the entry's index in the call's batch, the hypothesis (0, or the sample's
   * index when random sampling returns num_hypotheses > 1), the token id, its log-probability
   * when return_scores is set (has_score) and whether it is the entry's last token. A non-zero
   * return value ends that entry's decoding there. beam_size 1 only (the reference's rule and error
   * text). With a callback the host examines every step (instead of groups of 8) while the device
   * runs one step ahead, so a token reaches the callback at most two steps after it was sampled;
   * the calls come from the thread that made the translate / generate call (num_streams <= 1 is
   * forced). NULL = none. */
  int (*step_callback)(void* user_data, int64_t step, int64_t batch_id, int64_t hypothesis_id,
                       int32_t token_id, int32_t has_score, float score, int32_t is_last);
  void* step_callback_data;
  /* TranslationOptions / GenerationOptions::return_logits_vocab (include/ctranslate2/translation.h:71,
   * generation.h:57; GreedySearch::search, src/decoding.cc:868-872, 905-908): the vocabulary logits
   * every decoding step of an entry sampled from - after the logits processors and the
   * minimum-length mask of the end token (lowest()), before the log-softmax. One vector per step
   * the entry ran: forced prefix steps and the end token's step included, so the count can exceed
   * the hypothesis length. HOST buffer [batch, num_hypotheses, logits_steps, vocabulary size]
   * (steps past logits_steps are dropped), logits_counts_out [batch, num_hypotheses] = vectors
   * written. beam_size 1 only (greedy search, random sampling). NULL = off. */
  float* logits_out;
  int32_t* logits_counts_out;
  int32_t logits_steps;
  /* ct2x_generate_batch only - GenerationOptions::include_prompt_in_result (include/ctranslate2/
   * generation.h:61, the reference's default; DecoderReplica::run_generation,
   * src/models/language_model.cc:150-282): non-zero = no sequence-mode pass over the prompt; the
   * first prompt token starts the decoding, the rest of the prompt is a hard decoding prefix that
   * is RETURNED (return_prefix: it counts towards max / min_decoding_length, forced tokens score 0,
   * the step callback sees them). The returned sequences then begin with prompt token 1; putting
   * token 0 in front of them when it is not the BOS token (language_model.cc:262-267) is the
   * caller's side (include/ct2_mi355.hpp, the Level 1.5 replica). Not with return_alternatives.
   * A static prompt: static_prompt_length above.
   * 0 = the prompt is consumed by one sequence-mode pass and not returned. */
  int32_t include_prompt_in_result;
};

void ct2x_translation_options_default(struct ct2x_translation_options* opt);

/* ---- tensor parallelism (one process per GPU, RCCL over xGMI) ------------------------------
 * Reference: tensor_parallel=True in ctranslate2.Translator -> ScopedMPISetter / ncclComm
 * (src/devices.cc:141-229), weight sharding in Model::load (src/models/model.cc:662-743),
 * ops::ReduceAll after every row-parallel Dense (src/layers/common.cc:348-401,
 * src/layers/attention.cc:606-612, src/layers/transformer.cc:21-51).
 * Here: heads and the FFN inner dimension are split over `world` ranks; a row-parallel Dense
 * max-reduces the per-row absmax (global Quantize scale), sum-reduces the exact int32 partial
 * products and dequantises once, so the sharded model is bit-identical to the unsharded one.
 * Launch: rank 0 calls ct2x_tp_unique_id, the 128 bytes travel to every rank by the launcher's
 * own means (torch.distributed / MPI / a file), every rank calls ct2x_translator_create_tp with
 * its own device. Every rank must then issue the same translate / generate calls. */
int ct2x_tp_unique_id(void* id_out_128_bytes);
void* ct2x_translator_create_tp(const char* model_dir, int device_index, const char* compute_type,
                                int rank, int world, const void* unique_id_128_bytes);
/* The same sharded engine on a node-local communicator WITHOUT RCCL: one process per rank, bootstrap
 * through the POSIX shared-memory segment "/<segment_name>" (rank 0 creates it; the name must be
 * unique per group and agreed by the launcher), hipIpc peer-mapped mailboxes for the one-shot
 * exchanges of the decode steps, host-staged collectives for the few larger ones. Ranks may share
 * a device: this is how the cross-process exchange path runs on a single-GPU host (RCCL refuses
 * two ranks on one device). */
void* ct2x_translator_create_tp_shm(const char* model_dir, int device_index, const char* compute_type,
                                    int rank, int world, const char* segment_name);
/* What the handle's engine exchanges through: rank / world, whether the peer-mapped mailboxes of
 * the one-shot exchanges are in place (0: every exchange is a collective call of the backend),
 * and the communicator's name ("none", "rccl", "shared-memory", "in-process"). */
int ct2x_tp_info(void* translator, int32_t* rank, int32_t* world, int32_t* has_mailboxes,
                 char* backend, int backend_size);
/* The same sharded engines with an in-process communicator: `world` ranks in THIS process on ONE
 * device, each handle driven by its own host thread (the single-GPU test double of the above). */
/* Host-only: the rows of a column-parallel weight [n, k] owned by `rank` (Model::load,
 * src/models/model.cc:662-743). `blocks` are the sizes of the fused sub-matrices split separately
 * (Q | K | V); num_blocks = 0 means one block of n rows. first_out / count_out get one entry per
 * block. A row-parallel weight owns input columns [rank * k / world, (rank + 1) * k / world). */
int ct2x_tp_shard_rows(int64_t n, const int64_t* blocks, int32_t num_blocks, int32_t rank,
                       int32_t world, int64_t* first_out, int64_t* count_out);
/* One-rank RCCL communicator + all-reduces on `device_index` (identity): checks the RCCL binding
 * of the multi-GPU path where only one GPU is visible. */
int ct2x_tp_selftest_rccl(int device_index);
/* The in-process communicator's collectives (all-reduce int32 sum / float max, GatherAll) with
 * `world` ranks as threads on one device, verified on every rank. */
int ct2x_tp_selftest_local(int device_index, int world);
/* The one-shot exchanges of a row-parallel int8 Dense (peer-mapped mailboxes, 8-byte {payload, tag}
 * granules stored and polled inside the producer / consumer kernels; SURVEY 8(e) "custom <= 1 MiB
 * one-shot path" instead of ops::ReduceAll, src/ops/nccl_ops_gpu.cu:53-68) on `world` in-process
 * ranks of one device: results checked bit for bit against the unsharded row, then `iters` exchange
 * pairs timed; *us_per_dense_out = microseconds per (absmax exchange + sum exchange), worst rank. */
int ct2x_tp_selftest_oneshot(int device_index, int world, int64_t rows, int64_t depth, int64_t n,
                             int iters, float* us_per_dense_out);
/* The same checks across REAL ranks, one process per GPU (round 6): every rank calls it with its
 * own device and rank and either the RCCL unique id of rank 0 or the name of a shared-memory
 * segment (the two ways ct2x_translator_create_tp / _tp_shm connect). All-reduce (int32 sum,
 * float sum, float max) and all-gather on rank-dependent data with closed-form results, then - if
 * the communicator has peer-mapped mailboxes (*mailboxes_out) - the one-shot exchanges of a
 * row-parallel int8 Dense checked bit for bit against the whole row, 50 pairs timed
 * (*us_per_dense_out). `bench.py --gpus N` runs it before its timed region. */
int ct2x_tp_selftest_multi(int device_index, int rank, int world, const void* unique_id_128_bytes,
                           const char* shm_segment_name, int32_t* mailboxes_out, float* us_per_dense_out);
int ct2x_translator_create_tp_local(const char* model_dir, int device_index,
                                    const char* compute_type, int world, void** handles_out);

/* Translator::translate_batch on token ids (src/translator.cc:8-28 ->
 * src/models/sequence_to_sequence.cc:305-432). src_ids [batch, src_stride] raw ids
 * WITHOUT the implicit source BOS/EOS (added per config.json like Vocabulary::to_ids).
 * out_ids [batch, num_hypotheses, max_decoding_length], out_lengths/out_scores
 * [batch, num_hypotheses]. */
int ct2x_translate_batch(void* translator, const int32_t* src_ids, const int32_t* src_lengths,
                         int32_t batch, int32_t src_stride,
                         const struct ct2x_translation_options* opt, int32_t* out_ids,
                         int32_t* out_lengths, float* out_scores);

/* translate_batch with a target prefix per sentence (Translator::translate_batch(source,
 * target_prefix, options), src/translator.cc:8-28): prefix_ids [batch, prefix_stride] raw target
 * ids forced right after the decoder start token (hard prefix, src/decoding.cc:21-66); they are
 * part of the returned hypotheses and of max_decoding_length, and contribute 0 to the score.
 * prefix_lengths may contain zeros. */
int ct2x_translate_batch_prefix(void* translator, const int32_t* src_ids,
                                const int32_t* src_lengths, int32_t batch, int32_t src_stride,
                                const int32_t* prefix_ids, const int32_t* prefix_lengths,
                                int32_t prefix_stride,
                                const struct ct2x_translation_options* opt, int32_t* out_ids,
                                int32_t* out_lengths, float* out_scores);

/* Generator::generate_batch for decoder-only models (src/generator.cc:8-24 ->
 * src/models/language_model.cc:150-282); prompts are consumed in one prefill call. Outputs as
 * ct2x_translate_batch: out_ids [batch, num_hypotheses, max_decoding_length], out_lengths /
 * out_scores [batch, num_hypotheses] (num_hypotheses > 1: the best hypotheses of a beam search, or
 * - beam_size 1 with sampling_topk != 1 - that many samples per prompt ordered by score,
 * GreedySearch::search src/decoding.cc:751-814; a length of 0 past the first: fewer finished). */
int ct2x_generate_batch(void* translator, const int32_t* prompt_ids,
                        const int32_t* prompt_lengths, int32_t batch, int32_t stride,
                        const struct ct2x_translation_options* opt, int32_t* out_ids,
                        int32_t* out_lengths, float* out_scores);

/* Translator::score_batch (include/ctranslate2/translator.h:62-76 -> EncoderDecoderReplica::
 * run_scoring, src/models/sequence_to_sequence.cc:236-261 -> score_sequences, src/scoring.cc):
 * the log-probability of every target token (and of the final </s>) given the source, teacher
 * forced. src_ids / tgt_ids are raw ids WITHOUT special tokens; the decoder start token and </s>
 * are added and both sides truncated to max_input_length like Vocabulary::to_ids (0 = no limit).
 * out_ids / out_scores [batch, out_stride] receive the scored tokens (target + </s>) and their
 * log-probabilities, out_lengths [batch] their count (<= out_stride, or the call fails). An
 * example whose source is empty scores 0 for every token (skip_scoring, :263-283). */
int ct2x_score_batch(void* translator, const int32_t* src_ids, const int32_t* src_lengths,
                     int32_t batch, int32_t src_stride, const int32_t* tgt_ids,
                     const int32_t* tgt_lengths, int32_t tgt_stride, int32_t max_input_length,
                     int32_t* out_ids, float* out_scores, int32_t out_stride,
                     int32_t* out_lengths);

/* Generator::forward_batch (include/ctranslate2/generator.h:31-35; SequenceGeneratorReplica::
 * forward, src/models/language_model.cc:73-100), decoder-only models: ONE sequence-mode pass over
 * ids [batch, stride] (lengths [batch]); out is a HOST buffer [batch, max(lengths), vocab] of logits,
 * or of log-probabilities (ops::LogSoftMax) when return_log_probs != 0. Positions past a row's
 * length hold the values of the padding token 0, as in the reference. */
int ct2x_forward_batch(void* translator, const int32_t* ids, const int32_t* lengths, int32_t batch,
                       int32_t stride, int32_t return_log_probs, float* out);

/* Generator::score_batch (include/ctranslate2/generator.h:21-25; DecoderReplica::run_scoring ->
 * score_sequences, src/models/language_model.cc:110-135, src/scoring.cc:9-60): out_scores
 * [batch, stride] receives log P(ids[b][k + 1] | ids[b][..k]) for k < out_lengths[b] = length - 1
 * (after the max_input_length truncation of Vocabulary::to_ids; 0 = no limit). Sequences of fewer
 * than 2 tokens are skipped (out_lengths 0), as skip_scoring does. */
int ct2x_generator_score_batch(void* translator, const int32_t* ids, const int32_t* lengths,
                               int32_t batch, int32_t stride, int32_t max_input_length,
                               float* out_scores, int32_t* out_lengths);

/* Teacher-forced step logits (layers::Decoder::operator()(step, ids, state, logits),
 * include/ctranslate2/layers/decoder.h:25-29): src_ids INCLUDE special tokens here;
 * logits_out is a HOST buffer [T, batch, vocab]. */
int ct2x_step_logits(void* translator, const int32_t* src_ids, const int32_t* src_lengths,
                     int32_t batch, int32_t src_stride, const int32_t* tgt_in, int32_t T,
                     float* logits_out);

/* Host side of the path: the reference builds its sinusoidal position encodings and RoPE tables
 * on the CPU with ops::Sin / ops::Cos (src/layers/common.cc:204-229,
 * src/layers/attention_layer.cc:252-343), which an MKL build routes to VML (vsSin / vsCos,
 * src/cpu/primitives.cc:378-394). VML is neither libm nor correctly rounded, and a one-ulp table
 * difference flips int8 activations, so this backend carries VML's two kernels RESTATED (from the
 * library's disassembly, pinned bit for bit against it; arguments up to 10000 rad): token parity
 * does not depend on an MKL runtime on the host. CT2X_HOST_MATH=mkl uses an installed MKL instead
 * (dlopen of libmkl_rt; CT2X_MKL_PATH / MKLROOT / CONDA_PREFIX searched), CT2X_HOST_MATH=libm the
 * C library (not bit-identical to the reference).
 * ct2x_host_math_backend: "vml-restated" (default), "mkl-vml" or "libm".
 * ct2x_host_sin_cos: y = sin / cos of n floats through that backend (host only; tests).
 * ct2x_host_sinusoidal_table: host only, out [positions, depth]. */
const char* ct2x_host_math_backend(void);
int ct2x_host_sin_cos(int64_t n, const float* x, float* sin_out, float* cos_out);
int ct2x_host_sinusoidal_table(int64_t positions, int64_t depth, float* out);

/* Host side of RotaryEmbeddings::initialize (src/layers/attention_layer.cc:254-330): sin / cos
 * tables [positions, dim] for scaling type None (-1), Linear (0) or Llama3 (2), built through the
 * same host math backend as the engine's tables (no GPU needed). */
int ct2x_host_rotary_tables(int64_t dim, int interleave, int scaling_type, float scaling_factor,
                            float base, float low_freq_factor, float high_freq_factor,
                            int64_t original_max_position_embeddings, int64_t positions,
                            float* sin_out, float* cos_out);

/* Tests: beam search normally follows an ancestry table instead of moving the self-attention
 * caches after every step (Decoder::update_state, src/layers/decoder.cc:33-56) wherever the
 * attention kernels support it; 0 switches the table off for sessions created afterwards (the
 * caches are gathered every step), -1 restores the default. Results are identical either way. */
int ct2x_debug_set_beam_ancestry(int enabled);

/* Tests: the MFMA-bound GEMM's epilogue divides by 1 + exp(..) (Swish, GELUSigmoid, Sigmoid:
 * src/cpu/kernels.cc:146-207) with the compiler's own IEEE division sequence minus the steps that
 * are the identity inside the admitted operand range; this runs that sequence against the plain
 * division on `pairs` operand pairs on the device and returns the number of differing results
 * (must be 0). */
int ct2x_debug_fast_div_check(uint64_t pairs, uint32_t seed, uint64_t* mismatches);

/* Tests / A-B runs: which prologue forms the persistent streaming Dense takes (the value of
 * CT2X_STREAM_PROLOGUE, process wide): bit 0 RMSNorm + Quantize and bit 1 ops::Quantize redone by
 * every workgroup (round 4, off), bit 2 RMSNorm + Quantize and bit 3 ops::Quantize done once by the
 * launch's first sixteen workgroups, one row each (round 6, on: default 12). Every setting returns
 * the same bits (src/layers/common.cc:339-401 is one arithmetic). Returns the previous mask. */
int ct2x_debug_set_stream_prologue(int mask);

/* ---- debugging taps (tests only; not part of the reference boundary) ----------------------
 * ct2x_debug_step_taps = ct2x_step_logits with every producer on the path (embedding, norm +
 * Quantize, Quantize, Dense, attention) copying its output to the host, named
 * "<scope>.<index>.<producer>"; tests line them up against the same points of the reference's
 * layers and report the first tensor that differs. logits_out may be NULL. dtype: 0 fp32, 1 int8.
 * ct2x_debug_set_position_table replaces the sinusoidal position table (the reference's comes
 * from MKL's closed vsSin / vsCos; injecting it isolates that residual). */
int ct2x_debug_step_taps(void* translator, const int32_t* src_ids, const int32_t* src_lengths,
                         int32_t batch, int32_t src_stride, const int32_t* tgt_in, int32_t T,
                         float* logits_out);
int ct2x_debug_tap_count(void);
int ct2x_debug_tap_info(int index, char* name, int name_size, int32_t* dtype, int64_t* rows,
                        int64_t* cols);
int ct2x_debug_tap_data(int index, void* out);
int ct2x_debug_set_position_table(void* translator, const float* table, int64_t positions,
                                  int64_t depth);

/* Model facts used by bindings and tests. */
int ct2x_model_info(void* translator, int32_t* is_seq2seq, int32_t* vocab_size,
                    int32_t* d_model, int32_t* num_layers_dec, int32_t* num_heads,
                    int32_t* bos_id, int32_t* eos_id, char* compute_type, int ct_size);
/* Vocabulary lookups (include/ctranslate2/vocabulary.h:16-70); which: 0 source, 1 target. */
int ct2x_vocab_size(void* translator, int which);
const char* ct2x_vocab_token(void* translator, int which, int32_t id);
int32_t ct2x_vocab_id(void* translator, int which, const char* token);

/* ExecutionStats-style counters of the last translate/generate call
 * (include/ctranslate2/translator.h:9-13) + device timing of the decode loop. */
struct ct2x_exec_stats {
  int64_t num_tokens;
  int64_t num_examples;
  int64_t decode_steps;
  int64_t kernels_per_step;
  double total_ms;
  double encode_ms;
  double decode_ms;
};
int ct2x_last_stats(void* translator, struct ct2x_exec_stats* stats);
/* How many times the last translate / generate call dropped its finished rows from the running
 * batch (greedy decoding; the reference removes finished sentences at every step,
 * src/decoding.cc:916-956; here rows leave in bulk once at most half of a session's rows are alive
 * and the session has at least 64 rows - CT2X_COMPACT_MIN_ROWS overrides, 0 = never). -1: bad handle. */
int64_t ct2x_last_compactions(void* translator);
/* Session reuse: a Translator keeps the decoding session (buffers + captured step graphs) of its
 * last PLAIN ct2x_translate_batch call - greedy or beam search with arg-max sampling, no prefix, no
 * logits processors, no attention / logits / callback, one stream, the whole batch in one group -
 * and a call with the same padded shape (rows, longest source) and the same device-side options
 * (beam size, max / min decoding length, return_scores, end token, use_graph) re-runs its encoder into
 * those buffers and replays the graphs: no allocation, capture or instantiation (~0.45 ms of a 20 ms
 * Transformer-base call). Results are those of a fresh session (tests/test_session_reuse_gpu.py);
 * CT2X_SESSION_CACHE=0 switches it off; the session's device memory is released with the
 * translator or by ct2x_release_session. ct2x_session_hits: calls served that way so far (-1: bad
 * handle). */
int64_t ct2x_session_hits(void* translator);

/* The decoder-stack launch (one persistent launch per decode step, csrc/decoder_cluster.hip) needs
 * its workgroups co-resident. A session uses it only while it holds the device's gate (one holder
 * per GPU across threads and processes; a session that does not get the gate decodes with the
 * launch-per-op step - same tokens). If a cluster nevertheless waits in vain, the group's results
 * are discarded, the handle switches to the launch-per-op step for good and the group is decoded
 * again: ct2x_cluster_fallbacks counts those re-runs (-1: bad handle).
 * ct2x_debug_inject_cluster_timeout (tests): the next session that ran the launch reports a
 * starved cluster. Several workers per device is a supported deployment of the reference
 * (include/ctranslate2/replica_pool.h:339-359). */
int64_t ct2x_cluster_fallbacks(void* translator);
int ct2x_debug_inject_cluster_timeout(void* translator);
int ct2x_release_session(void* translator);

/* Per-kernel profile of the next translate/generate calls, measured with HIP events on the
 * engine's stream around every launch (graph replay is disabled while profiling). profile_end
 * writes a JSON object {label: {"launches":n,"total_ms":t,"bytes":b,"ops":o}, ...} where bytes /
 * ops are the ALGORITHMIC traffic / work of the launches ("Roofline accounting"). */
int ct2x_profile_begin(void* translator);
int ct2x_profile_end(void* translator, char* json_out, int json_size);

/* ---- host-only helpers (no GPU needed; exercised by the CPU test-suite) --------- */

/* Parses model.bin exactly as Model::load does (src/models/model.cc:594-661,767-777) and
 * reports what it found. Returns number of variables, <0 on error. */
int ct2x_inspect_model(const char* model_dir, char* spec_name, int spec_size,
                       int32_t* binary_version, int32_t* revision, int32_t* num_aliases);
/* i-th variable of model.bin: name, dtype id, rank, dims (up to 8), payload bytes. */
int ct2x_inspect_variable(const char* model_dir, int32_t index, char* name, int name_size,
                          int32_t* dtype, int32_t* rank, int64_t* dims, int64_t* nbytes);

/* The host side of a beam-search step (the bookkeeping BeamSearch::search does on the TopK
 * candidates, src/decoding.cc:596-676), as replayed by the decoding loop on the candidates the
 * device recorded: which of the first beam_size candidates finished (EOS at step >= prefix_length,
 * or the last step) and which candidate occupies each beam next. word_ids [2*beam_size];
 * finished [beam_size] (candidate slots, -1 padded), next [beam_size]. Returns the number of
 * finished candidates, <0 on error. */
int ct2x_host_beam_step_select(int32_t beam_size, int32_t step, int32_t is_last_step,
                               int32_t end_id, int32_t prefix_length, const int32_t* word_ids,
                               int32_t* finished, int32_t* next);
/* finalize_hypothesis_score without coverage penalty (src/decoding.cc:192-207). */
float ct2x_host_finalize_score(float score, float length, float length_penalty);

/* Host side of Whisper::generate's prompt handling (check_prompts / get_sot_index /
 * get_prompt_length, src/models/whisper.cc:149-200): prompts [batch, stride] must all hold
 * <|startoftranscript|> at the same position and the same number of task tokens (ids in
 * [sot_id, no_timestamps_id]) after it. Writes the common sot index and prompt length (the
 * tokens before the text tokens); non-zero with the reference's message in ct2x_last_error()
 * otherwise. No device needed. */
int ct2x_host_whisper_check_prompts(const int32_t* prompts, const int32_t* lengths, int32_t batch,
                                    int32_t stride, int32_t sot_id, int32_t no_timestamps_id,
                                    int32_t* sot_index, int32_t* prompt_length);

/* negative_dtw (src/dtw.cc:40-82; Whisper word alignment, src/models/whisper.cc:418): dynamic time
 * warping over the NEGATED matrix x [n, m] (host memory), i.e. the monotone path from (0, 0) to
 * (n - 1, m - 1) that maximises the summed x, with the reference's tie rule (diagonal only when
 * strictly cheaper than both, then "up" only when strictly cheaper than both, else "left").
 * path_i / path_j [n + m] receive the path in order; returns its length (<= n + m - 1), <0 on error. */
int64_t ct2x_host_negative_dtw(const float* x, int64_t n, int64_t m, int64_t* path_i, int64_t* path_j);

/* ---- Whisper (include/ctranslate2/models/whisper.h; python/cpp/whisper.cc binds these) ---------
 * models::Whisper on the fused engine: the convolutional front end as im2col + MFMA GEMMs, the
 * encoder / decoder of the Translator path, Whisper's logits processors (ApplyTimestampRules,
 * SuppressTokensBegin, GetNoSpeechProbs) inside the decoding step's graph. Features are float32
 * [batch, n_mels, frames] on the host. */

/* models::WhisperOptions (whisper.h:11-58), field for field (return_logits_vocab excluded). */
struct ct2x_whisper_options {
  int32_t beam_size;                    /* 5 */
  float patience;                       /* 1 */
  float length_penalty;                 /* 1 */
  float repetition_penalty;             /* 1 */
  int32_t no_repeat_ngram_size;         /* 0 */
  int32_t max_length;                   /* 448 */
  int32_t sampling_topk;                /* 1 */
  float sampling_temperature;           /* 1 */
  int32_t num_hypotheses;               /* 1 */
  int32_t return_scores;                /* 0 */
  int32_t return_no_speech_prob;        /* 0 */
  int32_t max_initial_timestamp_index;  /* 50 */
  int32_t suppress_blank;               /* 1 */
  int32_t num_suppress_tokens;          /* 1 */
  const int32_t* suppress_tokens;       /* {-1}: the model's config.json suppress_ids */
};
void ct2x_whisper_options_default(struct ct2x_whisper_options* options);

/* models::Whisper(model_path, Device::CUDA, compute_type) (replica_pool.h:33-45). NULL on error. */
void* ct2x_whisper_create(const char* model_dir, int device_index, const char* compute_type);
void ct2x_whisper_destroy(void* whisper);
/* Whisper::is_multilingual / n_mels / num_languages (whisper.h:166-168). */
int ct2x_whisper_info(void* whisper, int32_t* is_multilingual, int32_t* n_mels,
                      int32_t* num_languages);
/* Vocabulary of the model (WhisperModel::get_vocabulary): size, id -> token (NULL when out of
 * range), token -> id (the unknown token's id when absent). */
int ct2x_whisper_vocab_size(void* whisper);
int64_t ct2x_whisper_d_model(void* whisper); /* depth of the encoder output */
const char* ct2x_whisper_vocab_token(void* whisper, int32_t id);
int32_t ct2x_whisper_vocab_id(void* whisper, const char* token);
/* Whisper::encode (src/models/whisper.cc:80-106): out [batch, (frames - 1) / 2 + 1, d_model]
 * float32 on the host (to_cpu = true); out_dims[3] receives the shape. */
int ct2x_whisper_encode(void* whisper, const float* features, int32_t batch, int32_t n_mels,
                        int32_t frames, float* out, int64_t out_capacity, int64_t* out_dims);
/* Whisper::generate on id prompts (whisper.cc:232-371): prompts [batch, prompt_stride]. out_ids
 * [batch, num_hypotheses, max_len], out_lengths (-1: fewer hypotheses) / out_scores [batch,
 * num_hypotheses], out_no_speech [batch] (return_no_speech_prob). */
int ct2x_whisper_generate(void* whisper, const float* features, int32_t batch, int32_t n_mels,
                          int32_t frames, const int32_t* prompts, const int32_t* prompt_lengths,
                          int32_t prompt_stride, const struct ct2x_whisper_options* options,
                          int32_t max_len, int32_t* out_ids, int32_t* out_lengths,
                          float* out_scores, float* out_no_speech);
/* Whisper::detect_language (whisper.cc:562-630): the language token ids of every batch entry by
 * decreasing probability; out_ids / out_probs [batch, num_languages]. */
int ct2x_whisper_detect_language(void* whisper, const float* features, int32_t batch,
                                 int32_t n_mels, int32_t frames, int32_t num_languages,
                                 int32_t* out_ids, float* out_probs);
/* Whisper::align (whisper.cc:425-560): text_tokens [batch, text_stride], num_frames [batch].
 * out_pairs [batch, pair_capacity, 2] = (text token index, frame index) of the DTW path,
 * out_num_pairs [batch], out_token_probs [batch, text_stride]. */
int ct2x_whisper_align(void* whisper, const float* features, int32_t batch, int32_t n_mels,
                       int32_t frames, const int32_t* start_sequence, int32_t num_start,
                       const int32_t* text_tokens, const int32_t* text_lengths, int32_t text_stride,
                       const int32_t* num_frames, int32_t median_filter_width,
                       int32_t pair_capacity, int32_t* out_pairs, int32_t* out_num_pairs,
                       float* out_token_probs);

#ifdef __cplusplus
}
#endif
#endif /* CT2_MI355_H_ */
