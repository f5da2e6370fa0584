// extern "C" surface of the op-level boundary declared in include/ct2_mi355.h.
// Argument checking + launch + error translation only; no arithmetic happens on the host.

#include "../../include/ct2_mi355.h"

#include <memory>
#include <algorithm>
#include <atomic>
#include <cstring>

#include "kernels.h"
#include "runtime.h"

using namespace ct2x;

#define API_TRY try {
#define API_CATCH                                     \
  }                                                   \
  catch (const std::exception& e) {                   \
    set_last_error(e.what());                         \
    return 1;                                         \
  }                                                   \
  return 0;

static void require_f32(int dtype, const char* what) {
  if (dtype != CT2X_F32)
    throw Error(std::string(what) + ": activations are float32 in this backend (the float16 / "
                "bfloat16 compute types keep 16-bit WEIGHTS, see ct2x_gemm_f16w)");
}

extern "C" {

const char* ct2x_last_error(void) { return last_error(); }
int ct2x_version(void) { return CT2X_VERSION; }

int ct2x_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int ct2x_set_device(int device_index) {
  API_TRY
  CT2X_HIP_CHECK(hipSetDevice(device_index));
  API_CATCH
}

int ct2x_get_device(int* device_index) {
  API_TRY
  CT2X_HIP_CHECK(hipGetDevice(device_index));
  API_CATCH
}

int ct2x_device_info(int device_index, char* name, int name_size, char* arch, int arch_size,
                     int* compute_units, int64_t* hbm_bytes) {
  API_TRY
  hipDeviceProp_t prop;
  CT2X_HIP_CHECK(hipGetDeviceProperties(&prop, device_index));
  if (name && name_size > 0) {
    std::strncpy(name, prop.name, name_size - 1);
    name[name_size - 1] = 0;
  }
  if (arch && arch_size > 0) {
    std::strncpy(arch, prop.gcnArchName, arch_size - 1);
    arch[arch_size - 1] = 0;
  }
  if (compute_units) *compute_units = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = static_cast<int64_t>(prop.totalGlobalMem);
  API_CATCH
}

void* ct2x_thread_stream(void) {
  try {
    return thread_stream();
  } catch (const std::exception& e) {
    set_last_error(e.what());
    return nullptr;
  }
}

int ct2x_stream_synchronize(void* stream) {
  API_TRY
  CT2X_HIP_CHECK(hipStreamSynchronize(resolve_stream(stream)));
  API_CATCH
}

int ct2x_device_synchronize(void) {
  API_TRY
  CT2X_HIP_CHECK(hipDeviceSynchronize());
  API_CATCH
}

int ct2x_malloc(void** d_ptr, size_t bytes) {
  API_TRY
  *d_ptr = device_malloc(bytes);
  API_CATCH
}

int ct2x_free(void* d_ptr) {
  API_TRY
  device_free(d_ptr);
  API_CATCH
}

int ct2x_clear_cache(void) {
  API_TRY
  clear_cache();
  API_CATCH
}

int64_t ct2x_allocated_bytes(void) { return allocated_bytes(); }

int ct2x_copy_to_device(const void* h_src, void* d_dst, size_t bytes, void* stream) {
  API_TRY
  CT2X_HIP_CHECK(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice,
                                resolve_stream(stream)));
  // Pageable host memory: the call returns once the source buffer may be reused.
  API_CATCH
}

int ct2x_copy_to_host(const void* d_src, void* h_dst, size_t bytes, void* stream) {
  API_TRY
  hipStream_t s = resolve_stream(stream);
  CT2X_HIP_CHECK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, s));
  CT2X_HIP_CHECK(hipStreamSynchronize(s));  // callers read the host buffer immediately
  API_CATCH
}

int ct2x_copy_device(const void* d_src, void* d_dst, size_t bytes, void* stream) {
  API_TRY
  CT2X_HIP_CHECK(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice,
                                resolve_stream(stream)));
  API_CATCH
}

int ct2x_event_create(void** event) {
  API_TRY
  hipEvent_t e;
  CT2X_HIP_CHECK(hipEventCreate(&e));
  *event = e;
  API_CATCH
}

int ct2x_event_destroy(void* event) {
  API_TRY
  CT2X_HIP_CHECK(hipEventDestroy(static_cast<hipEvent_t>(event)));
  API_CATCH
}

int ct2x_event_record(void* event, void* stream) {
  API_TRY
  CT2X_HIP_CHECK(hipEventRecord(static_cast<hipEvent_t>(event), resolve_stream(stream)));
  API_CATCH
}

int ct2x_event_elapsed_ms(void* start, void* stop, float* ms) {
  API_TRY
  CT2X_HIP_CHECK(hipEventSynchronize(static_cast<hipEvent_t>(stop)));
  CT2X_HIP_CHECK(hipEventElapsedTime(ms, static_cast<hipEvent_t>(start),
                                     static_cast<hipEvent_t>(stop)));
  API_CATCH
}

// ---- ops ------------------------------------------------------------------------------

int ct2x_quantize_s8(const void* d_x, int dtype, int64_t rows, int64_t depth,
                     int round_before_cast, int8_t* d_q, float* d_scales, void* stream) {
  API_TRY
  require_f32(dtype, "ct2x_quantize_s8");
  launch_quantize_s8(static_cast<const float*>(d_x), rows, depth, round_before_cast, d_q,
                     d_scales, resolve_stream(stream));
  check_last_launch("quantize_s8");
  API_CATCH
}

int ct2x_gemm_s8s8s32(const int8_t* d_a, const int8_t* d_b, int64_t m, int64_t n, int64_t k,
                      int32_t* d_c, void* stream) {
  API_TRY
  GemmS8Epilogue ep;
  ep.mode = 0;
  ep.c_i32 = d_c;
  ep.ldy = n;
  launch_gemm_s8(d_a, d_b, m, n, k, ep, resolve_stream(stream));
  check_last_launch("gemm_s8s8s32");
  API_CATCH
}

int ct2x_dequantize_gemm_output(const int32_t* d_c, const float* d_a_scale,
                                const float* d_b_scale, const float* d_bias, int64_t m,
                                int64_t n, int activation, void* d_y, int dtype, void* stream) {
  API_TRY
  require_f32(dtype, "ct2x_dequantize_gemm_output");
  launch_dequantize_gemm_output(d_c, d_a_scale, d_b_scale, d_bias, m, n, activation,
                                static_cast<float*>(d_y), resolve_stream(stream));
  check_last_launch("dequantize_gemm_output");
  API_CATCH
}

int ct2x_gemm_s8_dequant(const int8_t* d_qa, const float* d_a_scale, const int8_t* d_w,
                         const float* d_w_scale, const float* d_bias, const void* d_residual,
                         int64_t m, int64_t n, int64_t k, int activation, void* d_y, int dtype,
                         void* stream) {
  API_TRY
  require_f32(dtype, "ct2x_gemm_s8_dequant");
  GemmS8Epilogue ep;
  ep.mode = 1;
  ep.a_scale = d_a_scale;
  ep.w_scale = d_w_scale;
  ep.bias = d_bias;
  ep.residual = static_cast<const float*>(d_residual);
  ep.y = static_cast<float*>(d_y);
  ep.act = activation;
  ep.ldy = n;
  launch_gemm_s8(d_qa, d_w, m, n, k, ep, resolve_stream(stream));
  check_last_launch("gemm_s8_dequant");
  API_CATCH
}

// scratch layout: [m, kSubSlots] packed maxima | ticket, step | [m] history ids | [m] history scores
int64_t ct2x_gemm_s8_argmax_scratch_bytes(int64_t m) {
  return static_cast<int64_t>(sizeof(unsigned long long)) * m * GreedyFuse::kSubSlots + 64 + 8 * m;
}

int ct2x_gemm_s8_argmax(const int8_t* d_qa, const float* d_a_scale, const int8_t* d_w,
                        const float* d_w_scale, const float* d_bias, int64_t m, int64_t n, int64_t k,
                        int32_t* d_ids_out, void* d_scratch, void* stream) {
  API_TRY
  CT2X_CHECK(gemm_s8_argmax_applicable(m, n, k),
             "ct2x_gemm_s8_argmax: needs m <= 32, n >= 8192, k % 16 == 0, k < 4096");
  char* base = static_cast<char*>(d_scratch);
  GemmS8Epilogue ep;
  ep.mode = 1;
  ep.a_scale = d_a_scale;
  ep.w_scale = d_w_scale;
  ep.bias = d_bias;
  ep.y = reinterpret_cast<float*>(base);  // never written in this form
  ep.ldy = n;
  ep.gf.slots = reinterpret_cast<unsigned long long*>(base);
  base += sizeof(unsigned long long) * m * GreedyFuse::kSubSlots;
  ep.gf.ticket = reinterpret_cast<unsigned int*>(base);
  ep.gf.step_ptr = reinterpret_cast<const int32_t*>(base + 16);  // stays 0: row 0 of the history
  base += 64;
  ep.gf.next_ids = d_ids_out;
  ep.gf.hist_ids = reinterpret_cast<int32_t*>(base);
  ep.gf.hist_scores = reinterpret_cast<float*>(base + 4 * m);
  hipStream_t s = resolve_stream(stream);
  launch_gemm_s8(d_qa, d_w, m, n, k, ep, s);
  check_last_launch("gemm_s8_argmax");
  CT2X_HIP_CHECK(hipMemsetAsync(ep.gf.hist_ids, 0, 8 * m, s));
  API_CATCH
}

int ct2x_gemm_s8_glu(const int8_t* d_qa, const float* d_a_scale, const int8_t* d_w0,
                     const float* d_w0_scale, const float* d_b0, const int8_t* d_w1,
                     const float* d_w1_scale, const float* d_b1, int64_t m, int64_t n, int64_t k,
                     int activation, void* d_y, int dtype, void* stream) {
  API_TRY
  require_f32(dtype, "ct2x_gemm_s8_glu");
  CT2X_CHECK(m <= 32 && k % 16 == 0, "ct2x_gemm_s8_glu: decode rows only (m <= 32, k % 16 == 0)");
  GemmS8Epilogue ep;
  ep.mode = 1;
  ep.a_scale = d_a_scale;
  ep.w_scale = d_w0_scale;
  ep.bias = d_b0;
  ep.b2 = d_w1;
  ep.w_scale2 = d_w1_scale;
  ep.bias2 = d_b1;
  ep.y = static_cast<float*>(d_y);
  ep.act = activation;
  ep.ldy = n;
  launch_gemm_s8(d_qa, d_w0, m, n, k, ep, resolve_stream(stream));
  check_last_launch("gemm_s8_glu");
  API_CATCH
}

int ct2x_gemm_s8_glu_amax(const int8_t* d_qa, const float* d_a_scale, const int8_t* d_w0,
                          const float* d_w0_scale, const float* d_b0, const int8_t* d_w1,
                          const float* d_w1_scale, const float* d_b1, int64_t m, int64_t n, int64_t k,
                          int activation, float* d_y, float* d_row_amax_out, void* stream) {
  API_TRY
  CT2X_CHECK(m <= 32 && k % 16 == 0, "ct2x_gemm_s8_glu_amax: decode rows only (m <= 32, k % 16 == 0)");
  CT2X_CHECK(d_row_amax_out != nullptr, "ct2x_gemm_s8_glu_amax: d_row_amax_out is NULL");
  GemmS8Epilogue ep;
  ep.mode = 1;
  ep.a_scale = d_a_scale;
  ep.w_scale = d_w0_scale;
  ep.bias = d_b0;
  ep.b2 = d_w1;
  ep.w_scale2 = d_w1_scale;
  ep.bias2 = d_b1;
  ep.y = d_y;
  ep.act = activation;
  ep.ldy = n;
  ep.row_amax_out = d_row_amax_out;
  launch_gemm_s8(d_qa, d_w0, m, n, k, ep, resolve_stream(stream));
  check_last_launch("gemm_s8_glu_amax");
  API_CATCH
}

int ct2x_gemm_s8_stream_packable(int64_t n, int64_t k, int dual) {
  try {
    return gemm_s8_stream_pack_applicable(n, k, dual != 0) ? 1 : 0;
  } catch (...) {
    return 0;
  }
}

int ct2x_gemm_s8_stream_pack(const int8_t* d_w, int64_t n, int64_t k, int dual, int8_t* d_packed,
                             void* stream) {
  API_TRY
  CT2X_CHECK(gemm_s8_stream_pack_applicable(n, k, dual != 0),
             "ct2x_gemm_s8_stream_pack: this (n, k) has no streaming form");
  launch_pack_stream_weight(d_w, n, k, dual != 0, d_packed, resolve_stream(stream));
  check_last_launch("gemm_s8_stream_pack");
  API_CATCH
}

int ct2x_gemm_s8_dequant_packed(const int8_t* d_qa, const float* d_a_scale, const int8_t* d_w_packed,
                                const float* d_w_scale, const float* d_bias, const float* d_residual,
                                int64_t m, int64_t n, int64_t k, int activation, float* d_y,
                                void* stream) {
  API_TRY
  CT2X_CHECK(gemm_s8_stream_applicable(m, n, k, false),
             "ct2x_gemm_s8_dequant_packed: needs m <= 16 and a (n, k) with a streaming form");
  GemmS8Epilogue ep;
  ep.mode = 1;
  ep.a_scale = d_a_scale;
  ep.w_scale = d_w_scale;
  ep.bias = d_bias;
  ep.residual = d_residual;
  ep.y = d_y;
  ep.act = activation;
  ep.ldy = n;
  ep.packed_b = true;
  launch_gemm_s8(d_qa, d_w_packed, m, n, k, ep, resolve_stream(stream));
  check_last_launch("gemm_s8_dequant_packed");
  API_CATCH
}

int ct2x_gemm_s8_glu_packed(const int8_t* d_qa, const float* d_a_scale, const int8_t* d_w0_packed,
                            const float* d_w0_scale, const float* d_b0, const int8_t* d_w1_packed,
                            const float* d_w1_scale, const float* d_b1, int64_t m, int64_t n,
                            int64_t k, int activation, float* d_y, float* d_row_amax_out,
                            void* stream) {
  API_TRY
  CT2X_CHECK(gemm_s8_stream_applicable(m, n, k, true),
             "ct2x_gemm_s8_glu_packed: needs m <= 16 and a (n, k) with a gated streaming form");
  GemmS8Epilogue ep;
  ep.mode = 1;
  ep.a_scale = d_a_scale;
  ep.w_scale = d_w0_scale;
  ep.bias = d_b0;
  ep.b2 = d_w1_packed;
  ep.w_scale2 = d_w1_scale;
  ep.bias2 = d_b1;
  ep.y = d_y;
  ep.act = activation;
  ep.ldy = n;
  ep.row_amax_out = d_row_amax_out;
  ep.packed_b = true;
  launch_gemm_s8(d_qa, d_w0_packed, m, n, k, ep, resolve_stream(stream));
  check_last_launch("gemm_s8_glu_packed");
  API_CATCH
}

namespace {
// Row buffers of the leader forms (gemm_s8_stream.hip AMODE 3 / 4) for the duration of one op call:
// the int8 rows, their scales and the zeroed tag words (the engine keeps these per session).
struct LeadScratch {
  void* base = nullptr;
  LeadScratch(GemmS8Epilogue& ep, int64_t k, hipStream_t s) {
    const size_t qbytes = static_cast<size_t>(16) * k;
    base = device_malloc(qbytes + 2 * 16 * sizeof(float));
    ep.lead_q = static_cast<int8_t*>(base);
    ep.lead_scale = reinterpret_cast<float*>(static_cast<char*>(base) + qbytes);
    ep.lead_flag = reinterpret_cast<unsigned*>(ep.lead_scale + 16);
    ep.lead_tag = 1;
    CT2X_HIP_CHECK(hipMemsetAsync(ep.lead_flag, 0, 16 * sizeof(unsigned), s));
  }
  ~LeadScratch() { device_free(base); }
  LeadScratch(const LeadScratch&) = delete;
  LeadScratch& operator=(const LeadScratch&) = delete;
};
}  // namespace

int ct2x_debug_set_stream_prologue(int mask) {
  return gemm_s8_stream_set_prologue_mask(mask);
}

int ct2x_norm_gemm_s8_dequant_packed(const float* d_x, const float* d_gamma, float eps,
                                     int rms_use_residual, int quant_flags, const int8_t* d_w_packed,
                                     const float* d_w_scale, const float* d_bias,
                                     const float* d_residual, int64_t m, int64_t n, int64_t k,
                                     int activation, float* d_y, float* d_row_amax_out, void* stream) {
  API_TRY
  CT2X_CHECK(gemm_s8_stream_norm_applicable(m, n, k, false, false),
             "ct2x_norm_gemm_s8_dequant_packed: needs m <= 16, k = 4096 and a n with a streaming form");
  CT2X_CHECK(d_x != nullptr && d_gamma != nullptr, "ct2x_norm_gemm_s8_dequant_packed: d_x / d_gamma is NULL");
  GemmS8Epilogue ep;
  ep.mode = 1;
  ep.w_scale = d_w_scale;
  ep.bias = d_bias;
  ep.residual = d_residual;
  ep.y = d_y;
  ep.act = activation;
  ep.ldy = n;
  ep.norm_x = d_x;
  ep.norm_gamma = d_gamma;
  ep.norm_eps = eps;
  ep.norm_rms_residual = rms_use_residual != 0;
  ep.quant_mode = quant_flags;
  ep.row_amax_out = d_row_amax_out;
  ep.packed_b = true;
  std::unique_ptr<LeadScratch> lead;  // CT2X_STREAM_PROLOGUE bit 2: the norm by sixteen leader workgroups
  if ((gemm_s8_stream_prologue_mask() & 4) && gemm_s8_stream_lead_norm_applicable(m, n, k, false, false))
    lead.reset(new LeadScratch(ep, k, resolve_stream(stream)));
  launch_gemm_s8(nullptr, d_w_packed, m, n, k, ep, resolve_stream(stream));
  check_last_launch("norm_gemm_s8_dequant_packed");
  API_CATCH
}

int ct2x_norm_gemm_s8_glu_packed(const float* d_x, const float* d_gamma, float eps,
                                 int rms_use_residual, int quant_flags, const int8_t* d_w0_packed,
                                 const float* d_w0_scale, const float* d_b0, const int8_t* d_w1_packed,
                                 const float* d_w1_scale, const float* d_b1, int64_t m, int64_t n,
                                 int64_t k, int activation, float* d_y, float* d_row_amax_out,
                                 void* stream) {
  API_TRY
  CT2X_CHECK(gemm_s8_stream_norm_applicable(m, n, k, true, false),
             "ct2x_norm_gemm_s8_glu_packed: needs m <= 16, k = 4096 and a n with a gated streaming form");
  CT2X_CHECK(d_x != nullptr && d_gamma != nullptr, "ct2x_norm_gemm_s8_glu_packed: d_x / d_gamma is NULL");
  GemmS8Epilogue ep;
  ep.mode = 1;
  ep.w_scale = d_w0_scale;
  ep.bias = d_b0;
  ep.b2 = d_w1_packed;
  ep.w_scale2 = d_w1_scale;
  ep.bias2 = d_b1;
  ep.y = d_y;
  ep.act = activation;
  ep.ldy = n;
  ep.norm_x = d_x;
  ep.norm_gamma = d_gamma;
  ep.norm_eps = eps;
  ep.norm_rms_residual = rms_use_residual != 0;
  ep.quant_mode = quant_flags;
  ep.row_amax_out = d_row_amax_out;
  ep.packed_b = true;
  std::unique_ptr<LeadScratch> lead;
  if ((gemm_s8_stream_prologue_mask() & 4) && gemm_s8_stream_lead_norm_applicable(m, n, k, true, false))
    lead.reset(new LeadScratch(ep, k, resolve_stream(stream)));
  launch_gemm_s8(nullptr, d_w0_packed, m, n, k, ep, resolve_stream(stream));
  check_last_launch("norm_gemm_s8_glu_packed");
  API_CATCH
}

int ct2x_gemm_s8_dequant_quantizing_packed(const float* d_x, const float* d_row_amax, int amax_parts,
                                           int quant_flags, const int8_t* d_w_packed,
                                           const float* d_w_scale, const float* d_bias,
                                           const float* d_residual, int64_t m, int64_t n, int64_t k,
                                           int activation, float* d_y, void* stream) {
  API_TRY
  // (CT2X_STREAM_PROLOGUE bit 3: ops::Quantize by sixteen leader workgroups - any K with a streaming form)
  const bool lead_q = (gemm_s8_stream_prologue_mask() & 8) && gemm_s8_stream_lead_quant_applicable(m, n, k);
  CT2X_CHECK(lead_q || gemm_s8_stream_qol_applicable(m, n, k),
             "ct2x_gemm_s8_dequant_quantizing_packed: needs m <= 16, k = 4096 and a n with a streaming form");
  CT2X_CHECK(d_x != nullptr && d_row_amax != nullptr && amax_parts >= 1,
             "ct2x_gemm_s8_dequant_quantizing_packed: d_x / d_row_amax is NULL or amax_parts < 1");
  GemmS8Epilogue ep;
  ep.mode = 1;
  ep.w_scale = d_w_scale;
  ep.bias = d_bias;
  ep.residual = d_residual;
  ep.y = d_y;
  ep.act = activation;
  ep.ldy = n;
  ep.a_f32 = d_x;
  ep.a_amax = d_row_amax;
  ep.a_amax_parts = amax_parts;
  ep.quant_mode = quant_flags;
  ep.packed_b = true;
  std::unique_ptr<LeadScratch> lead;
  if (lead_q) lead.reset(new LeadScratch(ep, k, resolve_stream(stream)));
  launch_gemm_s8(nullptr, d_w_packed, m, n, k, ep, resolve_stream(stream));
  check_last_launch("gemm_s8_dequant_quantizing_packed");
  API_CATCH
}

int ct2x_gemm_f16w(const float* d_a, const void* d_w, int w_dtype, int64_t m, int64_t n, int64_t k,
                   const float* d_bias, const float* d_residual, int activation, float* d_c,
                   void* stream) {
  API_TRY
  CT2X_CHECK(w_dtype == CT2X_F16 || w_dtype == CT2X_BF16,
             "ct2x_gemm_f16w: the weight type must be CT2X_F16 or CT2X_BF16");
  // decode rows: K slices over workgroups need a partial-sum workspace (the engine owns one per
  // session; here it comes from the stream-ordered allocator for the duration of the launch)
  float* ws = nullptr;
  size_t ws_floats = 0;
  unsigned* tickets = nullptr;
  const size_t n_tickets = 1024;
  if (m <= 256 && m > 0 && n > 0) {  // (129 .. 256 rows: up to 16 K slices, gemm_h_rows256_kernel)
    ws_floats = static_cast<size_t>(m <= 64 ? 20 : (m <= 128 ? 8 : 16)) * m * n;
    ws = static_cast<float*>(device_malloc(ws_floats * sizeof(float) + n_tickets * sizeof(unsigned)));
    // (the K slices of decode rows are summed inside the launch: a ticket per 64-column tile)
    tickets = reinterpret_cast<unsigned*>(ws + ws_floats);
    CT2X_HIP_CHECK(hipMemsetAsync(tickets, 0, n_tickets * sizeof(unsigned), resolve_stream(stream)));
  }
  launch_gemm_h(d_a, d_w, w_dtype == CT2X_BF16, m, n, k, d_bias, d_residual, activation, d_c,
                resolve_stream(stream), ws, ws_floats, tickets, tickets ? n_tickets : 0);
  if (ws) device_free(ws);
  check_last_launch("gemm_f16w");
  API_CATCH
}

int ct2x_norm_gemm_s8_dequant(const float* d_x, const float* d_gamma, const float* d_beta, float eps,
                              int rms_use_residual, int quant_flags, const int8_t* d_w,
                              const float* d_w_scale, const float* d_bias, const float* d_residual,
                              int64_t m, int64_t n, int64_t k, int activation, float* d_y,
                              void* stream) {
  API_TRY
  CT2X_CHECK(gemm_s8_norm_applicable(m, n, k),
             "ct2x_norm_gemm_s8_dequant: needs m <= 128, k in {64, 128, 256, 512}, n < 8192");
  GemmS8Epilogue ep;
  ep.mode = 1;
  ep.w_scale = d_w_scale;
  ep.bias = d_bias;
  ep.residual = d_residual;
  ep.y = d_y;
  ep.act = activation;
  ep.ldy = n;
  ep.norm_x = d_x;
  ep.norm_gamma = d_gamma;
  ep.norm_beta = d_beta;
  ep.norm_eps = eps;
  ep.norm_rms_residual = rms_use_residual != 0;
  ep.quant_mode = quant_flags;
  launch_gemm_s8(nullptr, d_w, m, n, k, ep, resolve_stream(stream));
  check_last_launch("norm_gemm_s8_dequant");
  API_CATCH
}

int ct2x_norm_gemm_s8_dequant_amax(const float* d_x, const float* d_gamma, const float* d_beta,
                                   float eps, int rms_use_residual, int quant_flags,
                                   const int8_t* d_w, const float* d_w_scale, const float* d_bias,
                                   const float* d_residual, int64_t m, int64_t n, int64_t k,
                                   int activation, float* d_y, float* d_row_amax_out, void* stream) {
  API_TRY
  CT2X_CHECK(gemm_s8_norm_applicable(m, n, k),
             "ct2x_norm_gemm_s8_dequant_amax: needs m <= 128, k in {64, 128, 256, 512}, n < 8192");
  CT2X_CHECK(d_row_amax_out != nullptr, "ct2x_norm_gemm_s8_dequant_amax: d_row_amax_out is NULL");
  GemmS8Epilogue ep;
  ep.mode = 1;
  ep.w_scale = d_w_scale;
  ep.bias = d_bias;
  ep.residual = d_residual;
  ep.y = d_y;
  ep.act = activation;
  ep.ldy = n;
  ep.norm_x = d_x;
  ep.norm_gamma = d_gamma;
  ep.norm_beta = d_beta;
  ep.norm_eps = eps;
  ep.norm_rms_residual = rms_use_residual != 0;
  ep.quant_mode = quant_flags;
  ep.row_amax_out = d_row_amax_out;
  launch_gemm_s8(nullptr, d_w, m, n, k, ep, resolve_stream(stream));
  check_last_launch("norm_gemm_s8_dequant_amax");
  API_CATCH
}

int ct2x_gemm_s8_dequant_quantizing(const float* d_x, const float* d_row_amax, int quant_flags,
                                    const int8_t* d_w, const float* d_w_scale, const float* d_bias,
                                    const float* d_residual, int64_t m, int64_t n, int64_t k,
                                    int activation, float* d_y, void* stream) {
  API_TRY
  CT2X_CHECK(gemm_s8_qol_applicable(m, n, k),
             "ct2x_gemm_s8_dequant_quantizing: needs m <= 128, k % 64 == 0, n <= 4096, m * k <= 262144");
  CT2X_CHECK(d_x != nullptr && d_row_amax != nullptr,
             "ct2x_gemm_s8_dequant_quantizing: d_x / d_row_amax is NULL");
  GemmS8Epilogue ep;
  ep.mode = 1;
  ep.w_scale = d_w_scale;
  ep.bias = d_bias;
  ep.residual = d_residual;
  ep.y = d_y;
  ep.act = activation;
  ep.ldy = n;
  ep.a_f32 = d_x;
  ep.a_amax = d_row_amax;
  ep.quant_mode = quant_flags;
  launch_gemm_s8(nullptr, d_w, m, n, k, ep, resolve_stream(stream));
  check_last_launch("gemm_s8_dequant_quantizing");
  API_CATCH
}

int ct2x_awq_dequantize(const int32_t* d_qweight, const void* d_scales_f16, const int32_t* d_qzeros,
                        int64_t in_c, int64_t out_c, int64_t group, void* d_out_f16, void* stream) {
  API_TRY
  launch_awq_dequantize_gemm(d_qweight, d_scales_f16, d_qzeros, in_c, out_c, group, d_out_f16,
                             resolve_stream(stream));
  check_last_launch("awq_dequantize");
  API_CATCH
}

int ct2x_awq_gemm(const void* d_x_f16, const int32_t* d_qweight, const void* d_scales_f16,
                  const int32_t* d_qzeros, int64_t m, int64_t in_c, int64_t out_c, int64_t group,
                  void* d_y_f16, float* d_y_f32, void* stream) {
  API_TRY
  CT2X_CHECK(d_y_f16 != nullptr || d_y_f32 != nullptr, "ct2x_awq_gemm: no output given");
  launch_awq_gemm(d_x_f16, d_qweight, d_scales_f16, d_qzeros, m, in_c, out_c, group, d_y_f16,
                  d_y_f32, resolve_stream(stream));
  check_last_launch("awq_gemm");
  API_CATCH
}

int ct2x_awq_gemv(const void* d_x_f16, const int32_t* d_qweight, const void* d_scales_f16,
                  const int32_t* d_qzeros, int64_t m, int64_t in_c, int64_t out_c, int64_t group,
                  int64_t zeros_w, int64_t sf_w, void* d_y_f16, float* d_y_f32, void* stream) {
  API_TRY
  CT2X_CHECK(d_y_f16 != nullptr || d_y_f32 != nullptr, "ct2x_awq_gemv: no output given");
  launch_awq_gemv(d_x_f16, d_qweight, d_scales_f16, d_qzeros, m, in_c, out_c, group, zeros_w, sf_w,
                  d_y_f16, d_y_f32, resolve_stream(stream));
  check_last_launch("awq_gemv");
  API_CATCH
}

int ct2x_awq_dequantize_gemv(const int32_t* d_qweight, const void* d_scales_f16,
                             const int32_t* d_qzeros, int64_t out_c, int64_t in_c, int64_t group,
                             int64_t zeros_w, int64_t sf_w, void* d_out_f16, void* stream) {
  API_TRY
  launch_awq_dequantize_gemv(d_qweight, d_scales_f16, d_qzeros, out_c, in_c, group, zeros_w, sf_w,
                             d_out_f16, resolve_stream(stream));
  check_last_launch("awq_dequantize_gemv");
  API_CATCH
}

int ct2x_awq_dequantize_bf16(const int32_t* d_qweight, const void* d_scales_bf16, const int32_t* d_qzeros,
                             int64_t in_c, int64_t out_c, int64_t group, void* d_out_bf16, void* stream) {
  API_TRY
  launch_awq_dequantize_gemm(d_qweight, d_scales_bf16, d_qzeros, in_c, out_c, group, d_out_bf16,
                             resolve_stream(stream), true);
  check_last_launch("awq_dequantize_bf16");
  API_CATCH
}

int ct2x_awq_gemm_bf16(const void* d_x_bf16, const int32_t* d_qweight, const void* d_scales_bf16,
                       const int32_t* d_qzeros, int64_t m, int64_t in_c, int64_t out_c, int64_t group,
                       void* d_y_bf16, float* d_y_f32, void* stream) {
  API_TRY
  CT2X_CHECK(d_y_bf16 != nullptr || d_y_f32 != nullptr, "ct2x_awq_gemm_bf16: no output given");
  launch_awq_gemm(d_x_bf16, d_qweight, d_scales_bf16, d_qzeros, m, in_c, out_c, group, d_y_bf16,
                  d_y_f32, resolve_stream(stream), nullptr, true);
  check_last_launch("awq_gemm_bf16");
  API_CATCH
}

int ct2x_awq_gemv_bf16(const void* d_x_bf16, const int32_t* d_qweight, const void* d_scales_bf16,
                       const int32_t* d_qzeros, int64_t m, int64_t in_c, int64_t out_c, int64_t group,
                       int64_t zeros_w, int64_t sf_w, void* d_y_bf16, float* d_y_f32, void* stream) {
  API_TRY
  CT2X_CHECK(d_y_bf16 != nullptr || d_y_f32 != nullptr, "ct2x_awq_gemv_bf16: no output given");
  launch_awq_gemv(d_x_bf16, d_qweight, d_scales_bf16, d_qzeros, m, in_c, out_c, group, zeros_w, sf_w,
                  d_y_bf16, d_y_f32, resolve_stream(stream), true);
  check_last_launch("awq_gemv_bf16");
  API_CATCH
}

int ct2x_awq_dequantize_gemv_bf16(const int32_t* d_qweight, const void* d_scales_bf16,
                                  const int32_t* d_qzeros, int64_t out_c, int64_t in_c, int64_t group,
                                  int64_t zeros_w, int64_t sf_w, void* d_out_bf16, void* stream) {
  API_TRY
  launch_awq_dequantize_gemv(d_qweight, d_scales_bf16, d_qzeros, out_c, in_c, group, zeros_w, sf_w,
                             d_out_bf16, resolve_stream(stream), true);
  check_last_launch("awq_dequantize_gemv_bf16");
  API_CATCH
}

int ct2x_dequantize_rows(const int8_t* d_x, const float* d_scale, int64_t rows, int64_t depth,
                         void* d_y, int dtype, void* stream) {
  API_TRY
  require_f32(dtype, "ct2x_dequantize_rows");
  launch_dequantize_rows(d_x, d_scale, rows, depth, static_cast<float*>(d_y),
                         resolve_stream(stream));
  check_last_launch("dequantize_rows");
  API_CATCH
}

int ct2x_gemm_f32(const float* d_a, const float* d_b, int64_t m, int64_t n, int64_t k,
                  int trans_b, const float* d_bias, const float* d_residual, int activation,
                  float* d_c, void* stream) {
  API_TRY
  launch_gemm_f32(d_a, d_b, 1, m, n, k, trans_b != 0, 1.f, 0, 0, 0, d_bias, d_residual,
                  activation, d_c, resolve_stream(stream));
  check_last_launch("gemm_f32");
  API_CATCH
}

int ct2x_layer_norm(const void* d_x, const float* d_gamma, const float* d_beta, int64_t rows,
                    int64_t depth, float epsilon, void* d_y, int dtype, void* stream) {
  API_TRY
  require_f32(dtype, "ct2x_layer_norm");
  CT2X_CHECK(d_beta != nullptr, "ct2x_layer_norm: beta is required (use ct2x_rms_norm)");
  CT2X_CHECK(depth * 4 <= 64 * 1024, "ct2x_layer_norm: depth > 16384 is not supported");
  launch_norm(static_cast<const float*>(d_x), d_gamma, d_beta, rows, depth, epsilon, false, 1,
              static_cast<float*>(d_y), nullptr, nullptr, resolve_stream(stream));
  check_last_launch("layer_norm");
  API_CATCH
}

int ct2x_rms_norm(const void* d_x, const float* d_gamma, int64_t rows, int64_t depth,
                  float epsilon, int use_residual, void* d_y, int dtype, void* stream) {
  API_TRY
  require_f32(dtype, "ct2x_rms_norm");
  CT2X_CHECK(depth * 4 <= 64 * 1024, "ct2x_rms_norm: depth > 16384 is not supported");
  launch_norm(static_cast<const float*>(d_x), d_gamma, nullptr, rows, depth, epsilon,
              use_residual != 0, 1, static_cast<float*>(d_y), nullptr, nullptr,
              resolve_stream(stream));
  check_last_launch("rms_norm");
  API_CATCH
}

int ct2x_norm_quantize_s8(const float* d_x, const float* d_gamma, const float* d_beta,
                          int64_t rows, int64_t depth, float epsilon, int use_residual,
                          int round_before_cast, float* d_y, int8_t* d_q, float* d_scales,
                          void* stream) {
  API_TRY
  CT2X_CHECK(depth * 4 <= 64 * 1024, "ct2x_norm_quantize_s8: depth > 16384 is not supported");
  launch_norm(d_x, d_gamma, d_beta, rows, depth, epsilon, use_residual != 0,
              round_before_cast, d_y, d_q, d_scales, resolve_stream(stream));
  check_last_launch("norm_quantize_s8");
  API_CATCH
}

int ct2x_softmax(const void* d_x, const int32_t* d_lengths, int64_t rows, int64_t depth,
                 int log, void* d_y, int dtype, void* stream) {
  API_TRY
  require_f32(dtype, "ct2x_softmax");
  launch_softmax(static_cast<const float*>(d_x), d_lengths, rows, depth, log != 0,
                 static_cast<float*>(d_y), resolve_stream(stream));
  check_last_launch("softmax");
  API_CATCH
}

int ct2x_rotary(const float* d_x, const float* d_sin, const float* d_cos, int64_t batch,
                int64_t time, int64_t depth, int64_t ndims, int interleave, float* d_y,
                void* stream) {
  API_TRY
  CT2X_CHECK(ndims <= depth && ndims % 2 == 0, "ct2x_rotary: invalid ndims");
  launch_rotary(d_x, d_sin, d_cos, batch, time, depth, ndims, interleave != 0, d_y,
                resolve_stream(stream));
  check_last_launch("rotary");
  API_CATCH
}

int ct2x_topk(const void* d_x, int dtype, int64_t rows, int64_t depth, int64_t k,
              void* d_values, int32_t* d_indices, void* stream) {
  API_TRY
  require_f32(dtype, "ct2x_topk");
  CT2X_CHECK(k >= 1 && k <= depth, "ct2x_topk: k must be in [1, depth]");
  // scratch of the two-launch form: one grow-only buffer per calling thread (= per stream),
  // like the temporaries ops::TopK takes from the allocator (src/ops/topk_gpu.cu:35-53)
  struct Scratch {
    void* ptr = nullptr;
    size_t bytes = 0;
    ~Scratch() {
      try { if (ptr) device_free(ptr); } catch (...) {}
    }
  };
  static thread_local Scratch scratch;
  const size_t need = topk_workspace_bytes(rows, depth, k);
  if (need > scratch.bytes) {
    if (scratch.ptr) {  // an earlier launch of this thread may still read the old block
      CT2X_HIP_CHECK(hipStreamSynchronize(resolve_stream(stream)));
      device_free(scratch.ptr);
    }
    scratch.ptr = device_malloc(need);
    scratch.bytes = need;
  }
  launch_topk(static_cast<const float*>(d_x), rows, depth, k, static_cast<float*>(d_values),
              d_indices, resolve_stream(stream), scratch.ptr, scratch.bytes);
  check_last_launch("topk");
  API_CATCH
}

int ct2x_process_logits(float* d_logits, int64_t rows, int64_t vocab, const int32_t* d_sequences,
                        int32_t length, float repetition_penalty, int32_t no_repeat_ngram_size,
                        const int32_t* d_suppress_ids, const int32_t* d_suppress_offsets,
                        int32_t num_suppress, void* stream) {
  API_TRY
  CT2X_CHECK(length >= 0 && length <= 16000, "ct2x_process_logits: length must be in [0, 16000]");
  CT2X_CHECK(repetition_penalty > 0.f, "ct2x_process_logits: repetition_penalty must be > 0");
  CT2X_CHECK(no_repeat_ngram_size >= 0 && num_suppress >= 0, "ct2x_process_logits: bad sizes");
  CT2X_CHECK(length == 0 || d_sequences != nullptr, "ct2x_process_logits: sequences missing");
  launch_process_logits(d_logits, rows, vocab, d_sequences, 1, length, nullptr, 0, length,
                        std::max(length, 1), repetition_penalty, no_repeat_ngram_size,
                        d_suppress_ids, d_suppress_offsets, num_suppress, resolve_stream(stream));
  check_last_launch("process_logits");
  API_CATCH
}

int ct2x_convert(const void* d_x, int src_dtype, void* d_y, int dst_dtype, int64_t size, void* stream) {
  API_TRY
  auto is_float = [](int d) { return d == CT2X_F32 || d == CT2X_F16 || d == CT2X_BF16; };
  CT2X_CHECK(is_float(src_dtype) && is_float(dst_dtype) && size >= 0,
             "ct2x_convert: float32 / float16 / bfloat16 only");
  launch_convert(d_x, src_dtype, d_y, dst_dtype, size, resolve_stream(stream));
  check_last_launch("convert");
  API_CATCH
}

int ct2x_conv1d(const float* d_x, const float* d_w, int64_t batch, int64_t in_channels, int64_t length,
                int64_t out_channels, int64_t kernel, int64_t stride, int64_t padding,
                int64_t dilation, int64_t groups, float* d_y, void* stream) {
  API_TRY
  CT2X_CHECK(batch >= 0 && in_channels >= 1 && out_channels >= 1 && kernel >= 1 && stride >= 1 &&
                 dilation >= 1 && padding >= 0 && groups >= 1 && in_channels % groups == 0 &&
                 out_channels % groups == 0,
             "ct2x_conv1d: bad geometry");
  launch_conv1d(d_x, d_w, batch, in_channels, length, out_channels, kernel, stride, padding, dilation,
                groups, d_y, resolve_stream(stream));
  check_last_launch("conv1d");
  API_CATCH
}

int ct2x_median_filter(const float* d_x, int64_t rows, int64_t depth, int64_t width, float* d_y,
                       void* stream) {
  API_TRY
  CT2X_CHECK(width >= 1 && width <= 63 && width % 2 == 1, "ct2x_median_filter: odd width <= 63");
  CT2X_CHECK(depth > width / 2, "ct2x_median_filter: the axis must be longer than half the window");
  launch_median_filter(d_x, rows, depth, width, d_y, resolve_stream(stream));
  check_last_launch("median_filter");
  API_CATCH
}

int ct2x_alibi_add(const float* d_x, const float* d_alibi, int64_t batch, int64_t heads,
                   int64_t queries, int64_t keys, int64_t alibi_keys, int64_t alibi_offset,
                   float* d_y, void* stream) {
  API_TRY
  CT2X_CHECK(alibi_offset >= 0 && alibi_offset + keys <= alibi_keys,
             "ct2x_alibi_add: the bias table is shorter than offset + keys");
  launch_alibi_add(d_x, d_alibi, batch, heads, queries, keys, alibi_keys, alibi_offset, d_y,
                   resolve_stream(stream));
  check_last_launch("alibi_add");
  API_CATCH
}

int ct2x_topp_mask(const float* d_x, const float* d_probs, int64_t rows, int64_t depth, float p,
                   float mask_value, float* d_y, void* stream) {
  API_TRY
  CT2X_CHECK(rows >= 0 && depth >= 1 && d_x && d_probs && d_y, "ct2x_topp_mask: bad arguments");
  launch_topp_mask(d_x, d_probs, rows, depth, p, mask_value, d_y, resolve_stream(stream));
  check_last_launch("topp_mask");
  API_CATCH
}

int ct2x_multinomial(const float* d_weights, int64_t rows, int64_t classes, int64_t samples,
                     uint32_t seed, int32_t* d_out, void* stream) {
  API_TRY
  CT2X_CHECK(rows >= 0 && classes >= 1 && samples >= 1 && d_weights && d_out,
             "ct2x_multinomial: bad arguments");
  static std::atomic<uint32_t> calls{0};  // successive calls draw fresh numbers for one seed
  launch_multinomial(d_weights, rows, classes, samples, seed, calls.fetch_add(1), d_out,
                     resolve_stream(stream));
  check_last_launch("multinomial");
  API_CATCH
}

int ct2x_gumbel_noise(const float* d_x, float* d_y, int64_t size, uint32_t seed, void* stream) {
  API_TRY
  CT2X_CHECK(size >= 0 && (size == 0 || (d_x && d_y)), "ct2x_gumbel_noise: bad arguments");
  static std::atomic<uint32_t> calls{0};  // successive calls draw fresh numbers for one seed
  launch_gumbel_noise(d_x, d_y, size, seed, calls.fetch_add(1), resolve_stream(stream));
  check_last_launch("gumbel_noise");
  API_CATCH
}

int ct2x_debug_fast_div_check(uint64_t pairs, uint32_t seed, uint64_t* mismatches) {
  API_TRY
  CT2X_CHECK(mismatches != nullptr, "ct2x_debug_fast_div_check: mismatches is null");
  unsigned long long* d = static_cast<unsigned long long*>(device_malloc(sizeof(unsigned long long)));
  hipStream_t stream = resolve_stream(nullptr);
  CT2X_HIP_CHECK(hipMemsetAsync(d, 0, sizeof(unsigned long long), stream));
  launch_fast_div_check(pairs, seed, d, stream);
  check_last_launch("fast_div_check");
  unsigned long long h = 0;
  CT2X_HIP_CHECK(hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, stream));
  CT2X_HIP_CHECK(hipStreamSynchronize(stream));
  device_free(d);
  *mismatches = h;
  API_CATCH
}

int ct2x_random_sample(const float* d_logits, int64_t rows, int64_t vocab, int32_t sampling_topk,
                       float sampling_topp, float temperature, uint32_t seed,
                       const float* d_uniforms, const int32_t* d_step, int32_t* d_ids,
                       float* d_scores, void* stream) {
  API_TRY
  CT2X_CHECK(rows >= 0 && vocab >= 1, "ct2x_random_sample: bad shape");
  CT2X_CHECK(temperature > 0.f && sampling_topp > 0.f && sampling_topp <= 1.f,
             "ct2x_random_sample: temperature > 0, 0 < sampling_topp <= 1");
  CT2X_CHECK(d_step != nullptr && d_ids != nullptr && d_scores != nullptr,
             "ct2x_random_sample: the step counter and the outputs are device pointers");
  // history layout of the fused sampler with a one-step history: hist[step * rows + row]; the
  // caller passes *d_step == 0
  launch_random_sample(d_logits, rows, vocab, -1, 0, true, sampling_topk, sampling_topp, temperature,
                       seed, d_step, d_ids, d_ids, d_scores, resolve_stream(stream), nullptr, nullptr,
                       0, 0, true, nullptr, nullptr, d_uniforms);
  check_last_launch("random_sample");
  API_CATCH
}

int ct2x_beam_select(const float* d_topk_scores, const int32_t* d_topk_ids, int64_t batch,
                     int32_t beam_size, int64_t vocab, int32_t end_id,
                     const int32_t* d_prefix_ids, const int32_t* d_prefix_lens,
                     int32_t prefix_stride, const int32_t* d_step, int32_t first_step,
                     int32_t* d_cand_word, float* d_cand_score, int32_t* d_cand_origin,
                     int32_t* d_next_ids, float* d_next_scores, int32_t* d_parents, void* stream) {
  API_TRY
  CT2X_CHECK(beam_size >= 1 && beam_size <= 32, "ct2x_beam_select: beam_size must be in [1, 32]");
  CT2X_CHECK(d_step != nullptr, "ct2x_beam_select: the step counter is a device pointer");
  launch_beam_select(d_topk_scores, d_topk_ids, batch, beam_size, vocab, end_id, d_prefix_ids,
                     d_prefix_lens, prefix_stride, d_step, first_step, d_cand_word, d_cand_score,
                     d_cand_origin, d_next_ids, d_parents, d_next_scores, nullptr, nullptr, 0,
                     resolve_stream(stream));
  check_last_launch("beam_select");
  API_CATCH
}

int ct2x_binary(int op, const void* d_a, const void* d_b, void* d_c, int64_t size, int dtype,
                void* stream) {
  API_TRY
  launch_binary(op, d_a, d_b, d_c, size, dtype, resolve_stream(stream));
  check_last_launch("binary");
  API_CATCH
}

int ct2x_binary_scalar(int op, double a, const void* d_x, void* d_y, int64_t size, int dtype,
                       void* stream) {
  API_TRY
  launch_binary_scalar(op, a, d_x, d_y, size, dtype, resolve_stream(stream));
  check_last_launch("binary_scalar");
  API_CATCH
}

int ct2x_broadcast(int op, int kind, const void* d_a, const void* d_b, void* d_c, int64_t a_size,
                   int64_t b_size, int64_t block, int dtype, void* stream) {
  API_TRY
  launch_broadcast(op, kind, d_a, d_b, d_c, a_size, b_size, block, dtype, resolve_stream(stream));
  check_last_launch("broadcast");
  API_CATCH
}

int ct2x_fill(void* d_x, double value, int64_t size, int64_t stride, int dtype, void* stream) {
  API_TRY
  CT2X_CHECK(stride >= 1, "ct2x_fill: stride must be >= 1");
  launch_fill(d_x, value, size, stride, dtype, resolve_stream(stream));
  check_last_launch("fill");
  API_CATCH
}

int ct2x_indexed_fill(void* d_x, double value, const int32_t* d_indices, int64_t num_indices,
                      int dtype, void* stream) {
  API_TRY
  launch_indexed_fill(d_x, value, d_indices, num_indices, dtype, resolve_stream(stream));
  check_last_launch("indexed_fill");
  API_CATCH
}

int ct2x_unary(int fn, const float* d_x, float* d_y, int64_t size, void* stream) {
  API_TRY
  launch_unary(fn, d_x, d_y, size, resolve_stream(stream));
  check_last_launch("unary");
  API_CATCH
}

int ct2x_transpose(const void* d_a, const int64_t* dims, const int64_t* perm, int rank, void* d_b,
                   int dtype, void* stream) {
  API_TRY
  CT2X_CHECK(dims != nullptr && perm != nullptr, "ct2x_transpose: dims / perm missing");
  launch_transpose(d_a, dims, perm, rank, d_b, dtype, resolve_stream(stream));
  check_last_launch("transpose");
  API_CATCH
}

int ct2x_block_copy(const void* d_src, void* d_dst, int64_t outer, int64_t block_bytes,
                    int64_t src_stride_bytes, int64_t dst_stride_bytes, void* stream) {
  API_TRY
  CT2X_CHECK(outer >= 0 && block_bytes >= 0, "ct2x_block_copy: negative extent");
  launch_block_copy(d_src, d_dst, outer, block_bytes, src_stride_bytes, dst_stride_bytes,
                    resolve_stream(stream));
  check_last_launch("block_copy");
  API_CATCH
}

int ct2x_mean(const float* d_x, int64_t outer, int64_t axis, int64_t inner, int get_sum, float* d_y,
              void* stream) {
  API_TRY
  launch_mean(d_x, outer, axis, inner, get_sum != 0, d_y, resolve_stream(stream));
  check_last_launch("mean");
  API_CATCH
}

int ct2x_prepare_length_mask(const int32_t* d_lengths, int64_t batch, int64_t heads,
                             int64_t queries, int mask_future, int multi_query, int32_t* d_mask,
                             void* stream) {
  API_TRY
  launch_prepare_length_mask(d_lengths, batch, heads, queries, mask_future != 0, multi_query != 0,
                             d_mask, resolve_stream(stream));
  check_last_launch("prepare_length_mask");
  API_CATCH
}

int ct2x_gather_rows(const void* d_data, const int32_t* d_idx, int64_t n_idx,
                     int64_t row_bytes, void* d_out, void* stream) {
  API_TRY
  launch_gather_rows(d_data, d_idx, n_idx, row_bytes, d_out, resolve_stream(stream));
  check_last_launch("gather_rows");
  API_CATCH
}

int ct2x_matmul_f32(const float* d_a, const float* d_b, int64_t batch, int64_t m, int64_t n,
                    int64_t k, int trans_b, float alpha, float* d_c, void* stream) {
  API_TRY
  if (matmul_exact_applicable(m, n, k, trans_b != 0)) {
    // the reference CPU path's (MKL) summation order, bit for bit (attention_exact.hip)
    launch_matmul_exact(d_a, d_b, batch, m, n, k, trans_b != 0, alpha, d_c, resolve_stream(stream));
  } else {
    launch_gemm_f32(d_a, d_b, batch, m, n, k, trans_b != 0, alpha, m * k, n * k, m * n, nullptr,
                    nullptr, -1, d_c, resolve_stream(stream));
  }
  check_last_launch("matmul_f32");
  API_CATCH
}

int ct2x_attention_decode(const float* d_q, const float* d_k_cache, const float* d_v_cache,
                          const int32_t* d_lengths, int64_t length, int64_t rows, int64_t heads,
                          int64_t heads_kv, int64_t dhead, int64_t t_max, int64_t kv_group,
                          float scale, float* d_out, void* stream) {
  API_TRY
  CT2X_CHECK(dhead % 4 == 0 && dhead <= 512, "ct2x_attention_decode: dhead must be a multiple "
             "of 4 and <= 512");
  CT2X_CHECK(heads_kv >= 1 && heads % heads_kv == 0, "ct2x_attention_decode: heads must be a "
             "multiple of heads_kv");
  CT2X_CHECK(kv_group >= 1 && rows % kv_group == 0, "ct2x_attention_decode: rows must be a "
             "multiple of kv_group");
  CT2X_CHECK(t_max + 1 <= 16384, "ct2x_attention_decode: t_max > 16383 is not supported");
  AttnDecodeParams p;
  p.q = d_q;
  p.q_row_stride = heads * dhead;
  p.k_cache = const_cast<float*>(d_k_cache);
  p.v_cache = const_cast<float*>(d_v_cache);
  p.kv_row_stride = heads_kv * t_max * dhead;
  p.kv_head_stride = t_max * dhead;
  p.kv_time_stride = dhead;
  p.lengths = d_lengths;
  p.length = length;
  p.rows = rows;
  p.heads = heads;
  p.heads_kv = heads_kv;
  p.dhead = dhead;
  p.kv_group = kv_group;
  p.t_max = t_max;
  p.scale = scale;
  p.out = d_out;
  launch_attention_decode(p, resolve_stream(stream));
  check_last_launch("attention_decode");
  API_CATCH
}

int ct2x_cross_attention_h(const float* d_q, const void* d_kv, int kv_dtype, const int32_t* d_lengths,
                           int64_t batch, int64_t beam, int64_t heads, int64_t heads_kv,
                           int64_t s_max, float scale, float* d_out, void* stream) {
  return ct2x_cross_attention_hd(d_q, d_kv, kv_dtype, d_lengths, batch, beam, heads, heads_kv, 64,
                                 s_max, scale, d_out, stream);
}

static int cross_attention_entry(const float* d_q, const void* d_kv, int kv_dtype, const int32_t* d_lengths,
                                 int64_t batch, int64_t beam, int64_t heads, int64_t heads_kv,
                                 int64_t dhead, int64_t s_max, float scale, float* d_out,
                                 void* d_scratch, int64_t scratch_bytes, void* stream) {
  API_TRY
  CT2X_CHECK(kv_dtype == CT2X_F16 || kv_dtype == CT2X_BF16,
             "ct2x_cross_attention_h: the memory type must be CT2X_F16 or CT2X_BF16");
  CrossAttnHParams p;
  p.q = d_q;
  p.q_row_stride = heads * dhead;
  p.kv = static_cast<const uint16_t*>(d_kv);
  p.bf16 = kv_dtype == CT2X_BF16;
  p.lengths = d_lengths;
  p.batch = batch;
  p.beam = beam;
  p.heads = heads;
  p.heads_kv = heads_kv;
  p.dhead = dhead;
  p.s_max = s_max;
  p.scale = scale;
  p.out = d_out;
  p.out_row_stride = heads * dhead;
  CT2X_CHECK(cross_attention_h_applicable(p), "ct2x_cross_attention_h: unsupported shape (d_head 64 or 128, "
             "beam <= 8, heads a multiple of heads_kv, memory length within the LDS budget)");
  hipStream_t s = resolve_stream(stream);
  if (d_scratch != nullptr) {
    const int64_t floats = cross_attention_h_scratch_floats(batch, beam, heads, dhead, s_max);
    const int64_t need = floats * 4 + batch * heads * 4;
    CT2X_CHECK(scratch_bytes >= need, "ct2x_cross_attention_hd_split: scratch smaller than "
               "ct2x_cross_attention_split_scratch_bytes()");
    if (floats > 0) {
      p.split_scratch = static_cast<float*>(d_scratch);
      p.split_tickets = reinterpret_cast<unsigned*>(p.split_scratch + floats);
      CT2X_HIP_CHECK(hipMemsetAsync(p.split_tickets, 0, sizeof(unsigned) * batch * heads, s));
    }
  }
  launch_cross_attention_h(p, s);
  check_last_launch("cross_attention_h");
  API_CATCH
}

int ct2x_cross_attention_hd(const float* d_q, const void* d_kv, int kv_dtype, const int32_t* d_lengths,
                            int64_t batch, int64_t beam, int64_t heads, int64_t heads_kv,
                            int64_t dhead, int64_t s_max, float scale, float* d_out, void* stream) {
  return cross_attention_entry(d_q, d_kv, kv_dtype, d_lengths, batch, beam, heads, heads_kv, dhead,
                               s_max, scale, d_out, nullptr, 0, stream);
}

int64_t ct2x_cross_attention_split_scratch_bytes(int64_t batch, int64_t beam, int64_t heads,
                                                 int64_t dhead, int64_t s_max) {
  return cross_attention_h_scratch_floats(batch, beam, heads, dhead, s_max) * 4 + batch * heads * 4;
}

int ct2x_cross_attention_hd_split(const float* d_q, const void* d_kv, int kv_dtype,
                                  const int32_t* d_lengths, int64_t batch, int64_t beam,
                                  int64_t heads, int64_t heads_kv, int64_t dhead, int64_t s_max,
                                  float scale, float* d_out, void* d_scratch, int64_t scratch_bytes,
                                  void* stream) {
  return cross_attention_entry(d_q, d_kv, kv_dtype, d_lengths, batch, beam, heads, heads_kv, dhead,
                               s_max, scale, d_out, d_scratch, scratch_bytes, stream);
}

#ifdef CT2X_TUNING
// (TUNING builds only, not in include/ct2_mi355.h: phase stamps of the last 16-bit cross-attention launch)
int ct2x_tuning_xattn_stamps(unsigned long long* host16) {
  API_TRY
  xattn_h_read_stamps(host16);
  API_CATCH
}
#endif

#ifdef CT2X_TUNING
int ct2x_tuning_gemm_rows_stamps(unsigned long long* host16) {
  API_TRY
  gemm_h_rows_read_stamps(host16);
  API_CATCH
}
#endif

static int attention_full_entry(const char* name, const float* d_q, const float* d_k, const float* d_v,
                                const int32_t* d_lengths, int64_t batch, int64_t tq, int64_t tk,
                                int64_t heads, int64_t heads_kv, int64_t dhead, int causal,
                                int64_t q_offset, float scale, float* d_out, int half_kind,
                                void* stream) {
  API_TRY
  CT2X_CHECK(dhead % 4 == 0 && dhead <= 512, std::string(name) + ": dhead must be a multiple of 4 and <= 512");
  CT2X_CHECK(heads_kv >= 1 && heads % heads_kv == 0, std::string(name) + ": heads must be a multiple of heads_kv");
  CT2X_CHECK(tk + 1 <= 16384, std::string(name) + ": tk > 16383 is not supported");
  CT2X_CHECK(half_kind >= 0 && half_kind <= 2, std::string(name) + ": compute kind must be 1 (float16) or 2 (bfloat16)");
  AttnFullParams f;
  f.q = d_q;
  f.k = d_k;
  f.v = d_v;
  f.q_time_stride = heads * dhead;
  f.q_batch_stride = tq * heads * dhead;
  f.k_time_stride = f.v_time_stride = heads_kv * dhead;
  f.k_batch_stride = f.v_batch_stride = tk * heads_kv * dhead;
  f.k_head_stride = f.v_head_stride = dhead;
  f.lengths = d_lengths;
  f.batch = batch;
  f.tq = tq;
  f.tk = tk;
  f.heads = heads;
  f.heads_kv = heads_kv;
  f.dhead = dhead;
  f.causal = causal != 0;
  f.q_offset = q_offset;
  f.scale = scale;
  f.out = d_out;
  f.out_time_stride = heads * dhead;
  f.out_batch_stride = tq * heads * dhead;
  f.allow_inexact = half_kind != 0;
  f.half_kind = half_kind;
  launch_attention_full(f, resolve_stream(stream));
  check_last_launch(name);
  API_CATCH
}

int ct2x_attention_full(const float* d_q, const float* d_k, const float* d_v,
                        const int32_t* d_lengths, int64_t batch, int64_t tq, int64_t tk,
                        int64_t heads, int64_t heads_kv, int64_t dhead, int causal,
                        int64_t q_offset, float scale, float* d_out, void* stream) {
  return attention_full_entry("ct2x_attention_full", d_q, d_k, d_v, d_lengths, batch, tq, tk, heads,
                              heads_kv, dhead, causal, q_offset, scale, d_out, 0, stream);
}

int ct2x_attention_full_half(const float* d_q, const float* d_k, const float* d_v,
                             const int32_t* d_lengths, int64_t batch, int64_t tq, int64_t tk,
                             int64_t heads, int64_t heads_kv, int64_t dhead, int causal,
                             int64_t q_offset, float scale, int compute_kind, float* d_out,
                             void* stream) {
  if (compute_kind != 1 && compute_kind != 2) {
    API_TRY
    CT2X_CHECK(false, "ct2x_attention_full_half: compute kind must be 1 (float16) or 2 (bfloat16)");
    API_CATCH
  }
  return attention_full_entry("ct2x_attention_full_half", d_q, d_k, d_v, d_lengths, batch, tq, tk,
                              heads, heads_kv, dhead, causal, q_offset, scale, d_out, compute_kind,
                              stream);
}

}  // extern "C"
