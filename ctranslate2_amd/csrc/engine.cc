#include "engine.h"
#include "host_math.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>

#include "runtime.h"

namespace ct2x {

std::string compute_type_name(ComputeType t) {
  switch (t) {
    case ComputeType::kFloat32: return "float32";
    case ComputeType::kInt8Float32: return "int8_float32";
    case ComputeType::kFloat16: return "float16";
    case ComputeType::kBFloat16: return "bfloat16";
  }
  return "?";
}

namespace {

// The stream of the session the calling thread is driving (one worker thread per session).
thread_local hipStream_t tls_stream = nullptr;
thread_local int64_t tls_launches = 0;  // begin_kernel calls of this thread (kernels per step)

bool ends_with(const std::string& s, const std::string& suffix) {
  return s.size() >= suffix.size() && s.compare(s.size() - suffix.size(), suffix.size(), suffix) == 0;
}

// Host restatement of ops::Quantize for weights converted at load time
// (Model::ensure_dtype, src/models/model.cc:304-369 uses the model's rounding mode).
void host_quantize_rows(const std::vector<float>& w, int64_t rows, int64_t depth, bool round,
                        std::vector<int8_t>& q, std::vector<float>& scales) {
  q.resize(rows * depth);
  scales.resize(rows);
  for (int64_t i = 0; i < rows; ++i) {
    const float* x = w.data() + i * depth;
    float amax = 0.f;
    for (int64_t j = 0; j < depth; ++j) amax = std::max(amax, std::fabs(x[j]));
    const float scale = amax != 0.f ? 127.f / amax : 1.f;
    scales[i] = scale;
    for (int64_t j = 0; j < depth; ++j) {
      float v = x[j] * scale;
      if (round) v = std::nearbyintf(v);
      q[i * depth + j] = static_cast<int8_t>(static_cast<int32_t>(v));
    }
  }
}

// Resolution of the requested compute type against what this backend implements
// (reference: resolve_compute_type, src/types.cc:156-295).
// int8_float16 / int8_bfloat16 (int8 weights, 16-bit activations) resolve to int8_float32: the
// activations of this backend are fp32 throughout, a superset of the requested precision - the
// reference downgrades the same way when the 16-bit float type is not supported by the device.
ComputeType resolve(const std::string& requested, int model_weight_dtype) {
  const bool model_is_quantized = model_weight_dtype == kI8 || model_weight_dtype == kI16;
  if (requested == "float32") return ComputeType::kFloat32;
  if (requested == "float16") return ComputeType::kFloat16;
  if (requested == "bfloat16") return ComputeType::kBFloat16;
  if (requested == "int8" || requested == "int8_float32" || requested == "auto" ||
      requested == "int8_float16" || requested == "int8_bfloat16")
    return ComputeType::kInt8Float32;
  if (requested == "default" || requested.empty()) {
    if (model_is_quantized) return ComputeType::kInt8Float32;
    if (model_weight_dtype == kF16) return ComputeType::kFloat16;
    if (model_weight_dtype == kBF16) return ComputeType::kBFloat16;
    return ComputeType::kFloat32;
  }
  if (requested == "int16")
    throw std::invalid_argument("Requested int16 compute type, but this backend build only "
                                "implements int8_float32, float16, bfloat16 and float32");
  throw std::invalid_argument("Invalid compute type: " + requested);
}

uint16_t float_to_bf16_bits(float f) {  // round to nearest even, NaN kept quiet
  uint32_t u;
  std::memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return static_cast<uint16_t>((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return static_cast<uint16_t>(u >> 16);
}

uint16_t float_to_half_bits(float f) {  // IEEE binary16, round to nearest even
  uint32_t x;
  std::memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  x &= 0x7fffffffu;
  if (x >= 0x7f800000u) return static_cast<uint16_t>(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0));
  if (x >= 0x477ff000u) return static_cast<uint16_t>(sign | 0x7c00u);  // overflows to inf
  if (x < 0x33000001u) return static_cast<uint16_t>(sign);              // underflows to zero
  int32_t exp = static_cast<int32_t>(x >> 23) - 127 + 15;
  uint32_t mant = x & 0x7fffffu;
  if (exp <= 0) {  // subnormal half
    mant |= 0x800000u;
    const int shift = 14 - exp;
    const uint32_t half_mant = mant >> shift;
    const uint32_t rem = mant & ((1u << shift) - 1), halfway = 1u << (shift - 1);
    uint32_t r = half_mant;
    if (rem > halfway || (rem == halfway && (half_mant & 1u))) ++r;
    return static_cast<uint16_t>(sign | r);
  }
  uint32_t r = (static_cast<uint32_t>(exp) << 10) | (mant >> 13);
  const uint32_t rem = mant & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) ++r;  // carries into the exponent correctly
  return static_cast<uint16_t>(sign | r);
}

}  // namespace

thread_local DecodeState* Engine::pf_state_ = nullptr;

hipStream_t Engine::cur_stream() const { return tls_stream ? tls_stream : stream_; }

// ---------------------------------------------------------------------------------------------
// Loading
// ---------------------------------------------------------------------------------------------

Engine::Engine(std::shared_ptr<ModelFile> model, const std::string& compute_type,
               int device_index, std::shared_ptr<TpComm> tp)
    : model_(std::move(model)), tp_(std::move(tp)), device_(device_index) {
  if (const char* e = std::getenv("CT2X_TP_FAST")) tp_fast_ = e[0] == '1';
  if (const char* e = std::getenv("CT2X_TP_GRAPH")) tp_graph_ = e[0] != '0';
  if (tp_ && tp_->world() <= 1) tp_.reset();
  CT2X_HIP_CHECK(hipSetDevice(device_));
  stream_ = thread_stream();
  round_before_cast_ = model_->round_before_cast_in_quantization();
  // Activation quantisation reproduces the reference CPU backends' u8-shifted arithmetic by
  // default (device_utils.h, kQuantU8Shift); CT2X_S8_ACTIVATIONS=1 selects the plain signed
  // form the reference's own GPU kernels use (src/ops/quantize_gpu.cu:46-55).
  quant_mode_ = (round_before_cast_ ? 1 : 0) | 2;
  if (const char* env = std::getenv("CT2X_S8_ACTIVATIONS"))
    if (env[0] == '1') quant_mode_ &= ~2;

  int weight_dtype = kF32;
  for (const auto& v : model_->variables)
    if (ends_with(v.name, "/weight") && v.shape.size() == 2) {
      if (v.dtype == kI8 || v.dtype == kI16) { weight_dtype = v.dtype; break; }
      if (v.dtype == kF16 || v.dtype == kBF16) weight_dtype = v.dtype;
    }
  compute_type_ = resolve(compute_type, weight_dtype);

  const std::string& spec = model_->spec;
  if (spec == "TransformerSpec" || spec == "TransformerBase" || spec == "TransformerBig" ||
      spec.empty()) {
    has_encoder_ = true;
  } else if (spec == "WhisperSpec") {
    has_encoder_ = true;  // convolutional front end instead of source embeddings (load_stack)
  } else if (spec == "TransformerDecoderSpec") {
    has_encoder_ = false;
  } else {
    throw std::runtime_error("Model specification '" + spec + "' is not handled by this backend "
                             "(Transformer encoder-decoder, decoder-only and Whisper models are)");
  }
  load_weights();
}

Engine::~Engine() {
  (void)hipSetDevice(device_);
  (void)hipStreamSynchronize(stream_);
  for (void* p : device_blocks_) {
    try { device_free(p); } catch (...) {}
  }
  if (sin_pos_table_) { try { device_free(sin_pos_table_); } catch (...) {} }
  if (rope_sin_) { try { device_free(rope_sin_); } catch (...) {} }
  if (rope_cos_) { try { device_free(rope_cos_); } catch (...) {} }
}

const void* Engine::upload(const void* host, size_t bytes) {
  void* d = device_malloc(bytes);
  device_blocks_.push_back(d);
  CT2X_HIP_CHECK(hipMemcpy(d, host, bytes, hipMemcpyHostToDevice));
  weight_bytes_ += static_cast<int64_t>(bytes);
  return d;
}

const float* Engine::upload_floats(const std::vector<float>& v) {
  return static_cast<const float*>(upload(v.data(), v.size() * sizeof(float)));
}

namespace {

// Row ranges [first, first + count) of a [n, *] matrix owned by `rank` when every block of
// `blocks` rows (fused sub-matrices; one block = the whole matrix) is split evenly.
std::vector<std::pair<int64_t, int64_t>> column_shard_rows(int64_t n, const std::vector<int64_t>& blocks,
                                                           int rank, int world,
                                                           const std::string& scope) {
  std::vector<int64_t> sizes = blocks.empty() ? std::vector<int64_t>{n} : blocks;
  int64_t total = 0;
  for (int64_t b : sizes) total += b;
  if (total != n)
    throw std::runtime_error(scope + ": fused blocks do not add up to the weight's rows");
  std::vector<std::pair<int64_t, int64_t>> out;
  int64_t base = 0;
  for (int64_t b : sizes) {
    if (b % world != 0)
      throw std::invalid_argument(scope + ": output dimension " + std::to_string(b) +
                                  " is not divisible by the tensor-parallel world size " +
                                  std::to_string(world));
    out.emplace_back(base + rank * (b / world), b / world);
    base += b;
  }
  return out;
}

template <typename T>
std::vector<T> take_rows(const T* src, int64_t k, const std::vector<std::pair<int64_t, int64_t>>& rows) {
  std::vector<T> out;
  for (const auto& r : rows) out.insert(out.end(), src + r.first * k, src + (r.first + r.second) * k);
  return out;
}

template <typename T>
std::vector<T> take_cols(const T* src, int64_t n, int64_t k, int64_t k0, int64_t k1) {
  std::vector<T> out(static_cast<size_t>(n) * (k1 - k0));
  for (int64_t i = 0; i < n; ++i)
    std::copy(src + i * k + k0, src + i * k + k1, out.begin() + i * (k1 - k0));
  return out;
}

}  // namespace

DenseW Engine::load_dense(const std::string& scope, bool required, Shard shard,
                          const std::vector<int64_t>& blocks) {
  DenseW d;
  const Variable* w = model_->find(scope + "/weight");
  if (!w) {
    if (model_->has(scope + "/weight_packed"))
      throw std::runtime_error(scope + ": pre-packed CPU weights cannot be loaded on the GPU");
    if (required) throw std::out_of_range("variable " + scope + "/weight not found");
    return d;
  }
  // Conv1D weights [out, in, kernel] are the [out, in * kernel] matrix of the convolution run as a
  // GEMM over im2col rows (Conv1D::compute_with_gemm, src/ops/conv1d_cpu.cc:139-207; Model::load
  // reshapes them the same way to convert them, src/models/model.cc:208-227)
  const bool is_conv = w->shape.size() == 3 && scope.find("conv") != std::string::npos;
  if (w->shape.size() != 2 && !is_conv) throw std::runtime_error(scope + "/weight is not a matrix");
  if (const Variable* zero = model_->find(scope + "/weight_zero")) {
    // AWQ int4 (layers::Dense AWQ branch, src/layers/common.cc:402-438; formats: awq.hip). The
    // reference forces FLOAT16 for such models (src/models/model.cc:754-757): here the fp32
    // activation is rounded to float16 as it enters the GEMM, everything else stays fp32.
    const int method = static_cast<int>(model_->config_number("quantization_type", 0));
    const Variable* scale = model_->find(scope + "/weight_scale");
    if ((method != 1 && method != 2) || !scale || w->dtype != kI32 || zero->dtype != kI32 ||
        scale->dtype != kF16 || scale->shape.size() != 2 || zero->shape.size() != 2)
      throw std::invalid_argument(scope + ": malformed AWQ weight (needs quantization_type 1 / 2, "
                                  "int32 weight / zeros, float16 scales)");
    if (tp_) throw std::invalid_argument(scope + ": AWQ weights are not supported under tensor parallelism");
    d.present = true;
    d.awq = method;
    if (method == 1) {  // GEMM layout: weight [in_c, out_c / 8]
      d.k = w->shape[0];
      d.n = w->shape[1] * 8;
      d.aq_group = d.k / scale->shape[0];
    } else {            // GEMV layout: weight [out_c, in_c / 8]
      d.n = w->shape[0];
      d.k = w->shape[1] * 8;
      d.aq_sw = scale->shape[1];
      d.aq_zw = zero->shape[1];
      // the reference recovers the group size as in_c / sf_w (src/ops/awq/gemv_gpu.cu:447), which
      // only holds when in_c / G fills the padded width: config.json's quantization_group_size wins
      d.aq_group = static_cast<int64_t>(
          model_->config_number("quantization_group_size", static_cast<double>(d.k / d.aq_sw)));
    }
    d.aq_w = static_cast<const int32_t*>(upload(w->data.data(), w->data.size()));
    d.aq_s = upload(scale->data.data(), scale->data.size());
    d.aq_z = static_cast<const int32_t*>(upload(zero->data.data(), zero->data.size()));
    if (const Variable* b = model_->find(scope + "/bias")) d.bias = upload_floats(b->to_float());
    has_awq_ = true;
    return d;
  }
  d.present = true;
  d.n = w->shape[0];
  d.k = is_conv ? w->shape[1] * w->shape[2] : w->shape[1];
  const Variable* scale = model_->find(scope + "/weight_scale");
  const bool want_int8 = (compute_type_ == ComputeType::kInt8Float32);
  if (!tp_) shard = Shard::kNone;

  // The whole matrix is brought to its device form on the host first (int8 + per-row scales, or
  // fp32); the tensor-parallel slice is cut afterwards so that a row-parallel layer keeps the
  // scales of the FULL rows (Model::load shards after conversion too, src/models/model.cc:662-743).
  const int8_t* q_src = nullptr;
  std::vector<int8_t> q_own;
  std::vector<float> s_full, wf;
  if (want_int8 && w->dtype == kI8 && scale && scale->size() == d.n) {
    d.quantized = true;
    q_src = reinterpret_cast<const int8_t*>(w->data.data());
    s_full = scale->to_float();
  } else {
    // Bring the weight to float first (Dequantize: x / scale, src/ops/dequantize_cpu.cc).
    wf = w->to_float();
    if (w->dtype == kI8 || w->dtype == kI16) {
      if (!scale) {
        if (w->dtype == kI16) {
          for (auto& v : wf) v = v / 1000.f;  // ops::Quantize::global_int16_scale
        } else {
          throw std::runtime_error("variable " + scope + "/weight_scale not found");
        }
      } else {
        const std::vector<float> s = scale->to_float();
        if (scale->size() == 1) {
          for (auto& v : wf) v = v / s[0];
        } else {
          for (int64_t i = 0; i < d.n; ++i)
            for (int64_t j = 0; j < d.k; ++j) wf[i * d.k + j] = wf[i * d.k + j] / s[i];
        }
      }
    }
    if (want_int8) {
      host_quantize_rows(wf, d.n, d.k, round_before_cast_, q_own, s_full);
      d.quantized = true;
      q_src = q_own.data();
      wf.clear();
    }
  }
  // 16-bit compute types: the (dequantised / converted) weight is rounded to the type once here
  // (Model::ensure_dtype, src/models/model.cc:304-369); K must suit the 8-element MFMA fragments.
  const bool want_half = (compute_type_ == ComputeType::kFloat16 ||
                          compute_type_ == ComputeType::kBFloat16) && d.k % 8 == 0;
  std::vector<uint16_t> wh;
  if (want_half) {
    const bool bf = compute_type_ == ComputeType::kBFloat16;
    wh.resize(wf.size());
    for (size_t i = 0; i < wf.size(); ++i)
      wh[i] = bf ? float_to_bf16_bits(wf[i]) : float_to_half_bits(wf[i]);
    wf.clear();
  }
  std::vector<float> bias;
  if (const Variable* b = model_->find(scope + "/bias")) bias = b->to_float();

  if (shard == Shard::kColumn) {
    const auto rows = column_shard_rows(d.n, blocks, tp_->rank(), tp_->world(), scope);
    if (d.quantized) {
      q_own = take_rows(q_src, d.k, rows);
      q_src = q_own.data();
      s_full = take_rows(s_full.data(), 1, rows);
    } else if (want_half) {
      wh = take_rows(wh.data(), d.k, rows);
    } else {
      wf = take_rows(wf.data(), d.k, rows);
    }
    if (!bias.empty()) bias = take_rows(bias.data(), 1, rows);
    d.n = 0;
    for (const auto& r : rows) d.n += r.second;
  } else if (shard == Shard::kRow) {
    if (d.k % tp_->world() != 0)
      throw std::invalid_argument(scope + ": input dimension " + std::to_string(d.k) +
                                  " is not divisible by the tensor-parallel world size");
    const int64_t kl = d.k / tp_->world(), k0 = tp_->rank() * kl;
    if (d.quantized) {
      q_own = take_cols(q_src, d.n, d.k, k0, k0 + kl);
      q_src = q_own.data();
    } else if (want_half) {
      wh = take_cols(wh.data(), d.n, d.k, k0, k0 + kl);
    } else {
      wf = take_cols(wf.data(), d.n, d.k, k0, k0 + kl);
    }
    d.k = kl;
    d.row_parallel = true;  // scales and bias stay whole: applied once, after the rank sum
  }

  if (d.quantized) {
    d.wq = static_cast<const int8_t*>(upload(q_src, static_cast<size_t>(d.n) * d.k));
    d.wscale = upload_floats(s_full);
  } else if (want_half) {
    if (d.k % 8 != 0)  // a row-parallel slice may have lost the divisibility
      throw std::invalid_argument(scope + ": the 16-bit GEMM needs an input dimension divisible by 8");
    d.wh = upload(wh.data(), wh.size() * sizeof(uint16_t));
    d.wh_bf16 = compute_type_ == ComputeType::kBFloat16;
  } else {
    d.wf = upload_floats(wf);
  }
  if (!bias.empty()) d.bias = upload_floats(bias);
  return d;
}

NormW Engine::load_norm(const std::string& scope) {
  NormW n;
  const Variable* gamma = model_->find(scope + "/gamma");
  if (!gamma) return n;
  n.present = true;
  n.gamma = upload_floats(gamma->to_float());
  if (const Variable* beta = model_->find(scope + "/beta")) n.beta = upload_floats(beta->to_float());
  n.use_residual = model_->flag(scope + "/layer_norm_use_residual", false);
  // layers::LayerNorm ctor (src/layers/common.cc:444-454).
  if (model_->config_has_number("layer_norm_epsilon"))
    n.eps = static_cast<float>(model_->config_number("layer_norm_epsilon", 0));
  else
    n.eps = n.beta ? 1e-5f : 1e-6f;
  return n;
}

AttnW Engine::load_attention(const std::string& scope, bool self_attention, int64_t num_heads) {
  AttnW a;
  for (const char* unsupported : {"/relative_attention_bias", "/relative_position_keys",
                                  "/relative_asymmetric_position_keys",
                                  "/relative_position_values", "/q_norm/gamma", "/k_norm/gamma",
                                  "/v_norm/gamma", "/memory_kv/weight"})
    if (model_->has(scope + unsupported))
      throw std::runtime_error("attention variant with " + scope + unsupported +
                               " is not supported by this backend yet");
  a.ln = load_norm(scope + "/layer_norm");
  a.n_lin = self_attention ? 2 : 3;
  a.heads = num_heads;
  // Geometry from the unsharded variables.
  const Variable* w_out = model_->find(scope + "/linear_" + std::to_string(a.n_lin - 1) + "/weight");
  if (!w_out || w_out->shape.size() != 2)
    throw std::out_of_range("variable " + scope + "/linear_" + std::to_string(a.n_lin - 1) +
                            "/weight not found");
  const int64_t d_model = w_out->shape[0];
  a.dhead = static_cast<int64_t>(model_->attribute(scope + "/head_dim", double(d_model / num_heads)));
  {
    const bool mq = model_->flag(scope + "/multi_query", false);
    const int64_t hkv = mq ? 1
                           : static_cast<int64_t>(
                                 model_->attribute(scope + "/num_heads_kv", double(num_heads)));
    const int64_t qdim = num_heads * a.dhead, kvdim = hkv * a.dhead;
    // Tensor parallel: heads are split over the ranks - column-parallel input projections
    // (Q | K | V split block-wise), row-parallel output projection.
    if (self_attention) {
      a.lin[0] = load_dense(scope + "/linear_0", true, Shard::kColumn, {qdim, kvdim, kvdim});
      a.lin[1] = load_dense(scope + "/linear_1", true, Shard::kRow);
    } else {
      a.lin[0] = load_dense(scope + "/linear_0", true, Shard::kColumn, {qdim});
      a.lin[1] = load_dense(scope + "/linear_1", true, Shard::kColumn, {kvdim, kvdim});
      a.lin[2] = load_dense(scope + "/linear_2", true, Shard::kRow);
    }
  }
  a.queries_scale = static_cast<float>(
      model_->attribute(scope + "/queries_scale", 1.0 / std::sqrt(static_cast<float>(a.dhead))));
  if (!model_->has(scope + "/queries_scale"))
    a.queries_scale = 1.f / std::sqrt(static_cast<float>(a.dhead));
  const bool multi_query = model_->flag(scope + "/multi_query", false);
  a.heads_kv = multi_query ? 1
                           : static_cast<int64_t>(model_->attribute(scope + "/num_heads_kv",
                                                                    double(num_heads)));
  if (tp_) {
    // This rank's share of the heads. The column split of the projections only checks the
    // DIMENSIONS (load_dense); the split must also fall on head boundaries, and a KV head cannot
    // be divided (the reference's own TP path requires num_heads % world == 0 as well,
    // src/layers/attention_layer.cc / SAFE_DIVIDE).
    const int64_t world = tp_->world();
    if (a.heads % world != 0 || a.heads_kv % world != 0)
      throw std::runtime_error("tensor parallelism: " + scope + " has " + std::to_string(a.heads) +
                               " query heads and " + std::to_string(a.heads_kv) +
                               " key/value heads, which cannot be divided over " +
                               std::to_string(world) + " ranks on head boundaries");
    a.heads /= world;
    a.heads_kv /= world;
  }
  if (model_->attribute(scope + "/sliding_window", 0) != 0)
    throw std::runtime_error("sliding window attention is not supported by this backend yet");
  // make_rotary_embeddings (src/layers/attention_layer.cc:68-110).
  const int64_t rotary_dim = static_cast<int64_t>(model_->attribute(scope + "/rotary_dim", -1));
  if (rotary_dim >= 0) {
    a.rotary = true;
    a.rotary_dim = rotary_dim;
    a.rotary_interleave = model_->flag(scope + "/rotary_interleave", true);
    a.rotary_base = static_cast<float>(model_->attribute(scope + "/rotary_base", 10000.0));
    a.rotary_scaling_type = static_cast<int>(model_->attribute(scope + "/rotary_scaling_type", -1));
    a.rotary_scaling_factor =
        static_cast<float>(model_->attribute(scope + "/rotary_scaling_factor", 1.0));
    a.rotary_low_freq_factor =
        static_cast<float>(model_->attribute(scope + "/rotary_low_freq_factor", 1.0));
    a.rotary_high_freq_factor =
        static_cast<float>(model_->attribute(scope + "/rotary_high_freq_factor", 4.0));
    a.original_max_position_embeddings = static_cast<int64_t>(
        model_->attribute(scope + "/original_max_position_embeddings", 0));
    a.max_position_embeddings =
        static_cast<int64_t>(model_->attribute(scope + "/max_position_embeddings", 0));
    if (a.rotary_scaling_type == 1) {
      const Variable* lf = model_->find(scope + "/rotary_scaling_long_factor");
      const Variable* sf = model_->find(scope + "/rotary_scaling_short_factor");
      if (!lf || !sf)
        throw std::runtime_error(scope + ": Su rotary scaling without its long / short factors");
      a.rotary_long_factor = lf->to_float();
      a.rotary_short_factor = sf->to_float();
    }
  }
  if (a.dhead % 4 != 0 || a.dhead > 512)
    throw std::runtime_error("head dimension must be a multiple of 4 and <= 512");
  return a;
}

FfnW Engine::load_ffn(const std::string& scope, int activation) {
  FfnW f;
  f.ln = load_norm(scope + "/layer_norm");
  f.l0 = load_dense(scope + "/linear_0", true, Shard::kColumn);
  f.l0_noact = load_dense(scope + "/linear_0_noact", false, Shard::kColumn);
  f.l1 = load_dense(scope + "/linear_1", true, Shard::kRow);
  f.activation = activation;
  return f;
}

void Engine::load_stack(const std::string& scope, bool decoder, StackW& out) {
  const ModelFile& m = *model_;
  // Spec revisions < 5 keep these attributes at the model root
  // (TransformerModel::initialize, src/models/transformer.cc:62-81).
  auto attr = [&](const std::string& name, double def) {
    if (m.has(scope + "/" + name)) return m.attribute(scope + "/" + name, def);
    return m.attribute(name, def);
  };
  int64_t default_heads = 8;
  if (m.spec == "TransformerBig") default_heads = 16;
  out.num_heads = static_cast<int64_t>(attr("num_heads", double(default_heads)));
  out.pre_norm = attr("pre_norm", 1) != 0;
  int activation = static_cast<int>(attr("activation", 0));
  if (!decoder && m.has(scope + "/conv1/weight")) activation = 3;  // GELU (src/layers/whisper.cc:7)

  for (const char* unsupported : {"/project_in/weight", "/project_out/weight", "/scale_outputs"})
    if (m.has(scope + unsupported))
      throw std::runtime_error(scope + unsupported + " is not supported by this backend yet");
  if (m.flag(scope + "/alibi", false))
    throw std::runtime_error("ALiBi models are not supported by this backend yet");
  if (m.attribute(scope + "/sliding_window", 0) != 0)
    throw std::runtime_error("sliding window models are not supported by this backend yet");
  if (m.attribute(scope + "/final_logit_softcapping", 0) != 0)
    throw std::runtime_error("final logit soft-capping is not supported by this backend yet");
  if (m.flag(scope + "/start_from_zero_embedding", false))
    throw std::runtime_error("start_from_zero_embedding is not supported by this backend yet");

  const bool conv_front_end = !decoder && m.has(scope + "/conv1/weight");
  if (conv_front_end) {
    // layers::WhisperEncoder (src/layers/whisper.cc:9-24): conv1 (stride 1) and conv2 (stride 2),
    // kernel 3, padding 1, GELU; GELU encoder layers, pre-norm; learned positions
    for (const char* cs : {"/conv1/weight", "/conv2/weight"}) {
      const Variable& cw = m.get(scope + cs);
      if (cw.shape.size() != 3 || cw.shape[2] != 3)
        throw std::runtime_error(scope + cs + ": expected a [out, in, 3] convolution weight");
    }
    out.has_conv = true;
    out.conv1 = load_dense(scope + "/conv1", true);
    out.conv2 = load_dense(scope + "/conv2", true);
    out.n_mels = out.conv1.k / 3;
    out.d_model = out.conv2.n;
    if (out.conv2.k != 3 * out.conv1.n)
      throw std::runtime_error(scope + "/conv2: input channels differ from conv1's outputs");
  }
  // Embeddings (layers::Embeddings / ParallelEmbeddings, src/layers/common.cc:49-148).
  std::string emb_scope = scope + "/embeddings";
  if (conv_front_end) {
    // (no embeddings)
  } else {
  if (!m.has(emb_scope + "/weight")) {
    emb_scope = scope + "/embeddings_0";
    if (m.has(scope + "/embeddings_1/weight"))
      throw std::runtime_error("multiple input features are not supported by this backend yet");
  }
  const Variable& ew = m.get(emb_scope + "/weight");
  out.emb.vocab = ew.shape.at(0);
  out.emb.depth = ew.shape.at(1);
  out.d_model = out.emb.depth;
  {
    const Variable* es = m.find(emb_scope + "/weight_scale");
    const bool want_int8 = (compute_type_ == ComputeType::kInt8Float32);
    if (want_int8 && ew.dtype == kI8 && es && es->size() == out.emb.vocab) {
      out.emb.is_int8 = true;
      out.emb.table = upload(ew.data.data(), ew.data.size());
      out.emb.scale = upload_floats(es->to_float());
    } else {
      std::vector<float> wf = ew.to_float();
      if (ew.dtype == kI8 || ew.dtype == kI16) {
        std::vector<float> s = es ? es->to_float() : std::vector<float>{1000.f};
        for (int64_t i = 0; i < out.emb.vocab; ++i)
          for (int64_t j = 0; j < out.emb.depth; ++j)
            wf[i * out.emb.depth + j] = wf[i * out.emb.depth + j] / (s.size() == 1 ? s[0] : s[i]);
      }
      if (want_int8) {
        std::vector<int8_t> q;
        std::vector<float> s;
        host_quantize_rows(wf, out.emb.vocab, out.emb.depth, round_before_cast_, q, s);
        out.emb.is_int8 = true;
        out.emb.table = upload(q.data(), q.size());
        out.emb.scale = upload_floats(s);
      } else {
        out.emb.table = upload_floats(wf);
      }
    }
  }
  }  // (!conv_front_end)

  // build_embeddings_scale (src/layers/transformer.cc:378-399).
  if (!conv_front_end) {
    const Variable* scale = m.find(scope + "/scale_embeddings");
    if (!scale) scale = m.find(scope + "/embeddings/multiply_by_sqrt_depth");
    if (!scale || (scale->dtype == kI8 && scale->as_number() != 0)) {
      out.scale_embeddings = true;
      out.embeddings_scale = std::sqrt(static_cast<float>(out.emb.depth));
    } else if (scale->dtype != kI8 && scale->as_number() != 1.0) {
      out.scale_embeddings = true;
      out.embeddings_scale = static_cast<float>(scale->as_number());
    }
  }

  out.layernorm_embedding = load_norm(scope + "/layernorm_embedding");
  out.output_norm = load_norm(scope + "/layer_norm");

  for (int64_t l = 0;; ++l) {
    const std::string ls = scope + "/layer_" + std::to_string(l);
    if (!m.has(ls + "/self_attention/linear_0/weight")) break;
    for (const char* unsupported : {"/external_pre_encoder_attention_layer_norm/gamma",
                                    "/external_post_encoder_attention_layer_norm/gamma"})
      if (m.has(ls + unsupported))
        throw std::runtime_error("layer variant with " + ls + unsupported +
                                 " is not supported by this backend yet");
    LayerW layer;
    layer.self = load_attention(ls + "/self_attention", true, out.num_heads);
    if (decoder && m.has(ls + "/attention/linear_0/weight")) {
      layer.has_cross = true;
      layer.cross = load_attention(ls + "/attention", false, out.num_heads);
    }
    layer.ffn = load_ffn(ls + "/ffn", activation);
    // Layer variants (TransformerDecoderLayer, src/layers/transformer.cc:170-323). The encoder
    // layer has none of these norms (TransformerEncoderLayer, :54-67).
    const NormW shared_ln = load_norm(ls + "/shared_layer_norm");
    const NormW input_ln = load_norm(ls + "/input_layer_norm");
    const NormW post_attn_ln = load_norm(ls + "/post_attention_layer_norm");
    const NormW pre_ff_ln = load_norm(ls + "/pre_feedforward_layer_norm");
    const NormW post_ff_ln = load_norm(ls + "/post_feedforward_layer_norm");
    const bool any_variant = shared_ln.present || input_ln.present || post_attn_ln.present ||
                             pre_ff_ln.present || post_ff_ln.present;
    if (any_variant) {
      if (!decoder)
        throw std::runtime_error(ls + ": layer-variant norms on an encoder layer");
      if (layer.has_cross || !out.pre_norm || layer.self.ln.present || layer.ffn.ln.present || tp_)
        throw std::runtime_error("layer variant of " + ls + " with encoder-decoder attention, "
                                 "post-norm, sub-layer norms of its own or tensor parallelism is "
                                 "not supported by this backend yet");
      if (post_ff_ln.present && pre_ff_ln.present) {
        if (!input_ln.present || !post_attn_ln.present)
          throw std::runtime_error(ls + ": pre / post feed-forward norms without input / "
                                   "post-attention norms");
        layer.variant = 2;
        layer.self.ln = input_ln;
        layer.post_attn_ln = post_attn_ln;
        layer.ffn.ln = pre_ff_ln;
        layer.post_ff_ln = post_ff_ln;
        layer.layer_scalar = static_cast<float>(m.attribute(ls + "/layer_scalar", 1.0));
      } else if (shared_ln.present || input_ln.present) {
        layer.variant = 1;
        layer.self.ln = shared_ln.present ? shared_ln : input_ln;
        // the feed-forward block reads post_attention_layer_norm(x) when that norm exists, else the
        // same normalised input as the attention (transformer.cc:300-306)
        layer.ffn.ln = post_attn_ln.present ? post_attn_ln : layer.self.ln;
      } else {
        throw std::runtime_error(ls + ": unsupported combination of layer-variant norms");
      }
      layer.self.no_residual = true;
      layer.ffn.no_residual = true;
    } else if (m.has(ls + "/layer_scalar") && m.get(ls + "/layer_scalar").as_number() != 1.0) {
      // (only the pre + post norm arm applies it, transformer.cc:281-283)
    }
    out.layers.emplace_back(std::move(layer));
  }
  if (out.layers.empty()) throw std::runtime_error("no layers found under " + scope);

  // Position encodings (build_position_encoder, src/layers/transformer.cc:366-376): learned table
  // if present, else sinusoidal -- unless the attention layers carry their own (rotary).
  if (!out.layers.front().self.rotary) {
    if (const Variable* pe = m.find(scope + "/position_encodings/encodings")) {
      out.position_table = upload_floats(pe->to_float());
      out.num_positions = pe->shape.at(0);
    } else {
      out.sinusoidal_positions = true;
    }
  }

  if (decoder) out.projection = load_dense(scope + "/projection", true);
}

void Engine::load_weights() {
  if (has_encoder_) load_stack("encoder", false, enc_);
  load_stack("decoder", true, dec_);
  if (has_encoder_ != dec_.layers.front().has_cross)
    throw std::runtime_error("inconsistent model: encoder presence and cross-attention differ");
  // TransformerDecoder ctor (src/layers/transformer.cc:516-528): which encoder-decoder attention
  // return_attention reports - the first alignment_heads heads (0 = all) of alignment_layer
  alignment_layer_ = static_cast<int64_t>(model_->attribute("decoder/alignment_layer", -1));
  alignment_heads_ = static_cast<int64_t>(model_->attribute("decoder/alignment_heads", 1));
  if (alignment_layer_ < 0) alignment_layer_ += static_cast<int64_t>(dec_.layers.size());
  if (alignment_heads_ == 0) alignment_heads_ = dec_.num_heads;
  if (alignment_layer_ < 0 || alignment_layer_ >= static_cast<int64_t>(dec_.layers.size()) ||
      alignment_heads_ < 0 || alignment_heads_ > dec_.num_heads)
    throw std::runtime_error("decoder/alignment_layer or alignment_heads is out of range");
  pack_stream_weights();
}

// Decode-step weights of models that cannot stay cache resident get a second, fragment-major copy
// (gemm_s8_stream.hip: every request of the persistent streaming GEMM is one contiguous 1 KB block;
// Llama-3-8B shape 3.49 -> 3.28 ms per step). The row-major weight stays for prefill and for
// batches of more than 16 rows: HBM capacity (288 GB) is not what bounds this workload.
void Engine::pack_stream(DenseW& w, bool dual) {
  if (!w.present || !w.quantized || w.row_parallel || w.wq == nullptr) return;
  if (!gemm_s8_stream_pack_applicable(w.n, w.k, dual)) return;
  int8_t* packed = static_cast<int8_t*>(device_malloc(static_cast<size_t>(w.n) * w.k));
  device_blocks_.push_back(packed);
  launch_pack_stream_weight(w.wq, w.n, w.k, dual, packed, stream_);
  w.wq_stream = packed;
  w.wq_stream_dual = dual;
}

// HBM sits idle during the row-wide launches of a layer (norm + Quantize, element-wise Quantize:
// 48 of the Llama shape's 102 us per layer). The weight of the Dense that FOLLOWS such a launch is
// requested by spare workgroups appended to that launch (kernels.h PrefetchArgs), so the GEMM finds
// (a leading fraction of every wave's stream of) its weight in the memory-side cache.
// MEASURED NEGATIVE on the Llama-3-8B shape, both ways, and therefore off by default
// (CT2X_WEIGHT_PREFETCH=1 turns it on): as a parallel branch of the step's hipGraph (side stream,
// event fork / join per Dense) every cross-queue dependency cost ~18 us - 5.6 instead of 3.3 ms per
// step (`profiles/r02b_l8_side_stream_prefetch_negative.txt`); as spare workgroups of the row-wide
// launches 3.45 - 3.48 against 3.27 - 3.28 ms, 3.39 with half the bytes
// (`profiles/r02b_l8_inkernel_prefetch_negative.txt`): the carrying launches grow by more than the
// GEMMs shrink - a GEMM whose weight is MALL resident is only 20 % faster
// (`profiles/r02_skinny_hbm_vs_mall.txt`), and the requests compete with the norm's own chain.
void Engine::prefetch_weight(DecodeState& st, const DenseW& w, int site) {
  if (!weight_prefetch_ || !st.pf_active || profiling_ || taps_ || w.wq_stream == nullptr ||
      st.rows > 16 || prefetch_bytes_[site] <= 0)
    return;
  for (PrefetchRange& r : st.pf_pending.r) {
    if (r.p != nullptr) continue;
    if (weight_prefetch_range(w.wq_stream, w.n, w.k, w.wq_stream_dual,
                              prefetch_bytes_[site] / (w.wq_stream_dual ? 2 : 1), &r))
      st.pf_pending.wgs = prefetch_wgs_;
    return;
  }
}

void Engine::pack_stream_weights() {
  if (tp_) return;
  for (LayerW& layer : dec_.layers) {
    for (int i = 0; i < layer.self.n_lin; ++i) pack_stream(layer.self.lin[i], false);
    if (layer.has_cross) {
      pack_stream(layer.cross.lin[0], false);
      pack_stream(layer.cross.lin[2], false);
    }
    const bool gated = layer.ffn.l0_noact.present && layer.ffn.l0.quantized &&
                       layer.ffn.l0_noact.quantized && layer.ffn.l0_noact.n == layer.ffn.l0.n &&
                       layer.ffn.l0_noact.k == layer.ffn.l0.k;
    pack_stream(layer.ffn.l0, gated);
    if (gated) pack_stream(layer.ffn.l0_noact, true);
    pack_stream(layer.ffn.l1, false);
  }
  pack_stream(dec_.projection, false);
  {
    // CT2X_WEIGHT_PREFETCH=1 switches the weight prefetch on (measured NEGATIVE, off by default: see
    // Engine::prefetch_weight); CT2X_PREFETCH_MB="qkv,out,ffn_in,ffn_out" sets the megabytes requested
    // next to the launch in front of each Dense, CT2X_PREFETCH_WGS the spare workgroups
    const char* e = tuning_env("CT2X_WEIGHT_PREFETCH");
    bool any = false;
    for (const LayerW& layer : dec_.layers) any = any || layer.ffn.l1.wq_stream != nullptr;
    weight_prefetch_ = any && e != nullptr && e[0] == '1';
    double mb[4] = {25, 17, 36, 18};
    if (const char* m = tuning_env("CT2X_PREFETCH_MB"))
      (void)std::sscanf(m, "%lf,%lf,%lf,%lf", &mb[0], &mb[1], &mb[2], &mb[3]);
    for (int i = 0; i < 4; ++i) prefetch_bytes_[i] = static_cast<int64_t>(mb[i] * 1048576.0);
    if (const char* g = tuning_env("CT2X_PREFETCH_WGS")) prefetch_wgs_ = std::max(1, std::atoi(g));
  }
  CT2X_HIP_CHECK(hipStreamSynchronize(stream_));
  check_last_launch("pack_stream_weights");
}

// generate_sinusoidal_position_encoding (src/layers/common.cc:204-229).
std::vector<float> make_sinusoidal_table(int64_t positions, int64_t depth) {
  const int64_t half = depth / 2;
  // depth 2 would divide by zero here exactly as the reference does (inf increment): guarded
  const float log_timescale_increment =
      std::log(10000.f) / static_cast<float>(half > 1 ? half - 1 : 1);
  std::vector<float> timescales(half);
  for (int64_t i = 0; i < half; ++i)
    timescales[i] = std::exp(-log_timescale_increment * static_cast<float>(i));
  // ops::Sin / ops::Cos on the CPU = the host math backend (MKL VML when installed, host_math.h)
  std::vector<float> scaled(positions * half), sv(positions * half), cv(positions * half);
  for (int64_t p = 0; p < positions; ++p)
    for (int64_t j = 0; j < half; ++j)
      scaled[p * half + j] = static_cast<float>(p + 1) * timescales[j];
  host_vsin(positions * half, scaled.data(), sv.data());
  host_vcos(positions * half, scaled.data(), cv.data());
  std::vector<float> table(positions * depth);
  for (int64_t p = 0; p < positions; ++p) {
    for (int64_t j = 0; j < half; ++j) {
      table[p * depth + j] = sv[p * half + j];
      table[p * depth + half + j] = cv[p * half + j];
    }
  }
  return table;
}

const float* Engine::sinusoidal_table(int64_t positions, int64_t depth) {
  std::lock_guard<std::mutex> lock(table_mutex_);
  if (sin_pos_table_ && positions <= sin_pos_positions_ && depth == sin_pos_depth_)
    return sin_pos_table_;
  positions = std::max<int64_t>(positions, 500);
  const std::vector<float> table = make_sinusoidal_table(positions, depth);
  // A superseded table is kept until the engine dies: graphs of live sessions may point at it.
  if (sin_pos_table_) device_blocks_.push_back(sin_pos_table_);
  sin_pos_table_ = static_cast<float*>(device_malloc(table.size() * sizeof(float)));
  CT2X_HIP_CHECK(hipMemcpy(sin_pos_table_, table.data(), table.size() * sizeof(float),
                           hipMemcpyHostToDevice));
  sin_pos_positions_ = positions;
  sin_pos_depth_ = depth;
  return sin_pos_table_;
}

// RotaryEmbeddings::initialize (src/layers/attention_layer.cc:252-343), default / linear /
// Llama-3 scaling. Tables are [positions, dim].
// RotaryEmbeddings::initialize (src/layers/attention_layer.cc:254-330) on the host: inv_freq with
// None (-1) / Linear (0) / Llama3 (2) scaling, the position x frequency products, then sin / cos
// through the host math backend (host_math.h: MKL VML like the reference's ops::Sin / ops::Cos
// when an MKL runtime is installed). Tables are [positions, dim], halves or pairs duplicated.
void make_rotary_tables(int64_t dim, bool interleave, int scaling_type, float scaling_factor,
                        float base, float low_freq_factor, float high_freq_factor,
                        int64_t original_max_position_embeddings, int64_t positions,
                        std::vector<float>& s, std::vector<float>& c,
                        const std::vector<float>* su_long, const std::vector<float>* su_short,
                        int64_t max_position_embeddings) {
  const int64_t half = dim / 2;
  std::vector<float> inv_freq(half);
  for (int64_t i = 0; i < half; ++i)
    inv_freq[i] = 1.f / std::pow(base, static_cast<float>(i * 2) / static_cast<float>(dim));
  if (scaling_type == 1) {  // Su (attention_layer.cc:257-267): long factors once the table is
                            // longer than the original context, else the short ones
    const std::vector<float>* f = positions > original_max_position_embeddings ? su_long : su_short;
    if (!f || static_cast<int64_t>(f->size()) < half)
      throw std::invalid_argument("Su rotary scaling needs rotary_scaling_long_factor / "
                                  "rotary_scaling_short_factor with rotary_dim / 2 entries");
    for (int64_t i = 0; i < half; ++i)
      inv_freq[i] = 1.f / ((*f)[i] * std::pow(base, static_cast<float>(i * 2) / static_cast<float>(dim)));
  }
  if (scaling_type == 2) {  // Llama3
    std::vector<float> new_freqs = inv_freq;
    const float factor = scaling_factor;
    const float old_context_len = static_cast<float>(original_max_position_embeddings);
    const float low_freq_wavelen = old_context_len / low_freq_factor;
    const float high_freq_wavelen = old_context_len / high_freq_factor;
    for (int64_t i = 0; i < half; ++i) {
      // `float wavelen = 2.0f * M_PI / inv_freq` (attention_layer.cc:283): M_PI is a double, so the
      // quotient is formed in double precision and rounded once
      const float wavelen = static_cast<float>(2.0 * M_PI / static_cast<double>(inv_freq[i]));
      if (wavelen < high_freq_wavelen) {
      } else if (wavelen > low_freq_wavelen) {
        new_freqs[i] /= factor;
      } else {
        const float smooth = (old_context_len / wavelen - low_freq_factor) /
                             (high_freq_factor - low_freq_factor);
        new_freqs[i] = ((1 - smooth) * inv_freq[i] / factor + smooth * inv_freq[i]);
      }
    }
    inv_freq = new_freqs;
  }
  s.assign(positions * dim, 0.f);
  c.assign(positions * dim, 0.f);
  std::vector<float> freqs(positions * half), fs(positions * half), fc(positions * half);
  for (int64_t p = 0; p < positions; ++p) {
    const float t = (scaling_type == 0) ? static_cast<float>(p) / scaling_factor
                                        : static_cast<float>(p);
    for (int64_t i = 0; i < half; ++i) freqs[p * half + i] = t * inv_freq[i];
  }
  host_vsin(positions * half, freqs.data(), fs.data());
  host_vcos(positions * half, freqs.data(), fc.data());
  for (int64_t p = 0; p < positions; ++p) {
    for (int64_t i = 0; i < half; ++i) {
      const float sf = fs[p * half + i], cf = fc[p * half + i];
      if (interleave) {
        s[p * dim + 2 * i] = s[p * dim + 2 * i + 1] = sf;
        c[p * dim + 2 * i] = c[p * dim + 2 * i + 1] = cf;
      } else {
        s[p * dim + i] = s[p * dim + half + i] = sf;
        c[p * dim + i] = c[p * dim + half + i] = cf;
      }
    }
  }
  if (original_max_position_embeddings != 0 && max_position_embeddings != 0 && scaling_type != 2) {
    // `float scale = _max_position_embeddings / _original_max_position_embeddings` divides two
    // integers (attention_layer.cc:333)
    const float scale = static_cast<float>(max_position_embeddings / original_max_position_embeddings);
    if (scale > 1) {
      const float factor = static_cast<float>(std::sqrt(
          1 + std::log(scale) / std::log(static_cast<double>(original_max_position_embeddings))));
      for (auto& v : s) v = v * factor;
      for (auto& v : c) v = v * factor;
    }
  }
}

void Engine::rotary_tables(const AttnW& a, int64_t positions, const float** sin_out,
                           const float** cos_out) {
  std::lock_guard<std::mutex> lock(table_mutex_);
  const int64_t dim = a.rotary_dim == 0 ? a.dhead : a.rotary_dim;
  // the cached tables belong to ONE rotary configuration: another layer asking with a different
  // one (dimension, base, scaling, interleave) gets its own tables
  const std::string key = std::to_string(dim) + "/" + std::to_string(a.rotary_base) + "/" +
                          std::to_string(a.rotary_scaling_type) + "/" +
                          std::to_string(a.rotary_scaling_factor) + "/" +
                          std::to_string(a.rotary_low_freq_factor) + "/" +
                          std::to_string(a.rotary_high_freq_factor) + "/" +
                          std::to_string(a.original_max_position_embeddings) + "/" +
                          std::to_string(a.max_position_embeddings) + "/" +
                          (a.rotary_interleave ? "i" : "h");
  if (rope_sin_ && key != rope_key_) rope_positions_ = 0;  // rebuild below (old tables stay alive)
  rope_key_ = key;
  if (rope_sin_ && positions <= rope_positions_) {
    *sin_out = rope_sin_;
    *cos_out = rope_cos_;
    return;
  }
  // RotaryEmbeddings::apply (attention_layer.cc:220-224) grows its tables by at least
  // num_initial_positions = 2048 at a time
  positions = std::max<int64_t>(positions, rope_positions_ + 2048);
  std::vector<float> s, c;
  make_rotary_tables(dim, a.rotary_interleave, a.rotary_scaling_type, a.rotary_scaling_factor,
                     a.rotary_base, a.rotary_low_freq_factor, a.rotary_high_freq_factor,
                     a.original_max_position_embeddings, positions, s, c, &a.rotary_long_factor,
                     &a.rotary_short_factor, a.max_position_embeddings);
  if (rope_sin_) {  // kept alive for sessions that captured the old pointers
    device_blocks_.push_back(rope_sin_);
    device_blocks_.push_back(rope_cos_);
  }
  rope_sin_ = static_cast<float*>(device_malloc(s.size() * sizeof(float)));
  rope_cos_ = static_cast<float*>(device_malloc(c.size() * sizeof(float)));
  CT2X_HIP_CHECK(hipMemcpy(rope_sin_, s.data(), s.size() * sizeof(float), hipMemcpyHostToDevice));
  CT2X_HIP_CHECK(hipMemcpy(rope_cos_, c.data(), c.size() * sizeof(float), hipMemcpyHostToDevice));
  rope_positions_ = positions;
  *sin_out = rope_sin_;
  *cos_out = rope_cos_;
}

// ---------------------------------------------------------------------------------------------
// Building blocks
// ---------------------------------------------------------------------------------------------

void Engine::begin_kernel(const char* kind, double bytes, double ops, int64_t m, int64_t n,
                          int64_t k) {
  ++launch_counter_;
  ++tls_launches;  // what step() differences: launches of THIS thread (sessions run concurrently)
  if (!profiling_) return;
  PendingKernel pk;
  pk.label = kind;
  if (m || n || k)
    pk.label += "[m=" + std::to_string(m) + ",n=" + std::to_string(n) + ",k=" + std::to_string(k) + "]";
  pk.bytes = bytes;
  pk.ops = ops;
  auto get_event = [this]() {
    if (!event_pool_.empty()) {
      hipEvent_t e = event_pool_.back();
      event_pool_.pop_back();
      return e;
    }
    hipEvent_t e;
    CT2X_HIP_CHECK(hipEventCreate(&e));
    return e;
  };
  pk.start = get_event();
  pk.stop = get_event();
  CT2X_HIP_CHECK(hipEventRecord(pk.start, cur_stream()));
  pending_.push_back(pk);
}

void Engine::end_kernel() {
  if (!profiling_) return;
  CT2X_HIP_CHECK(hipEventRecord(pending_.back().stop, cur_stream()));
}

void Engine::profile_begin() {
  CT2X_HIP_CHECK(hipStreamSynchronize(cur_stream()));
  pending_.clear();
  profiling_ = true;
  // Calibration of the measurement itself: the same event bracket around an EMPTY kernel
  // (one thread incrementing a scratch word), queued back to back like the real launches. The
  // label "noop[calibration]" tells readers of the profile what a bracket costs on this stack
  // when the kernel inside does nothing (event records + one dependent launch).
  if (profile_scratch_ == nullptr) {
    profile_scratch_ = static_cast<int32_t*>(device_malloc(sizeof(int32_t)));
    CT2X_HIP_CHECK(hipMemsetAsync(profile_scratch_, 0, sizeof(int32_t), cur_stream()));
  }
  const int64_t counted = launch_counter_;
  for (int i = 0; i < 64; ++i) {
    begin_kernel("noop[calibration]");
    launch_advance_step(profile_scratch_, cur_stream());
    end_kernel();
  }
  launch_counter_ = counted;  // not part of a decoder step
}

std::map<std::string, Engine::KernelStat> Engine::profile_end() {
  std::map<std::string, KernelStat> stats;
  CT2X_HIP_CHECK(hipStreamSynchronize(cur_stream()));
  for (auto& pk : pending_) {
    float ms = 0;
    CT2X_HIP_CHECK(hipEventElapsedTime(&ms, pk.start, pk.stop));
    KernelStat& st = stats[pk.label];
    st.launches += 1;
    st.total_ms += ms;
    st.bytes += pk.bytes;
    st.ops += pk.ops;
    event_pool_.push_back(pk.start);
    event_pool_.push_back(pk.stop);
  }
  pending_.clear();
  profiling_ = false;
  return stats;
}

void Engine::tap(const char* producer, const void* dev, int dtype, int64_t rows, int64_t cols,
                 bool indexed) {
  if (!taps_ || dev == nullptr) return;
  Tap t;
  (void)indexed;
  t.name = tap_scope_ + "." + std::to_string(tap_index_++) + "." + producer;
  t.dtype = dtype;
  t.rows = rows;
  t.cols = cols;
  t.data.resize(static_cast<size_t>(rows * cols) * (dtype == 1 ? 1 : 4));
  CT2X_HIP_CHECK(hipStreamSynchronize(cur_stream()));
  CT2X_HIP_CHECK(hipMemcpy(t.data.data(), dev, t.data.size(), hipMemcpyDeviceToHost));
  taps_->push_back(std::move(t));
}

void Engine::debug_set_position_table(const float* host, int64_t positions, int64_t depth) {
  std::lock_guard<std::mutex> lock(table_mutex_);
  if (sin_pos_table_) device_blocks_.push_back(sin_pos_table_);
  sin_pos_table_ = static_cast<float*>(device_malloc(static_cast<size_t>(positions * depth) * sizeof(float)));
  CT2X_HIP_CHECK(hipMemcpy(sin_pos_table_, host, static_cast<size_t>(positions * depth) * sizeof(float),
                           hipMemcpyHostToDevice));
  sin_pos_positions_ = positions;
  sin_pos_depth_ = depth;
}

void Engine::quant(const float* x, int64_t rows, int64_t depth, int8_t* q, float* s) {
  begin_kernel("quantize_s8", double(rows) * depth * 5 + rows * 4, 0, rows, 0, depth);
  launch_quantize_s8(x, rows, depth, quant_mode_, q, s, cur_stream());
  end_kernel();
  if (taps_) {
    tap("f", x, 0, rows, depth);
    tap("q", q, 1, rows, depth);
    tap("qs", s, 0, rows, 1);
  }
}

void Engine::norm(const NormW& w, const float* x, int64_t rows, int64_t depth, float* y) {
  begin_kernel("norm", double(rows) * depth * 8 + depth * 8, 0, rows, 0, depth);
  launch_norm(x, w.gamma, w.beta, rows, depth, w.eps, w.use_residual, quant_mode_, y,
              nullptr, nullptr, cur_stream());
  end_kernel();
}

void Engine::norm_quant(const NormW& w, const float* x, int64_t rows, int64_t depth, float* y,
                        int8_t* q, float* s) {
  begin_kernel("norm_quantize", double(rows) * depth * (y ? 9 : 5) + depth * 8 + rows * 4, 0, rows,
               0, depth);
  float* tap_y = nullptr;
  if (taps_ && y == nullptr) {  // tap mode: also materialise the fp32 row the Quantize consumes
    tap_y = static_cast<float*>(device_malloc(static_cast<size_t>(rows * depth) * sizeof(float)));
    y = tap_y;
  }
  DecodeState* pst = pf_state_;
  const PrefetchArgs* pf = (pst != nullptr && pst->pf_pending.wgs > 0) ? &pst->pf_pending : nullptr;
  launch_norm(x, w.gamma, w.beta, rows, depth, w.eps, w.use_residual, quant_mode_, y, q, s,
              cur_stream(), pf);
  if (pf != nullptr) pst->pf_pending = PrefetchArgs();
  end_kernel();
  if (taps_) {
    tap("f", y, 0, rows, depth);
    tap("q", q, 1, rows, depth);
    tap("qs", s, 0, rows, 1);
    if (tap_y) {
      CT2X_HIP_CHECK(hipStreamSynchronize(cur_stream()));
      device_free(tap_y);
    }
  }
}

// layers::Dense::operator() (src/layers/common.cc:339-442).
bool Engine::dense_norm(const NormW& ln, const DenseW& w, const float* x, int64_t m, int act,
                        const float* residual, float* y, float* row_amax_out) {
  if (!gemm_s8_norm_fusion_enabled() || taps_ || !ln.present || !w.quantized || w.row_parallel ||
      !gemm_s8_norm_applicable(m, w.n, w.k))
    return false;
  GemmS8Epilogue ep;
  ep.mode = 1;
  ep.w_scale = w.wscale;
  ep.bias = w.bias;
  ep.residual = residual;
  ep.y = y;
  ep.act = act;
  ep.ldy = w.n;
  ep.norm_x = x;
  ep.norm_gamma = ln.gamma;
  ep.norm_beta = ln.beta;
  ep.norm_eps = ln.eps;
  ep.norm_rms_residual = ln.use_residual;
  ep.quant_mode = quant_mode_;
  ep.row_amax_out = row_amax_out;
  const double M = double(m), N = double(w.n), K = double(w.k);
  begin_kernel("norm_gemm_s8_dequant", 4 * M * K + N * K + 4 * N + 8 * K + 4 * M * N +
                                           (residual ? 4 * M * N : 0) + (w.bias ? 4 * N : 0),
               2 * M * N * K, m, w.n, w.k);
  launch_gemm_s8(nullptr, w.wq, m, w.n, w.k, ep, cur_stream());
  end_kernel();
  return true;
}

// An int8 Dense whose ops::Quantize is folded into the A-fragment fetch (gemm_s8.hip, QOL): x is the
// fp32 activation, row_amax its row absmax left by the producing launch.
void Engine::dense_qol(const DenseW& w, const float* x, const float* row_amax, int64_t m, int act,
                       const float* residual, float* y) {
  GemmS8Epilogue ep;
  ep.mode = 1;
  ep.w_scale = w.wscale;
  ep.bias = w.bias;
  ep.residual = residual;
  ep.y = y;
  ep.act = act;
  ep.ldy = w.n;
  ep.a_f32 = x;
  ep.a_amax = row_amax;
  ep.quant_mode = quant_mode_;
  const double M = double(m), N = double(w.n), K = double(w.k);
  begin_kernel("qgemm_s8_dequant", 4 * M * K + N * K + 4 * N + 4 * M + 4 * M * N +
                                       (residual ? 4 * M * N : 0) + (w.bias ? 4 * N : 0),
               2 * M * N * K, m, w.n, w.k);
  launch_gemm_s8(nullptr, w.wq, m, w.n, w.k, ep, cur_stream());
  end_kernel();
}

// Output projection of an attention block + residual + the NEXT sub-layer's pre-norm + Quantize
// as one row-owning launch (rowblock.hip). Returns false when the shape / layer does not qualify
// (the caller then runs dense + norm_quant).
bool Engine::dense_out_next_norm(const DenseW& w, const QuantIn& cq, int64_t rows,
                                 const float* residual, float* y, const NormW* next_ln,
                                 const DenseW* next_dense, DecodeState& st) {
  if (tp_ || taps_ || !dec_.pre_norm || next_ln == nullptr || !next_ln->present ||
      next_dense == nullptr || !next_dense->quantized || !w.quantized || w.row_parallel ||
      next_dense->k != w.n || w.k != w.n || !rowgemv_ln_applicable(rows, w.n, w.k))
    return false;  // (w.k == w.n: the launch quantises in place, row strides must agree)
  const double M = double(rows), N = double(w.n), K = double(w.k);
  // algorithmic bytes: the weight once (every row re-reads it from L2), rows in / out
  begin_kernel("rowgemv_norm_quantize", M * K + N * K + 16 * N + 8 * M + 8 * M * N + M * N,
               2 * M * N * K, rows, w.n, w.k);
  launch_rowgemv_ln(cq.q, cq.s, w.wq, w.wscale, w.bias, residual, y, rows, w.n, w.k, next_ln->gamma,
                    next_ln->beta, next_ln->eps, next_ln->use_residual, quant_mode_, st.qbuf,
                    st.qscale, cur_stream());
  end_kernel();
  return true;
}

// layers::Dense, AWQ branch (src/layers/common.cc:402-438: GemmAwq / GemvAwq + bias, activation,
// residual; the reference's other arm - dequantise the whole weight, then a float16 GEMM, when
// m >= 1024 - computes the same product). x is rounded to float16 here.
void Engine::dense_awq(const DenseW& w, const float* x, int64_t m, int act, const float* residual,
                       float* y, DecodeState* ws) {
  CT2X_CHECK(ws != nullptr && ws->hbuf != nullptr, "AWQ Dense without its float16 scratch buffer");
  begin_kernel("convert_f16", double(m) * w.k * 6, 0, m, 0, w.k);
  launch_convert(x, kF32, ws->hbuf, kF16, m * w.k, cur_stream());  // DType values = ct2x_dtype
  end_kernel();
  const double bytes = double(w.n) * w.k / 2 + 2.0 * m * w.k + 4.0 * m * w.n;
  begin_kernel(w.awq == 1 ? "awq_gemm" : "awq_gemv", bytes, 2.0 * m * w.n * w.k, m, w.n, w.k);
  if (w.awq == 1)
    launch_awq_gemm(ws->hbuf, w.aq_w, w.aq_s, w.aq_z, m, w.k, w.n, w.aq_group, nullptr, y, cur_stream(),
                    ws->awq_ws);
  else
    launch_awq_gemv(ws->hbuf, w.aq_w, w.aq_s, w.aq_z, m, w.k, w.n, w.aq_group, w.aq_zw, w.aq_sw,
                    nullptr, y, cur_stream());
  end_kernel();
  // apply_bias_and_activation (src/ops/gemm.cc:11-25): bias, activation, then the residual
  if (act >= 0) {
    if (w.bias) {
      begin_kernel("bias_residual");
      launch_bias_residual(y, w.bias, nullptr, m, w.n, y, cur_stream());
      end_kernel();
    }
    begin_kernel("activation");
    launch_unary(act, y, y, m * w.n, cur_stream());
    end_kernel();
    if (residual) {
      begin_kernel("bias_residual");
      launch_bias_residual(y, nullptr, residual, m, w.n, y, cur_stream());
      end_kernel();
    }
  } else if (w.bias || residual) {
    begin_kernel("bias_residual");
    launch_bias_residual(y, w.bias, residual, m, w.n, y, cur_stream());
    end_kernel();
  }
  if (taps_) tap("d", y, 0, m, w.n);
}

void Engine::dense(const DenseW& w, const float* x, const QuantIn* qin, int64_t m, int act,
                   const float* residual, float* y, DecodeState* ws) {
  if (w.awq) {
    CT2X_CHECK(x != nullptr, "AWQ Dense takes the fp32 activation");
    dense_awq(w, x, m, act, residual, y, ws);
    return;
  }
  if (w.row_parallel && tp_) {
    // Row-parallel Dense (attention-out / FFN-out; layers::Dense TP arm, src/layers/common.cc:
    // 348-401): this rank holds K/world input columns. int8: global Quantize scale = max over
    // ranks of the slice absmax, exact int32 partial products summed over ranks, ONE dequantise
    // (+bias, +residual) on every rank - bit-identical to the unsharded layer. float: fp32 partial
    // products summed over ranks, then bias and residual.
    CT2X_CHECK(x != nullptr && act < 0, "row-parallel Dense takes the fp32 activation, no activation");
    CT2X_CHECK(ws && ws->tp_acc && ws->tp_amax, "tensor-parallel scratch buffers are missing");
    if (w.quantized && tp_fast_) {
      // The reference's exchange (src/layers/common.cc:348-401 + the ReduceAll of its callers):
      // every rank quantises ITS slice with its own row scale, dequantises its partial product
      // and the float outputs are summed - ONE collective per layer instead of two, not
      // bit-identical to the unsharded layer (CT2X_TP_FAST=1).
      quant(x, m, w.k, ws->qbuf, ws->qscale);
      float* acc = reinterpret_cast<float*>(ws->tp_acc);
      GemmS8Epilogue ep;
      ep.mode = 1;
      ep.a_scale = ws->qscale;
      ep.w_scale = w.wscale;
      ep.y = acc;
      ep.ldy = w.n;
      begin_kernel("gemm_s8_dequant", double(m) * w.k + double(w.n) * w.k + 4.0 * m * w.n,
                   2.0 * m * w.n * w.k, m, w.n, w.k);
      launch_gemm_s8(ws->qbuf, w.wq, m, w.n, w.k, ep, cur_stream());
      end_kernel();
      begin_kernel("tp_all_reduce_sum", double(m) * w.n * 4, 0, m, w.n, 0);
      tp_->all_reduce_sum_f32(acc, m * w.n, cur_stream());
      end_kernel();
      begin_kernel("bias_residual", double(m) * w.n * 8, 0, m, w.n, 0);
      launch_bias_residual(acc, w.bias, residual, m, w.n, y, cur_stream());
      end_kernel();
    } else if (w.quantized && tp_->peer_box() && tp_oneshot_fits(m, m * w.n)) {
      // One-shot exchanges inside the producer / consumer kernels (tp_oneshot.hip): 3 launches and
      // no collective call instead of 6 launches with two all-reduces.
      const TpPeerBox& pb = *tp_->peer_box();
      begin_kernel("tp_amax_quantize", double(m) * w.k * 5 + 16.0 * m * pb.world, 0, m, 0, w.k);
      launch_tp_amax_quantize(pb, x, m, w.k, quant_mode_, ws->qbuf, ws->qscale, cur_stream());
      end_kernel();
      GemmS8Epilogue ep;
      ep.mode = 0;
      ep.c_i32 = ws->tp_acc;
      ep.ldy = w.n;
      begin_kernel("gemm_s8s8s32", double(m) * w.k + double(w.n) * w.k + 4.0 * m * w.n,
                   2.0 * m * w.n * w.k, m, w.n, w.k);
      launch_gemm_s8(ws->qbuf, w.wq, m, w.n, w.k, ep, cur_stream());
      end_kernel();
      begin_kernel("tp_sum_dequant", double(m) * w.n * (8 + 16.0 * (pb.world - 1)), 0, m, w.n, 0);
      launch_tp_sum_dequant(pb, ws->tp_acc, ws->qscale, w.wscale, w.bias, residual, m, w.n, y,
                            cur_stream());
      end_kernel();
    } else if (w.quantized) {
      begin_kernel("tp_row_amax", double(m) * w.k * 4, 0, m, 0, w.k);
      launch_row_amax(x, m, w.k, ws->tp_amax, cur_stream());
      end_kernel();
      begin_kernel("tp_all_reduce_max", double(m) * 4, 0, m, 0, 0);
      tp_->all_reduce_max_f32(ws->tp_amax, m, cur_stream());
      end_kernel();
      begin_kernel("tp_quantize", double(m) * w.k * 5, 0, m, 0, w.k);
      launch_quantize_with_amax(x, ws->tp_amax, m, w.k, quant_mode_, ws->qbuf, ws->qscale,
                                cur_stream());
      end_kernel();
      GemmS8Epilogue ep;
      ep.mode = 0;
      ep.c_i32 = ws->tp_acc;
      ep.ldy = w.n;
      begin_kernel("gemm_s8s8s32", double(m) * w.k + double(w.n) * w.k + 4.0 * m * w.n,
                   2.0 * m * w.n * w.k, m, w.n, w.k);
      launch_gemm_s8(ws->qbuf, w.wq, m, w.n, w.k, ep, cur_stream());
      end_kernel();
      begin_kernel("tp_all_reduce_sum", double(m) * w.n * 4, 0, m, w.n, 0);
      tp_->all_reduce_sum_i32(ws->tp_acc, m * w.n, cur_stream());
      end_kernel();
      begin_kernel("dequantize_gemm_output", double(m) * w.n * 8, 0, m, w.n, 0);
      launch_dequantize_gemm_output(ws->tp_acc, ws->qscale, w.wscale, w.bias, m, w.n, -1, y,
                                    cur_stream(), residual);
      end_kernel();
    } else {
      float* acc = reinterpret_cast<float*>(ws->tp_acc);
      if (w.wh) {
        begin_kernel("gemm_h", 4.0 * m * w.k + 2.0 * w.n * w.k + 4.0 * m * w.n, 2.0 * m * w.n * w.k,
                     m, w.n, w.k);
        launch_gemm_h(x, w.wh, w.wh_bf16, m, w.n, w.k, nullptr, nullptr, -1, acc, cur_stream(),
                      ws ? ws->splitk_ws : nullptr, ws ? ws->splitk_floats : 0);
      } else {
        begin_kernel("gemm_f32", 4.0 * (double(m) * w.k + double(w.n) * w.k + double(m) * w.n),
                     2.0 * m * w.n * w.k, m, w.n, w.k);
        launch_gemm_f32(x, w.wf, 1, m, w.n, w.k, true, 1.f, 0, 0, 0, nullptr, nullptr, -1, acc,
                        cur_stream());
      }
      end_kernel();
      if (tp_->peer_box() && tp_oneshot_fits(m, m * w.n)) {
        begin_kernel("tp_sum_bias", double(m) * w.n * (8 + 16.0 * (tp_->world() - 1)), 0, m, w.n, 0);
        launch_tp_sum_bias(*tp_->peer_box(), acc, w.bias, residual, m, w.n, y, cur_stream());
        end_kernel();
      } else {
        begin_kernel("tp_all_reduce_sum", double(m) * w.n * 4, 0, m, w.n, 0);
        tp_->all_reduce_sum_f32(acc, m * w.n, cur_stream());
        end_kernel();
        begin_kernel("bias_residual", double(m) * w.n * 8, 0, m, w.n, 0);
        launch_bias_residual(acc, w.bias, residual, m, w.n, y, cur_stream());
        end_kernel();
      }
    }
    return;
  }
  if (w.quantized) {
    QuantIn local;
    if (!qin) {
      quant(x, m, w.k, ws->qbuf, ws->qscale);
      local.q = ws->qbuf;
      local.s = ws->qscale;
      qin = &local;
    }
    GemmS8Epilogue ep;
    ep.mode = 1;
    ep.a_scale = qin->s;
    ep.w_scale = w.wscale;
    ep.bias = w.bias;
    ep.residual = residual;
    ep.y = y;
    ep.act = act;
    ep.ldy = w.n;
    {
      const double M = double(m), N = double(w.n), K = double(w.k);
      begin_kernel("gemm_s8_dequant", M * K + N * K + 4 * N + 4 * M + 4 * M * N +
                                          (residual ? 4 * M * N : 0) + (w.bias ? 4 * N : 0),
                   2 * M * N * K, m, w.n, w.k);
    }
    const int8_t* wq = w.wq;
    if (w.wq_stream && !w.wq_stream_dual && gemm_s8_stream_applicable(m, w.n, w.k, false)) {
      wq = w.wq_stream;
      ep.packed_b = true;
    }
    launch_gemm_s8(qin->q, wq, m, w.n, w.k, ep, cur_stream());
    end_kernel();
  } else if (w.wh) {
    {
      const double M = double(m), N = double(w.n), K = double(w.k);
      begin_kernel("gemm_h", 4 * M * K + 2 * N * K + 4 * M * N + (residual ? 4 * M * N : 0),
                   2 * M * N * K, m, w.n, w.k);
    }
    launch_gemm_h(x, w.wh, w.wh_bf16, m, w.n, w.k, w.bias, residual, act, y, cur_stream(),
                  ws ? ws->splitk_ws : nullptr, ws ? ws->splitk_floats : 0);
    end_kernel();
  } else {
    {
      const double M = double(m), N = double(w.n), K = double(w.k);
      begin_kernel("gemm_f32", 4 * (M * K + N * K + M * N + (residual ? M * N : 0)), 2 * M * N * K,
                   m, w.n, w.k);
    }
    launch_gemm_f32(x, w.wf, 1, m, w.n, w.k, true, 1.f, 0, 0, 0, w.bias, residual, act, y,
                    cur_stream());
    end_kernel();
  }
  if (taps_) tap("d", y, 0, m, w.n);
}

// ---------------------------------------------------------------------------------------------
// Session
// ---------------------------------------------------------------------------------------------

DecodeState::~DecodeState() {
  if (engine) {
    (void)hipSetDevice(engine->device_index());
    (void)hipStreamSynchronize(stream ? stream : engine->stream());
  }
  for (int i = 0; i < 2; ++i) {
    if (graph_exec[i]) (void)hipGraphExecDestroy(graph_exec[i]);
    if (pin_ids[i]) (void)hipHostFree(pin_ids[i]);
    if (pin_scores[i]) (void)hipHostFree(pin_scores[i]);
    if (pin_event[i]) (void)hipEventDestroy(pin_event[i]);
    if (pin_cand_word[i]) (void)hipHostFree(pin_cand_word[i]);
    if (pin_cand_score[i]) (void)hipHostFree(pin_cand_score[i]);
    if (pin_cand_origin[i]) (void)hipHostFree(pin_cand_origin[i]);
    if (pin_cand_event[i]) (void)hipEventDestroy(pin_cand_event[i]);
  }
  if (graph) (void)hipGraphDestroy(graph);

  for (void* p : owned) {
    try { device_free(p); } catch (...) {}
  }
}

namespace {

template <typename T>
T* alloc(DecodeState& st, int64_t count) {
  void* p = device_malloc(static_cast<size_t>(std::max<int64_t>(count, 1)) * sizeof(T));
  st.owned.push_back(p);
  return static_cast<T*>(p);
}

}  // namespace

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
  if (compute_type_ == ComputeType::kFloat16 || compute_type_ == ComputeType::kBFloat16) {
    // split-K partials of the 16-bit GEMM at a few hundred decode rows (beam search on large
    // batches): session owned, so a captured step never allocates
    st->splitk_floats = static_cast<size_t>(16) * rows * std::max(std::max(d, wffn), wqkv);
    // (the vocabulary projection of a few rows: K slices of 512 over workgroups, gemm_h_rows_kernel)
    if (rows <= 64)
      st->splitk_floats = std::max(st->splitk_floats,
                                   static_cast<size_t>((d + 511) / 512) * rows * static_cast<size_t>(V));
    st->splitk_ws = alloc<float>(*st, static_cast<int64_t>(st->splitk_floats));
  }
  st->logits = alloc<float>(*st, rows * V);
  st->d_ids = alloc<int32_t>(*st, rows);
  st->d_step = alloc<int32_t>(*st, 1);
  st->d_done = alloc<int32_t>(*st, 1);
  CT2X_HIP_CHECK(hipMemsetAsync(st->d_done, 0, sizeof(int32_t), cur_stream()));
  st->hist_ids = alloc<int32_t>(*st, (t_max + 1) * rows);
  st->hist_scores = alloc<float>(*st, (t_max + 1) * rows);
  CT2X_HIP_CHECK(hipMemsetAsync(st->d_step, 0, sizeof(int32_t), cur_stream()));
  st->logprobs = alloc<float>(*st, rows * V);  // log-probabilities (beam search) / exp scratch
  if (beam_mode) {
    st->beam_scores = alloc<float>(*st, rows);
    st->topk_scores = alloc<float>(*st, batch * 2 * beam);
    st->topk_ids = alloc<int32_t>(*st, batch * 2 * beam);
    st->d_parents = alloc<int32_t>(*st, rows);
    CT2X_CHECK(2 * beam <= 64, "beam_size must be <= 32");
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
    bool ok = allowed;
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
      ok = attention_exact_decode_applicable(p, false);
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
    if (features)
      encode_features(*st, features, frames);
    else
      encode(*st, src_ids, src_lengths);
  }
  return st;
}

// TransformerEncoder::operator() (src/layers/transformer.cc:427-471) on padded rows followed by
// the cross-attention K/V projection of every decoder layer (process_cross_attention,
// src/layers/attention.cc:371-440), which the reference runs lazily at decoding step 0.
// Padded positions are computed but never observed: every op on this path is row-wise and the
// attention kernels mask keys beyond lengths[b].
void Engine::encode(DecodeState& st, const int32_t* src_ids, const int32_t* src_lengths) {
  const int64_t B = st.batch, S = st.s_max, d = enc_.d_model, rows = B * S;
  int32_t* d_src = alloc<int32_t>(st, rows);
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
  encode_layers(st, x, y);
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
  const int64_t B = st.batch, S = st.s_max, d = enc_.d_model, rows = B * S;
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
    AttnFullParams f;
    f.q = st.qkv;
    f.k = st.qkv + qdim;
    f.v = st.qkv + qdim + kvdim;
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
    f.scale = a.queries_scale;
    f.out = st.ctx;
    begin_kernel("attention_encoder", 0, 0, rows, a.heads, a.dhead);
    launch_attention_full(f, cur_stream());
    end_kernel();
    tap("ctx", st.ctx, 0, rows, a.heads * a.dhead);
    dense(a.lin[1], st.ctx, nullptr, rows, -1, a.ln.present ? x : nullptr, y, &st);
    if (a.ln.present && !enc_.pre_norm) norm(a.ln, y, rows, d, y);
    std::swap(x, y);

    // --- feed forward (FeedForwardNetwork::operator(), src/layers/transformer.cc:21-51)
    tap_scope("enc.l" + std::to_string(enc_layer_index) + ".ffn");
    run_ffn(layer.ffn, enc_.pre_norm, rows, d, x, y, st);
    ++enc_layer_index;
  }
  if (enc_.output_norm.present) norm(enc_.output_norm, x, rows, d, x);
  st.enc_out = x;
  tap_scope("enc.out");
  tap("x", x, 0, rows, d);

  // Cross-attention memory projections: linear_1(memory) -> [B*S, 2*kv_dim], kept in the GEMM
  // output layout (the attention kernel takes strides; no split_heads transpose).
  bool quantized_once = false;
  tap_scope("enc.kv");
  for (size_t l = 0; l < dec_.layers.size(); ++l) {
    const AttnW& a = dec_.layers[l].cross;
    QuantIn mq;
    if (a.lin[1].quantized) {
      if (!quantized_once) {  // the same memory feeds every layer: quantise it once
        quant(x, rows, d, st.qbuf, st.qscale);
        quantized_once = true;
      }
      mq.q = st.qbuf;
      mq.s = st.qscale;
    }
    dense(a.lin[1], x, mq.q ? &mq : nullptr, rows, -1, nullptr, st.cross_kv[l], &st);
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
  if (qin_ready) {
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
  if (ffn.l0_noact.present && fq.q && ffn.l0.quantized && ffn.l0_noact.quantized && rows <= 32 &&
      ffn.l0.k % 16 == 0 && ffn.l0_noact.n == ffn.l0.n) {
    // Decode rows: gate and up projection in ONE launch, y = act(gate) * up (no fp32
    // intermediates, no ops::Mul launch).
    GemmS8Epilogue ep;
    ep.mode = 1;
    ep.a_scale = fq.s;
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
    begin_kernel("gemm_s8_glu", M * K + 2 * N * K + 8 * N + 4 * M + 4 * M * N, 4 * M * N * K, rows,
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
  CT2X_HIP_CHECK(hipStreamSynchronize(cur_stream()));
  check_last_launch("prefill");
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

  st.pf_active = false;
  st.pf_pending = PrefetchArgs();
  pf_state_ = nullptr;
  // Final norm + vocabulary projection (transformer.cc:842-855).
  tap_scope("dec.out");
  QuantIn pq;
  const float* proj_in = x;
  if (!dense_norm(dec_.output_norm, dec_.projection, x, rows, -1, nullptr, st.logits)) {
    if (dec_.output_norm.present) {
      if (dec_.projection.quantized) {
        norm_quant(dec_.output_norm, x, rows, d, nullptr, st.qbuf, st.qscale);
        pq.q = st.qbuf;
        pq.s = st.qscale;
      } else {
        norm(dec_.output_norm, x, rows, d, st.ctx);
        proj_in = st.ctx;
      }
    }
    dense(dec_.projection, proj_in, pq.q ? &pq : nullptr, rows, -1, nullptr, st.logits, &st);
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
    if (st.g_topk == 1) {
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
    begin_kernel("beam_logprobs", double(st.rows) * V * 12, 0, st.rows, 0, V);
    launch_beam_logprobs(st.logits, st.beam_scores, st.rows, V, st.g_eos, st.g_min_length,
                         st.d_step, st.logprobs, cur_stream(),
                         st.prefix_extends_min ? st.d_bprefix_len : nullptr,
                         static_cast<int32_t>(st.beam), !half_compute(), st.d_bprefix, st.d_bprefix_len,
                         st.bprefix_stride, st.prefix_first_step, st.d_bprefix ? st.bias_beta : 0.f,
                         st.d_diverged);
    end_kernel();
    begin_kernel("beam_topk", double(st.rows) * V * 4, 0, st.batch, 2 * st.beam, st.beam * V);
    launch_topk(st.logprobs, st.batch, st.beam * V, 2 * st.beam, st.topk_scores, st.topk_ids,
                cur_stream(), st.topk_ws, st.topk_ws_bytes);
    end_kernel();
    begin_kernel("beam_select");
    launch_beam_select(st.topk_scores, st.topk_ids, st.batch, static_cast<int32_t>(st.beam), V,
                       st.g_eos, st.d_bprefix, st.d_bprefix_len, st.bprefix_stride, st.d_step,
                       st.prefix_first_step, st.hist_cand_word, st.hist_cand_score,
                       st.hist_cand_origin, st.d_ids, st.d_parents, st.beam_scores,
                       (st.lp_active || st.wh_active) ? st.alive_seq[par] : nullptr,
                       (st.lp_active || st.wh_active) ? st.alive_seq[par ^ 1] : nullptr,
                       static_cast<int32_t>(st.t_max), cur_stream(),
                       st.ancestry ? st.anc[par] : nullptr, st.ancestry ? st.anc[par ^ 1] : nullptr,
                       (st.d_bprefix && st.bias_beta > 0.f) ? st.d_diverged : nullptr);
    end_kernel();
    // Positions [0, step] are live; the gather copies the positions below the advanced counter.
    begin_kernel("advance_step");
    launch_advance_step(st.d_step, cur_stream());
    end_kernel();
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
  const int parity = st.cur;
  if (!st.graph_ready[parity]) {
    // One capture at a time per process: concurrent captures from several worker threads fail on
    // ROCm 7.2 ("operation failed due to a previous error during capture"); replays are free.
    static std::mutex capture_mutex;
    std::lock_guard<std::mutex> capture_lock(capture_mutex);
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
    CT2X_HIP_CHECK(hipGraphInstantiate(&st.graph_exec[parity], g, nullptr, nullptr, 0));
    CT2X_HIP_CHECK(hipGraphDestroy(g));
    st.graph_ready[parity] = true;
    }
  }
  CT2X_HIP_CHECK(hipGraphLaunch(st.graph_exec[parity], cur_stream()));
  if (st.fuse_beam) st.cur ^= 1;
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
      if (st.pin_cand_word[i]) (void)hipHostFree(st.pin_cand_word[i]);
      if (st.pin_cand_score[i]) (void)hipHostFree(st.pin_cand_score[i]);
      if (st.pin_cand_origin[i]) (void)hipHostFree(st.pin_cand_origin[i]);
      CT2X_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&st.pin_cand_word[i]), need * sizeof(int32_t)));
      CT2X_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&st.pin_cand_score[i]), need * sizeof(float)));
      CT2X_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&st.pin_cand_origin[i]), need * sizeof(int32_t)));
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
      if (st.pin_ids[i]) (void)hipHostFree(st.pin_ids[i]);
      if (st.pin_scores[i]) (void)hipHostFree(st.pin_scores[i]);
      CT2X_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&st.pin_ids[i]), need * sizeof(int32_t)));
      CT2X_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&st.pin_scores[i]), need * sizeof(float)));
      if (!st.pin_event[i]) CT2X_HIP_CHECK(hipEventCreateWithFlags(&st.pin_event[i], hipEventDisableTiming));
    }
    st.pin_capacity = need;
  }
  CT2X_HIP_CHECK(hipMemcpyAsync(st.pin_ids[slot], st.hist_ids + first * st.rows,
                                sizeof(int32_t) * need, hipMemcpyDeviceToHost, cur_stream()));
  CT2X_HIP_CHECK(hipMemcpyAsync(st.pin_scores[slot], st.hist_scores + first * st.rows,
                                sizeof(float) * need, hipMemcpyDeviceToHost, cur_stream()));
  CT2X_HIP_CHECK(hipEventRecord(st.pin_event[slot], cur_stream()));
}

void Engine::wait_history(DecodeState& st, int slot, const int32_t** h_ids,
                          const float** h_scores) {
  CT2X_HIP_CHECK(hipEventSynchronize(st.pin_event[slot]));
  *h_ids = st.pin_ids[slot];
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

void Engine::set_step(DecodeState& st, int32_t step) {
  tls_stream = st.stream;
  set_active_stream(tls_stream);
  CT2X_HIP_CHECK(hipMemcpyAsync(st.d_step, &step, sizeof(int32_t), hipMemcpyHostToDevice,
                                cur_stream()));
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
