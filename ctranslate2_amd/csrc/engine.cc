#include "engine.h"
#include "host_math.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>

#include "engine_internal.h"
#include "runtime.h"

namespace ct2x {

// (engine_internal.h) the stream of the session the calling thread is driving, and its launch count
thread_local hipStream_t tls_stream = nullptr;
thread_local int64_t tls_launches = 0;

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

float half_bits_to_float(uint16_t h) {  // IEEE binary16 -> float, exact
  const uint32_t sign = static_cast<uint32_t>(h & 0x8000u) << 16;
  const uint32_t exp = (h >> 10) & 31u, man = h & 0x3ffu;
  uint32_t bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else {  // subnormal: normalise
      int e = -1;
      uint32_t m = man;
      do {
        ++e;
        m <<= 1;
      } while ((m & 0x400u) == 0);
      bits = sign | (static_cast<uint32_t>(127 - 15 - e) << 23) | ((m & 0x3ffu) << 13);
    }
  } else if (exp == 31) {
    bits = sign | 0x7f800000u | (man << 13);
  } else {
    bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
  }
  float f;
  std::memcpy(&f, &bits, 4);
  return f;
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
thread_local const DecodeState* Engine::argmax_state_ = nullptr;

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
    // host copies: words, scales as 16-bit patterns, zero words
    std::vector<int32_t> qw(w->data.size() / 4), qz(zero->data.size() / 4);
    std::vector<uint16_t> sc(scale->data.size() / 2);
    std::memcpy(qw.data(), w->data.data(), qw.size() * 4);
    std::memcpy(qz.data(), zero->data.data(), qz.size() * 4);
    std::memcpy(sc.data(), scale->data.data(), sc.size() * 2);
    // Compute type bfloat16 (round 6; BASELINE configs[4]): the scales are rounded to bfloat16 once
    // here and the bf16 arm of the kernels runs (awq.hip) - the reference has float16 only.
    d.aq_bf16 = compute_type_ == ComputeType::kBFloat16;
    if (d.aq_bf16)
      for (uint16_t& v : sc) v = float_to_bf16_bits(half_bits_to_float(v));
    std::vector<float> bias;
    if (const Variable* b = model_->find(scope + "/bias")) bias = b->to_float();
    if (tp_ && shard != Shard::kNone) {
      // Tensor parallelism (round 6; Model::load shards every Dense alike, src/models/model.cc:
      // 662-743 - the reference never meets AWQ there because its CUDA-only kernels are not in the
      // ROCm build): the packed words, scales and zero points are cut along the same axis as a
      // float weight would be; cuts fall on words (8 channels) and on quantisation groups.
      const int world = tp_->world(), rank = tp_->rank();
      const int64_t G = d.aq_group, groups = d.k / G;
      if (shard == Shard::kColumn) {
        const auto rows = column_shard_rows(d.n, blocks, rank, world, scope);
        int64_t n_local = 0;
        for (const auto& r : rows) {
          if (r.first % 8 != 0 || r.second % 8 != 0)
            throw std::invalid_argument(scope + ": an AWQ column shard must start and end on a packed word (8 channels)");
          n_local += r.second;
        }
        if (method == 1) {
          std::vector<int32_t> qw2, qz2;
          std::vector<uint16_t> sc2;
          const int64_t qout = d.n / 8;
          for (int64_t kk = 0; kk < d.k; ++kk)
            for (const auto& r : rows)
              qw2.insert(qw2.end(), qw.begin() + kk * qout + r.first / 8,
                         qw.begin() + kk * qout + (r.first + r.second) / 8);
          for (int64_t g = 0; g < groups; ++g)
            for (const auto& r : rows) {
              qz2.insert(qz2.end(), qz.begin() + g * qout + r.first / 8,
                         qz.begin() + g * qout + (r.first + r.second) / 8);
              sc2.insert(sc2.end(), sc.begin() + g * d.n + r.first, sc.begin() + g * d.n + r.first + r.second);
            }
          qw.swap(qw2); qz.swap(qz2); sc.swap(sc2);
        } else {
          qw = take_rows(qw.data(), d.k / 8, rows);
          qz = take_rows(qz.data(), d.aq_zw, rows);
          sc = take_rows(sc.data(), d.aq_sw, rows);
        }
        if (!bias.empty()) bias = take_rows(bias.data(), 1, rows);
        d.n = n_local;
      } else {  // Shard::kRow: K / world input channels, whole groups
        if (d.k % world != 0 || (d.k / world) % G != 0 || (d.k / world) % 128 != 0)
          throw std::invalid_argument(scope + ": an AWQ row shard must hold whole quantisation groups "
                                              "and a multiple of 128 input channels");
        const int64_t kl = d.k / world, k0 = rank * kl, g0 = k0 / G, gl = kl / G;
        if (method == 1) {
          const int64_t qout = d.n / 8;
          qw = std::vector<int32_t>(qw.begin() + k0 * qout, qw.begin() + (k0 + kl) * qout);
          qz = std::vector<int32_t>(qz.begin() + g0 * qout, qz.begin() + (g0 + gl) * qout);
          sc = std::vector<uint16_t>(sc.begin() + g0 * d.n, sc.begin() + (g0 + gl) * d.n);
        } else {
          // GEMV layout: zero points are nibbles (group g = nibble g % 8 of word g / 8) and both
          // side tables are padded to the widths the layout prescribes for the LOCAL in_c
          // (src/ops/awq/gemv_gpu.cu:296-302, 365-369; oracle/awq_oracle.py gemv_widths)
          auto md = [](int64_t c, int64_t dv) { return (c + dv - 1) / dv; };
          const int64_t zw2 = G == 64 ? md(md(gl, 8), 2) * 2 : md(gl, 8), sw2 = zw2 * 8;
          std::vector<int32_t> qw2 = take_cols(qw.data(), d.n, d.k / 8, k0 / 8, (k0 + kl) / 8);
          std::vector<int32_t> qz2(static_cast<size_t>(d.n) * zw2, 0);
          std::vector<uint16_t> sc2(static_cast<size_t>(d.n) * sw2, 0);
          for (int64_t oc = 0; oc < d.n; ++oc)
            for (int64_t g = 0; g < gl; ++g) {
              const int64_t gs = g0 + g;
              const uint32_t z = (static_cast<uint32_t>(qz[oc * d.aq_zw + gs / 8]) >> (4 * (gs % 8))) & 15u;
              qz2[oc * zw2 + g / 8] = static_cast<int32_t>(static_cast<uint32_t>(qz2[oc * zw2 + g / 8]) |
                                                           (z << (4 * (g % 8))));
              sc2[oc * sw2 + g] = sc[oc * d.aq_sw + gs];
            }
          qw.swap(qw2); qz.swap(qz2); sc.swap(sc2);
          d.aq_zw = zw2;
          d.aq_sw = sw2;
        }
        d.k = kl;
        d.row_parallel = true;  // bias stays whole: applied once, after the rank sum
      }
    }
    d.aq_w = static_cast<const int32_t*>(upload(qw.data(), qw.size() * 4));
    d.aq_s = upload(sc.data(), sc.size() * 2);
    d.aq_z = static_cast<const int32_t*>(upload(qz.data(), qz.size() * 4));
    if (!bias.empty()) d.bias = upload_floats(bias);
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
  pack_cluster_weights();
  // the re-pack launches run on the engine's own stream, the sessions on their callers' streams:
  // nothing orders the two but this (a first decode call could otherwise read a half-packed weight)
  CT2X_HIP_CHECK(hipStreamSynchronize(stream_));
}

// The decoder stack as one launch per step (decoder_cluster.hip) serves the int8 pre-norm
// encoder-decoder Transformer at the geometry its slices are cut for; CT2X_CLUSTER_STACK=0 keeps
// the launch-per-op step (A/B runs, tests/test_decoder_cluster_gpu.py).
void Engine::pack_cluster_weights() {
  static const bool enabled = [] {
    const char* e = std::getenv("CT2X_CLUSTER_STACK");
    return e == nullptr || e[0] != '0';
  }();
  cluster_ready_ = false;
  if (!enabled || tp_ || !has_encoder_ || is_whisper() || compute_type_ != ComputeType::kInt8Float32 ||
      !dec_.pre_norm || dec_.layernorm_embedding.present || dec_.layers.empty() ||
      dec_.layers.size() > static_cast<size_t>(kClusterMaxLayers))
    return;
  auto dense_ok = [](const DenseW& w, int64_t n, int64_t k) {
    return w.present && w.quantized && !w.row_parallel && !w.awq && w.wq != nullptr && w.wscale != nullptr &&
           w.n == n && w.k == k;
  };
  for (const LayerW& layer : dec_.layers) {
    const AttnW& s = layer.self;
    const AttnW& c = layer.cross;
    if (layer.variant != 0 || !layer.has_cross || s.rotary || c.rotary || s.no_residual || c.no_residual ||
        layer.ffn.no_residual || layer.ffn.l0_noact.present || !s.ln.present || !c.ln.present ||
        !layer.ffn.ln.present || s.heads_kv != s.heads || c.heads_kv != c.heads || c.heads != s.heads ||
        c.dhead != s.dhead || !decoder_cluster_geometry_ok(dec_.d_model, s.heads, s.dhead, layer.ffn.l0.n))
      return;
    const int64_t d = dec_.d_model, f = layer.ffn.l0.n;
    if (!dense_ok(s.lin[0], 3 * d, d) || !dense_ok(s.lin[1], d, d) || !dense_ok(c.lin[0], d, d) ||
        !dense_ok(c.lin[1], 2 * d, d) || !dense_ok(c.lin[2], d, d) || !dense_ok(layer.ffn.l0, f, d) ||
        !dense_ok(layer.ffn.l1, d, f))
      return;
  }
  auto pack = [&](const DenseW& w, int sections) {
    int8_t* packed = static_cast<int8_t*>(device_malloc(static_cast<size_t>(w.n) * w.k));
    device_blocks_.push_back(packed);
    launch_pack_cluster_weight(w.wq, w.n, w.k, sections, packed, stream_);
    ClusterDenseW out;
    out.w = packed;
    out.wscale = w.wscale;
    out.bias = w.bias;
    return out;
  };
  auto norm_of = [](const NormW& n) {
    ClusterNormW out;
    out.gamma = n.gamma;
    out.beta = n.beta;
    out.eps = n.eps;
    out.rms_use_residual = n.use_residual ? 1 : 0;
    return out;
  };
  cluster_layers_.clear();
  for (const LayerW& layer : dec_.layers) {
    ClusterLayerArgs a;
    a.ln_self = norm_of(layer.self.ln);
    a.ln_cross = norm_of(layer.cross.ln);
    a.ln_ffn = norm_of(layer.ffn.ln);
    a.qkv = pack(layer.self.lin[0], 3);
    a.self_out = pack(layer.self.lin[1], 1);
    a.cross_q = pack(layer.cross.lin[0], 1);
    a.cross_out = pack(layer.cross.lin[2], 1);
    a.ffn0 = pack(layer.ffn.l0, 1);
    a.ffn1 = pack(layer.ffn.l1, 1);
    a.self_scale = layer.self.queries_scale;
    a.cross_scale = layer.cross.queries_scale;
    cluster_layers_.push_back(a);
  }
  check_last_launch("pack_cluster_weights");
  cluster_ready_ = true;
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
              nullptr, nullptr, cur_stream(), nullptr, half_compute() && !taps_);
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
  if (!gemm_s8_norm_fusion_enabled() || taps_ || !ln.present || !w.quantized || w.row_parallel)
    return false;
  // streaming form (weights too large to stay cache resident, <= 16 rows, RMSNorm rows of 4096)
  const bool streamable = w.wq_stream != nullptr && !w.wq_stream_dual && !tp_;
  // (round 6: the norm by sixteen leader workgroups of the launch, CT2X_STREAM_PROLOGUE bit 2)
  GemmS8Epilogue ep;
  const bool lead = streamable && (gemm_s8_stream_prologue_mask() & 4) &&
                    gemm_s8_stream_lead_norm_applicable(m, w.n, w.k, false, ln.beta != nullptr) &&
                    lead_fill(ep, m);
  const bool stream = lead || (streamable && (gemm_s8_stream_prologue_mask() & 1) &&
                               gemm_s8_stream_norm_applicable(m, w.n, w.k, false, ln.beta != nullptr));
  if (!stream && !gemm_s8_norm_applicable(m, w.n, w.k)) return false;
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
  ep.packed_b = stream;
  const double M = double(m), N = double(w.n), K = double(w.k);
  begin_kernel("norm_gemm_s8_dequant", 4 * M * K + N * K + 4 * N + 8 * K + 4 * M * N +
                                           (residual ? 4 * M * N : 0) + (w.bias ? 4 * N : 0),
               2 * M * N * K, m, w.n, w.k);
  launch_gemm_s8(nullptr, stream ? w.wq_stream : w.wq, m, w.n, w.k, ep, cur_stream());
  end_kernel();
  return true;
}

// An int8 Dense whose ops::Quantize is folded into the A-fragment fetch (gemm_s8.hip, QOL): x is the
// fp32 activation, row_amax its row absmax left by the producing launch.
void Engine::dense_qol(const DenseW& w, const float* x, const float* row_amax, int64_t m, int act,
                       const float* residual, float* y, int amax_parts) {
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
  ep.a_amax_parts = amax_parts;
  ep.quant_mode = quant_mode_;
  // streaming form: the fragment-major weight copy
  const bool stream = stream_qol_applicable(w, m);
  ep.packed_b = stream;
  if (stream && stream_lead_quant_applicable(w, m)) lead_fill(ep, m);
  const double M = double(m), N = double(w.n), K = double(w.k);
  begin_kernel("qgemm_s8_dequant", 4 * M * K + N * K + 4 * N + 4 * M + 4 * M * N +
                                       (residual ? 4 * M * N : 0) + (w.bias ? 4 * N : 0),
               2 * M * N * K, m, w.n, w.k);
  launch_gemm_s8(nullptr, stream ? w.wq_stream : w.wq, m, w.n, w.k, ep, cur_stream());
  end_kernel();
}

// Whether the Quantize in front of this Dense can run as the prologue of the streaming kernel
// (gemm_s8_stream.hip AMODE 1: <= 16 rows, K = 4096, packed weight present).
bool Engine::stream_qol_applicable(const DenseW& w, int64_t m) const {
  if (stream_lead_quant_applicable(w, m)) return true;
  return !tp_ && !taps_ && (gemm_s8_stream_prologue_mask() & 2) && w.quantized && !w.row_parallel &&
         w.wq_stream != nullptr && !w.wq_stream_dual && gemm_s8_stream_qol_applicable(m, w.n, w.k);
}

// ... or as the work of the launch's first sixteen workgroups (AMODE 4, round 6: any K with a
// streaming form; needs the session's tag words).
bool Engine::stream_lead_quant_applicable(const DenseW& w, int64_t m) const {
  const DecodeState* st = pf_state_;
  return !tp_ && !taps_ && (gemm_s8_stream_prologue_mask() & 8) && st != nullptr &&
         st->lead_flag != nullptr && st->rows <= 16 && w.quantized && !w.row_parallel &&
         w.wq_stream != nullptr && !w.wq_stream_dual && gemm_s8_stream_lead_quant_applicable(m, w.n, w.k);
}

bool Engine::lead_fill(GemmS8Epilogue& ep, int64_t m) {
  DecodeState* st = pf_state_;
  if (st == nullptr || st->lead_flag == nullptr || taps_ || m > 16 || st->rows > 16) return false;
  ep.lead_q = st->qbuf;
  ep.lead_scale = st->qscale;
  ep.lead_flag = st->lead_flag;
  ep.lead_step = st->d_step;
  ep.lead_tag = 1u + static_cast<unsigned>(st->lead_seq++ % 1023);
  static const int headstart = [] {  // CT2X_LEAD_HEADSTART (x 512 clocks, A/B runs)
    const char* e = std::getenv("CT2X_LEAD_HEADSTART");
    return e != nullptr ? std::atoi(e) : GemmS8Epilogue().lead_headstart;
  }();
  ep.lead_headstart = headstart;
  return true;
}

void Engine::reset_lead_flags(DecodeState& st) {
  if (st.lead_flag != nullptr)
    CT2X_HIP_CHECK(hipMemsetAsync(st.lead_flag, 0, 16 * sizeof(unsigned), cur_stream()));
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
  begin_kernel(w.aq_bf16 ? "convert_bf16" : "convert_f16", double(m) * w.k * 6, 0, m, 0, w.k);
  launch_convert(x, kF32, ws->hbuf, w.aq_bf16 ? kBF16 : kF16, m * w.k, cur_stream());  // DType values = ct2x_dtype
  end_kernel();
  // Row-parallel (tensor parallelism, round 6): this rank's K / world input channels give a partial
  // product; the ranks' fp32 partials are summed, then bias and residual once (as the float arm of
  // Engine::dense does).
  const bool rowp = w.row_parallel && tp_;
  if (rowp) {
    CT2X_CHECK(act < 0, "row-parallel Dense takes no activation");
    CT2X_CHECK(ws->tp_acc != nullptr, "tensor-parallel scratch buffers are missing");
  }
  float* out = rowp ? reinterpret_cast<float*>(ws->tp_acc) : y;
  const double bytes = double(w.n) * w.k / 2 + 2.0 * m * w.k + 4.0 * m * w.n;
  begin_kernel(w.awq == 1 ? "awq_gemm" : "awq_gemv", bytes, 2.0 * m * w.n * w.k, m, w.n, w.k);
  if (w.awq == 1)
    launch_awq_gemm(ws->hbuf, w.aq_w, w.aq_s, w.aq_z, m, w.k, w.n, w.aq_group, nullptr, out, cur_stream(),
                    ws->awq_ws, w.aq_bf16);
  else
    launch_awq_gemv(ws->hbuf, w.aq_w, w.aq_s, w.aq_z, m, w.k, w.n, w.aq_group, w.aq_zw, w.aq_sw,
                    nullptr, out, cur_stream(), w.aq_bf16);
  end_kernel();
  if (rowp) {
    begin_kernel("tp_all_reduce_sum", double(m) * w.n * 4, 0, m, w.n, 0);
    tp_->all_reduce_sum_f32(out, m * w.n, cur_stream());
    end_kernel();
    begin_kernel("bias_residual", double(m) * w.n * 8, 0, m, w.n, 0);
    launch_bias_residual(out, w.bias, residual, m, w.n, y, cur_stream());
    end_kernel();
    if (taps_) tap("d", y, 0, m, w.n);
    return;
  }
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
                      ws ? ws->splitk_ws : nullptr, ws ? ws->splitk_floats : 0,
                      ws ? ws->splitk_tickets : nullptr, ws && ws->splitk_tickets ? DecodeState::kSplitkTickets : 0);
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
    if (argmax_state_ != nullptr && y == argmax_state_->logits) {
      // the greedy sampler rides in this launch (GreedyFuse, kernels.h); no logits are stored
      const DecodeState& gs = *argmax_state_;
      ep.gf.slots = gs.gf_slots;
      ep.gf.ticket = gs.gf_ticket;
      ep.gf.step_ptr = gs.d_step;
      ep.gf.disable_id = gs.g_eos;
      ep.gf.min_length = gs.g_min_length;
      ep.gf.prefix_ids = gs.d_prefix;
      ep.gf.prefix_lens = gs.d_prefix_len;
      ep.gf.prefix_stride = gs.prefix_stride;
      ep.gf.prefix_first_step = gs.prefix_first_step;
      ep.gf.prefix_extends_min = gs.prefix_extends_min ? 1 : 0;
      ep.gf.next_ids = gs.d_ids;
      ep.gf.hist_ids = gs.hist_ids;
      ep.gf.hist_scores = gs.hist_scores;
      ep.gf.advance_step_ptr = gs.d_step;
      ep.gf.host_hist_ids = gs.host_hist_ids;
    }
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
                  ws ? ws->splitk_ws : nullptr, ws ? ws->splitk_floats : 0,
                  ws ? ws->splitk_tickets : nullptr, ws && ws->splitk_tickets ? DecodeState::kSplitkTickets : 0);
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

}  // namespace ct2x
