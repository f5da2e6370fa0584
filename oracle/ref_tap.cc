// TEST INFRASTRUCTURE ONLY. Op-level taps on the REAL reference CPU library (oracle/_ref), without
// touching its sources: this shared object defines the same (interposable) symbols as five of
// the reference's op front-ends and is loaded BEFORE libctranslate2_ref.so, so the calls the
// reference's own layers make through the PLT land here; each shim forwards to the original
// (dlsym RTLD_NEXT) and records what went in / came out while a recording is active:
//   ops::Quantize::operator()    (src/ops/quantize.cc:21-51)   fp32 input, int8 output, row scales
//                                                               = "the int8 activations + row scales
//                                                               entering each Dense" (layers::Dense,
//                                                               src/layers/common.cc:339-401)
//   ops::LayerNorm::operator()   (src/ops/layer_norm.cc:29-55)  fp32 output
//   ops::MatMul::operator()      (src/ops/matmul.cc:14-84)      fp32 output (QK^T scores, P.V context)
//   ops::SoftMax::operator()     (src/ops/softmax.cc:28-47)     fp32 output
//   ops::Dequantize::operator()  (src/ops/dequantize.cc:46-59)  fp32 Dense output before the residual
// The parity tests line the recorded sequence up against the engine's own taps
// (ct2x_debug_step_taps) and report the first tensor that differs.
#include <dlfcn.h>

#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "ctranslate2/ops/dequantize.h"
#include "ctranslate2/ops/layer_norm.h"
#include "ctranslate2/ops/matmul.h"
#include "ctranslate2/ops/quantize.h"
#include "ctranslate2/ops/softmax.h"
#include "ctranslate2/storage_view.h"

using ctranslate2::DataType;
using ctranslate2::Device;
using ctranslate2::StorageView;

namespace {

struct Rec {
  std::string kind;             // "quantize.in", "quantize.out", "quantize.scale", "layer_norm", ...
  int dtype;                    // 0 fp32, 1 int8 (raw bytes)
  std::vector<int64_t> shape;
  std::vector<char> data;
};

std::vector<Rec> g_recs;
bool g_on = false;
std::mutex g_mutex;

void record(const char* kind, const StorageView& x) {
  if (!g_on || x.device() != Device::CPU) return;
  int dtype;
  size_t item;
  if (x.dtype() == DataType::FLOAT32) {
    dtype = 0;
    item = 4;
  } else if (x.dtype() == DataType::INT8) {
    dtype = 1;
    item = 1;
  } else {
    return;
  }
  Rec r;
  r.kind = kind;
  r.dtype = dtype;
  for (ctranslate2::dim_t i = 0; i < x.rank(); ++i) r.shape.push_back(x.dim(i));
  r.data.resize(static_cast<size_t>(x.size()) * item);
  if (x.size() > 0) std::memcpy(r.data.data(), x.buffer(), r.data.size());
  std::lock_guard<std::mutex> lock(g_mutex);
  g_recs.push_back(std::move(r));
}

template <typename Fn>
Fn next(const char* mangled) {
  void* p = dlsym(RTLD_NEXT, mangled);
  if (!p) throw std::runtime_error(std::string("ref_tap: original not found: ") + mangled);
  return reinterpret_cast<Fn>(p);
}

}  // namespace

namespace ctranslate2 {
namespace ops {

void Quantize::operator()(const StorageView& input, StorageView& output, StorageView& scale) const {
  using Fn = void (*)(const Quantize*, const StorageView&, StorageView&, StorageView&);
  static Fn orig = next<Fn>("_ZNK11ctranslate23ops8QuantizeclERKNS_11StorageViewERS2_S5_");
  orig(this, input, output, scale);
  record("quantize.in", input);
  record("quantize.out", output);
  record("quantize.scale", scale);
}

void LayerNorm::operator()(const StorageView* beta, const StorageView* gamma,
                           const StorageView& input, StorageView& output) const {
  using Fn = void (*)(const LayerNorm*, const StorageView*, const StorageView*, const StorageView&,
                      StorageView&);
  static Fn orig = next<Fn>("_ZNK11ctranslate23ops9LayerNormclEPKNS_11StorageViewES4_RS3_RS2_");
  orig(this, beta, gamma, input, output);
  record("layer_norm", output);
}

void MatMul::operator()(const StorageView& a, const StorageView& b, StorageView& c) const {
  using Fn = void (*)(const MatMul*, const StorageView&, const StorageView&, StorageView&);
  static Fn orig = next<Fn>("_ZNK11ctranslate23ops6MatMulclERKNS_11StorageViewES4_RS2_");
  orig(this, a, b, c);
  record("matmul", c);
}

void SoftMax::operator()(const StorageView& x, const StorageView* lengths, StorageView& y) const {
  using Fn = void (*)(const SoftMax*, const StorageView&, const StorageView*, StorageView&);
  static Fn orig = next<Fn>("_ZNK11ctranslate23ops7SoftMaxclERKNS_11StorageViewEPS3_RS2_");
  orig(this, x, lengths, y);
  record("softmax", y);
}

void Dequantize::operator()(const StorageView& c, const StorageView& a_scale,
                            const StorageView& b_scale, const bool transpose_a,
                            const bool transpose_b, StorageView& y, const StorageView* bias) const {
  using Fn = void (*)(const Dequantize*, const StorageView&, const StorageView&, const StorageView&,
                      bool, bool, StorageView&, const StorageView*);
  static Fn orig = next<Fn>("_ZNK11ctranslate23ops10DequantizeclERKNS_11StorageViewES4_S4_bbRS2_PS3_");
  orig(this, c, a_scale, b_scale, transpose_a, transpose_b, y, bias);
  record("dense", y);
}

}  // namespace ops
}  // namespace ctranslate2

extern "C" {

void ct2tap_begin() {
  std::lock_guard<std::mutex> lock(g_mutex);
  g_recs.clear();
  g_on = true;
}

void ct2tap_end() {
  std::lock_guard<std::mutex> lock(g_mutex);
  g_on = false;
}

int ct2tap_count() {
  std::lock_guard<std::mutex> lock(g_mutex);
  return static_cast<int>(g_recs.size());
}

// kind -> name (NUL-terminated, truncated to name_size); rank and up to 8 dims -> shape.
int ct2tap_info(int i, char* name, int name_size, int32_t* dtype, int32_t* rank, int64_t* shape) {
  std::lock_guard<std::mutex> lock(g_mutex);
  if (i < 0 || i >= static_cast<int>(g_recs.size())) return -1;
  const Rec& r = g_recs[i];
  std::strncpy(name, r.kind.c_str(), name_size - 1);
  name[name_size - 1] = 0;
  *dtype = r.dtype;
  *rank = static_cast<int32_t>(r.shape.size());
  for (size_t d = 0; d < r.shape.size() && d < 8; ++d) shape[d] = r.shape[d];
  return 0;
}

int ct2tap_data(int i, void* out) {
  std::lock_guard<std::mutex> lock(g_mutex);
  if (i < 0 || i >= static_cast<int>(g_recs.size())) return -1;
  std::memcpy(out, g_recs[i].data.data(), g_recs[i].data.size());
  return 0;
}

}  // extern "C"
