// Level-1 integration stubs: the translation unit a CTranslate2 maintainer adds
// to the reference build IN PLACE OF src/ops/*_gpu.cu, src/cuda/primitives.cu and
// src/cuda/allocator.cc so that the reference's own front-end (layers::Dense,
// MultiHeadAttention, decoding.cc) runs on the MI355X kernels behind the C ABI of
// include/ct2_mi355.h. Every definition below is an explicit specialisation of a template the
// reference DECLARES (file:line given) - the CPU test-suite compiles this file against the
// reference's real headers (tests/test_host.py::test_integration_stubs_match_the_reference_headers,
// skipped where /root/reference is absent), which is the proof that the boundary fits.
//
// This file is written for this repository; it contains no reference code. float32 arm only
// (T = float): the 16-bit arms forward the same way with CT2X_F16 / CT2X_BF16.
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <vector>

#include "ctranslate2/allocator.h"
#include "ctranslate2/ops/ops.h"
#include "ctranslate2/primitives.h"

#include "ct2_mi355.h"

namespace ctranslate2 {

  static inline void check(int rc) {  // CUDA_CHECK equivalent (src/cuda/utils.h:68-74)
    if (rc != 0)
      throw std::runtime_error(ct2x_last_error());
  }

  static inline void* stream() {  // cuda::get_cuda_stream() (src/cuda/utils.cc:69-125)
    return ct2x_thread_stream();
  }

  // Scratch device memory of the composed (non hot path) forms below; stream-ordered allocator.
  struct Scratch {
    explicit Scratch(size_t bytes) { check(ct2x_malloc(&ptr, bytes ? bytes : 1)); }
    ~Scratch() { (void)ct2x_free(ptr); }
    float* f() const { return static_cast<float*>(ptr); }
    void* ptr = nullptr;
  };

  namespace ops {

    // include/ctranslate2/ops/quantize.h:27-30  <-  src/ops/quantize_gpu.cu:87-103
    template<>
    void Quantize::quantize<Device::CUDA, float, int8_t>(const StorageView& input,
                                                         StorageView& output,
                                                         StorageView& scale) const {
      if (_shift_to_uint8)
        throw std::invalid_argument("Shift to uin8_t is not defined on CUDA");
      const dim_t depth = input.dim(-1);
      // bit 1: round like the reference's CPU path (src/cpu/kernels.cc:592-604: fma(x, scale, 128),
      // round, cvtt, then the u8s8 GEMM's -128 compensation = an s8 operand), so that Device::CUDA
      // results equal Device::CPU results (ct2_mi355.h, ct2x_quantize_s8)
      // CT2X_S8_ACTIVATIONS=1 (the engine's switch of the same name) selects the reference's GPU
      // arithmetic instead - rint(x * scale) / truncation toward zero, src/ops/quantize_gpu.cu:
      // 36-57 - which is what the reference's own op tests expect of Device::CUDA
      static const bool gpu_form = [] {
        const char* e = std::getenv("CT2X_S8_ACTIVATIONS");
        return e != nullptr && e[0] == '1';
      }();
      check(ct2x_quantize_s8(input.data<float>(), CT2X_F32, scale.size(), depth,
                             (_round_before_cast ? 1 : 0) | (gpu_form ? 0 : 2),
                             output.data<int8_t>(), scale.data<float>(), stream()));
    }

    // include/ctranslate2/ops/dequantize.h:27-30  <-  src/ops/dequantize_gpu.cu:17-28
    template<>
    void Dequantize::dequantize<Device::CUDA, int8_t, float>(const StorageView& input,
                                                             const StorageView& scale,
                                                             StorageView& output) const {
      const dim_t depth = input.dim(-1);
      check(ct2x_dequantize_rows(input.data<int8_t>(), scale.data<float>(), input.size() / depth,
                                 depth, output.data<float>(), CT2X_F32, stream()));
    }

    // include/ctranslate2/ops/dequantize.h:32-39  <-  src/ops/dequantize_gpu.cu:67-121
    template<>
    void Dequantize::dequantize_gemm_output<Device::CUDA, float>(const StorageView& c,
                                                                 const StorageView& a_scale,
                                                                 const StorageView& b_scale,
                                                                 const bool transpose_a,
                                                                 const bool transpose_b,
                                                                 const StorageView* bias,
                                                                 StorageView& y) const {
      if (transpose_a || !transpose_b)
        throw std::invalid_argument("dequantize_gemm_output: only the Dense call shape (A.B^T)");
      check(ct2x_dequantize_gemm_output(c.data<int32_t>(), a_scale.data<float>(),
                                        b_scale.data<float>(),
                                        bias ? bias->data<float>() : nullptr,
                                        a_scale.size(), c.dim(-1),
                                        _activation_type ? int(*_activation_type) : CT2X_ACT_NONE,
                                        y.data<float>(), CT2X_F32, stream()));
    }

    // include/ctranslate2/ops/layer_norm.h:27-35  <-  src/ops/layer_norm_gpu.cu:174-213
    template<>
    void LayerNorm::compute<Device::CUDA, float>(const StorageView* beta,
                                                 const StorageView* gamma,
                                                 const StorageView& input,
                                                 const dim_t /*axis*/,
                                                 const dim_t outer_size,
                                                 const dim_t axis_size,
                                                 const dim_t inner_size,
                                                 StorageView& output) const {
      if (inner_size == 1 && beta && gamma) {  // layers::LayerNorm: last axis, affine
        check(ct2x_layer_norm(input.data<float>(), gamma->data<float>(), beta->data<float>(),
                              outer_size, axis_size, _epsilon, output.data<float>(), CT2X_F32,
                              stream()));
        return;
      }
      // Any axis, optional affine (src/ops/layer_norm_gpu.cu:174-213): bring the axis last with a
      // transpose, normalise with unit gamma / zero beta where none is given, transpose back.
      Scratch ones(size_t(axis_size) * sizeof (float)), zeros(size_t(axis_size) * sizeof (float));
      check(ct2x_fill(ones.ptr, 1.0, axis_size, 1, CT2X_F32, stream()));
      check(ct2x_fill(zeros.ptr, 0.0, axis_size, 1, CT2X_F32, stream()));
      const float* g = gamma ? gamma->data<float>() : ones.f();
      const float* b = beta ? beta->data<float>() : zeros.f();
      if (inner_size == 1) {
        check(ct2x_layer_norm(input.data<float>(), g, b, outer_size, axis_size, _epsilon,
                              output.data<float>(), CT2X_F32, stream()));
      } else {
        Scratch t_in(size_t(input.size()) * sizeof (float)), t_out(size_t(input.size()) * sizeof (float));
        const int64_t dims[3] = {outer_size, axis_size, inner_size}, perm[3] = {0, 2, 1};
        check(ct2x_transpose(input.data<float>(), dims, perm, 3, t_in.ptr, CT2X_F32, stream()));
        check(ct2x_layer_norm(t_in.f(), g, b, outer_size * inner_size, axis_size, _epsilon, t_out.f(),
                              CT2X_F32, stream()));
        const int64_t back[3] = {outer_size, inner_size, axis_size};
        check(ct2x_transpose(t_out.ptr, back, perm, 3, output.data<float>(), CT2X_F32, stream()));
      }
      check(ct2x_stream_synchronize(stream()));  // the scratch buffers go out of scope
    }

    // include/ctranslate2/ops/rms_norm.h:17-20  <-  src/ops/rms_norm_gpu.cu:20-63
    template<>
    void RMSNorm::compute<Device::CUDA, float>(const StorageView& gamma,
                                               const StorageView& input,
                                               StorageView& output) const {
      const dim_t depth = input.dim(-1);
      check(ct2x_rms_norm(input.data<float>(), gamma.data<float>(), input.size() / depth, depth,
                          _epsilon, _use_residual ? 1 : 0, output.data<float>(), CT2X_F32,
                          stream()));
    }

    // include/ctranslate2/ops/softmax.h:19-20  <-  src/ops/softmax_gpu.cu:196-272
    template<>
    void SoftMax::compute<Device::CUDA, float>(const StorageView& input,
                                               const StorageView* lengths,
                                               StorageView& output) const {
      const dim_t depth = input.dim(-1);
      check(ct2x_softmax(input.data<float>(), lengths ? lengths->data<int32_t>() : nullptr,
                         input.size() / depth, depth, _log ? 1 : 0, output.data<float>(), CT2X_F32,
                         stream()));
    }

    // include/ctranslate2/ops/rotary.h:22-27  <-  src/ops/rotary_gpu.cu:28-85
    template<>
    void Rotary::compute<Device::CUDA, float>(const StorageView& input,
                                              const StorageView& sin,
                                              const StorageView& cos,
                                              StorageView& output,
                                              bool is_transpose) const {
      if (!is_transpose)
        throw std::invalid_argument("Rotary: only the [batch, heads, time, depth] layout");
      const dim_t depth = input.dim(-1);
      const dim_t time = input.dim(-2);
      const dim_t ndims = _ndims == 0 ? depth : _ndims;
      check(ct2x_rotary(input.data<float>(), sin.data<float>(), cos.data<float>(),
                        input.size() / (time * depth), time, depth, ndims, _interleave ? 1 : 0,
                        output.data<float>(), stream()));
    }

    // include/ctranslate2/ops/topk.h:16-19  <-  src/ops/topk_gpu.cu:28-54
    template<>
    void TopK::compute<Device::CUDA, float, int32_t>(const StorageView& x,
                                                     StorageView& values,
                                                     StorageView& indices) const {
      const dim_t depth = x.dim(-1);
      check(ct2x_topk(x.data<float>(), CT2X_F32, x.size() / depth, depth, _k,
                      values.data<float>(), indices.data<int32_t>(), stream()));
    }

    // include/ctranslate2/ops/gather.h:19-24  <-  src/ops/gather_gpu.cu:53-91 (axis 0 form).
    // Rows are moved as bytes, so every payload type of the dispatch shares one device entry.
#define CT2X_GATHER_STUB(T)                                                                  \
    template<>                                                                               \
    void Gather::compute<Device::CUDA, T>(const StorageView& data,                           \
                                          const StorageView& input,                          \
                                          const dim_t axis,                                  \
                                          const dim_t batch_dims,                            \
                                          StorageView& output) const {                       \
      if (axis == 0 && batch_dims == 0) {                                                    \
        const dim_t row_size = data.size() / data.dim(0);                                    \
        check(ct2x_gather_rows(data.data<T>(), input.data<int32_t>(), input.size(),          \
                               row_size * dim_t(sizeof (T)), output.data<T>(), stream()));   \
        return;                                                                              \
      }                                                                                      \
      /* gather along the first non-batch dimension (src/ops/gather_gpu.cu:53-91): the same row   \
       * gather on data viewed as [batch * T, inner] with ids offset by batch * T (the offsets  \
       * are added on the host: this form is not on the decoding hot path) */               \
      if (axis != batch_dims)                                                                \
        throw std::invalid_argument("Gather: the axis must be the first non-batch dimension"); \
      dim_t batch = 1, inner = 1;                                                            \
      for (dim_t i = 0; i < axis; ++i) batch *= data.dim(i);                                 \
      for (dim_t i = axis + 1; i < data.rank(); ++i) inner *= data.dim(i);                   \
      const dim_t time = data.dim(axis), per_batch = input.size() / batch;                   \
      std::vector<int32_t> ids(input.size());                                                \
      check(ct2x_copy_to_host(input.data<int32_t>(), ids.data(), ids.size() * 4, stream())); \
      for (dim_t b = 0; b < batch; ++b)                                                      \
        for (dim_t j = 0; j < per_batch; ++j) ids[b * per_batch + j] += int32_t(b * time);   \
      void* d_ids = nullptr;                                                                 \
      check(ct2x_malloc(&d_ids, ids.size() * 4));                                            \
      check(ct2x_copy_to_device(ids.data(), d_ids, ids.size() * 4, stream()));               \
      check(ct2x_stream_synchronize(stream()));                                              \
      const int rc = ct2x_gather_rows(data.data<T>(), static_cast<const int32_t*>(d_ids),    \
                                      dim_t(ids.size()), inner * dim_t(sizeof (T)),          \
                                      output.data<T>(), stream());                           \
      check(ct2x_free(d_ids));                                                               \
      check(rc);                                                                             \
    }
    CT2X_GATHER_STUB(float)
    CT2X_GATHER_STUB(int8_t)
    CT2X_GATHER_STUB(int16_t)
    CT2X_GATHER_STUB(int32_t)
    CT2X_GATHER_STUB(float16_t)   // rows are moved as bytes
    CT2X_GATHER_STUB(bfloat16_t)
#undef CT2X_GATHER_STUB

    // include/ctranslate2/ops/bias_add.h:20-24  <-  src/ops/bias_add_gpu.cu:46-102 (the CPU body,
    // src/ops/bias_add_cpu.cc:6-34, composes the same three steps): broadcast add of the bias,
    // optional residual, optional activation - here over the element-wise entry points.
    template<>
    void BiasAdd::compute<Device::CUDA, float>(const StorageView& value,
                                               const StorageView& bias,
                                               StorageView& output,
                                               const StorageView* residual) const {
      if (_axis == -1 || _axis == value.rank() - 1) {
        check(ct2x_broadcast(CT2X_OP_ADD, 0, bias.data<float>(), value.data<float>(),
                             output.data<float>(), bias.size(), value.size(), 0, CT2X_F32, stream()));
      } else {
        const dim_t axis = _axis < 0 ? value.rank() + _axis : _axis;
        dim_t width = 1;
        for (dim_t i = axis + 1; i < value.rank(); ++i)
          width *= value.dim(i);
        check(ct2x_broadcast(CT2X_OP_ADD, 2, bias.data<float>(), value.data<float>(),
                             output.data<float>(), bias.size(), value.size(), width, CT2X_F32,
                             stream()));
      }
      if (residual)
        check(ct2x_binary(CT2X_OP_ADD, residual->data<float>(), output.data<float>(),
                          output.data<float>(), output.size(), CT2X_F32, stream()));
      if (_activation_type)
        check(ct2x_unary(int(*_activation_type), output.data<float>(), output.data<float>(),
                         output.size(), stream()));
    }

  }

  // include/ctranslate2/primitives.h (gemm)  <-  src/cuda/primitives.cu:571-597
  template<>
  template<>
  void primitives<Device::CUDA>::gemm(bool, bool,
                                      bool transpose_a, bool transpose_b,
                                      dim_t m, dim_t n, dim_t k,
                                      float alpha,
                                      const int8_t* a, dim_t lda,
                                      const int8_t* b, dim_t ldb,
                                      float beta,
                                      int32_t* c, dim_t ldc,
                                      const int32_t*) {
    if (alpha != 1 || beta != 0 || ldc != n)
      throw std::invalid_argument("int8 GEMM: alpha = 1, beta = 0 and a dense output only");
    if (!transpose_a && transpose_b && lda == k && ldb == k) {  // the Dense call shape: A.B^T
      check(ct2x_gemm_s8s8s32(a, b, m, n, k, c, stream()));
      return;
    }
    // Other operand layouts (the reference's op tests use A.B): brought to A.B^T through the
    // host - not a path any layer takes.
    std::vector<int8_t> ha(size_t(m) * k), hb(size_t(n) * k);
    {
      const dim_t a_rows = transpose_a ? k : m, b_rows = transpose_b ? n : k;
      std::vector<int8_t> ra(size_t(a_rows) * lda), rb(size_t(b_rows) * ldb);
      check(ct2x_copy_to_host(a, ra.data(), ra.size(), stream()));
      check(ct2x_copy_to_host(b, rb.data(), rb.size(), stream()));
      for (dim_t i = 0; i < m; ++i)
        for (dim_t t = 0; t < k; ++t)
          ha[i * k + t] = transpose_a ? ra[t * lda + i] : ra[i * lda + t];
      for (dim_t j = 0; j < n; ++j)
        for (dim_t t = 0; t < k; ++t)
          hb[j * k + t] = transpose_b ? rb[j * ldb + t] : rb[t * ldb + j];
    }
    void *da = nullptr, *db = nullptr;
    check(ct2x_malloc(&da, ha.size()));
    check(ct2x_malloc(&db, hb.size()));
    check(ct2x_copy_to_device(ha.data(), da, ha.size(), stream()));
    check(ct2x_copy_to_device(hb.data(), db, hb.size(), stream()));
    check(ct2x_stream_synchronize(stream()));
    const int rc = ct2x_gemm_s8s8s32(static_cast<const int8_t*>(da), static_cast<const int8_t*>(db),
                                     m, n, k, c, stream());
    check(ct2x_stream_synchronize(stream()));
    check(ct2x_free(da));
    check(ct2x_free(db));
    check(rc);
  }

  // include/ctranslate2/primitives.h (gemm<float, float>)  <-  src/cuda/primitives.cu:485-520:
  // the shapes ops::Gemm produces for layers::Dense (row-major, no padding, alpha 1, beta 0).
  template<>
  template<>
  void primitives<Device::CUDA>::gemm(bool, bool,
                                      bool transpose_a, bool transpose_b,
                                      dim_t m, dim_t n, dim_t k,
                                      float alpha,
                                      const float* a, dim_t lda,
                                      const float* b, dim_t ldb,
                                      float beta,
                                      float* c, dim_t ldc,
                                      const float*) {
    if (lda != (transpose_a ? m : k) || ldb != (transpose_b ? k : n) || ldc != n)
      throw std::invalid_argument("float GEMM: contiguous row-major operands only");
    if (!transpose_a && alpha == 1 && beta == 0) {  // what ops::Gemm produces for layers::Dense
      check(ct2x_gemm_f32(a, b, m, n, k, transpose_b ? 1 : 0, nullptr, nullptr, CT2X_ACT_NONE, c,
                          stream()));
      return;
    }
    // General form C = alpha * op(A) . op(B) + beta * C (src/cuda/primitives.cu:485-520, cublasSgemm)
    // composed from device entry points: transpose A when asked, product into scratch, then the
    // two scalings and the sum.
    const float* a_rm = a;
    std::unique_ptr<Scratch> at;
    if (transpose_a) {
      at.reset(new Scratch(size_t(m) * k * sizeof (float)));
      const int64_t dims[2] = {k, m}, perm[2] = {1, 0};
      check(ct2x_transpose(a, dims, perm, 2, at->ptr, CT2X_F32, stream()));
      a_rm = at->f();
    }
    Scratch prod(size_t(m) * n * sizeof (float));
    check(ct2x_gemm_f32(a_rm, b, m, n, k, transpose_b ? 1 : 0, nullptr, nullptr, CT2X_ACT_NONE,
                        prod.f(), stream()));
    if (alpha != 1)
      check(ct2x_binary_scalar(CT2X_OP_MUL, alpha, prod.f(), prod.f(), m * n, CT2X_F32, stream()));
    if (beta == 0) {
      check(ct2x_copy_device(prod.ptr, c, size_t(m) * n * sizeof (float), stream()));
    } else {
      if (beta != 1)
        check(ct2x_binary_scalar(CT2X_OP_MUL, beta, c, c, m * n, CT2X_F32, stream()));
      check(ct2x_binary(CT2X_OP_ADD, prod.f(), c, c, m * n, CT2X_F32, stream()));
    }
    check(ct2x_stream_synchronize(stream()));  // the scratch buffers go out of scope
  }

  // include/ctranslate2/primitives.h:224-232  <-  src/cuda/primitives.cu:599-683, as ops::MatMul
  // calls it (src/ops/matmul.cc:62-73): contiguous batches, beta 0.
  template<>
  template<>
  void primitives<Device::CUDA>::gemm_batch_strided(bool transpose_a, bool transpose_b,
                                                    dim_t m, dim_t n, dim_t k,
                                                    float alpha,
                                                    const float* a, dim_t lda, dim_t stridea,
                                                    const float* b, dim_t ldb, dim_t strideb,
                                                    float beta,
                                                    float* c, dim_t ldc, dim_t stridec,
                                                    dim_t batch_size) {
    if (transpose_a || beta != 0 || lda != k || ldb != (transpose_b ? k : n) || ldc != n
        || stridea != m * k || strideb != n * k || stridec != m * n)
      throw std::invalid_argument("batched float GEMM: only contiguous batches of A.B / A.B^T, beta=0");
    check(ct2x_matmul_f32(a, b, batch_size, m, n, k, transpose_b ? 1 : 0, alpha, c, stream()));
  }

  // include/ctranslate2/primitives.h:13-14  <-  src/cuda/primitives.cu:60-65: scalar read, the
  // copy has landed when the call returns.
#define CT2X_AT_STUB(T)                                                    \
  template<> template<>                                                    \
  T primitives<Device::CUDA>::at(const T* x, dim_t index) {                \
    T value = T();                                                         \
    check(ct2x_copy_to_host(x + index, &value, sizeof (T), stream()));     \
    return value;                                                          \
  }
  CT2X_AT_STUB(float)
  CT2X_AT_STUB(int8_t)
  CT2X_AT_STUB(int16_t)
  CT2X_AT_STUB(int32_t)
#undef CT2X_AT_STUB

  // include/ctranslate2/primitives.h:145-151  <-  src/cuda/primitives.cu (penalize_previous_tokens):
  // RepetitionPenalty has already gathered previous_scores; the one-launch processor redoes the
  // gather itself, so the stub passes the scores and the ids and leaves the other processors off.
  template<>
  template<>
  void primitives<Device::CUDA>::penalize_previous_tokens(float* scores,
                                                          const float*,
                                                          const int32_t* previous_ids,
                                                          float penalty,
                                                          dim_t batch_size,
                                                          dim_t length,
                                                          dim_t vocabulary_size) {
    check(ct2x_process_logits(scores, batch_size, vocabulary_size, previous_ids,
                              static_cast<int32_t>(length), penalty, 0, nullptr, nullptr, 0,
                              stream()));
  }

  // include/ctranslate2/primitives.h:24 (copy) and :235-239 (cross_device_primitives)
  //   <-  src/cuda/primitives.cu:97-101, 685-754. Byte copies: one macro over the payload types.
#define CT2X_COPY_STUBS(T)                                                                        \
  template<> template<>                                                                           \
  void primitives<Device::CUDA>::copy(const T* x, T* y, dim_t size) {                             \
    check(ct2x_copy_device(x, y, size * sizeof (T), stream()));                                   \
  }                                                                                               \
  template<> template<>                                                                           \
  void cross_device_primitives<Device::CUDA, Device::CPU>::copy(const T* x, T* y, dim_t size) {   \
    check(ct2x_copy_to_host(x, y, size * sizeof (T), stream())); /* returns when landed */        \
  }                                                                                               \
  template<> template<>                                                                           \
  void cross_device_primitives<Device::CPU, Device::CUDA>::copy(const T* x, T* y, dim_t size) {   \
    check(ct2x_copy_to_device(x, y, size * sizeof (T), stream()));                                \
  }
  CT2X_COPY_STUBS(float)
  CT2X_COPY_STUBS(int8_t)
  CT2X_COPY_STUBS(int16_t)
  CT2X_COPY_STUBS(int32_t)
#undef CT2X_COPY_STUBS

  // ---- element-wise / layout members (include/ctranslate2/primitives.h:16-185  <-
  //      src/cuda/primitives.cu:103-483): one device entry per family, see primitives.hip -----
  template <typename T> struct ct2x_type;
  template<> struct ct2x_type<float> { static constexpr int id = CT2X_F32; };
  template<> struct ct2x_type<int8_t> { static constexpr int id = CT2X_I8; };
  template<> struct ct2x_type<int16_t> { static constexpr int id = CT2X_I16; };
  template<> struct ct2x_type<int32_t> { static constexpr int id = CT2X_I32; };

#define CT2X_ARITH_STUBS(T)                                                                          \
  template<> template<>                                                                              \
  void primitives<Device::CUDA>::fill(T* x, T a, dim_t size) {                                       \
    check(ct2x_fill(x, double(a), size, 1, ct2x_type<T>::id, stream()));                             \
  }                                                                                                  \
  template<> template<>                                                                              \
  void primitives<Device::CUDA>::strided_fill(T* x, T a, dim_t inc_x, dim_t size) {                  \
    check(ct2x_fill(x, double(a), size, inc_x, ct2x_type<T>::id, stream()));                         \
  }                                                                                                  \
  template<> template<>                                                                              \
  void primitives<Device::CUDA>::indexed_fill(T* x, T a, const int32_t* indices, dim_t n) {          \
    check(ct2x_indexed_fill(x, double(a), indices, n, ct2x_type<T>::id, stream()));                  \
  }                                                                                                  \
  template<> template<>                                                                              \
  void primitives<Device::CUDA>::add(T a, const T* x, T* y, dim_t size) {                            \
    check(ct2x_binary_scalar(CT2X_OP_ADD, double(a), x, y, size, ct2x_type<T>::id, stream()));       \
  }                                                                                                  \
  template<> template<>                                                                              \
  void primitives<Device::CUDA>::add(const T* a, const T* b, T* c, dim_t size) {                     \
    check(ct2x_binary(CT2X_OP_ADD, a, b, c, size, ct2x_type<T>::id, stream()));                      \
  }                                                                                                  \
  template<> template<>                                                                              \
  void primitives<Device::CUDA>::sub(const T* a, const T* b, T* c, dim_t size) {                     \
    check(ct2x_binary(CT2X_OP_SUB, a, b, c, size, ct2x_type<T>::id, stream()));                      \
  }                                                                                                  \
  template<> template<>                                                                              \
  void primitives<Device::CUDA>::mul(T a, const T* x, T* y, dim_t size) {                            \
    check(ct2x_binary_scalar(CT2X_OP_MUL, double(a), x, y, size, ct2x_type<T>::id, stream()));       \
  }                                                                                                  \
  template<> template<>                                                                              \
  void primitives<Device::CUDA>::mul(const T* a, const T* b, T* c, dim_t size) {                     \
    check(ct2x_binary(CT2X_OP_MUL, a, b, c, size, ct2x_type<T>::id, stream()));                      \
  }                                                                                                  \
  template<> template<>                                                                              \
  void primitives<Device::CUDA>::min(T a, const T* x, T* y, dim_t size) {                            \
    check(ct2x_binary_scalar(CT2X_OP_MIN, double(a), x, y, size, ct2x_type<T>::id, stream()));       \
  }                                                                                                  \
  template<> template<>                                                                              \
  void primitives<Device::CUDA>::min(const T* a, const T* b, T* c, dim_t size) {                     \
    check(ct2x_binary(CT2X_OP_MIN, a, b, c, size, ct2x_type<T>::id, stream()));                      \
  }                                                                                                  \
  template<> template<>                                                                              \
  void primitives<Device::CUDA>::max(T a, const T* x, T* y, dim_t size) {                            \
    check(ct2x_binary_scalar(CT2X_OP_MAX, double(a), x, y, size, ct2x_type<T>::id, stream()));       \
  }                                                                                                  \
  template<> template<>                                                                              \
  void primitives<Device::CUDA>::max(const T* a, const T* b, T* c, dim_t size) {                     \
    check(ct2x_binary(CT2X_OP_MAX, a, b, c, size, ct2x_type<T>::id, stream()));                      \
  }                                                                                                  \
  template<> template<>                                                                              \
  void primitives<Device::CUDA>::add_batch_broadcast(const T* a, const T* b, T* c,                   \
                                                     dim_t a_size, dim_t b_size) {                   \
    check(ct2x_broadcast(CT2X_OP_ADD, 0, a, b, c, a_size, b_size, 0, ct2x_type<T>::id, stream()));   \
  }                                                                                                  \
  template<> template<>                                                                              \
  void primitives<Device::CUDA>::add_depth_broadcast(const T* a, const T* b, T* c,                   \
                                                     dim_t a_size, dim_t b_size) {                   \
    check(ct2x_broadcast(CT2X_OP_ADD, 1, a, b, c, a_size, b_size, 0, ct2x_type<T>::id, stream()));   \
  }                                                                                                  \
  template<> template<>                                                                              \
  void primitives<Device::CUDA>::add_block_broadcast(const T* a, const T* b, T* c, dim_t block,      \
                                                     dim_t a_size, dim_t b_size) {                   \
    check(ct2x_broadcast(CT2X_OP_ADD, 2, a, b, c, a_size, b_size, block, ct2x_type<T>::id, stream())); \
  }                                                                                                  \
  template<> template<>                                                                              \
  void primitives<Device::CUDA>::mul_batch_broadcast(const T* a, const T* b, T* c,                   \
                                                     dim_t a_size, dim_t b_size) {                   \
    check(ct2x_broadcast(CT2X_OP_MUL, 0, a, b, c, a_size, b_size, 0, ct2x_type<T>::id, stream()));   \
  }                                                                                                  \
  template<> template<>                                                                              \
  void primitives<Device::CUDA>::transpose_2d(const T* a, const dim_t* dims, T* b) {                 \
    const int64_t perm[2] = {1, 0};                                                                  \
    check(ct2x_transpose(a, dims, perm, 2, b, ct2x_type<T>::id, stream()));                          \
  }                                                                                                  \
  template<> template<>                                                                              \
  void primitives<Device::CUDA>::transpose_3d(const T* a, const dim_t* dims, const dim_t* perm,      \
                                              T* b) {                                                \
    check(ct2x_transpose(a, dims, perm, 3, b, ct2x_type<T>::id, stream()));                          \
  }                                                                                                  \
  template<> template<>                                                                              \
  void primitives<Device::CUDA>::transpose_4d(const T* a, const dim_t* dims, const dim_t* perm,      \
                                              T* b) {                                                \
    check(ct2x_transpose(a, dims, perm, 4, b, ct2x_type<T>::id, stream()));                          \
  }
  CT2X_ARITH_STUBS(float)
  CT2X_ARITH_STUBS(int8_t)
  CT2X_ARITH_STUBS(int16_t)
  CT2X_ARITH_STUBS(int32_t)
#undef CT2X_ARITH_STUBS

#define CT2X_UNARY_STUB(NAME, FN)                                                  \
  template<> template<>                                                            \
  void primitives<Device::CUDA>::NAME(const float* x, float* y, dim_t size) {      \
    check(ct2x_unary(FN, x, y, size, stream()));                                   \
  }
  CT2X_UNARY_STUB(relu, CT2X_ACT_RELU)
  CT2X_UNARY_STUB(gelu_tanh, CT2X_ACT_GELU_TANH)
  CT2X_UNARY_STUB(swish, CT2X_ACT_SWISH)
  CT2X_UNARY_STUB(gelu, CT2X_ACT_GELU)
  CT2X_UNARY_STUB(gelu_sigmoid, CT2X_ACT_GELU_SIGMOID)
  CT2X_UNARY_STUB(tanh, CT2X_ACT_TANH)
  CT2X_UNARY_STUB(sigmoid, CT2X_ACT_SIGMOID)
  CT2X_UNARY_STUB(exp, 7)
  CT2X_UNARY_STUB(log, 8)
  CT2X_UNARY_STUB(sin, 9)
  CT2X_UNARY_STUB(cos, 10)
#undef CT2X_UNARY_STUB

  template<>
  void primitives<Device::CUDA>::prepare_length_mask(const int32_t* lengths,
                                                     dim_t batch_size,
                                                     dim_t num_heads,
                                                     dim_t num_queries,
                                                     bool mask_future,
                                                     bool multi_query,
                                                     int32_t* mask) {
    check(ct2x_prepare_length_mask(lengths, batch_size, num_heads, num_queries, mask_future ? 1 : 0,
                                   multi_query ? 1 : 0, mask, stream()));
  }

  // include/ctranslate2/allocator.h:24-26  <-  src/cuda/allocator.cc
  class Mi355Allocator : public Allocator {
  public:
    void* allocate(size_t size, int) override {
      void* p = nullptr;
      check(ct2x_malloc(&p, size));
      return p;
    }
    void free(void* ptr, int) override {
      check(ct2x_free(ptr));
    }
    void clear_cache() override {
      check(ct2x_clear_cache());
    }
  };

  template<>
  Allocator& get_allocator<Device::CUDA>() {
    static Mi355Allocator allocator;
    return allocator;
  }

}
