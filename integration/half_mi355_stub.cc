// Level-1 integration stubs, 16-bit arms: the float16_t / bfloat16_t
// specialisations of primitives<Device::CUDA> and ops::X::compute<Device::CUDA, T> that the
// reference declares next to the float ones (src/cuda/primitives.cu:756-849 instantiates them for
// __half / __nv_bfloat16). This backend's device entry points compute in float32, so an arm
//   * that only MOVES elements (copies, fills, transposes, the layout ops) forwards the 2-byte
//     payload as it is;
//   * that does arithmetic converts its operands to float32 on the device (ct2x_convert), calls the
//     float32 arm defined in ops_mi355_stub.cc / host_mi355_stub.cc, and converts the result back -
//     the rounding the reference's own 16-bit kernels apply to their fp32 intermediates.
// These arms make `compute_type = float16 / bfloat16` reachable through the reference's front-end;
// the fused engine's 16-bit compute types (16-bit weights, fp32 activations) do not use them.
//
// This file is written for this repository; it contains no reference code.
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "ctranslate2/ops/ops.h"
#include "ctranslate2/primitives.h"
#include "ctranslate2/types.h"

#include "ct2_mi355.h"

namespace ctranslate2 {

  namespace {

    inline void check16(int rc) {
      if (rc != 0)
        throw std::runtime_error(ct2x_last_error());
    }
    inline void* stream16() { return ct2x_thread_stream(); }

    template <typename T> struct Ct2xType;
    template<> struct Ct2xType<float16_t> { static constexpr int value = CT2X_F16; };
    template<> struct Ct2xType<bfloat16_t> { static constexpr int value = CT2X_BF16; };

    template <typename T>
    inline int16_t bits_of(T v) {
      int16_t b;
      static_assert(sizeof (T) == 2, "16-bit float types");
      std::memcpy(&b, &v, 2);
      return b;
    }

    struct Buffer {
      explicit Buffer(size_t bytes) { check16(ct2x_malloc(&ptr, bytes ? bytes : 1)); }
      ~Buffer() { (void)ct2x_free(ptr); }  // stream-ordered: safe behind queued work on this stream
      Buffer(const Buffer&) = delete;
      void* ptr = nullptr;
    };

    // float32 copy of a 16-bit device array
    template <typename T>
    struct In32 {
      In32(const T* x, dim_t n) : buf(size_t(n) * 4) {
        check16(ct2x_convert(x, Ct2xType<T>::value, buf.ptr, CT2X_F32, n, stream16()));
      }
      const float* get() const { return static_cast<const float*>(buf.ptr); }
      Buffer buf;
    };
    // float32 scratch whose content is rounded into a 16-bit device array by commit()
    template <typename T>
    struct Out32 {
      Out32(T* y, dim_t n) : dst(y), size(n), buf(size_t(n) * 4) {}
      float* get() const { return static_cast<float*>(buf.ptr); }
      void commit() {
        check16(ct2x_convert(buf.ptr, CT2X_F32, dst, Ct2xType<T>::value, size, stream16()));
      }
      T* dst;
      dim_t size;
      Buffer buf;
    };

    // StorageView helpers for the op-level arms
    inline StorageView f32_like(const StorageView& x) {
      return StorageView(x.shape(), DataType::FLOAT32, Device::CUDA);
    }
    template <typename T>
    inline StorageView to_f32(const StorageView& x) {
      StorageView y = f32_like(x);
      if (x.size() > 0)
        check16(ct2x_convert(x.data<T>(), Ct2xType<T>::value, y.data<float>(), CT2X_F32, x.size(),
                             stream16()));
      return y;
    }
    template <typename T>
    inline void from_f32(const StorageView& x32, StorageView& y) {
      if (x32.size() > 0)
        check16(ct2x_convert(x32.data<float>(), CT2X_F32, y.data<T>(), Ct2xType<T>::value, x32.size(),
                             stream16()));
    }

    [[noreturn]] void not_provided16(const char* op) {
      throw std::runtime_error(std::string(op) + " is not provided by the MI355X backend "
                               "(outside the quantized decoder hot path, SURVEY section 8)");
    }

  }

  // ---- primitives<Device::CUDA>, 16-bit element types ------------------------------------------

  template<> template<>
  void primitives<Device::CUDA>::convert(const float* x, float16_t* y, dim_t size) {
    check16(ct2x_convert(x, CT2X_F32, y, CT2X_F16, size, stream16()));
  }
  template<> template<>
  void primitives<Device::CUDA>::convert(const float* x, bfloat16_t* y, dim_t size) {
    check16(ct2x_convert(x, CT2X_F32, y, CT2X_BF16, size, stream16()));
  }
  template<> template<>
  void primitives<Device::CUDA>::convert(const float16_t* x, float* y, dim_t size) {
    check16(ct2x_convert(x, CT2X_F16, y, CT2X_F32, size, stream16()));
  }
  template<> template<>
  void primitives<Device::CUDA>::convert(const bfloat16_t* x, float* y, dim_t size) {
    check16(ct2x_convert(x, CT2X_BF16, y, CT2X_F32, size, stream16()));
  }
  template<> template<>
  void primitives<Device::CUDA>::convert(const float16_t* x, bfloat16_t* y, dim_t size) {
    check16(ct2x_convert(x, CT2X_F16, y, CT2X_BF16, size, stream16()));
  }
  template<> template<>
  void primitives<Device::CUDA>::convert(const bfloat16_t* x, float16_t* y, dim_t size) {
    check16(ct2x_convert(x, CT2X_BF16, y, CT2X_F16, size, stream16()));
  }

#define CT2X_HALF_PRIMITIVES(T)                                                                     \
  /* element movers: the 2-byte payload as it is */                                                 \
  template<> template<>                                                                             \
  void primitives<Device::CUDA>::copy(const T* x, T* y, dim_t size) {                               \
    check16(ct2x_copy_device(x, y, size_t(size) * 2, stream16()));                                  \
  }                                                                                                 \
  template<> template<>                                                                             \
  void primitives<Device::CUDA>::fill(T* x, T a, dim_t size) {                                      \
    check16(ct2x_fill(x, double(bits_of(a)), size, 1, CT2X_I16, stream16()));                       \
  }                                                                                                 \
  template<> template<>                                                                             \
  void primitives<Device::CUDA>::indexed_fill(T* x, T a, const int32_t* indices, dim_t n) {         \
    check16(ct2x_indexed_fill(x, double(bits_of(a)), indices, n, CT2X_I16, stream16()));            \
  }                                                                                                 \
  template<> template<>                                                                             \
  void primitives<Device::CUDA>::transpose_2d(const T* a, const dim_t* dims, T* b) {                \
    const int64_t d[2] = {dims[0], dims[1]}, perm[2] = {1, 0};                                      \
    check16(ct2x_transpose(a, d, perm, 2, b, CT2X_I16, stream16()));                                \
  }                                                                                                 \
  template<> template<>                                                                             \
  void primitives<Device::CUDA>::transpose_3d(const T* a, const dim_t* dims, const dim_t* perm,     \
                                              T* b) {                                               \
    const int64_t d[3] = {dims[0], dims[1], dims[2]}, p[3] = {perm[0], perm[1], perm[2]};           \
    check16(ct2x_transpose(a, d, p, 3, b, CT2X_I16, stream16()));                                   \
  }                                                                                                 \
  template<> template<>                                                                             \
  void primitives<Device::CUDA>::transpose_4d(const T* a, const dim_t* dims, const dim_t* perm,     \
                                              T* b) {                                               \
    const int64_t d[4] = {dims[0], dims[1], dims[2], dims[3]};                                      \
    const int64_t p[4] = {perm[0], perm[1], perm[2], perm[3]};                                      \
    check16(ct2x_transpose(a, d, p, 4, b, CT2X_I16, stream16()));                                   \
  }                                                                                                 \
  template<> template<>                                                                             \
  void cross_device_primitives<Device::CPU, Device::CUDA>::copy(const T* x, T* y, dim_t size) {     \
    check16(ct2x_copy_to_device(x, y, size_t(size) * 2, stream16()));                               \
  }                                                                                                 \
  template<> template<>                                                                             \
  void cross_device_primitives<Device::CUDA, Device::CPU>::copy(const T* x, T* y, dim_t size) {     \
    check16(ct2x_copy_to_host(x, y, size_t(size) * 2, stream16()));                                 \
  }                                                                                                 \
  template<> template<>                                                                             \
  T primitives<Device::CUDA>::at(const T* x, dim_t index) {                                         \
    T value = T();                                                                                  \
    check16(ct2x_copy_to_host(x + index, &value, 2, stream16()));                                   \
    return value;                                                                                   \
  }                                                                                                 \
  /* arithmetic: through float32 */                                                                 \
  template<> template<>                                                                             \
  T primitives<Device::CUDA>::max(const T* x, dim_t size) {                                         \
    In32<T> a(x, size);                                                                             \
    return T(primitives<Device::CUDA>::max(a.get(), size));                                         \
  }                                                                                                 \
  template<> template<>                                                                             \
  float primitives<Device::CUDA>::logsumexp(const T* x, dim_t size) {                               \
    In32<T> a(x, size);                                                                             \
    return primitives<Device::CUDA>::logsumexp(a.get(), size);                                      \
  }                                                                                                 \
  template<> template<>                                                                             \
  void primitives<Device::CUDA>::add_batch_broadcast(const T* a, const T* b, T* c, dim_t a_size,    \
                                                     dim_t b_size) {                                \
    In32<T> fa(a, a_size), fb(b, b_size);                                                           \
    Out32<T> fc(c, b_size);                                                                         \
    primitives<Device::CUDA>::add_batch_broadcast(fa.get(), fb.get(), fc.get(), a_size, b_size);    \
    fc.commit();                                                                                    \
  }                                                                                                 \
  template<> template<>                                                                             \
  void primitives<Device::CUDA>::add_depth_broadcast(const T* a, const T* b, T* c, dim_t a_size,    \
                                                     dim_t b_size) {                                \
    In32<T> fa(a, a_size), fb(b, b_size);                                                           \
    Out32<T> fc(c, b_size);                                                                         \
    primitives<Device::CUDA>::add_depth_broadcast(fa.get(), fb.get(), fc.get(), a_size, b_size);    \
    fc.commit();                                                                                    \
  }                                                                                                 \
  template<> template<>                                                                             \
  void primitives<Device::CUDA>::penalize_previous_tokens(T* scores, const T* previous_scores,      \
                                                          const int32_t* previous_ids, T penalty,   \
                                                          dim_t batch_size, dim_t length,           \
                                                          dim_t vocabulary_size) {                  \
    In32<T> fs(scores, batch_size * vocabulary_size), fp(previous_scores, batch_size * length);     \
    Out32<T> out(scores, batch_size * vocabulary_size);                                             \
    check16(ct2x_copy_device(fs.get(), out.get(), size_t(batch_size) * vocabulary_size * 4,         \
                             stream16()));                                                          \
    primitives<Device::CUDA>::penalize_previous_tokens(out.get(), fp.get(), previous_ids,           \
                                                       float(penalty), batch_size, length,          \
                                                       vocabulary_size);                            \
    out.commit();                                                                                   \
  }                                                                                                 \
  template<> template<>                                                                             \
  void primitives<Device::CUDA>::gemm(bool a_is_packed, bool b_is_packed, bool transpose_a,         \
                                      bool transpose_b, dim_t m, dim_t n, dim_t k, float alpha,     \
                                      const T* a, dim_t lda, const T* b, dim_t ldb, float beta,     \
                                      T* c, dim_t ldc, const T*) {                                  \
    if (!transpose_a && transpose_b && lda == k && ldb == k && ldc == n && beta == 0 &&             \
        alpha == 1.f && k % 8 == 0) {                                                               \
      /* the layers::Dense layout (weight [n, k]): the 16-bit weight goes to the matrix cores as    \
         it is (MFMA, fp32 accumulation, src/cuda/primitives.cu:519-569); the activation was 16-bit \
         already, so its round trip through fp32 loses nothing */                                   \
      In32<T> fa(a, m * lda);                                                                       \
      Out32<T> fc(c, m * ldc);                                                                      \
      check16(ct2x_gemm_f16w(fa.get(), b, Ct2xType<T>::value, m, n, k, nullptr, nullptr, -1,        \
                             fc.get(), stream16()));                                                \
      fc.commit();                                                                                  \
      return;                                                                                       \
    }                                                                                               \
    In32<T> fa(a, (transpose_a ? k : m) * lda), fb(b, (transpose_b ? n : k) * ldb);                 \
    Out32<T> fc(c, m * ldc);                                                                        \
    if (beta != 0)                                                                                  \
      check16(ct2x_convert(c, Ct2xType<T>::value, fc.get(), CT2X_F32, m * ldc, stream16()));        \
    primitives<Device::CUDA>::gemm(a_is_packed, b_is_packed, transpose_a, transpose_b, m, n, k,     \
                                   alpha, fa.get(), lda, fb.get(), ldb, beta, fc.get(), ldc,        \
                                   static_cast<const float*>(nullptr));                             \
    fc.commit();                                                                                    \
  }                                                                                                 \
  template<> template<>                                                                             \
  void primitives<Device::CUDA>::gemm_batch_strided(bool transpose_a, bool transpose_b, dim_t m,    \
                                                    dim_t n, dim_t k, float alpha, const T* a,      \
                                                    dim_t lda, dim_t stridea, const T* b, dim_t ldb, \
                                                    dim_t strideb, float beta, T* c, dim_t ldc,     \
                                                    dim_t stridec, dim_t batch_size) {              \
    In32<T> fa(a, batch_size * stridea), fb(b, batch_size * strideb);                               \
    Out32<T> fc(c, batch_size * stridec);                                                           \
    if (beta != 0)                                                                                  \
      check16(ct2x_convert(c, Ct2xType<T>::value, fc.get(), CT2X_F32, batch_size * stridec,         \
                           stream16()));                                                            \
    primitives<Device::CUDA>::gemm_batch_strided(transpose_a, transpose_b, m, n, k, alpha, fa.get(), \
                                                 lda, stridea, fb.get(), ldb, strideb, beta,        \
                                                 fc.get(), ldc, stridec, batch_size);               \
    fc.commit();                                                                                    \
  }

  CT2X_HALF_PRIMITIVES(float16_t)
  CT2X_HALF_PRIMITIVES(bfloat16_t)
#undef CT2X_HALF_PRIMITIVES

#define CT2X_HALF_BINARY(T, NAME)                                                                   \
  template<> template<>                                                                             \
  void primitives<Device::CUDA>::NAME(const T* a, const T* b, T* c, dim_t size) {                   \
    In32<T> fa(a, size), fb(b, size);                                                               \
    Out32<T> fc(c, size);                                                                           \
    primitives<Device::CUDA>::NAME(fa.get(), fb.get(), fc.get(), size);                             \
    fc.commit();                                                                                    \
  }
#define CT2X_HALF_SCALAR(T, NAME)                                                                   \
  template<> template<>                                                                             \
  void primitives<Device::CUDA>::NAME(T a, const T* x, T* y, dim_t size) {                          \
    In32<T> fx(x, size);                                                                            \
    Out32<T> fy(y, size);                                                                           \
    primitives<Device::CUDA>::NAME(float(a), fx.get(), fy.get(), size);                             \
    fy.commit();                                                                                    \
  }
#define CT2X_HALF_UNARY(T, NAME)                                                                    \
  template<> template<>                                                                             \
  void primitives<Device::CUDA>::NAME(const T* x, T* y, dim_t size) {                               \
    In32<T> fx(x, size);                                                                            \
    Out32<T> fy(y, size);                                                                           \
    primitives<Device::CUDA>::NAME(fx.get(), fy.get(), size);                                       \
    fy.commit();                                                                                    \
  }
#define CT2X_HALF_ELEMENTWISE(T)                                                                    \
  CT2X_HALF_BINARY(T, add) CT2X_HALF_BINARY(T, sub) CT2X_HALF_BINARY(T, mul)                        \
  CT2X_HALF_BINARY(T, min) CT2X_HALF_BINARY(T, max)                                                 \
  CT2X_HALF_SCALAR(T, add) CT2X_HALF_SCALAR(T, mul) CT2X_HALF_SCALAR(T, min) CT2X_HALF_SCALAR(T, max) \
  CT2X_HALF_UNARY(T, relu) CT2X_HALF_UNARY(T, gelu) CT2X_HALF_UNARY(T, gelu_tanh)                   \
  CT2X_HALF_UNARY(T, gelu_sigmoid) CT2X_HALF_UNARY(T, sigmoid) CT2X_HALF_UNARY(T, swish)            \
  CT2X_HALF_UNARY(T, tanh) CT2X_HALF_UNARY(T, log) CT2X_HALF_UNARY(T, sin) CT2X_HALF_UNARY(T, cos)

  CT2X_HALF_ELEMENTWISE(float16_t)
  CT2X_HALF_ELEMENTWISE(bfloat16_t)
#undef CT2X_HALF_ELEMENTWISE
#undef CT2X_HALF_UNARY
#undef CT2X_HALF_SCALAR
#undef CT2X_HALF_BINARY

  // ---- ops::X::compute<Device::CUDA, T>, 16-bit element types --------------------------------------
  namespace ops {

#define CT2X_HALF_OPS(T)                                                                            \
    template<>                                                                                      \
    void Quantize::quantize<Device::CUDA, T, int8_t>(const StorageView& input, StorageView& output, \
                                                     StorageView& scale) const {                    \
      StorageView in32 = to_f32<T>(input);                                                          \
      quantize<Device::CUDA, float, int8_t>(in32, output, scale);                                   \
    }                                                                                               \
    template<>                                                                                      \
    void Dequantize::dequantize<Device::CUDA, int8_t, T>(const StorageView& input,                  \
                                                         const StorageView& scale,                  \
                                                         StorageView& output) const {               \
      StorageView out32 = f32_like(output);                                                         \
      dequantize<Device::CUDA, int8_t, float>(input, scale, out32);                                 \
      from_f32<T>(out32, output);                                                                   \
    }                                                                                               \
    template<>                                                                                      \
    void Dequantize::dequantize_gemm_output<Device::CUDA, T>(const StorageView& c,                  \
                                                             const StorageView& a_scale,            \
                                                             const StorageView& b_scale,            \
                                                             const bool transpose_a,                \
                                                             const bool transpose_b,                \
                                                             const StorageView* bias,               \
                                                             StorageView& y) const {                \
      StorageView y32 = f32_like(y);                                                                \
      if (bias) {                                                                                   \
        StorageView bias32 = to_f32<T>(*bias);                                                      \
        dequantize_gemm_output<Device::CUDA, float>(c, a_scale, b_scale, transpose_a, transpose_b,  \
                                                    &bias32, y32);                                  \
      } else {                                                                                      \
        dequantize_gemm_output<Device::CUDA, float>(c, a_scale, b_scale, transpose_a, transpose_b,  \
                                                    nullptr, y32);                                  \
      }                                                                                             \
      from_f32<T>(y32, y);                                                                          \
    }                                                                                               \
    template<>                                                                                      \
    void LayerNorm::compute<Device::CUDA, T>(const StorageView* beta, const StorageView* gamma,     \
                                             const StorageView& input, const dim_t axis,            \
                                             const dim_t outer_size, const dim_t axis_size,         \
                                             const dim_t inner_size, StorageView& output) const {   \
      StorageView in32 = to_f32<T>(input), out32 = f32_like(output), b32, g32;                      \
      if (beta) b32 = to_f32<T>(*beta);                                                             \
      if (gamma) g32 = to_f32<T>(*gamma);                                                           \
      compute<Device::CUDA, float>(beta ? &b32 : nullptr, gamma ? &g32 : nullptr, in32, axis,       \
                                   outer_size, axis_size, inner_size, out32);                       \
      from_f32<T>(out32, output);                                                                   \
    }                                                                                               \
    template<>                                                                                      \
    void RMSNorm::compute<Device::CUDA, T>(const StorageView& gamma, const StorageView& input,      \
                                           StorageView& output) const {                             \
      StorageView g32 = to_f32<T>(gamma), in32 = to_f32<T>(input), out32 = f32_like(output);        \
      compute<Device::CUDA, float>(g32, in32, out32);                                               \
      from_f32<T>(out32, output);                                                                   \
    }                                                                                               \
    template<>                                                                                      \
    void SoftMax::compute<Device::CUDA, T>(const StorageView& input, const StorageView* lengths,    \
                                           StorageView& output) const {                             \
      StorageView in32 = to_f32<T>(input), out32 = f32_like(output);                                \
      compute<Device::CUDA, float>(in32, lengths, out32);                                           \
      from_f32<T>(out32, output);                                                                   \
    }                                                                                               \
    template<>                                                                                      \
    void Rotary::compute<Device::CUDA, T>(const StorageView& input, const StorageView& sin,         \
                                          const StorageView& cos, StorageView& output,              \
                                          bool is_transposed) const {                               \
      StorageView in32 = to_f32<T>(input), s32 = to_f32<T>(sin), c32 = to_f32<T>(cos);              \
      StorageView out32 = f32_like(output);                                                         \
      compute<Device::CUDA, float>(in32, s32, c32, out32, is_transposed);                           \
      from_f32<T>(out32, output);                                                                   \
    }                                                                                               \
    template<>                                                                                      \
    void TopK::compute<Device::CUDA, T, int32_t>(const StorageView& x, StorageView& values,         \
                                                 StorageView& indices) const {                      \
      StorageView x32 = to_f32<T>(x), v32 = f32_like(values);                                       \
      compute<Device::CUDA, float, int32_t>(x32, v32, indices);                                     \
      from_f32<T>(v32, values);                                                                     \
    }                                                                                               \
    template<>                                                                                      \
    void TopPMask::compute<Device::CUDA, T>(const StorageView& input, const StorageView& probs,     \
                                            StorageView& output) const {                            \
      StorageView in32 = to_f32<T>(input), p32 = to_f32<T>(probs), out32 = f32_like(output);        \
      compute<Device::CUDA, float>(in32, p32, out32);                                               \
      from_f32<T>(out32, output);                                                                   \
    }                                                                                               \
    template<>                                                                                      \
    void Multinomial::compute<Device::CUDA, T>(const StorageView& input, StorageView& output) const { \
      StorageView in32 = to_f32<T>(input);                                                          \
      compute<Device::CUDA, float>(in32, output);                                                   \
    }                                                                                               \
    template<>                                                                                      \
    void BiasAdd::compute<Device::CUDA, T>(const StorageView& value, const StorageView& bias,       \
                                           StorageView& output, const StorageView* residual) const { \
      StorageView v32 = to_f32<T>(value), b32 = to_f32<T>(bias), out32 = f32_like(output), r32;     \
      if (residual) r32 = to_f32<T>(*residual);                                                     \
      compute<Device::CUDA, float>(v32, b32, out32, residual ? &r32 : nullptr);                     \
      from_f32<T>(out32, output);                                                                   \
    }                                                                                               \
    template<>                                                                                      \
    void Mean::compute<Device::CUDA, T>(const StorageView& input, const dim_t outer_size,           \
                                        const dim_t axis_size, const dim_t inner_size,              \
                                        const bool get_sum, StorageView& output) const {            \
      StorageView in32 = to_f32<T>(input), out32 = f32_like(output);                                \
      compute<Device::CUDA, float>(in32, outer_size, axis_size, inner_size, get_sum, out32);        \
      from_f32<T>(out32, output);                                                                   \
    }                                                                                               \
    template<>                                                                                      \
    void AlibiAdd::compute<Device::CUDA, T>(const StorageView& input, const StorageView& alibi,     \
                                            const dim_t alibi_offset, StorageView& output) const {  \
      StorageView in32 = to_f32<T>(input), a32 = to_f32<T>(alibi), out32 = f32_like(output);        \
      compute<Device::CUDA, float>(in32, a32, alibi_offset, out32);                                 \
      from_f32<T>(out32, output);                                                                   \
    }                                                                                               \
    template<>                                                                                      \
    void MedianFilter::compute<Device::CUDA, T>(const StorageView& input, const dim_t axis_size,    \
                                                StorageView& output) const {                        \
      StorageView in32 = to_f32<T>(input), out32 = f32_like(output);                                \
      compute<Device::CUDA, float>(in32, axis_size, out32);                                         \
      from_f32<T>(out32, output);                                                                   \
    }                                                                                               \
    template<>                                                                                      \
    void Conv1D::compute<Device::CUDA, T>(const StorageView& input, const StorageView& weight,      \
                                          const StorageView* bias, StorageView& output,             \
                                          const StorageView* qscale) const {                        \
      StorageView in32 = to_f32<T>(input), w32 = to_f32<T>(weight), out32 = f32_like(output), b32;  \
      if (bias) b32 = to_f32<T>(*bias);                                                             \
      compute<Device::CUDA, float>(in32, w32, bias ? &b32 : nullptr, out32, qscale);                \
      from_f32<T>(out32, output);                                                                   \
    }                                                                                               \
    template<>                                                                                      \
    void GumbelMax::add_gumbel_noise<Device::CUDA, T>(const StorageView& x, StorageView& y) const { \
      StorageView x32 = to_f32<T>(x), y32 = f32_like(y);                                            \
      add_gumbel_noise<Device::CUDA, float>(x32, y32);                                              \
      from_f32<T>(y32, y);                                                                          \
    }

    CT2X_HALF_OPS(float16_t)
    CT2X_HALF_OPS(bfloat16_t)
#undef CT2X_HALF_OPS

    // ---- AWQ int4 (include/ctranslate2/ops/awq/*.h; CUDA: src/ops/awq/dequantize_gpu.cu:66-98,
    // gemm_gpu.cu:470-530, gemv_gpu.cu:428-530). float16 only, as in the reference. ----------------
    template<>
    void DequantizeAwq::dequantize<Device::CUDA, int, float16_t>(const StorageView& input,
                                                                 const StorageView& scale,
                                                                 const StorageView& zero,
                                                                 StorageView& output) const {
      if (input.rank() != 2)
        throw std::invalid_argument("DequantizeAwq on this backend takes one [in_c, out_c / 8] weight");
      const dim_t in_c = input.dim(0), out_c = input.dim(1) * 8;
      const dim_t group = in_c / scale.dim(0);
      output.resize({in_c, out_c});
      check16(ct2x_awq_dequantize(input.data<int32_t>(), scale.data<float16_t>(), zero.data<int32_t>(),
                                  in_c, out_c, group, output.data<float16_t>(), stream16()));
    }
    // The reference returns eight split-K slices that its front-end adds up with ops::Sum(0)
    // (src/ops/awq/gemm.cc:24-27): this backend accumulates over the whole K in one launch and
    // returns ONE slice, so the front-end's Sum is the identity.
    template<>
    void GemmAwq::compute<Device::CUDA, float16_t, int>(const StorageView& a, const StorageView& b,
                                                        const StorageView& scale,
                                                        const StorageView& zero,
                                                        StorageView& c) const {
      const dim_t in_c = a.dim(-1), m = a.size() / in_c, out_c = b.dim(1) * 8;
      const dim_t group = in_c / scale.dim(0);
      if (a.rank() == 2) c.resize({1, m, out_c});
      else c.resize({1, a.dim(0), a.dim(1), out_c});
      check16(ct2x_awq_gemm(a.data<float16_t>(), b.data<int32_t>(), scale.data<float16_t>(),
                            zero.data<int32_t>(), m, in_c, out_c, group, c.data<float16_t>(), nullptr,
                            stream16()));
    }
    template<>
    void GemvAwq::compute_gemv<Device::CUDA, float16_t, int>(const StorageView& a, const StorageView& b,
                                                             const StorageView& scale,
                                                             const StorageView& zero,
                                                             StorageView& c) const {
      const dim_t in_c = a.dim(-1), m = a.size() / in_c, out_c = b.dim(0);
      const dim_t group = in_c / scale.dim(-1);  // gemv_gpu.cu:447
      if (a.rank() == 2) c.resize({m, out_c});
      else c.resize({a.dim(0), a.dim(1), out_c});
      check16(ct2x_awq_gemv(a.data<float16_t>(), b.data<int32_t>(), scale.data<float16_t>(),
                            zero.data<int32_t>(), m, in_c, out_c, group, zero.dim(-1), scale.dim(-1),
                            c.data<float16_t>(), nullptr, stream16()));
    }
    template<>
    void GemvAwq::compute_gemv2<Device::CUDA, float16_t, int>(const StorageView& a, const StorageView& b,
                                                              const StorageView& scale,
                                                              const StorageView& zero,
                                                              StorageView& c) const {
      const dim_t in_c = a.dim(-1), m = a.size() / in_c, out_c = b.dim(0);
      const dim_t group = in_c / scale.dim(-1);  // gemv_gpu.cu:493
      if (a.rank() == 2) c.resize({1, m, out_c});  // one slice: the front-end's Sum(0) is the identity
      else c.resize({1, a.dim(0), a.dim(1), out_c});
      check16(ct2x_awq_gemv(a.data<float16_t>(), b.data<int32_t>(), scale.data<float16_t>(),
                            zero.data<int32_t>(), m, in_c, out_c, group, zero.dim(-1), scale.dim(-1),
                            c.data<float16_t>(), nullptr, stream16()));
    }

  }

}
