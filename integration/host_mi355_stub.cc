// Level-1 integration, second translation unit (see ops_mi355_stub.cc): what the reference's
// HOST sources need from a GPU backend besides the hot ops - the layout ops of the unfused
// front-end (Concat / Split / Slide / Tile), Mean, the integer arms of penalize_previous_tokens,
// the single-rank forms of the collectives, explicit "not provided" bodies for the ops outside
// SURVEY section 8 (so that a call fails loudly instead of at symbol resolution), and the
// cuda:: helper set of src/cuda/utils.h:98-134 / src/cuda/random.h that devices.cc, types.cc,
// utils.cc and models/model.cc call. Compiled by a plain host compiler against the reference's
// real headers plus integration/shim_cuda/ (the CUDA runtime NAMES those headers leak);
// everything lands on the C ABI of include/ct2_mi355.h. Written for this repository.
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "ctranslate2/ops/ops.h"
#include "ctranslate2/primitives.h"
#include "ctranslate2/random.h"
#include "cuda/random.h"
#include "cuda/utils.h"

#include "ct2_mi355.h"

namespace ctranslate2 {

  static inline void check(int rc) {
    if (rc != 0)
      throw std::runtime_error(ct2x_last_error());
  }

  static inline void* stream() {
    return ct2x_thread_stream();
  }

  static dim_t copy_size_from(const StorageView& x, dim_t axis) {
    dim_t size = 1;
    for (dim_t i = axis; i < x.rank(); ++i)
      size *= x.dim(i);
    return size;
  }

  static dim_t iter_size_before(const StorageView& x, dim_t axis) {
    dim_t size = 1;
    for (dim_t i = 0; i < axis; ++i)
      size *= x.dim(i);
    return size;
  }

  namespace ops {

    // include/ctranslate2/ops/concat.h:18  <-  src/ops/concat_split_slide_gpu.cu:66-112.
    // Same decomposition as the CPU body (src/ops/concat_split_slide_cpu.cc:26-50): every input is
    // iter_size blocks of copy_size elements placed step_size apart in the output.
#define CT2X_LAYOUT_STUBS(T)                                                                        \
    template<>                                                                                      \
    void Concat::compute<Device::CUDA, T>(const std::vector<const StorageView*>& inputs,            \
                                          StorageView& output) const {                              \
      const dim_t axis = _axis < 0 ? output.rank() + _axis : _axis;                                 \
      const dim_t step_size = output.dim(axis) * output.stride(axis);                               \
      T* output_data = output.data<T>();                                                            \
      for (const StorageView* input : inputs) {                                                     \
        const dim_t copy_size = copy_size_from(*input, axis);                                       \
        if (copy_size == 0)                                                                         \
          continue;                                                                                 \
        check(ct2x_block_copy(input->data<T>(), output_data, iter_size_before(*input, axis),        \
                              copy_size * sizeof (T), copy_size * sizeof (T),                       \
                              step_size * sizeof (T), stream()));                                   \
        output_data += copy_size;                                                                   \
      }                                                                                             \
    }                                                                                               \
    template<>                                                                                      \
    void Split::compute<Device::CUDA, T>(const StorageView& input,                                  \
                                         std::vector<StorageView*>& outputs) const {                \
      const dim_t axis = _axis < 0 ? input.rank() + _axis : _axis;                                  \
      const dim_t step_size = input.dim(axis) * input.stride(axis);                                 \
      const T* input_data = input.data<T>();                                                        \
      for (StorageView* output : outputs) {                                                         \
        const dim_t copy_size = copy_size_from(*output, axis);                                      \
        if (copy_size == 0)                                                                         \
          continue;                                                                                 \
        check(ct2x_block_copy(input_data, output->data<T>(), iter_size_before(*output, axis),       \
                              copy_size * sizeof (T), step_size * sizeof (T),                       \
                              copy_size * sizeof (T), stream()));                                   \
        input_data += copy_size;                                                                    \
      }                                                                                             \
    }                                                                                               \
    template<>                                                                                      \
    void Slide::compute<Device::CUDA, T>(const StorageView& input, StorageView& output,             \
                                         const dim_t& index) const {                                \
      const dim_t axis = _axis < 0 ? input.rank() + _axis : _axis;                                  \
      const dim_t stride_axis = input.stride(axis) == 0 ? 1 : input.stride(axis);                   \
      const dim_t step_size = input.dim(axis) * stride_axis;                                        \
      const dim_t copy_size = copy_size_from(output, axis);                                         \
      if (copy_size == 0)                                                                           \
        return;                                                                                     \
      check(ct2x_block_copy(input.data<T>() + index * stride_axis, output.data<T>(),                \
                            iter_size_before(output, axis), copy_size * sizeof (T),                 \
                            step_size * sizeof (T), copy_size * sizeof (T), stream()));             \
    }                                                                                               \
    /* include/ctranslate2/ops/tile.h:17-21  <-  src/ops/tile_gpu.cu:36-70 */                       \
    template<>                                                                                      \
    void Tile::compute<Device::CUDA, T>(const StorageView& input, const dim_t outer_size,           \
                                        const dim_t inner_size, StorageView& output) const {        \
      for (dim_t t = 0; t < _num_tiles; ++t)                                                        \
        check(ct2x_block_copy(input.data<T>(), output.data<T>() + t * inner_size, outer_size,       \
                              inner_size * sizeof (T), inner_size * sizeof (T),                     \
                              _num_tiles * inner_size * sizeof (T), stream()));                     \
    }                                                                                               \
    /* include/ctranslate2/ops/nccl_ops.h:7-33  <-  src/ops/nccl_ops_gpu.cu:53-85. This build has */\
    /* no CT2_WITH_TENSOR_PARALLEL (ScopedMPISetter::getNRanks() == 1): one rank = a copy. The  */  \
    /* multi-rank exchange of this backend lives behind ct2x_translator_create_tp (tp_comm.cc). */  \
    template<>                                                                                      \
    void ReduceAll::compute<Device::CUDA, T>(const StorageView& input, StorageView& output) const { \
      check(ct2x_copy_device(input.data<T>(), output.data<T>(), input.size() * sizeof (T),          \
                             stream()));                                                            \
    }                                                                                               \
    template<>                                                                                      \
    void GatherAll::compute<Device::CUDA, T>(const StorageView& input, StorageView& output) const { \
      check(ct2x_copy_device(input.data<T>(), output.data<T>(), input.size() * sizeof (T),          \
                             stream()));                                                            \
    }

    CT2X_LAYOUT_STUBS(float)
    CT2X_LAYOUT_STUBS(int8_t)
    CT2X_LAYOUT_STUBS(int16_t)
    CT2X_LAYOUT_STUBS(int32_t)
    CT2X_LAYOUT_STUBS(float16_t)   // 2-byte payloads: the 16-bit float arms move bytes like int16
    CT2X_LAYOUT_STUBS(bfloat16_t)
#undef CT2X_LAYOUT_STUBS

    // include/ctranslate2/ops/mean.h:16-22  <-  src/ops/mean_gpu.cu:12-63
    template<>
    void Mean::compute<Device::CUDA, float>(const StorageView& input, const dim_t outer_size,
                                            const dim_t axis_size, const dim_t inner_size,
                                            const bool get_sum, StorageView& output) const {
      check(ct2x_mean(input.data<float>(), outer_size, axis_size, inner_size, get_sum ? 1 : 0,
                      output.data<float>(), stream()));
    }

    // Ops outside the hot path of SURVEY section 8 (Whisper front-end, ALiBi, FlashAttention, the
    // standalone sampling ops whose work this backend fuses into its samplers): defined so that the
    // library resolves, failing loudly if a model ever reaches them.
    [[noreturn]] static void not_provided(const char* op) {
      throw std::runtime_error(std::string(op) + " is not provided by the MI355X backend "
                               "(outside the quantized decoder hot path, SURVEY section 8)");
    }

    template<>
    void AlibiAdd::compute<Device::CUDA, float>(const StorageView& input, const StorageView& alibi,
                                                const dim_t alibi_offset, StorageView& output) const {
      check(ct2x_alibi_add(input.data<float>(), alibi.data<float>(), input.dim(0), input.dim(1),
                           input.dim(2), input.dim(3), alibi.dim(-1), alibi_offset,
                           output.data<float>(), ct2x_thread_stream()));
    }

    template<>
    void Conv1D::compute<Device::CUDA, float>(const StorageView& input, const StorageView& weight,
                                              const StorageView* bias, StorageView& output,
                                              const StorageView* qscale) const {
      if (qscale)
        not_provided("ops::Conv1D with quantised weights");
      // src/ops/conv1d_gpu.cu:12-131: the convolution, then bias + activation over the channel axis
      check(ct2x_conv1d(input.data<float>(), weight.data<float>(), input.dim(0), input.dim(1),
                        input.dim(2), weight.dim(0), weight.dim(2), _stride, _padding, _dilation,
                        _groups, output.data<float>(), ct2x_thread_stream()));
      apply_bias_and_activation(output, bias, _activation_type, /*residual=*/nullptr, /*axis=*/-2);
    }

    template<>
    void FlashAttention::compute<Device::CUDA>(StorageView&, StorageView&, StorageView&, StorageView&,
                                               StorageView*, StorageView*, StorageView*, bool,
                                               StorageView*, StorageView*, const bool, StorageView*,
                                               dim_t) const {
      not_provided("ops::FlashAttention");
    }

    template<>
    void GumbelMax::add_gumbel_noise<Device::CUDA, float>(const StorageView& x, StorageView& y) const {
      // src/ops/gumbel_max_gpu.cu:27-35
      check(ct2x_gumbel_noise(x.data<float>(), y.data<float>(), x.size(), get_random_seed(),
                              ct2x_thread_stream()));
    }

    template<>
    void MedianFilter::compute<Device::CUDA, float>(const StorageView& input, const dim_t axis_size,
                                                    StorageView& output) const {
      if (axis_size <= _width / 2) {  // src/ops/median_filter_cpu.cc:24-28: pass through
        if (&output != &input)
          output.copy_from(input);
        return;
      }
      check(ct2x_median_filter(input.data<float>(), input.size() / axis_size, axis_size, _width,
                               output.data<float>(), ct2x_thread_stream()));
    }

    template<>
    void Multinomial::compute<Device::CUDA, float>(const StorageView& input, StorageView& output) const {
      // src/ops/multinomial_gpu.cu: one inverse-CDF draw per (row, sample)
      const dim_t classes = input.dim(-1);
      check(ct2x_multinomial(input.data<float>(), input.size() / classes, classes, _sample_size,
                             get_random_seed(), output.data<int32_t>(), ct2x_thread_stream()));
    }

    template<>
    dim_t TopPMask::max_num_classes<Device::CUDA>() {
      return 1 << 20;
    }

    template<>
    void TopPMask::compute<Device::CUDA, float>(const StorageView& input, const StorageView& probs,
                                                StorageView& output) const {
      const dim_t depth = input.dim(-1);
      check(ct2x_topp_mask(input.data<float>(), probs.data<float>(), input.size() / depth, depth, _p,
                           _mask_value, output.data<float>(), ct2x_thread_stream()));
    }

  }

  // include/ctranslate2/primitives.h:108-116: integer arms of penalize_previous_tokens exist only
  // because DEVICE_AND_TYPE_DISPATCH instantiates every type (src/decoding_utils.cc:43-66 calls it
  // on the float scores).
#define CT2X_PENALIZE_STUB(T)                                                                   \
  template<> template<>                                                                         \
  void primitives<Device::CUDA>::penalize_previous_tokens(T*, const T*, const int32_t*, T,      \
                                                          dim_t, dim_t, dim_t) {                \
    throw std::invalid_argument("penalize_previous_tokens is defined on float scores");         \
  }
  CT2X_PENALIZE_STUB(int8_t)
  CT2X_PENALIZE_STUB(int16_t)
  CT2X_PENALIZE_STUB(int32_t)
#undef CT2X_PENALIZE_STUB

  // include/ctranslate2/primitives.h:46,100 (scalar reductions read back to the host; only the
  // Whisper timestamp rule calls them, src/models/whisper.cc:855-856): composed from the row ops -
  // max = TopK(k = 1) of the vector as one row; logsumexp = x[0] - log_softmax(x)[0].
  template<> template<>
  float primitives<Device::CUDA>::max(const float* x, dim_t size) {
    void* tmp = nullptr;
    check(ct2x_malloc(&tmp, 8));
    check(ct2x_topk(x, CT2X_F32, 1, size, 1, tmp, reinterpret_cast<int32_t*>(static_cast<char*>(tmp) + 4),
                    stream()));
    float value = 0.f;
    check(ct2x_copy_to_host(tmp, &value, sizeof (float), stream()));
    check(ct2x_free(tmp));
    return value;
  }

  template<> template<>
  float primitives<Device::CUDA>::logsumexp(const float* x, dim_t size) {
    // logsumexp = x[i] - log_softmax(x)[i] at the arg-max i: any other element may be a disabled
    // token (numeric_limits<float>::lowest()), where the difference cancels to 0
    void* tmp = nullptr;
    check(ct2x_malloc(&tmp, size * sizeof (float) + 8));
    char* top = static_cast<char*>(tmp) + size * sizeof (float);
    check(ct2x_topk(x, CT2X_F32, 1, size, 1, top, reinterpret_cast<int32_t*>(top + 4), stream()));
    check(ct2x_softmax(x, nullptr, 1, size, /*log=*/1, tmp, CT2X_F32, stream()));
    float best = 0.f, y = 0.f;
    int32_t index = 0;
    check(ct2x_copy_to_host(top, &best, sizeof (float), stream()));
    check(ct2x_copy_to_host(top + 4, &index, sizeof (int32_t), stream()));
    check(ct2x_copy_to_host(static_cast<float*>(tmp) + index, &y, sizeof (float), stream()));
    check(ct2x_free(tmp));
    return best - y;
  }

  // ---- cuda:: helper set (src/cuda/utils.h:98-134; bodies in src/cuda/utils.cc) ---------------
  namespace cuda {

    const char* cublasGetStatusName(cublasStatus_t) {
      return "no BLAS library on this backend";
    }

    // src/cuda/utils.cc:122-125: one stream per host thread.
    cudaStream_t get_cuda_stream() {
      return ct2x_thread_stream();
    }

    cublasHandle_t get_cublas_handle() {
      return nullptr;  // GEMMs are hand-written kernels (ct2x_gemm_*), there is no handle
    }

    int get_gpu_count() {
      return ct2x_device_count();
    }

    bool has_gpu() {
      return get_gpu_count() > 0;
    }

    // src/cuda/utils.cc:177-199: cached per device. MI355X = gfx950; reported to the reference's
    // capability checks (src/types.cc:99-139, 351-354) as a device with int8 / fp16 / bf16 matrix
    // cores: compute capability 9.5.
    const cudaDeviceProp& get_device_properties(int device) {
      static std::mutex mutex;
      static std::vector<cudaDeviceProp> cache;
      std::lock_guard<std::mutex> lock(mutex);
      if (device < 0)
        check(ct2x_get_device(&device));
      if (cache.empty())
        cache.resize(get_gpu_count());
      if (device >= static_cast<int>(cache.size()))
        throw std::invalid_argument("invalid device index " + std::to_string(device));
      cudaDeviceProp& prop = cache[device];
      if (prop.major == 0) {
        char arch[64];
        int cus = 0;
        int64_t hbm = 0;
        std::memset(&prop, 0, sizeof (prop));
        check(ct2x_device_info(device, prop.name, sizeof (prop.name), arch, sizeof (arch), &cus, &hbm));
        prop.major = 9;
        prop.minor = 5;
        prop.multiProcessorCount = cus;
        prop.totalGlobalMem = static_cast<size_t>(hbm);
      }
      return prop;
    }

    bool gpu_supports_int8(int) { return true; }            // v_mfma_i32_16x16x64_i8
    bool gpu_has_int8_tensor_cores(int) { return true; }
    bool gpu_has_fp16_tensor_cores(int) { return true; }
    bool have_same_compute_capability(const std::vector<int>&) { return true; }

    static thread_local bool true_fp16_gemm = true;
    bool use_true_fp16_gemm() { return true_fp16_gemm; }
    void use_true_fp16_gemm(bool use) { true_fp16_gemm = use; }

    // src/cuda/random.h: the fused samplers draw counter-based uniforms (decode_ops.hip); there
    // are no generator states to allocate or free.
    curandStatePhilox4_32_10_t* get_curand_states(size_t) { return nullptr; }
    void free_curand_states() {}

  }

}
