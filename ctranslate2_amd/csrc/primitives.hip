// Element-wise / layout members of primitives<Device::CUDA> that the reference's UNFUSED front-end
// calls between the hot ops (include/ctranslate2/primitives.h:10-233; CUDA bodies in
// src/cuda/primitives.cu, CPU bodies in src/cpu/primitives.cc): add / sub / mul / min / max in
// their vector and scalar forms, the batch / depth / block broadcasts, fill / strided_fill /
// indexed_fill, the unary float functions, transpose_2d/3d/4d and prepare_length_mask. The fused
// engine uses none of them (the work is folded into its kernels); they exist so that the Level-1
// stubs (integration/ops_mi355_stub.cc, BOUNDARY.md) cover what a maintainer's op-by-op
// build links against. All of them are plain HBM-bound scans: one element per lane and
// grid-stride step, consecutive lanes on consecutive addresses.
#include <algorithm>

#include <hip/hip_fp16.h>

#include "device_utils.h"
#include "kernels.h"
#include "runtime.h"

namespace ct2x {

namespace {

constexpr int kThreads = 256;

inline unsigned blocks_for(int64_t size) {
  const int64_t b = (size + kThreads - 1) / kThreads;
  return static_cast<unsigned>(b < 1 ? 1 : (b > 65535 ? 65535 : b));
}

template <typename T>
__device__ __forceinline__ T apply_binary(int op, T a, T b) {
  switch (op) {
    case 0: return static_cast<T>(a + b);
    case 1: return static_cast<T>(a - b);
    case 2: return static_cast<T>(a * b);
    case 3: return a < b ? a : b;
    default: return a > b ? a : b;
  }
}

// c[i] = a[i] op b[i]
// (no __restrict__ on these element-wise kernels: the reference calls them in place, e.g.
// primitives::add(x, y, size) = add(x, y, y, size), include/ctranslate2/primitives.h:36-38)
template <typename T>
__global__ void binary_kernel(int op, const T* a, const T* b, T* c, int64_t size) {
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < size;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
    c[i] = apply_binary<T>(op, a[i], b[i]);
}

// y[i] = x[i] op a   (add / mul / min / max with a scalar; commutative forms only)
template <typename T>
__global__ void binary_scalar_kernel(int op, T a, const T* x, T* y, int64_t size) {
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < size;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
    y[i] = apply_binary<T>(op, x[i], a);
}

// kind 0: batch broadcast  c[i] = a[i % a_size] op b[i]        (add_/mul_batch_broadcast)
// kind 1: depth broadcast  c[i] = a[i / depth] op b[i], depth = b_size / a_size
// kind 2: block broadcast  c[i] = a[(i / block) % a_size] op b[i]
template <typename T>
__global__ void broadcast_kernel(int op, int kind, const T* __restrict__ a, const T* b, T* c,
                                 int64_t a_size, int64_t b_size, int64_t block) {
  const int64_t depth = a_size > 0 ? b_size / a_size : 1;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < b_size;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    int64_t ai;
    if (kind == 0)
      ai = i % a_size;
    else if (kind == 1)
      ai = i / depth;
    else
      ai = (i / block) % a_size;
    c[i] = apply_binary<T>(op, a[ai], b[i]);
  }
}

template <typename T>
__global__ void fill_kernel(T* __restrict__ x, T value, int64_t size, int64_t stride) {
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < size;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
    x[i * stride] = value;
}

template <typename T>
__global__ void indexed_fill_kernel(T* __restrict__ x, T value, const int32_t* __restrict__ idx,
                                    int64_t n) {
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
    x[idx[i]] = value;
}

// fn: 0..6 = ops::ActivationType (ReLU, GELUTanh, Swish, GELU, GELUSigmoid, Tanh, Sigmoid),
// 7 exp, 8 log, 9 sin, 10 cos.
__global__ void unary_kernel(int fn, const float* x, float* y, int64_t size) {
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < size;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const float v = x[i];
    float r;
    if (fn <= 6)
      r = apply_activation(v, fn);
    else if (fn == 7)
      r = expf(v);
    else if (fn == 8)
      r = logf(v);
    else if (fn == 9)
      r = sinf(v);
    else
      r = cosf(v);
    y[i] = r;
  }
}

// b = transpose(a, perm): rank padded to 4, out shape[k] = dims[perm[k]], consecutive lanes
// write consecutive output elements.
struct TransposeParams {
  int64_t out_dims[4];
  int64_t in_strides_of_out[4];  // stride in `a` of the input axis that output axis k walks
};

template <typename T>
__global__ void transpose_kernel(const T* __restrict__ a, T* __restrict__ b, int64_t size,
                                 TransposeParams p) {
  for (int64_t o = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; o < size;
       o += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    int64_t rem = o, in = 0;
#pragma unroll
    for (int k = 3; k >= 0; --k) {
      const int64_t j = rem % p.out_dims[k];
      rem /= p.out_dims[k];
      in += j * p.in_strides_of_out[k];
    }
    b[o] = a[in];
  }
}

// AttentionLayer::prepare_length_mask (src/cpu/primitives.cc:431-448)
__global__ void length_mask_kernel(const int32_t* __restrict__ lengths, int64_t heads,
                                   int64_t queries, bool mask_future, bool multi_query,
                                   int32_t* __restrict__ mask, int64_t total) {
  const int64_t per_batch = heads * queries;
  for (int64_t o = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; o < total;
       o += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t b = o / per_batch, i = o % per_batch;
    const int32_t length = lengths[b];
    int32_t v = length;
    if (mask_future) {
      const int32_t q = static_cast<int32_t>(multi_query ? i / heads : i % queries) + 1;
      v = length < q ? length : q;
    }
    mask[o] = v;
  }
}

template <typename F>
void by_dtype(int dtype, F&& f) {
  switch (dtype) {
    case 0: f(float()); break;
    case 1: f(int8_t()); break;
    case 2: f(int16_t()); break;
    case 3: f(int32_t()); break;
    default: throw Error("element-wise primitive: dtype must be float32, int8, int16 or int32");
  }
}

// Layout ops of the unfused front-end (ops::Concat / Split / Slide: src/ops/
// concat_split_slide_gpu.cu:66-209; ops::Tile: src/ops/tile_gpu.cu:36-70): all of them are
// "outer x block" strided copies. block_copy moves `outer` blocks of `block` bytes, source block o
// at src + o*src_stride, destination block o at dst + o*dst_stride; 16-byte lanes when the
// pointers, the strides and the block size allow, bytes otherwise.
template <typename V>
__global__ void block_copy_kernel(const char* __restrict__ src, char* __restrict__ dst,
                                  int64_t outer, int64_t block_v, int64_t src_stride,
                                  int64_t dst_stride) {
  const int64_t total = outer * block_v;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t o = i / block_v, e = i % block_v;
    reinterpret_cast<V*>(dst + o * dst_stride)[e] = reinterpret_cast<const V*>(src + o * src_stride)[e];
  }
}

// ops::Mean / Sum over the middle axis of [outer, axis, inner] (src/ops/mean_gpu.cu:12-63): one
// thread per (outer, inner) element walks the axis in order (fp32 accumulation, as the CPU's
// sequential loop, src/ops/mean_cpu.cc).
__global__ void mean_kernel(const float* __restrict__ x, int64_t outer, int64_t axis, int64_t inner,
                            bool get_sum, float* __restrict__ y) {
  const int64_t total = outer * inner;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t o = i / inner, e = i % inner;
    const float* p = x + o * axis * inner + e;
    float acc = 0.f;
    for (int64_t a = 0; a < axis; ++a) acc += p[a * inner];
    y[i] = get_sum ? acc : acc / static_cast<float>(axis);
  }
}

}  // namespace

void launch_block_copy(const void* src, void* dst, int64_t outer, int64_t block_bytes,
                       int64_t src_stride_bytes, int64_t dst_stride_bytes, hipStream_t stream) {
  if (outer <= 0 || block_bytes <= 0) return;
  const uintptr_t bits = reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) |
                         static_cast<uintptr_t>(block_bytes) | static_cast<uintptr_t>(src_stride_bytes) |
                         static_cast<uintptr_t>(dst_stride_bytes);
  const int64_t unit = (bits & 15) == 0 ? 16 : ((bits & 3) == 0 ? 4 : 1);
  const int64_t total = outer * (block_bytes / unit);
  const unsigned grid = static_cast<unsigned>(std::min<int64_t>((total + 255) / 256, 65535));
  const char* s = static_cast<const char*>(src);
  char* d = static_cast<char*>(dst);
  if (unit == 16)
    hipLaunchKernelGGL(block_copy_kernel<int4>, dim3(grid), dim3(256), 0, stream, s, d, outer,
                       block_bytes / 16, src_stride_bytes, dst_stride_bytes);
  else if (unit == 4)
    hipLaunchKernelGGL(block_copy_kernel<int32_t>, dim3(grid), dim3(256), 0, stream, s, d, outer,
                       block_bytes / 4, src_stride_bytes, dst_stride_bytes);
  else
    hipLaunchKernelGGL(block_copy_kernel<char>, dim3(grid), dim3(256), 0, stream, s, d, outer,
                       block_bytes, src_stride_bytes, dst_stride_bytes);
}

void launch_mean(const float* x, int64_t outer, int64_t axis, int64_t inner, bool get_sum, float* y,
                 hipStream_t stream) {
  const int64_t total = outer * inner;
  if (total <= 0) return;
  const unsigned grid = static_cast<unsigned>(std::min<int64_t>((total + 255) / 256, 65535));
  hipLaunchKernelGGL(mean_kernel, dim3(grid), dim3(256), 0, stream, x, outer, axis, inner, get_sum, y);
}

void launch_binary(int op, const void* a, const void* b, void* c, int64_t size, int dtype,
                   hipStream_t stream) {
  if (size == 0) return;
  if (op < 0 || op > 4) throw Error("launch_binary: op must be add, sub, mul, min or max");
  by_dtype(dtype, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL(binary_kernel<T>, dim3(blocks_for(size)), dim3(kThreads), 0, stream, op,
                       static_cast<const T*>(a), static_cast<const T*>(b), static_cast<T*>(c), size);
  });
}

void launch_binary_scalar(int op, double scalar, const void* x, void* y, int64_t size, int dtype,
                          hipStream_t stream) {
  if (size == 0) return;
  if (op != 0 && op != 2 && op != 3 && op != 4)
    throw Error("launch_binary_scalar: op must be add, mul, min or max");
  by_dtype(dtype, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL(binary_scalar_kernel<T>, dim3(blocks_for(size)), dim3(kThreads), 0, stream,
                       op, static_cast<T>(scalar), static_cast<const T*>(x), static_cast<T*>(y), size);
  });
}

void launch_broadcast(int op, int kind, const void* a, const void* b, void* c, int64_t a_size,
                      int64_t b_size, int64_t block, int dtype, hipStream_t stream) {
  if (b_size == 0) return;
  if (a_size <= 0 || kind < 0 || kind > 2 || (kind == 2 && block <= 0) || (op != 0 && op != 2))
    throw Error("launch_broadcast: bad arguments");
  by_dtype(dtype, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL(broadcast_kernel<T>, dim3(blocks_for(b_size)), dim3(kThreads), 0, stream, op,
                       kind, static_cast<const T*>(a), static_cast<const T*>(b), static_cast<T*>(c),
                       a_size, b_size, block);
  });
}

void launch_fill(void* x, double value, int64_t size, int64_t stride, int dtype, hipStream_t stream) {
  if (size == 0) return;
  by_dtype(dtype, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL(fill_kernel<T>, dim3(blocks_for(size)), dim3(kThreads), 0, stream,
                       static_cast<T*>(x), static_cast<T>(value), size, stride);
  });
}

void launch_indexed_fill(void* x, double value, const int32_t* indices, int64_t n, int dtype,
                         hipStream_t stream) {
  if (n == 0) return;
  by_dtype(dtype, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL(indexed_fill_kernel<T>, dim3(blocks_for(n)), dim3(kThreads), 0, stream,
                       static_cast<T*>(x), static_cast<T>(value), indices, n);
  });
}

void launch_unary(int fn, const float* x, float* y, int64_t size, hipStream_t stream) {
  if (size == 0) return;
  if (fn < 0 || fn > 10) throw Error("launch_unary: unknown function");
  hipLaunchKernelGGL(unary_kernel, dim3(blocks_for(size)), dim3(kThreads), 0, stream, fn, x, y, size);
}

void launch_transpose(const void* a, const int64_t* dims, const int64_t* perm, int rank, void* b,
                      int dtype, hipStream_t stream) {
  if (rank < 2 || rank > 4) throw Error("launch_transpose: rank must be 2, 3 or 4");
  int64_t d[4] = {1, 1, 1, 1}, pm[4] = {0, 1, 2, 3};
  const int pad = 4 - rank;  // leading unit axes
  bool seen[4] = {false, false, false, false};
  for (int i = 0; i < rank; ++i) {
    if (perm[i] < 0 || perm[i] >= rank || seen[perm[i]]) throw Error("launch_transpose: bad permutation");
    seen[perm[i]] = true;
    d[pad + i] = dims[i];
    pm[pad + i] = pad + perm[i];
  }
  int64_t in_stride[4];
  in_stride[3] = 1;
  for (int i = 2; i >= 0; --i) in_stride[i] = in_stride[i + 1] * d[i + 1];
  TransposeParams p;
  int64_t size = 1;
  for (int k = 0; k < 4; ++k) {
    p.out_dims[k] = d[pm[k]];
    p.in_strides_of_out[k] = in_stride[pm[k]];
    size *= d[k];
  }
  if (size == 0) return;
  by_dtype(dtype, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL(transpose_kernel<T>, dim3(blocks_for(size)), dim3(kThreads), 0, stream,
                       static_cast<const T*>(a), static_cast<T*>(b), size, p);
  });
}

void launch_prepare_length_mask(const int32_t* lengths, int64_t batch, int64_t heads,
                                int64_t queries, bool mask_future, bool multi_query,
                                int32_t* mask, hipStream_t stream) {
  const int64_t total = batch * heads * queries;
  if (total == 0) return;
  hipLaunchKernelGGL(length_mask_kernel, dim3(blocks_for(total)), dim3(kThreads), 0, stream, lengths,
                     heads, queries, mask_future, multi_query, mask, total);
}

// primitives::convert between float32, float16 and bfloat16 (src/cuda/primitives.cu:160-196):
// round to nearest even on the way down (the hardware conversion for fp16, the usual bias trick
// for bf16 with quiet NaNs), exact on the way up; 16 -> 16 goes through float.
__device__ __forceinline__ float load_as_float(const void* p, int64_t i, int dtype) {
  if (dtype == 0) return static_cast<const float*>(p)[i];
  const uint16_t b = static_cast<const uint16_t*>(p)[i];
  if (dtype == 4) return __half2float(__ushort_as_half(b));
  return __uint_as_float(static_cast<uint32_t>(b) << 16);
}
__device__ __forceinline__ void store_from_float(void* p, int64_t i, int dtype, float v) {
  if (dtype == 0) {
    static_cast<float*>(p)[i] = v;
  } else if (dtype == 4) {
    static_cast<uint16_t*>(p)[i] = __half_as_ushort(__float2half_rn(v));
  } else {
    uint32_t u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u)
      u = 0x7fc00000u;  // NaN
    else
      u += 0x7fffu + ((u >> 16) & 1u);
    static_cast<uint16_t*>(p)[i] = static_cast<uint16_t>(u >> 16);
  }
}
__global__ void convert_kernel(const void* __restrict__ x, int src, void* __restrict__ y, int dst,
                               int64_t n) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i < n) store_from_float(y, i, dst, load_as_float(x, i, src));
}

void launch_convert(const void* x, int src_dtype, void* y, int dst_dtype, int64_t n,
                    hipStream_t stream) {
  if (n == 0) return;
  hipLaunchKernelGGL(convert_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, stream, x, src_dtype, y,
                     dst_dtype, n);
}

// ---- ops outside the decoder hot path that the reference's front-end can still ask a device for
// (Whisper's Conv1D front-end and MedianFilter alignment smoothing, ALiBi position bias): direct,
// one thread per output element - correctness arms of the Level-1 boundary, not tuned kernels.

// ops::Conv1D (src/ops/conv1d_cpu.cc / conv1d_gpu.cu; bias and activation are applied by the
// caller through apply_bias_and_activation): x [batch, in_ch, len], w [out_ch, in_ch / groups,
// kernel], y [batch, out_ch, out_len].
__global__ void conv1d_kernel(const float* __restrict__ x, const float* __restrict__ w, int64_t in_ch,
                              int64_t len, int64_t out_ch, int64_t kernel, int64_t stride,
                              int64_t padding, int64_t dilation, int64_t groups, int64_t out_len,
                              int64_t total, float* __restrict__ y) {
  const int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (e >= total) return;
  const int64_t t = e % out_len, oc = (e / out_len) % out_ch, b = e / (out_len * out_ch);
  const int64_t icg = in_ch / groups, ocg = out_ch / groups, g = oc / ocg;
  const float* xb = x + (b * in_ch + g * icg) * len;
  const float* wo = w + oc * icg * kernel;
  float acc = 0.f;
  for (int64_t ic = 0; ic < icg; ++ic)
    for (int64_t k = 0; k < kernel; ++k) {
      const int64_t pos = t * stride - padding + k * dilation;
      if (pos >= 0 && pos < len) acc = fmaf(wo[ic * kernel + k], xb[ic * len + pos], acc);
    }
  y[e] = acc;
}

void launch_conv1d(const float* x, const float* w, int64_t batch, int64_t in_ch, int64_t len,
                   int64_t out_ch, int64_t kernel, int64_t stride, int64_t padding, int64_t dilation,
                   int64_t groups, float* y, hipStream_t stream) {
  const int64_t out_len = (len + 2 * padding - (dilation * (kernel - 1) + 1)) / stride + 1;
  const int64_t total = batch * out_ch * out_len;
  if (total <= 0) return;
  hipLaunchKernelGGL(conv1d_kernel, dim3(blocks_for(total)), dim3(kThreads), 0, stream, x, w, in_ch,
                     len, out_ch, kernel, stride, padding, dilation, groups, out_len, total, y);
}

// ops::MedianFilter (src/ops/median_filter_cpu.cc:12-61): median of a window of `width` (odd,
// <= 63) around every element of the last axis, reflected at both ends.
__global__ void median_filter_kernel(const float* __restrict__ x, int64_t depth, int64_t width,
                                     int64_t total, float* __restrict__ y) {
  const int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (e >= total) return;
  const int64_t j = e % depth;
  const float* in = x + (e - j);
  const int64_t rank = width / 2;
  float window[63];
  for (int64_t k = -rank; k <= rank; ++k) {
    int64_t read = j + k < 0 ? -(j + k) : j + k;
    if (read >= depth) read = depth - (read - depth) - 2;
    window[k + rank] = in[read];
  }
  for (int64_t a = 1; a < width; ++a) {  // insertion sort of the (short) window
    const float v = window[a];
    int64_t b = a - 1;
    while (b >= 0 && window[b] > v) {
      window[b + 1] = window[b];
      --b;
    }
    window[b + 1] = v;
  }
  y[e] = window[rank];
}

void launch_median_filter(const float* x, int64_t rows, int64_t depth, int64_t width, float* y,
                          hipStream_t stream) {
  const int64_t total = rows * depth;
  if (total == 0) return;
  hipLaunchKernelGGL(median_filter_kernel, dim3(blocks_for(total)), dim3(kThreads), 0, stream, x, depth,
                     width, total, y);
}

// ops::AlibiAdd (src/ops/alibi_add_cpu.cc:8-31): y[b, h, q, k] = x[b, h, q, k] +
// alibi[h, alibi_offset + k], alibi [1, heads, 1, alibi_keys].
__global__ void alibi_add_kernel(const float* __restrict__ x, const float* __restrict__ alibi,
                                 int64_t heads, int64_t queries, int64_t keys, int64_t alibi_keys,
                                 int64_t offset, int64_t total, float* __restrict__ y) {
  const int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (e >= total) return;
  const int64_t k = e % keys, h = (e / (keys * queries)) % heads;
  y[e] = x[e] + alibi[h * alibi_keys + offset + k];
}

void launch_alibi_add(const float* x, const float* alibi, int64_t batch, int64_t heads,
                      int64_t queries, int64_t keys, int64_t alibi_keys, int64_t offset, float* y,
                      hipStream_t stream) {
  const int64_t total = batch * heads * queries * keys;
  if (total == 0) return;
  hipLaunchKernelGGL(alibi_add_kernel, dim3(blocks_for(total)), dim3(kThreads), 0, stream, x, alibi,
                     heads, queries, keys, alibi_keys, offset, total, y);
}

}  // namespace ct2x
