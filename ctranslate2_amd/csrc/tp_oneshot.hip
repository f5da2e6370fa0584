// One-shot tensor-parallel exchanges over peer-mapped mailboxes ("a custom <= 1 MiB
// one-shot path"; replaces the two RCCL calls per row-parallel Dense - ops::ReduceAll /
// GatherAll, src/ops/nccl_ops_gpu.cu:53-85 - with peer stores and polls INSIDE the kernels that
// produce and consume the exchanged values: no collective launch, no host call, graph-capturable).
//
// A row-parallel int8 Dense needs two values from the other ranks (engine.cc, Engine::dense):
//   (1) before its GEMM: the max over ranks of every row's absmax (the global Quantize scale);
//   (2) after its GEMM:  the sum over ranks of the exact int32 partial products.
// Both travel as 8-byte GRANULES {payload, tag} written by ONE system-scope store each into the
// receiver's mailbox (xGMI peer store between GPUs; the same device for the in-process test
// communicator) and polled by the consumer until the tag is the tag of this exchange
// (MI355X_MICROARCH.md, hand-off price list: a granule needs no fence and no separate flag):
//   tp_amax_quantize_kernel: row absmax of the local slice -> granule to every rank -> max of the
//                            R granules -> Quantize of the slice with the global scale
//   tp_sum_dequant_kernel:   int32 partial -> granule to every rank -> rank-ordered sum ->
//                            Dequantize (+ bias, + residual), the reference's CPU formula order
// (float layers: tp_sum_bias_kernel sums fp32 partials in rank order - identical on every rank).
// The tag is a per-communicator counter in device memory, advanced by the last workgroup of every
// exchange kernel (so a captured graph replays with fresh tags); mailboxes have two parity halves:
// a rank can run at most one exchange ahead of its slowest peer (it needs that peer's granules to
// finish its own), so exchange k + 1 never overwrites granules of k that are still being read.
// Polls are bounded: a peer that never arrives sets the status word instead of hanging the GPU.

#include "device_utils.h"
#include "kernels.h"
#include "runtime.h"
#include "tp_comm.h"

namespace ct2x {

namespace {

__device__ __forceinline__ void granule_store(uint64_t* p, uint32_t payload, uint32_t tag) {
  __hip_atomic_store(p, (static_cast<uint64_t>(tag) << 32) | payload, __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_SYSTEM);
}
// Polls until the granule carries `tag`; false (and status = 1) after a few seconds without it.
__device__ __forceinline__ bool granule_wait(const uint64_t* p, uint32_t tag, uint32_t* payload,
                                             uint32_t* status) {
  // (fail fast: once any poll of this rank has given up, nobody waits again - a dead peer costs one
  // time-out, not one per granule)
  const uint32_t limit = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1u : (1u << 22);
  for (uint32_t spin = 0; spin < limit; ++spin) {
    const uint64_t g = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (static_cast<uint32_t>(g >> 32) == tag) {
      *payload = static_cast<uint32_t>(g);
      return true;
    }
    if (spin > 64) __builtin_amdgcn_s_sleep(32);
  }
  __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  *payload = 0;
  return false;
}
// The last workgroup of the launch advances the communicator's tag counter.
__device__ __forceinline__ void finish_exchange(TpBoxControl* ctl) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t done = atomicAdd(&ctl->done, 1u);
    if (done == gridDim.x * gridDim.y - 1) {
      ctl->done = 0;
      __hip_atomic_store(&ctl->seq, ctl->seq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

struct PeerBoxes {
  char* box[kTpMaxWorld];
};

// One workgroup per row: local absmax (launch_row_amax's reduction), granules, global scale,
// Quantize (quantize_s8_kernel's element form; src/cpu/kernels.cc:577-651).
__global__ void tp_amax_quantize_kernel(const float* __restrict__ x, int64_t depth, int quant_mode,
                                        int8_t* __restrict__ q, float* __restrict__ scales,
                                        PeerBoxes peers, int rank, int world) {
  __shared__ float scratch[16];
  __shared__ float s_scale;
  const int row = blockIdx.x;
  TpBoxControl* ctl = reinterpret_cast<TpBoxControl*>(peers.box[rank]);
  const uint32_t tag = ctl->seq + 1;
  const float* xr = x + static_cast<int64_t>(row) * depth;
  float m = 0.f;
  for (int64_t i = threadIdx.x; i < depth; i += blockDim.x) m = fmaxf(m, fabsf(xr[i]));
  m = block_max(m, scratch);
  if (threadIdx.x < world) {
    // lane r carries this rank's maximum to rank r (its own box included) ...
    uint64_t* slot = reinterpret_cast<uint64_t*>(peers.box[threadIdx.x] + tp_box_amax_offset(tag & 1, rank)) + row;
    granule_store(slot, __float_as_uint(m), tag);
  }
  if (threadIdx.x == 0) {
    // ... and lane 0 collects the R maxima that arrive in this rank's box
    float g = 0.f;
    for (int r = 0; r < world; ++r) {
      uint32_t bits;
      granule_wait(reinterpret_cast<const uint64_t*>(peers.box[rank] + tp_box_amax_offset(tag & 1, r)) + row,
                   tag, &bits, &ctl->status);
      g = fmaxf(g, __uint_as_float(bits));
    }
    const float scale = (g != 0.f) ? 127.f / g : 1.f;
    s_scale = scale;
    scales[row] = scale;
  }
  __syncthreads();
  const float scale = s_scale;
  int8_t* qr = q + static_cast<int64_t>(row) * depth;
  for (int64_t i = threadIdx.x; i < depth; i += blockDim.x) qr[i] = to_s8(xr[i], scale, quant_mode);
  finish_exchange(ctl);
}

// Element-wise: int32 (kInt) or fp32 partial -> granules -> sum in rank order -> epilogue.
//   kInt:  y = (sum * (1 / a_scale[row])) / w_scale[col] + bias[col] (+ residual)
//          (dequantize_gemm_output_kernel; CPU formula order src/cpu/kernels.cc:666-673)
//   float: y = residual + (sum + bias[col])  (bias_residual_kernel)
template <bool kInt>
__global__ void tp_sum_epilogue_kernel(const void* __restrict__ partial, const float* __restrict__ a_scale,
                                       const float* __restrict__ w_scale, const float* __restrict__ bias,
                                       const float* residual, int64_t n, int64_t total, float* y,
                                       PeerBoxes peers, int rank, int world) {
  TpBoxControl* ctl = reinterpret_cast<TpBoxControl*>(peers.box[rank]);
  const uint32_t tag = ctl->seq + 1;
  const int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (e < total) {
    const uint32_t mine = static_cast<const uint32_t*>(partial)[e];
    for (int r = 0; r < world; ++r) {
      if (r == rank) continue;
      granule_store(reinterpret_cast<uint64_t*>(peers.box[r] + tp_box_sum_offset(tag & 1, rank)) + e, mine, tag);
    }
    int32_t isum = 0;
    float fsum = 0.f;
    for (int r = 0; r < world; ++r) {  // rank order: the float sum is the same on every rank
      uint32_t bits = mine;
      if (r != rank)
        granule_wait(reinterpret_cast<const uint64_t*>(peers.box[rank] + tp_box_sum_offset(tag & 1, r)) + e,
                     tag, &bits, &ctl->status);
      if (kInt) isum += static_cast<int32_t>(bits);
      else fsum = (r == 0) ? __uint_as_float(bits) : fsum + __uint_as_float(bits);
    }
    const int64_t row = e / n, col = e % n;
    float v;
    if (kInt) {
      const float r_a = 1.f / a_scale[row];
      v = static_cast<float>(isum) * r_a;
      v = v / w_scale[col];
      if (bias) v = v + bias[col];
      if (residual) v = residual[e] + v;
    } else {
      v = fsum;
      if (bias) v = v + bias[col];
      if (residual) v = residual[e] + v;
    }
    y[e] = v;
  }
  finish_exchange(ctl);
}

PeerBoxes boxes_of(const TpPeerBox& pb) {
  PeerBoxes p;
  for (int r = 0; r < kTpMaxWorld; ++r) p.box[r] = r < pb.world ? pb.box[r] : nullptr;
  return p;
}

}  // namespace

bool tp_oneshot_fits(int64_t rows, int64_t elements) {
  return rows <= kTpBoxMaxRows && elements <= kTpBoxMaxElements;
}

void launch_tp_amax_quantize(const TpPeerBox& pb, const float* x, int64_t rows, int64_t depth,
                             int quant_mode, int8_t* q, float* scales, hipStream_t stream) {
  if (rows == 0) return;
  CT2X_CHECK(rows <= kTpBoxMaxRows, "one-shot absmax exchange: too many rows for the mailbox");
  int block = 64;
  while (block < depth && block < 1024) block <<= 1;
  hipLaunchKernelGGL(tp_amax_quantize_kernel, dim3(static_cast<unsigned>(rows)), dim3(block), 0, stream,
                     x, depth, quant_mode, q, scales, boxes_of(pb), pb.rank, pb.world);
}

void launch_tp_sum_dequant(const TpPeerBox& pb, const int32_t* partial, const float* a_scale,
                           const float* w_scale, const float* bias, const float* residual,
                           int64_t m, int64_t n, float* y, hipStream_t stream) {
  const int64_t total = m * n;
  if (total == 0) return;
  CT2X_CHECK(total <= kTpBoxMaxElements, "one-shot sum exchange: message larger than the mailbox");
  hipLaunchKernelGGL((tp_sum_epilogue_kernel<true>), dim3(static_cast<unsigned>((total + 255) / 256)),
                     dim3(256), 0, stream, partial, a_scale, w_scale, bias, residual, n, total, y,
                     boxes_of(pb), pb.rank, pb.world);
}

void launch_tp_sum_bias(const TpPeerBox& pb, const float* partial, const float* bias,
                        const float* residual, int64_t m, int64_t n, float* y, hipStream_t stream) {
  const int64_t total = m * n;
  if (total == 0) return;
  CT2X_CHECK(total <= kTpBoxMaxElements, "one-shot sum exchange: message larger than the mailbox");
  hipLaunchKernelGGL((tp_sum_epilogue_kernel<false>), dim3(static_cast<unsigned>((total + 255) / 256)),
                     dim3(256), 0, stream, partial, nullptr, nullptr, bias, residual, n, total, y,
                     boxes_of(pb), pb.rank, pb.world);
}

}  // namespace ct2x
