// Collectives of the tensor-parallel Dense layers (reference ops::ReduceAll /
// ops::GatherAll, src/ops/nccl_ops_gpu.cu:53-85, bootstrap src/devices.cc:141-229).
//
// Two exchanges exist per row-parallel Dense, both on [m, ...] buffers that every rank holds:
//   * max  over ranks of the per-row absmax  (m floats)      -> global Quantize scale
//   * sum  over ranks of the int32 GEMM accumulators [m, d]  -> dequantised ONCE, after the sum
// Summing the exact int32 partials (instead of the reference's per-rank float outputs) keeps the
// tensor-parallel result bit-identical to the single-GPU result. Float (non-quantised) layers sum
// fp32 partial products.
//
// Backends: RCCL over xGMI (one process per GPU, ncclAllReduce on the engine's stream, library
// resolved with dlopen so single-GPU hosts do not need it) and an in-process communicator for
// world_size ranks that live as threads of ONE process on one device - the test double that lets
// the sharding and the placement of the collectives be verified on a single-GPU box.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <memory>
#include <string>
#include <vector>

namespace ct2x {

class TpComm {
public:
  virtual ~TpComm() = default;
  virtual int rank() const = 0;
  virtual int world() const = 0;
  // In-place all-reduce of `count` elements at `buf` (device memory of this rank) on `stream`.
  virtual void all_reduce_sum_i32(int32_t* buf, int64_t count, hipStream_t stream) = 0;
  virtual void all_reduce_sum_f32(float* buf, int64_t count, hipStream_t stream) = 0;
  virtual void all_reduce_max_f32(float* buf, int64_t count, hipStream_t stream) = 0;
  // ops::GatherAll (src/ops/nccl_ops_gpu.cu:70-85): every rank contributes `bytes` bytes at `send`;
  // recv [world][bytes] holds the contributions in rank order on every rank. send may not alias recv.
  virtual void all_gather(const void* send, void* recv, size_t bytes, hipStream_t stream) = 0;
  virtual const char* backend() const = 0;
  // true when the collectives are plain stream work that a hipGraph can capture and replay
  virtual bool capturable() const = 0;
};

// 128-byte ncclUniqueId created on rank 0 and handed to every rank by the launcher.
constexpr int kTpUniqueIdBytes = 128;
void rccl_get_unique_id(void* id_out);
std::shared_ptr<TpComm> make_rccl_comm(int rank, int world, const void* unique_id);

// `world` communicators sharing one rendezvous object; comm i must be used by exactly one thread.
std::vector<std::shared_ptr<TpComm>> make_local_comms(int world);

}  // namespace ct2x
