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

// ---- one-shot exchanges over peer-mapped mailboxes (tp_oneshot.hip) ---------------------------
// Every rank owns one mailbox ("box") that every other rank can store into: peer-mapped device
// memory between GPUs (hipIpc handles exchanged at bootstrap), plain device memory for the
// in-process communicator. Layout: control block | absmax granules [2 parity][R sources][rows] |
// sum granules [2 parity][R sources][elements]; a granule is 8 bytes {payload, tag}.
constexpr int kTpMaxWorld = 8;
constexpr int64_t kTpBoxMaxRows = 4096;            // rows per absmax exchange
constexpr int64_t kTpBoxMaxElements = 1 << 18;     // 4-byte elements per sum exchange (1 MiB)
struct TpBoxControl {
  uint32_t seq;     // tag of the last finished exchange (advanced by the exchange kernels)
  uint32_t done;    // workgroups of the running exchange kernel that have finished
  uint32_t status;  // 1: a poll gave up (a peer never arrived)
  uint32_t pad[61];
};
static_assert(sizeof(TpBoxControl) == 256, "control block");
inline __host__ __device__ size_t tp_box_amax_offset(int parity, int src) {
  return sizeof(TpBoxControl) + (static_cast<size_t>(parity) * kTpMaxWorld + src) * kTpBoxMaxRows * 8;
}
inline __host__ __device__ size_t tp_box_sum_offset(int parity, int src) {
  return sizeof(TpBoxControl) + 2 * static_cast<size_t>(kTpMaxWorld) * kTpBoxMaxRows * 8 +
         (static_cast<size_t>(parity) * kTpMaxWorld + src) * kTpBoxMaxElements * 8;
}
inline size_t tp_box_bytes() { return tp_box_sum_offset(2, 0); }
struct TpPeerBox {
  int rank = 0, world = 1;
  char* box[kTpMaxWorld] = {nullptr};  // box[r]: rank r's mailbox as THIS rank addresses it
};

class TpComm {
public:
  virtual ~TpComm() = default;
  // Mailboxes for the one-shot exchanges, or nullptr when this communicator has none (the engine
  // then uses the collectives below).
  virtual const TpPeerBox* peer_box() const { return nullptr; }
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

// tp_oneshot.hip: fused producer / consumer kernels of a row-parallel Dense (see the file header).
bool tp_oneshot_fits(int64_t rows, int64_t elements);
void launch_tp_amax_quantize(const TpPeerBox& pb, const float* x, int64_t rows, int64_t depth,
                             int quant_mode, int8_t* q, float* scales, hipStream_t stream);
void launch_tp_sum_dequant(const TpPeerBox& pb, const int32_t* partial, const float* a_scale,
                           const float* w_scale, const float* bias, const float* residual,
                           int64_t m, int64_t n, float* y, hipStream_t stream);
void launch_tp_sum_bias(const TpPeerBox& pb, const float* partial, const float* bias,
                        const float* residual, int64_t m, int64_t n, float* y, hipStream_t stream);

// 128-byte ncclUniqueId created on rank 0 and handed to every rank by the launcher.
constexpr int kTpUniqueIdBytes = 128;
void rccl_get_unique_id(void* id_out);
std::shared_ptr<TpComm> make_rccl_comm(int rank, int world, const void* unique_id);

// Node-local communicator without RCCL (tp_comm.cc, ShmComm): one PROCESS per rank, bootstrap
// through the POSIX shared-memory segment "/<name>" (created by rank 0), hipIpc mailboxes for the
// one-shot exchanges, host-staged fallback collectives. Ranks may share a device.
std::shared_ptr<TpComm> make_shm_comm(int rank, int world, const std::string& name);

// `world` communicators sharing one rendezvous object; comm i must be used by exactly one thread.
std::vector<std::shared_ptr<TpComm>> make_local_comms(int world);

}  // namespace ct2x
