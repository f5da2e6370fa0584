// Communicators behind the tensor-parallel Dense layers (see tp_comm.h).

#include "tp_comm.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <thread>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "kernels.h"
#include "runtime.h"

namespace ct2x {

// ---------------------------------------------------------------------------------------------
// RCCL over xGMI. The library is resolved at run time: the single-GPU product path must load on
// hosts without RCCL, and nothing here is a shim over NCCL semantics - these are RCCL's own entry
// points (RCCL keeps the nccl* symbol names).
// ---------------------------------------------------------------------------------------------
namespace {

struct RcclApi {
  using Comm = void*;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(Comm*, int, const void* /* ncclUniqueId by value, 128 bytes */, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  void* handle = nullptr;
};

// ncclDataType_t / ncclRedOp_t values (rccl.h): ncclInt32 = 2, ncclFloat32 = 7; ncclSum = 0,
// ncclMax = 2.
constexpr int kNcclInt8 = 0, kNcclInt32 = 2, kNcclFloat32 = 7, kNcclSum = 0, kNcclMax = 2;

struct UniqueIdByValue {
  char bytes[kTpUniqueIdBytes];
};

RcclApi& rccl() {
  static RcclApi api = [] {
    RcclApi a;
    // A process that already carries an RCCL (torch.distributed maps its own librccl.so) must not
    // get a second copy: first look for the library that is ALREADY loaded (RTLD_NOLOAD, sonames
    // in the order a loader would have registered them), only then load one.
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      a.handle = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
      if (a.handle) break;
    }
    if (!a.handle && dlsym(RTLD_DEFAULT, "ncclAllReduce") != nullptr)
      a.handle = dlopen(nullptr, RTLD_NOW);  // linked into the process under another path
    if (!a.handle) {
      for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1",
                               "/opt/rocm/lib/librccl.so"}) {
        a.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (a.handle) break;
      }
    }
    if (!a.handle)
      throw Error(std::string("tensor parallelism needs RCCL (librccl.so): ") + dlerror());
    auto sym = [&](const char* n) {
      void* p = dlsym(a.handle, n);
      if (!p) throw Error(std::string("librccl.so lacks ") + n);
      return p;
    };
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(sym("ncclAllReduce"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(sym("ncclAllGather"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
    return a;
  }();
  return api;
}

void rccl_check(int rc, const char* what) {
  if (rc != 0) throw Error(std::string(what) + ": " + rccl().GetErrorString(rc));
}

class RcclComm : public TpComm {
public:
  RcclComm(int rank, int world, const void* unique_id) : rank_(rank), world_(world) {
    // ncclCommInitRank takes the 128-byte id BY VALUE.
    using InitFn = int (*)(RcclApi::Comm*, int, UniqueIdByValue, int);
    UniqueIdByValue id;
    std::memcpy(id.bytes, unique_id, kTpUniqueIdBytes);
    rccl_check(reinterpret_cast<InitFn>(rccl().CommInitRank)(&comm_, world, id, rank),
               "ncclCommInitRank");
    const char* e = std::getenv("CT2X_TP_ONESHOT");
    if (world > 1 && world <= kTpMaxWorld && !(e && e[0] == '0')) {
      try {
        setup_peer_boxes();
      } catch (const std::exception&) {
        // peer mapping refused (no xGMI / IPC between these devices): RCCL collectives only
        (void)hipGetLastError();
        has_box_ = false;
      }
    }
  }
  ~RcclComm() override {
    for (int r = 0; r < world_; ++r)
      if (r != rank_ && has_box_ && pb_.box[r]) (void)hipIpcCloseMemHandle(pb_.box[r]);
    if (own_box_) (void)hipFree(own_box_);
    if (comm_) (void)rccl().CommDestroy(comm_);
  }
  const TpPeerBox* peer_box() const override { return has_box_ ? &pb_ : nullptr; }
  int rank() const override { return rank_; }
  int world() const override { return world_; }
  const char* backend() const override { return "rccl"; }
  bool capturable() const override { return true; }  // RCCL collectives are stream-ordered
  void all_gather(const void* send, void* recv, size_t bytes, hipStream_t stream) override {
    rccl_check(rccl().AllGather(send, recv, bytes, kNcclInt8, comm_, stream), "ncclAllGather");
  }
  void all_reduce_sum_i32(int32_t* buf, int64_t count, hipStream_t stream) override {
    rccl_check(rccl().AllReduce(buf, buf, count, kNcclInt32, kNcclSum, comm_, stream),
               "ncclAllReduce(int32, sum)");
  }
  void all_reduce_sum_f32(float* buf, int64_t count, hipStream_t stream) override {
    rccl_check(rccl().AllReduce(buf, buf, count, kNcclFloat32, kNcclSum, comm_, stream),
               "ncclAllReduce(float, sum)");
  }
  void all_reduce_max_f32(float* buf, int64_t count, hipStream_t stream) override {
    rccl_check(rccl().AllReduce(buf, buf, count, kNcclFloat32, kNcclMax, comm_, stream),
               "ncclAllReduce(float, max)");
  }

private:
  // One mailbox per rank in fine-grained device memory (peer stores must become visible without a
  // kernel boundary), its IPC handle all-gathered over the communicator that was just created,
  // every peer's box opened and mapped. Any refusal leaves the communicator on RCCL collectives.
  void setup_peer_boxes() {
    const size_t bytes = tp_box_bytes();
    CT2X_HIP_CHECK(hipExtMallocWithFlags(reinterpret_cast<void**>(&own_box_), bytes,
                                         hipDeviceMallocFinegrained));
    CT2X_HIP_CHECK(hipMemset(own_box_, 0, bytes));
    hipIpcMemHandle_t mine;
    CT2X_HIP_CHECK(hipIpcGetMemHandle(&mine, own_box_));
    void *d_send = nullptr, *d_recv = nullptr;
    CT2X_HIP_CHECK(hipMalloc(&d_send, sizeof(mine)));
    CT2X_HIP_CHECK(hipMalloc(&d_recv, sizeof(mine) * world_));
    CT2X_HIP_CHECK(hipMemcpy(d_send, &mine, sizeof(mine), hipMemcpyHostToDevice));
    rccl_check(rccl().AllGather(d_send, d_recv, sizeof(mine), kNcclInt8, comm_, nullptr), "ncclAllGather(ipc)");
    CT2X_HIP_CHECK(hipStreamSynchronize(nullptr));
    std::vector<hipIpcMemHandle_t> all(world_);
    CT2X_HIP_CHECK(hipMemcpy(all.data(), d_recv, sizeof(mine) * world_, hipMemcpyDeviceToHost));
    (void)hipFree(d_send);
    (void)hipFree(d_recv);
    pb_.rank = rank_;
    pb_.world = world_;
    for (int r = 0; r < world_; ++r) {
      if (r == rank_) {
        pb_.box[r] = own_box_;
      } else {
        void* p = nullptr;
        CT2X_HIP_CHECK(hipIpcOpenMemHandle(&p, all[r], hipIpcMemLazyEnablePeerAccess));
        pb_.box[r] = static_cast<char*>(p);
      }
    }
    has_box_ = true;
  }

  int rank_, world_;
  RcclApi::Comm comm_ = nullptr;
  char* own_box_ = nullptr;
  bool has_box_ = false;
  TpPeerBox pb_;
};

// ---------------------------------------------------------------------------------------------
// In-process communicator: `world` ranks = `world` host threads driving their own streams on ONE
// device. An all-reduce is two host rendezvous around a rank-order combine kernel:
//   1. every rank publishes its buffer and an event recorded on its stream, then waits (host);
//   2. every rank makes its stream wait for the other ranks' events and combines ALL buffers, in
//      rank order (identical result on every rank), into a private scratch buffer; records a
//      second event; waits (host);
//   3. every rank makes its stream wait for the others' second events (nobody still reads its
//      buffer) and copies the scratch back in place.
// ---------------------------------------------------------------------------------------------
struct LocalShared {
  explicit LocalShared(int w) : world(w), bufs(w, nullptr), ready(w, nullptr), done(w, nullptr) {}
  int world;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  uint64_t generation = 0;
  std::vector<void*> bufs;
  std::vector<hipEvent_t> ready, done;
  bool poisoned = false;
  char* boxes[kTpMaxWorld] = {nullptr};  // one-shot mailboxes of the ranks (same device)
  ~LocalShared() {
    for (char* b : boxes)
      if (b) (void)hipFree(b);
  }

  void barrier() {
    std::unique_lock<std::mutex> lock(mu);
    const uint64_t gen = generation;
    if (++arrived == world) {
      arrived = 0;
      ++generation;
      cv.notify_all();
    } else {
      // A rank that failed elsewhere never arrives: give up instead of hanging the test run.
      if (!cv.wait_for(lock, std::chrono::seconds(60), [&] { return generation != gen; })) {
        poisoned = true;
        cv.notify_all();
        throw Error("in-process tensor-parallel communicator: a rank did not reach the collective");
      }
    }
    if (poisoned) throw Error("in-process tensor-parallel communicator: aborted");
  }
};

class LocalComm : public TpComm {
public:
  LocalComm(std::shared_ptr<LocalShared> shared, int rank) : s_(std::move(shared)), rank_(rank) {
    CT2X_HIP_CHECK(hipEventCreateWithFlags(&s_->ready[rank], hipEventDisableTiming));
    CT2X_HIP_CHECK(hipEventCreateWithFlags(&s_->done[rank], hipEventDisableTiming));
  }
  // (the boxes are allocated by make_local_comms once every rank exists)
  void set_boxes() {
    pb_.rank = rank_;
    pb_.world = s_->world;
    for (int r = 0; r < s_->world; ++r) pb_.box[r] = s_->boxes[r];
    has_box_ = s_->boxes[0] != nullptr;
  }
  const TpPeerBox* peer_box() const override { return has_box_ ? &pb_ : nullptr; }
  ~LocalComm() override {
    if (scratch_) { try { device_free(scratch_); } catch (...) {} }
    (void)hipEventDestroy(s_->ready[rank_]);
    (void)hipEventDestroy(s_->done[rank_]);
  }
  int rank() const override { return rank_; }
  int world() const override { return s_->world; }
  const char* backend() const override { return "in-process"; }
  bool capturable() const override { return false; }  // host rendezvous inside every collective
  void all_gather(const void* send, void* recv, size_t bytes, hipStream_t stream) override {
    s_->bufs[rank_] = const_cast<void*>(send);
    CT2X_HIP_CHECK(hipEventRecord(s_->ready[rank_], stream));
    s_->barrier();
    for (int r = 0; r < s_->world; ++r) {
      if (r != rank_) CT2X_HIP_CHECK(hipStreamWaitEvent(stream, s_->ready[r], 0));
      CT2X_HIP_CHECK(hipMemcpyAsync(static_cast<char*>(recv) + static_cast<size_t>(r) * bytes,
                                    s_->bufs[r], bytes, hipMemcpyDeviceToDevice, stream));
    }
    CT2X_HIP_CHECK(hipEventRecord(s_->done[rank_], stream));
    s_->barrier();
    // nobody may overwrite its send buffer before every rank has copied it
    for (int r = 0; r < s_->world; ++r)
      if (r != rank_) CT2X_HIP_CHECK(hipStreamWaitEvent(stream, s_->done[r], 0));
    s_->barrier();
  }
  void all_reduce_sum_i32(int32_t* buf, int64_t count, hipStream_t stream) override {
    run(buf, count, true, false, stream);
  }
  void all_reduce_sum_f32(float* buf, int64_t count, hipStream_t stream) override {
    run(buf, count, false, false, stream);
  }
  void all_reduce_max_f32(float* buf, int64_t count, hipStream_t stream) override {
    run(buf, count, false, true, stream);
  }

private:
  void run(void* buf, int64_t count, bool is_int32, bool is_max, hipStream_t stream) {
    const size_t bytes = static_cast<size_t>(count) * 4;
    if (bytes > scratch_bytes_) {
      if (scratch_) device_free(scratch_);
      scratch_ = device_malloc(bytes);
      scratch_bytes_ = bytes;
    }
    s_->bufs[rank_] = buf;
    CT2X_HIP_CHECK(hipEventRecord(s_->ready[rank_], stream));
    s_->barrier();
    for (int r = 0; r < s_->world; ++r)
      if (r != rank_) CT2X_HIP_CHECK(hipStreamWaitEvent(stream, s_->ready[r], 0));
    launch_reduce_ranks(s_->bufs.data(), s_->world, count, is_int32, is_max, scratch_, stream);
    CT2X_HIP_CHECK(hipEventRecord(s_->done[rank_], stream));
    s_->barrier();
    for (int r = 0; r < s_->world; ++r)
      if (r != rank_) CT2X_HIP_CHECK(hipStreamWaitEvent(stream, s_->done[r], 0));
    CT2X_HIP_CHECK(hipMemcpyAsync(buf, scratch_, bytes, hipMemcpyDeviceToDevice, stream));
    s_->barrier();  // bufs[] / events may be reused by the next collective
  }

  std::shared_ptr<LocalShared> s_;
  int rank_;
  void* scratch_ = nullptr;
  size_t scratch_bytes_ = 0;
  TpPeerBox pb_;
  bool has_box_ = false;
};


// ---------------------------------------------------------------------------------------------
// Node-local communicator WITHOUT RCCL: one process per rank, bootstrap and the (rare) large
// collectives through a POSIX shared-memory segment, the decode-step exchanges through the same
// hipIpc peer-mapped mailboxes and one-shot kernels as the RCCL communicator. Two uses: hosts
// where RCCL is not installed, and - what the single-GPU test tier runs - TWO PROCESSES ON ONE
// GPU: RCCL refuses two ranks on one device, this communicator does not, so the real
// cross-process path (hipIpcGetMemHandle / hipIpcOpenMemHandle, system-scope granule stores into
// another process's mailbox, polls inside kernels of two different processes that run
// concurrently) is exercised where only one GPU is visible.
// Segment layout: header (barrier words, one IPC handle per rank) | world x kShmChunk bytes.
// The fallback collectives stage through host memory chunk by chunk (D2H, barrier, combine in
// rank order on every rank, barrier, H2D): slow, deterministic, identical on every rank; they
// serve the shapes outside the one-shot envelope (encoder / prefill rows), never a decode step.
// ---------------------------------------------------------------------------------------------
constexpr size_t kShmChunk = 8u << 20;
struct ShmHeader {
  std::atomic<uint32_t> ready;       // rank 0 has initialised the segment
  std::atomic<uint32_t> arrived;     // barrier: arrivals of the running generation
  std::atomic<uint32_t> generation;  // barrier: bumped by the last arriver
  std::atomic<uint32_t> poisoned;    // a rank gave up
  hipIpcMemHandle_t handles[kTpMaxWorld];
  // bootstrap handshake: rank r writes a token of its own making into hello[r], rank 0 echoes it
  // into ack[r]. A rank that opened a STALE segment of the same name (left by a crashed run: its
  // `ready` word is already set) never sees its token come back and opens the name again.
  std::atomic<uint64_t> hello[kTpMaxWorld];
  std::atomic<uint64_t> ack[kTpMaxWorld];
  char pad[4096 - 16 - sizeof(hipIpcMemHandle_t) * kTpMaxWorld - 16 * kTpMaxWorld];
};
static_assert(sizeof(ShmHeader) == 4096, "segment header");

class ShmComm : public TpComm {
public:
  ShmComm(int rank, int world, const std::string& name) : rank_(rank), world_(world), name_("/" + name) {
    if (world < 2 || world > kTpMaxWorld) throw Error("shared-memory communicator: world size must be 2..8");
    bytes_ = sizeof(ShmHeader) + static_cast<size_t>(world) * kShmChunk;
    const auto t0 = std::chrono::steady_clock::now();
    auto expired = [&] { return std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120); };
    auto map_segment = [&](int fd) {
      void* p = mmap(nullptr, bytes_, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
      close(fd);
      if (p == MAP_FAILED) throw Error("shared-memory communicator: mmap failed");
      hdr_ = static_cast<ShmHeader*>(p);
      data_ = static_cast<char*>(p) + sizeof(ShmHeader);
    };
    if (rank == 0) {
      (void)shm_unlink(name_.c_str());
      const int fd = shm_open(name_.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
      if (fd < 0 || ftruncate(fd, static_cast<off_t>(bytes_)) != 0)
        throw Error("shared-memory communicator: cannot create " + name_);
      map_segment(fd);
      // (a fresh segment is zero filled)
      hdr_->ready.store(0x43543258u, std::memory_order_release);
      // echo every peer's token: only then is it known to sit on THIS run's segment
      int acked = 0;
      std::vector<bool> done(world, false);
      while (acked < world - 1) {
        for (int r = 1; r < world; ++r) {
          if (done[r]) continue;
          const uint64_t token = hdr_->hello[r].load(std::memory_order_acquire);
          if (token != 0) {
            hdr_->ack[r].store(token, std::memory_order_release);
            done[r] = true;
            ++acked;
          }
        }
        if (acked < world - 1) {
          if (expired()) throw Error("shared-memory communicator: a rank never opened " + name_);
          std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
      }
    } else {
      const uint64_t token = ((static_cast<uint64_t>(getpid()) << 32) ^
                              static_cast<uint64_t>(std::chrono::steady_clock::now().time_since_epoch().count())) | 1u;
      for (;;) {
        if (expired()) throw Error("shared-memory communicator: rank 0 never created " + name_);
        struct stat st;
        const int fd = shm_open(name_.c_str(), O_RDWR, 0600);
        if (fd < 0 || fstat(fd, &st) != 0 || static_cast<size_t>(st.st_size) < bytes_) {
          if (fd >= 0) close(fd);
          std::this_thread::sleep_for(std::chrono::milliseconds(5));
          continue;
        }
        map_segment(fd);
        // this run's segment answers within moments of rank 0 reaching its handshake loop; a stale
        // one (or one rank 0 is about to replace) never does: let go of it and open the name again
        const auto t1 = std::chrono::steady_clock::now();
        bool ok = false, said_hello = false;
        while (std::chrono::steady_clock::now() - t1 < std::chrono::milliseconds(500)) {
          if (!said_hello && hdr_->ready.load(std::memory_order_acquire) == 0x43543258u) {
            hdr_->hello[rank].store(token, std::memory_order_release);
            said_hello = true;
          }
          if (said_hello && hdr_->ack[rank].load(std::memory_order_acquire) == token) {
            ok = true;
            break;
          }
          std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        if (ok) break;
        munmap(hdr_, bytes_);
        hdr_ = nullptr;
        data_ = nullptr;
      }
    }
    const char* e = std::getenv("CT2X_TP_ONESHOT");
    bool want_boxes = !(e && e[0] == '0');
    // Every rank publishes a handle (or zeros) and takes part in both barriers, whatever happens
    // to its own allocation: a refusal on one rank must not leave the others waiting.
    hipIpcMemHandle_t mine;
    std::memset(&mine, 0, sizeof(mine));
    if (want_boxes) {
      const size_t bytes = tp_box_bytes();
      if (hipExtMallocWithFlags(reinterpret_cast<void**>(&own_box_), bytes, hipDeviceMallocFinegrained) != hipSuccess ||
          hipMemset(own_box_, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
          hipIpcGetMemHandle(&mine, own_box_) != hipSuccess) {
        (void)hipGetLastError();
        std::memset(&mine, 0, sizeof(mine));
        want_boxes = false;
      }
    }
    hdr_->handles[rank] = mine;
    barrier();
    pb_.rank = rank;
    pb_.world = world;
    bool ok = want_boxes;
    const hipIpcMemHandle_t zero = {};
    for (int r = 0; r < world && ok; ++r) {
      if (r == rank) {
        pb_.box[r] = own_box_;
      } else if (std::memcmp(&hdr_->handles[r], &zero, sizeof(zero)) == 0) {
        ok = false;
      } else {
        void* q = nullptr;
        if (hipIpcOpenMemHandle(&q, hdr_->handles[r], hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
          (void)hipGetLastError();
          ok = false;
        } else {
          pb_.box[r] = static_cast<char*>(q);
          opened_[r] = true;
        }
      }
    }
    // all ranks or none: a rank without mailboxes would take the collective path alone
    if (!ok) hdr_->poisoned.fetch_or(2u);
    barrier();
    has_box_ = ok && (hdr_->poisoned.load() & 2u) == 0;
    barrier();
    if (rank == 0) (void)shm_unlink(name_.c_str());  // the mapping outlives the name
  }
  ~ShmComm() override {
    for (int r = 0; r < world_; ++r)
      if (opened_[r]) (void)hipIpcCloseMemHandle(pb_.box[r]);
    if (own_box_) (void)hipFree(own_box_);
    if (hdr_) munmap(hdr_, bytes_);
  }
  const TpPeerBox* peer_box() const override { return has_box_ ? &pb_ : nullptr; }
  int rank() const override { return rank_; }
  int world() const override { return world_; }
  const char* backend() const override { return "shared-memory"; }
  bool capturable() const override { return false; }  // host staging inside every collective
  void all_gather(const void* send, void* recv, size_t bytes, hipStream_t stream) override {
    for (size_t off = 0; off < bytes; off += kShmChunk) {
      const size_t n = std::min(kShmChunk, bytes - off);
      CT2X_HIP_CHECK(hipMemcpyAsync(data_ + rank_ * kShmChunk, static_cast<const char*>(send) + off, n,
                                    hipMemcpyDeviceToHost, stream));
      CT2X_HIP_CHECK(hipStreamSynchronize(stream));
      barrier();
      for (int r = 0; r < world_; ++r)
        CT2X_HIP_CHECK(hipMemcpyAsync(static_cast<char*>(recv) + static_cast<size_t>(r) * bytes + off,
                                      data_ + r * kShmChunk, n, hipMemcpyHostToDevice, stream));
      CT2X_HIP_CHECK(hipStreamSynchronize(stream));
      barrier();
    }
  }
  void all_reduce_sum_i32(int32_t* buf, int64_t count, hipStream_t stream) override {
    reduce<int32_t>(buf, count, stream, [](int32_t a, int32_t b) { return a + b; });
  }
  void all_reduce_sum_f32(float* buf, int64_t count, hipStream_t stream) override {
    reduce<float>(buf, count, stream, [](float a, float b) { return a + b; });
  }
  void all_reduce_max_f32(float* buf, int64_t count, hipStream_t stream) override {
    reduce<float>(buf, count, stream, [](float a, float b) { return a > b ? a : b; });
  }

private:
  template <typename F>
  void wait_for(F&& cond, const char* what) {
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (!cond()) {
      if (hdr_->poisoned.load() & 1u) throw Error("shared-memory communicator: aborted by a peer");
      if ((++spins & 1023u) == 0) {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) {
          hdr_->poisoned.fetch_or(1u);
          throw Error(std::string("shared-memory communicator: ") + what);
        }
        std::this_thread::yield();
      }
    }
  }
  void barrier() {
    const uint32_t gen = hdr_->generation.load(std::memory_order_acquire);
    if (hdr_->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == static_cast<uint32_t>(world_)) {
      hdr_->arrived.store(0, std::memory_order_relaxed);
      hdr_->generation.store(gen + 1, std::memory_order_release);
    } else {
      wait_for([&] { return hdr_->generation.load(std::memory_order_acquire) != gen; },
               "a rank did not reach the collective");
    }
  }
  template <typename T, typename Op>
  void reduce(T* buf, int64_t count, hipStream_t stream, Op op) {
    const int64_t per = static_cast<int64_t>(kShmChunk / sizeof(T));
    std::vector<T> out;
    for (int64_t off = 0; off < count; off += per) {
      const int64_t n = std::min(per, count - off);
      CT2X_HIP_CHECK(hipMemcpyAsync(data_ + rank_ * kShmChunk, buf + off, n * sizeof(T),
                                    hipMemcpyDeviceToHost, stream));
      CT2X_HIP_CHECK(hipStreamSynchronize(stream));
      barrier();
      out.resize(n);
      const T* first = reinterpret_cast<const T*>(data_);
      for (int64_t i = 0; i < n; ++i) out[i] = first[i];
      for (int r = 1; r < world_; ++r) {  // rank order: the same value on every rank
        const T* src = reinterpret_cast<const T*>(data_ + r * kShmChunk);
        for (int64_t i = 0; i < n; ++i) out[i] = op(out[i], src[i]);
      }
      barrier();  // everyone has read the chunk before the next one overwrites it
      CT2X_HIP_CHECK(hipMemcpyAsync(buf + off, out.data(), n * sizeof(T), hipMemcpyHostToDevice, stream));
      CT2X_HIP_CHECK(hipStreamSynchronize(stream));
    }
  }

  int rank_, world_;
  std::string name_;
  size_t bytes_ = 0;
  ShmHeader* hdr_ = nullptr;
  char* data_ = nullptr;
  char* own_box_ = nullptr;
  bool has_box_ = false;
  bool opened_[kTpMaxWorld] = {false};
  TpPeerBox pb_;
};

}  // namespace

std::shared_ptr<TpComm> make_shm_comm(int rank, int world, const std::string& name) {
  return std::make_shared<ShmComm>(rank, world, name);
}

void rccl_get_unique_id(void* id_out) {
  rccl_check(rccl().GetUniqueId(id_out), "ncclGetUniqueId");
}

std::shared_ptr<TpComm> make_rccl_comm(int rank, int world, const void* unique_id) {
  return std::make_shared<RcclComm>(rank, world, unique_id);
}

std::vector<std::shared_ptr<TpComm>> make_local_comms(int world) {
  if (world < 1 || world > 8) throw Error("in-process communicator: world size must be 1..8");
  auto shared = std::make_shared<LocalShared>(world);
  std::vector<std::shared_ptr<TpComm>> comms;
  std::vector<std::shared_ptr<LocalComm>> locals;
  for (int r = 0; r < world; ++r) locals.push_back(std::make_shared<LocalComm>(shared, r));
  const char* e = std::getenv("CT2X_TP_ONESHOT");
  if (world == 2 && !(e && e[0] == '0')) {
    // the same mailboxes and kernels as between GPUs, on one device: what the single-GPU box can
    // verify of the one-shot exchanges (protocol, tags, parity, exactness) and time as a loopback.
    // TWO ranks only: their kernels wait for each other, so they must sit on different hardware
    // queues - guaranteed by giving the two rank threads streams of different priority
    // (thread_stream_with_priority); more in-process ranks keep the host-rendezvous collectives.
    for (int r = 0; r < world; ++r) {
      CT2X_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&shared->boxes[r]), tp_box_bytes()));
      CT2X_HIP_CHECK(hipMemset(shared->boxes[r], 0, tp_box_bytes()));
    }
    CT2X_HIP_CHECK(hipDeviceSynchronize());
  }
  for (auto& l : locals) {
    l->set_boxes();
    comms.push_back(l);
  }
  return comms;
}

}  // namespace ct2x
