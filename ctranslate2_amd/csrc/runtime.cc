#include "runtime.h"

#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace ct2x {

static thread_local std::string g_last_error;

void set_last_error(const std::string& msg) { g_last_error = msg; }
const char* last_error() { return g_last_error.c_str(); }

namespace {

struct ThreadStream {
  hipStream_t stream = nullptr;
  int device = -1;
  ~ThreadStream() {
    // Streams are intentionally leaked at thread exit when the runtime may already be
    // torn down (same choice as a process-lifetime handle).
  }
};

thread_local ThreadStream t_stream;

size_t size_class(size_t bytes) {
  if (bytes <= 256) return 256;
  if (bytes <= (1u << 20)) {  // next power of two up to 1 MiB
    size_t s = 256;
    while (s < bytes) s <<= 1;
    return s;
  }
  const size_t step = size_t(2) << 20;  // 2 MiB granules above
  return (bytes + step - 1) / step * step;
}

struct FreeBlock {
  void* ptr;
  hipStream_t stream;  // the stream the freeing thread was enqueueing on
  hipEvent_t event;    // recorded on it at free time (nullptr: nothing was pending)
};

thread_local hipStream_t t_active = nullptr;

struct Pool {
  std::mutex mutex;
  // (device, size class) -> free blocks
  std::map<std::pair<int, size_t>, std::vector<FreeBlock>> free_blocks;
  std::vector<hipEvent_t> spare_events;
  std::unordered_map<void*, std::pair<int, size_t>> live;  // ptr -> (device, class)
  int64_t allocated = 0;
};

Pool& pool() {
  static Pool* p = new Pool();  // leaked on purpose: outlives static destruction order
  return *p;
}

}  // namespace

hipStream_t thread_stream() {
  int device = 0;
  CT2X_HIP_CHECK(hipGetDevice(&device));
  if (t_stream.stream == nullptr || t_stream.device != device) {
    CT2X_HIP_CHECK(hipStreamCreateWithFlags(&t_stream.stream, hipStreamNonBlocking));
    t_stream.device = device;
  }
  return t_stream.stream;
}

// The calling thread's stream re-created at priority level `level` (0 = default, 1 = highest).
// Streams of different priorities never share a hardware queue; the in-process tensor-parallel
// double needs that (its two ranks run kernels that wait for each other: on one shared queue the
// second would sit behind the first forever).
hipStream_t thread_stream_with_priority(int level) {
  int device = 0;
  CT2X_HIP_CHECK(hipGetDevice(&device));
  int least = 0, greatest = 0;
  CT2X_HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
  const int prio = level > 0 ? greatest : 0;
  if (t_stream.stream != nullptr && t_stream.device == device) {
    int cur = 0;
    if (hipStreamGetPriority(t_stream.stream, &cur) == hipSuccess && cur == prio) return t_stream.stream;
    (void)hipStreamSynchronize(t_stream.stream);
    (void)hipStreamDestroy(t_stream.stream);
    t_stream.stream = nullptr;
  }
  CT2X_HIP_CHECK(hipStreamCreateWithPriority(&t_stream.stream, hipStreamNonBlocking, prio));
  t_stream.device = device;
  return t_stream.stream;
}

void set_active_stream(hipStream_t stream) { t_active = stream; }
hipStream_t active_stream() { return t_active ? t_active : thread_stream(); }

void* device_malloc(size_t bytes) {
  if (bytes == 0) bytes = 1;
  int device = 0;
  CT2X_HIP_CHECK(hipGetDevice(&device));
  const size_t cls = size_class(bytes);
  Pool& p = pool();
  {
    const hipStream_t mine = active_stream();
    std::lock_guard<std::mutex> lock(p.mutex);
    auto it = p.free_blocks.find({device, cls});
    if (it != p.free_blocks.end()) {
      auto& blocks = it->second;
      // newest first: same-stream blocks are reusable at once, others once their event is done
      for (size_t i = blocks.size(); i-- > 0;) {
        FreeBlock& b = blocks[i];
        const bool ready = b.event == nullptr || b.stream == mine || hipEventQuery(b.event) == hipSuccess;
        if (!ready) continue;
        void* ptr = b.ptr;
        if (b.event) p.spare_events.push_back(b.event);
        blocks.erase(blocks.begin() + static_cast<std::ptrdiff_t>(i));
        p.live[ptr] = {device, cls};
        return ptr;
      }
    }
  }
  void* ptr = nullptr;
  hipError_t e = hipMalloc(&ptr, cls);
  if (e != hipSuccess) {
    clear_cache();
    CT2X_HIP_CHECK(hipMalloc(&ptr, cls));
  }
  std::lock_guard<std::mutex> lock(p.mutex);
  p.live[ptr] = {device, cls};
  p.allocated += static_cast<int64_t>(cls);
  return ptr;
}

void device_free(void* ptr) {
  if (!ptr) return;
  Pool& p = pool();
  std::lock_guard<std::mutex> lock(p.mutex);
  auto it = p.live.find(ptr);
  if (it == p.live.end()) throw Error("ct2x_free: pointer was not allocated by ct2x_malloc");
  FreeBlock b{ptr, nullptr, nullptr};
  // Work that still uses the block may be queued on the caller's stream: remember the stream and
  // an event behind that work (no device means nothing can be pending).
  int device = 0;
  if (hipGetDevice(&device) == hipSuccess && device == it->second.first) {
    hipStream_t stream = t_active ? t_active : t_stream.stream;
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (stream != nullptr) (void)hipStreamIsCapturing(stream, &capturing);
    if (stream != nullptr && capturing == hipStreamCaptureStatusNone) {
      hipEvent_t ev = nullptr;
      if (!p.spare_events.empty()) {
        ev = p.spare_events.back();
        p.spare_events.pop_back();
      } else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
        ev = nullptr;
      }
      if (ev != nullptr && hipEventRecord(ev, stream) == hipSuccess) {
        b.stream = stream;
        b.event = ev;
      } else if (ev != nullptr) {
        p.spare_events.push_back(ev);
      }
    }
  }
  p.free_blocks[it->second].push_back(b);
  p.live.erase(it);
}

namespace {
struct PinnedPool {
  std::mutex mutex;
  std::map<size_t, std::vector<void*>> free_blocks;  // size class -> buffers
  std::map<void*, size_t> live;
};
PinnedPool& pinned_pool() {
  static PinnedPool* p = new PinnedPool();  // (never destroyed: buffers may outlive static teardown)
  return *p;
}
size_t pinned_class(size_t bytes) {
  size_t c = 4096;
  while (c < bytes) c <<= 1;
  return c;
}
}  // namespace

void* pinned_malloc(size_t bytes) {
  const size_t c = pinned_class(bytes);
  PinnedPool& p = pinned_pool();
  {
    std::lock_guard<std::mutex> lock(p.mutex);
    auto it = p.free_blocks.find(c);
    if (it != p.free_blocks.end() && !it->second.empty()) {
      void* ptr = it->second.back();
      it->second.pop_back();
      p.live[ptr] = c;
      return ptr;
    }
  }
  void* ptr = nullptr;
  CT2X_HIP_CHECK(hipHostMalloc(&ptr, c));
  std::lock_guard<std::mutex> lock(p.mutex);
  p.live[ptr] = c;
  return ptr;
}

void pinned_free(void* ptr) {
  if (ptr == nullptr) return;
  PinnedPool& p = pinned_pool();
  std::lock_guard<std::mutex> lock(p.mutex);
  auto it = p.live.find(ptr);
  if (it == p.live.end()) return;
  p.free_blocks[it->second].push_back(ptr);
  p.live.erase(it);
}

void clear_cache() {
  {
    PinnedPool& pp = pinned_pool();
    std::vector<void*> host_blocks;
    {
      std::lock_guard<std::mutex> lock(pp.mutex);
      for (auto& kv : pp.free_blocks) {
        host_blocks.insert(host_blocks.end(), kv.second.begin(), kv.second.end());
        kv.second.clear();
      }
    }
    for (void* ptr : host_blocks) (void)hipHostFree(ptr);
  }
  Pool& p = pool();
  std::vector<void*> to_free;
  {
    std::lock_guard<std::mutex> lock(p.mutex);
    for (auto& kv : p.free_blocks) {
      for (const FreeBlock& b : kv.second) {
        to_free.push_back(b.ptr);
        if (b.event) p.spare_events.push_back(b.event);
        p.allocated -= static_cast<int64_t>(kv.first.second);
      }
      kv.second.clear();
    }
  }
  if (!to_free.empty()) {
    (void)hipDeviceSynchronize();
    for (void* ptr : to_free) (void)hipFree(ptr);
  }
}

int64_t allocated_bytes() {
  Pool& p = pool();
  std::lock_guard<std::mutex> lock(p.mutex);
  return p.allocated;
}

void check_last_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess)
    throw Error(std::string("kernel launch failed (") + what + "): " + hipGetErrorString(e));
}

}  // namespace ct2x
