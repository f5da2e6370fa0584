// Host runtime of the MI355X backend: error plumbing, per-thread HIP stream, caching device
// allocator, events. Mirrors the contracts of the reference's CUDA runtime glue
// (src/cuda/utils.cc:75-130 stream-per-thread; src/cuda/allocator.cc:34-185 caching
// allocator; src/cuda/utils.h:68-96 error macros) without any of its code.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <stdexcept>
#include <string>

namespace ct2x {

class Error : public std::runtime_error {
public:
  explicit Error(const std::string& msg) : std::runtime_error(msg) {}
};

#define CT2X_HIP_CHECK(expr)                                                              \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess)                                                                 \
      throw ::ct2x::Error(std::string("HIP failure in " #expr ": ") +                     \
                          hipGetErrorString(_e) + " (" __FILE__ ":" +                     \
                          std::to_string(__LINE__) + ")");                                \
  } while (0)

#define CT2X_CHECK(cond, msg)                                                             \
  do {                                                                                    \
    if (!(cond)) throw ::ct2x::Error(std::string(msg));                                   \
  } while (0)

void set_last_error(const std::string& msg);
const char* last_error();

// The calling thread's stream (created on first use on the current device).
hipStream_t thread_stream();
hipStream_t thread_stream_with_priority(int level);
inline hipStream_t resolve_stream(void* s) {
  return s ? static_cast<hipStream_t>(s) : thread_stream();
}

// The stream the calling thread is enqueueing on right now: the engine sets it while it drives a
// session on that session's stream; otherwise it is thread_stream(). The allocator orders the
// reuse of freed blocks against it.
void set_active_stream(hipStream_t stream);  // nullptr: back to thread_stream()
hipStream_t active_stream();

// Stream-ordered caching allocator (the contract of the reference's CUDA allocator,
// src/cuda/allocator.cc:34-185): blocks are recycled per size class and never returned to the
// driver until clear_cache(). device_free() may be called while work that uses the block is still
// queued on the calling thread's active stream: the block remembers that stream and an event
// recorded behind the queued work; it is handed out again either to a request on the SAME stream
// (stream order makes that safe) or, to any other stream, only once the event has completed.
void* device_malloc(size_t bytes);
void device_free(void* ptr);
void clear_cache();
int64_t allocated_bytes();

// Page-locked host staging buffers (the read-back slots of the decoding loops), recycled by size
// class (powers of two) and returned to the driver by clear_cache() only: hipHostMalloc /
// hipHostFree cost 100 - 250 us each on this stack - a decoding session took four of them, 0.9 ms
// of a 21 ms Transformer-base call in its destructor alone (measured). The caller
// makes sure no copy into the buffer is in flight when it releases it.
void* pinned_malloc(size_t bytes);
void pinned_free(void* ptr);

// Experiment switches (kernel-variant A/B knobs of tools/*): read only by a library built with
// `make TUNING=1`; the shipped library ignores them (returns nullptr) and runs the measured winners.
inline const char* tuning_env(const char* name) {
#ifdef CT2X_TUNING
  return std::getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

// Launch bookkeeping used by the engine to report kernels/step.
void check_last_launch(const char* what);

}  // namespace ct2x
