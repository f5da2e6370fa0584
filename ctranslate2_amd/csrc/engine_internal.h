// Shared by the translation units of the engine (engine.cc: loading and building blocks;
// engine_session.cc: sessions, encoder and sequence passes; engine_step.cc: the decoding step).
#pragma once

#include <algorithm>
#include <atomic>
#include <cstdint>

#include "engine.h"
#include "runtime.h"

namespace ct2x {

// The stream of the session the calling thread is driving (one worker thread per session) and the
// begin_kernel calls of this thread (kernels per step). Defined in engine.cc.
extern std::atomic<int> g_debug_beam_ancestry;  // engine_session.cc (tests: 0 = no ancestry table)
extern thread_local hipStream_t tls_stream;
extern thread_local int64_t tls_launches;

// Device memory owned by a session (released with it through the caching allocator).
template <typename T>
T* alloc(DecodeState& st, int64_t count) {
  void* p = device_malloc(static_cast<size_t>(std::max<int64_t>(count, 1)) * sizeof(T));
  st.owned.push_back(p);
  return static_cast<T*>(p);
}

}  // namespace ct2x
