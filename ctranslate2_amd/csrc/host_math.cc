#include "host_math.h"

#include <dlfcn.h>

#include <cmath>
#include <cstdlib>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

namespace ct2x {

namespace {

// MKL's single dynamic library (LP64 interface: MKL_INT = int).
using VmlFn = void (*)(int n, const float* a, float* r);

struct Vml {
  VmlFn vs_sin = nullptr;
  VmlFn vs_cos = nullptr;
  bool tried = false;
};

Vml& vml() {
  static Vml v;
  static std::once_flag once;
  std::call_once(once, [] {
    v.tried = true;
    const char* mode = std::getenv("CT2X_HOST_MATH");
    const std::string want = mode ? mode : "auto";
    if (want == "libm") return;
    std::vector<std::string> dirs = {""};  // "" = the loader's own search path
    for (const char* env : {"CT2X_MKL_PATH", "MKLROOT", "CONDA_PREFIX"}) {
      if (const char* d = std::getenv(env)) {
        dirs.push_back(std::string(d) + "/");
        dirs.push_back(std::string(d) + "/lib/");
        dirs.push_back(std::string(d) + "/lib/intel64/");
      }
    }
    dirs.push_back("/opt/conda/lib/");
    dirs.push_back("/opt/intel/oneapi/mkl/latest/lib/");
    dirs.push_back("/opt/intel/oneapi/mkl/latest/lib/intel64/");
    void* h = nullptr;
    for (const auto& d : dirs) {
      for (const char* name : {"libmkl_rt.so.2", "libmkl_rt.so.1", "libmkl_rt.so"}) {
        h = dlopen((d + name).c_str(), RTLD_LAZY | RTLD_LOCAL);
        if (h) break;
      }
      if (h) break;
    }
    if (h) {
      // VML needs no threads for a few hundred kB of arguments; the sequential layer also avoids
      // pulling an OpenMP runtime into the process.
      using SetLayer = int (*)(int);
      if (auto set_layer = reinterpret_cast<SetLayer>(dlsym(h, "MKL_Set_Threading_Layer")))
        set_layer(1 /* MKL_THREADING_SEQUENTIAL */);
      v.vs_sin = reinterpret_cast<VmlFn>(dlsym(h, "vsSin"));
      v.vs_cos = reinterpret_cast<VmlFn>(dlsym(h, "vsCos"));
      if (!v.vs_sin || !v.vs_cos) v.vs_sin = v.vs_cos = nullptr;
    }
    if (want == "mkl" && !v.vs_sin)
      throw std::runtime_error("CT2X_HOST_MATH=mkl but no MKL runtime (libmkl_rt.so) was found; set "
                               "CT2X_MKL_PATH");
  });
  return v;
}

void run(VmlFn fn, float (*fallback)(float), int64_t n, const float* x, float* y) {
  if (fn) {
    constexpr int64_t kChunk = 1 << 20;
    for (int64_t i = 0; i < n; i += kChunk)
      fn(static_cast<int>(n - i < kChunk ? n - i : kChunk), x + i, y + i);
    return;
  }
  for (int64_t i = 0; i < n; ++i) y[i] = fallback(x[i]);
}

}  // namespace

void host_vsin(int64_t n, const float* x, float* y) { run(vml().vs_sin, sinf, n, x, y); }
void host_vcos(int64_t n, const float* x, float* y) { run(vml().vs_cos, cosf, n, x, y); }
const char* host_math_backend() { return vml().vs_sin ? "mkl-vml" : "libm"; }

}  // namespace ct2x
