// TEST INFRASTRUCTURE ONLY (oracle/_ref build).
// Minimal no-op stand-in for the spdlog API surface that the reference's CPU
// sources use (third_party/spdlog is an empty submodule in /root/reference).
// Written from the call sites (src/logging.cc:12-81, src/utils.cc, src/models/model.cc).
#pragma once
#include <memory>
#include <mutex>
#include <string>
#include <thread>

namespace spdlog {
  namespace level {
    enum level_enum { trace = 0, debug = 1, info = 2, warn = 3, err = 4, critical = 5, off = 6 };
  }

  class logger {
  public:
    template <typename... Args> void set_pattern(Args&&...) {}
  };

  inline level::level_enum& _level_ref() {
    static level::level_enum lvl = level::warn;
    return lvl;
  }
  inline void set_level(level::level_enum l) { _level_ref() = l; }
  inline level::level_enum get_level() { return _level_ref(); }
  inline bool should_log(level::level_enum l) { return l >= _level_ref(); }
  inline void set_default_logger(std::shared_ptr<logger>) {}

  template <typename... Args> inline void trace(Args&&...) {}
  template <typename... Args> inline void debug(Args&&...) {}
  template <typename... Args> inline void info(Args&&...) {}
  template <typename... Args> inline void warn(Args&&...) {}
  template <typename... Args> inline void error(Args&&...) {}
  template <typename... Args> inline void critical(Args&&...) {}
}
