// TEST INFRASTRUCTURE ONLY (oracle/_ref build): see ../spdlog.h
#pragma once
#include "../spdlog.h"
namespace spdlog {
  inline std::shared_ptr<logger> stderr_logger_mt(const std::string&) {
    return std::make_shared<logger>();
  }
}
