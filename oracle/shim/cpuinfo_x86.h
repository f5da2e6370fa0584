// TEST INFRASTRUCTURE ONLY (oracle/_ref build).
// Stand-in for google/cpu_features' cpuinfo_x86.h (empty submodule in the
// reference checkout); provides exactly the fields src/cpu/cpu_info.cc:12-41 reads.
#pragma once
#include <cpuid.h>
#include <cstring>

#define CPU_FEATURES_VENDOR_GENUINE_INTEL "GenuineIntel"
#define CPU_FEATURES_VENDOR_AUTHENTIC_AMD "AuthenticAMD"

namespace cpu_features {
  struct X86Features {
    int sse4_1, avx, avx2, avx512f, avx512cd, avx512vl, avx512dq, avx512bw;
  };
  struct X86Info {
    X86Features features;
    char vendor[13];
  };
  inline X86Info GetX86Info() {
    X86Info info;
    std::memset(&info, 0, sizeof(info));
    unsigned eax = 0, ebx = 0, ecx = 0, edx = 0;
    __get_cpuid(0, &eax, &ebx, &ecx, &edx);
    std::memcpy(info.vendor + 0, &ebx, 4);
    std::memcpy(info.vendor + 4, &edx, 4);
    std::memcpy(info.vendor + 8, &ecx, 4);
    info.vendor[12] = 0;
    __builtin_cpu_init();
    info.features.sse4_1 = __builtin_cpu_supports("sse4.1");
    info.features.avx = __builtin_cpu_supports("avx");
    info.features.avx2 = __builtin_cpu_supports("avx2");
    info.features.avx512f = __builtin_cpu_supports("avx512f");
    info.features.avx512cd = __builtin_cpu_supports("avx512cd");
    info.features.avx512vl = __builtin_cpu_supports("avx512vl");
    info.features.avx512dq = __builtin_cpu_supports("avx512dq");
    info.features.avx512bw = __builtin_cpu_supports("avx512bw");
    return info;
  }
}
