// What a meeting of an eight-workgroup cluster costs on the MI355X by WHERE its members sit and HOW
// the slots are stored (the decoder-stack launch, decoder_cluster.hip, meets 42 times per step):
//
//   placement 0: member m of a cluster is workgroup 8 * cluster + m - the dispatcher deals
//                consecutive workgroups to consecutive XCDs, so the members sit on EIGHT XCDs and a
//                slot can only be handed over at the memory side (sc1 store, sc1 load);
//   placement 1: the members of a cluster share ONE XCD (cluster = b % 8 + 8 * (b / 64), member =
//                b / 8 % 8): the XCD's L2 is the point of coherence for them - a plain (or sc0)
//                store leaves the line in that L2 and an L1-bypassing load is served from it.
//
// Every round each member publishes 64 tagged 8-byte slots {tag, value} (threads 0 .. 63), then
// all 512 threads poll one slot each until it carries the round's tag and check the value (a stale
// or torn slot is counted). Optionally every thread also requests `stream` 16-byte lines of a big
// buffer per round, in four parts between the polls, as the launch does with its weights.
//
//   hipcc --offload-arch=gfx950 -O3 -o build/tools/xcd_local_handover_probe tools/xcd_local_handover_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); exit(1);} } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));

struct Ctl {
  unsigned stale, timeouts, xcd_mismatch, polls;
  unsigned long long sink;
};

constexpr unsigned kSpinLimit = 1u << 14;

template <int STORE>  // 0 = agent-scope atomic store (sc1), 1 = plain store, 2 = sc0 store
__device__ __forceinline__ void put(unsigned long long* p, unsigned long long v) {
  if (STORE == 0) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else if (STORE == 1) {
    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
  } else {
    asm volatile("global_store_dwordx2 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
  }
}

template <int STORE>
__global__ __launch_bounds__(512) void meetings(Ctl* ctl, unsigned long long* slots, const v4i* stream_buf,
                                                size_t stream_lines, int rounds, int placement, int stream,
                                                unsigned base) {
  const int b = blockIdx.x, t = threadIdx.x;
  int cluster, member;
  if (placement == 0) {
    cluster = b / 8;
    member = b % 8;
  } else {
    cluster = b % 8 + 8 * (b / 64);
    member = (b / 8) % 8;
  }
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 0xf;
  if (t == 0 && static_cast<int>(xcc) != b % 8) atomicAdd(&ctl->xcd_mismatch, 1u);
  unsigned long long* mine = slots + static_cast<size_t>(cluster) * 1024;  // two areas of 512 slots, alternating
  v4i acc = {0, 0, 0, 0};
  size_t line = (static_cast<size_t>(b) * 512 + t) % stream_lines;
  unsigned polls = 0;
  for (int r = 0; r < rounds; ++r) {
    if (__hip_atomic_load(&ctl->timeouts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;  // (a lost member: everybody leaves)
    const unsigned tag = base + static_cast<unsigned>(r) + 1u;
    unsigned long long* area = mine + (r & 1) * 512;
    if (t < 64) {
      const unsigned val = tag * 2654435761u + static_cast<unsigned>(member * 64 + t);
      put<STORE>(area + member * 64 + t, (static_cast<unsigned long long>(tag) << 32) | val);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    unsigned long long u = 0;
    bool ok = false;
    unsigned spins = 0;
    int part = 0;
    while (!ok) {
      if (part < 4) {
        for (int i = 0; i < stream / 4; ++i) {
          const v4i w = __builtin_nontemporal_load(stream_buf + line);
          acc += w;
          line += 512 * 256;
          if (line >= stream_lines) line -= stream_lines;
        }
        ++part;
      } else {
        __builtin_amdgcn_s_sleep(1);
      }
      u = __hip_atomic_load(area + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ++polls;
      ok = static_cast<unsigned>(u >> 32) == tag;
      if (!ok && (++spins > kSpinLimit ||
                  ((spins & 255) == 0 && __hip_atomic_load(&ctl->timeouts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0))) {
        atomicAdd(&ctl->timeouts, 1u);
        break;
      }
    }
    if (ok && static_cast<unsigned>(u) != tag * 2654435761u + static_cast<unsigned>(t)) atomicAdd(&ctl->stale, 1u);
  }
  if (t == 0 && b == 0) ctl->polls = polls;
  if (acc.x + acc.y + acc.z + acc.w == 0x7fffffff) ctl->sink = 1;
}

template <int STORE>
static void run(const char* label, Ctl* ctl, unsigned long long* slots, const v4i* buf, size_t lines, int placement,
                int stream, unsigned& base) {
  const int rounds = 2000;
  CK(hipMemset(ctl, 0, sizeof(Ctl)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  meetings<STORE><<<256, 512>>>(ctl, slots, buf, lines, 50, placement, stream, base);  // warm
  base += 50;
  CK(hipDeviceSynchronize());
  CK(hipMemset(ctl, 0, sizeof(Ctl)));
  CK(hipEventRecord(e0));
  meetings<STORE><<<256, 512>>>(ctl, slots, buf, lines, rounds, placement, stream, base);
  base += rounds;
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  Ctl h;
  CK(hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
  printf("%-44s placement %d stream %2d lines: %.3f us per meeting, polls/meeting %.2f, stale %u timeouts %u xcd-mismatch %u\n",
         label, placement, stream, ms * 1000.0 / rounds, static_cast<double>(h.polls) / rounds, h.stale, h.timeouts,
         h.xcd_mismatch);
}

int main() {
  Ctl* ctl;
  unsigned long long* slots;
  v4i* buf;
  const size_t lines = (64u << 20) / 16;  // 64 MB: does not fit an L2, fits the Infinity Cache
  CK(hipMalloc(&ctl, sizeof(Ctl)));
  CK(hipMalloc(&slots, sizeof(unsigned long long) * 1024 * 32));
  CK(hipMalloc(&buf, lines * 16));
  CK(hipMemset(slots, 0, sizeof(unsigned long long) * 1024 * 32));
  CK(hipMemset(buf, 1, lines * 16));
  unsigned base = 0;
  for (int stream : {0, 16}) {
    for (int rep = 0; rep < 2; ++rep) {
      run<0>("sc1 store, members on eight XCDs (today)", ctl, slots, buf, lines, 0, stream, base);
      run<0>("sc1 store, members on one XCD", ctl, slots, buf, lines, 1, stream, base);
      run<1>("plain store, members on one XCD", ctl, slots, buf, lines, 1, stream, base);
      run<2>("sc0 store, members on one XCD", ctl, slots, buf, lines, 1, stream, base);
      run<1>("plain store, members on eight XCDs (invalid)", ctl, slots, buf, lines, 0, stream, base);
    }
  }
  return 0;
}
