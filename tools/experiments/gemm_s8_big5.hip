// MFMA-bound int8 projection GEMM, two-workgroups-per-CU form (round 4 experiment -> see DESIGN):
// the same contract as gemm_s8_big3.hip, but a workgroup is FOUR waves (one per SIMD) that own a
// 128 x 256 output tile with a three-stage LDS ring of 64-byte K-steps (24 KiB per stage, 76 KiB
// per workgroup), so that TWO workgroups are resident per CU. Why: in the one-workgroup forms the
// epilogue of a tile (the fp32 result leaving at HBM-write speed, plus ~10 - 35 VALU operations
// per element for the exact Dequantize + activation) runs with the matrix pipes idle - 15 - 40 %
// of a launch at K = 4096. With two independent workgroups per CU the hardware scheduler overlaps
// one workgroup's epilogue (and its pipeline fill) with the other's K loop, and the 2 x finer
// tiles balance a ragged last round dynamically. Price: the weight tile is shared by 128 instead
// of 256 activation rows, 1.5 x the L2 -> LDS bytes per MAC.
//
// LDS image of a stage: rows of 64 bytes (one MFMA K-step); 16-byte chunk c of tile row r is
// stored at slot c ^ f((r >> 2) & 3) with f = {0, 2, 3, 1}: the four 16-lane groups of a
// ds_read_b128 fragment read ({0-3, 12-15, 20-27}, ...) then touch 16 distinct 16-byte slots of
// the 256-byte bank row. The DMA (global_load_lds_dwordx4) writes lane-linearly, so the swizzle
// is applied to the source address.

#include "device_utils.h"
#include "kernels.h"
#include "runtime.h"

#include <atomic>
#include <cstdlib>

namespace ct2x {

namespace {

constexpr int F_BM = 128, F_BN = 256;
constexpr int F_BK = 64;                                  // K bytes per stage = one MFMA K-step
constexpr int F_THREADS = 256;
constexpr int F_STAGES = 3;
constexpr int F_STAGE_BYTES = (F_BM + F_BN) * F_BK;       // 24 KiB
constexpr int F_RING_BYTES = F_STAGES * F_STAGE_BYTES;    // 72 KiB
// 1/a_scale [128] | w_scale [256] | bias [256] | 1/w_scale [256]
constexpr int F_PAR_W = 128, F_PAR_B = 384, F_PAR_WI = 640;
constexpr int F_PARAM_BYTES = 896 * 4;
constexpr int F_FLAG_OFFSET = F_RING_BYTES + F_PARAM_BYTES;
constexpr int F_LDS_BYTES = F_FLAG_OFFSET + 16;           // ~75.5 KiB: two workgroups per CU

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__device__ __forceinline__ void glds16(const int8_t* src, int8_t* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((glb_void*)(src), (lds_void*)(lds_wave_base), 16, 0, 0);
}
__device__ __forceinline__ void glds4(const float* src, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((glb_void*)(src), (lds_void*)(lds_wave_base), 4, 0, 0);
}
__device__ __forceinline__ int swz4(int x) { return (0x78 >> (2 * (x & 3))) & 3; }  // {0, 2, 3, 1}

template <int ACT, int VAR = 0>
__global__ __launch_bounds__(F_THREADS, 2) void gemm_s8_big5_kernel(const int8_t* __restrict__ A,
                                                                    const int8_t* __restrict__ B,
                                                                    int M, int N, int K,
                                                                    int tiles_m, int tiles_n,
                                                                    GemmS8Epilogue ep) {
  extern __shared__ __attribute__((aligned(16))) int8_t lds[];  // the ONLY LDS object

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave;
  const int r = lane & 15, g = lane >> 4;

  // XCD-aware, bijective remap of the hardware workgroup id (round-robin over the 8 XCDs) to a
  // logical id whose consecutive values sit on the same XCD, then bands of 16 row-tiles walked
  // column by column so neighbours share operands in that XCD's L2.
  const int nwg = tiles_m * tiles_n;
  int wg = blockIdx.x;
  {
    const int xcd = wg & 7, q = nwg >> 3, rem = nwg & 7;
    wg = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (wg >> 3);
  }
  constexpr int BAND = 16;
  const int band = wg / (BAND * tiles_n);
  const int band_rows = min(BAND, tiles_m - band * BAND);
  const int in_band = wg - band * BAND * tiles_n;
  const int tm = band * BAND + in_band % band_rows;
  const int tn = in_band / band_rows;

  // DMA map: one instruction moves 256 x 16 B = 64 tile rows of 64 B; thread -> row tid / 4,
  // slot tid % 4, which receives source chunk (tid % 4) ^ f(row >> 2). Rows past M / N are
  // clamped (their products are never stored).
  const int ld_row = tid >> 2;
  const uint32_t ld_chunk = ((tid & 3) ^ swz4(tid >> 4)) * 16;
  uint32_t a_rel[2], b_rel[4];
  {
    const int m_last = M - 1 - tm * F_BM, n_last = N - 1 - tn * F_BN;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      a_rel[i] = static_cast<uint32_t>(min(i * 64 + ld_row, m_last)) * static_cast<uint32_t>(K) + ld_chunk;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      b_rel[i] = static_cast<uint32_t>(min(i * 64 + ld_row, n_last)) * static_cast<uint32_t>(K) + ld_chunk;
  }
  const int8_t* a_base = A + static_cast<int64_t>(tm) * F_BM * K;
  const int8_t* b_base = B + static_cast<int64_t>(tn) * F_BN * K;
  auto stage = [&](int buf_off, int k0) {
    int8_t* base = lds + buf_off + wave * 1024;
    const int8_t* ak = a_base + k0;
    const int8_t* bk = b_base + k0;
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(ak + a_rel[i], base + i * 4096);
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(bk + b_rel[i], base + F_BM * F_BK + i * 4096);
  };

  v4i acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = v4i{0, 0, 0, 0};

  // Fragment addresses: tile row = base16 + r, K chunk g stored at slot g ^ f(r >> 2).
  const int f_slot = (g ^ swz4(r >> 2)) << 4;
  const int x_off = r * F_BK + f_slot;
  const int w_off = F_BM * F_BK + (wn * 64 + r) * F_BK + f_slot;
  v4i xf[8], wf[4];
  auto load_frags = [&](int buf_off) {
    const int8_t* tile = lds + buf_off;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      wf[j] = *reinterpret_cast<const v4i*>(tile + w_off + j * 16 * F_BK);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      xf[i] = *reinterpret_cast<const v4i*>(tile + x_off + i * 16 * F_BK);
  };
  auto mfmas = [&]() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[j], xf[i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  auto phase_barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  int* slow_div = reinterpret_cast<int*>(lds + F_FLAG_OFFSET);
  if (tid == 0) *slow_div = ep.plain_division ? 1 : 0;
  float* par = reinterpret_cast<float*>(lds + F_RING_BYTES);
  if (ep.mode == 1) {  // epilogue operands: 128 row scales, 256 column scales, 256 biases
    const int col = min(tn * F_BN + wave * 64 + lane, N - 1);
    glds4(ep.w_scale + col, par + F_PAR_W + wave * 64);
    if (ep.bias) glds4(ep.bias + col, par + F_PAR_B + wave * 64);
    if (wave < 2) {
      const int row = min(tm * F_BM + wave * 64 + lane, M - 1);
      glds4(ep.a_scale + row, par + wave * 64);
    }
  }
  const int steps = K / F_BK;  // >= 4 (gemm_s8_big_applicable: K % 128 == 0, K >= 256)
  stage(0, 0);
  stage(F_STAGE_BYTES, F_BK);

  // Three-stage ring, one barrier per K-step: step s waits (counted, in order) for its own
  // share of stage s, the barrier makes every wave's share visible and proves that stage s - 1
  // has been read by all four waves, whose buffer then takes the request for stage s + 2.
  int cur = 0, nxt2 = 2 * F_STAGE_BYTES;  // byte offsets of stage s and stage s + 2 in the ring
  for (int s = 0; s < steps; ++s) {
    if (s + 1 < steps) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    phase_barrier();
    if (s + 2 < steps) stage(nxt2, (s + 2) * F_BK);
    load_frags(cur);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    mfmas();
    cur = cur + F_STAGE_BYTES == F_RING_BYTES ? 0 : cur + F_STAGE_BYTES;
    nxt2 = nxt2 + F_STAGE_BYTES == F_RING_BYTES ? 0 : nxt2 + F_STAGE_BYTES;
  }

  // ---- epilogue ---------------------------------------------------------------------------
  // C/D layout of the 16x16 MFMA: col j = lane & 15, row i = (lane >> 4) * 4 + reg; with the
  // operands swapped (A-operand = weight rows) i is the output column and j the output row: a
  // lane holds four consecutive columns of one row per accumulator -> 16-byte stores.
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  const int re = lane_e & 15, ge = lane_e >> 4;
  const int64_t m0 = static_cast<int64_t>(tm) * F_BM;
  const int64_t n0 = static_cast<int64_t>(tn) * F_BN;
  // 16-byte stores need the row pitch and the base addresses to allow them.
  const bool full = (m0 + F_BM <= M) && (n0 + F_BN <= N) && ((ep.ldy & 3) == 0) &&
                    (((reinterpret_cast<uintptr_t>(ep.y) | reinterpret_cast<uintptr_t>(ep.c_i32) |
                       reinterpret_cast<uintptr_t>(ep.residual)) & 15) == 0);
  if (VAR & 2) {  // K loops only: one conditional store keeps the accumulators alive
    int keep = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) keep ^= acc[i][j][0] ^ acc[i][j][1] ^ acc[i][j][2] ^ acc[i][j][3];
    if (keep == 0x7fffffff && ep.c_i32) ep.c_i32[0] = keep;
  } else if (ep.mode == 0) {
    int32_t* __restrict__ c = ep.c_i32;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int cl = wn * 64 + j * 16 + ge * 4;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int64_t row = m0 + i * 16 + re;
        int32_t* p = c + row * ep.ldy + n0 + cl;
        if (full) {
          *reinterpret_cast<v4i*>(p) = acc[i][j];
        } else if (row < M) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n0 + cl + e < N) p[e] = acc[i][j][e];
        }
      }
    }
  } else {
    // One reciprocal per row and per column for the whole tile, and the range check of
    // exact_div (device_utils.h): a scale outside it sends this workgroup to plain division.
    {
      const float w = par[F_PAR_W + tid];
      par[F_PAR_WI + tid] = 1.f / w;
      bool ok = exact_div_divisor_ok(w);
      if (tid < F_BM) {
        const float f = 1.f / par[tid];
        par[tid] = f;
        ok = ok && exact_div_factor_ok(f);
      }
      if (!ok) *slow_div = 1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    phase_barrier();
    const bool fast_div = __builtin_amdgcn_readfirstlane(*slow_div) == 0;
    const bool has_bias = ep.bias != nullptr;
    const bool has_res = ep.residual != nullptr;
    float* __restrict__ y = ep.y;
    const float* __restrict__ res = ep.residual;
    float ra[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) ra[i] = par[i * 16 + re];
#define CT2X_BIG5_EPILOGUE(FAST)                                                                  \
_Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                 \
  const int cl = wn * 64 + j * 16 + ge * 4;                                                     \
  const v4f ws = *reinterpret_cast<const v4f*>(par + F_PAR_W + cl);                                 \
  const v4f wi = *reinterpret_cast<const v4f*>(par + F_PAR_WI + cl);                                 \
  const v4f bs = has_bias ? *reinterpret_cast<const v4f*>(par + F_PAR_B + cl) : v4f{0, 0, 0, 0};    \
  _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                               \
    const int64_t row = m0 + i * 16 + re;                                            \
    const int64_t at = row * ep.ldy + n0 + cl;                                                  \
    v4f rs = v4f{0, 0, 0, 0};                                                                   \
    if (has_res) {                                                                              \
      if (full) rs = *reinterpret_cast<const v4f*>(res + at);                                   \
      else if (row < M) {                                                                       \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) if (n0 + cl + e < N) rs[e] = res[at + e]; \
      }                                                                                         \
    }                                                                                           \
    v4f out;                                                                                    \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                             \
      /* CPU formula order: c * (1/a_scale) / b_scale + bias. */                                \
      float v = static_cast<float>(acc[i][j][e]) * ra[i];                                       \
      v = FAST ? exact_div(v, ws[e], wi[e]) : v / ws[e];                                        \
      if (has_bias) v = v + bs[e];                                                              \
      v = ACT > 0 ? apply_activation(v, ACT) : (ep.act == 0 /* ReLU */ ? fmaxf(v, 0.f) : v);    \
      if (has_res) v = rs[e] + v;                                                               \
      out[e] = v;                                                                               \
    }                                                                                           \
    if (full) *reinterpret_cast<v4f*>(y + at) = out;                                            \
    else if (row < M) {                                                                         \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) if (n0 + cl + e < N) y[at + e] = out[e];    \
    }                                                                                           \
  }                                                                                             \
}
    if (fast_div) {
      CT2X_BIG5_EPILOGUE(true)
    } else {
      CT2X_BIG5_EPILOGUE(false)
    }
#undef CT2X_BIG5_EPILOGUE
  }

}

template <int ACT, int VAR = 0>
void launch_big5(const int8_t* a, const int8_t* b, int M, int N, int K, int tiles_m, int tiles_n,
                 const GemmS8Epilogue& ep, hipStream_t stream) {
  static std::atomic<bool> configured[64];
  int dev = 0;
  CT2X_HIP_CHECK(hipGetDevice(&dev));
  if (!configured[dev & 63].load(std::memory_order_acquire)) {
    CT2X_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_s8_big5_kernel<ACT, VAR>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS_BYTES));
    configured[dev & 63].store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL((gemm_s8_big5_kernel<ACT, VAR>), dim3(tiles_m * tiles_n), dim3(F_THREADS),
                     F_LDS_BYTES, stream, a, b, M, N, K, tiles_m, tiles_n, ep);
}

}  // namespace

void launch_gemm_s8_big5(const int8_t* a, const int8_t* b, int M, int N, int K,
                         const GemmS8Epilogue& ep_in, hipStream_t stream) {
  static const bool plain_division = [] {
    const char* e = tuning_env("CT2X_EXACT_DIV");
    return e != nullptr && e[0] == '0';
  }();
  GemmS8Epilogue ep = ep_in;
  ep.plain_division = plain_division;
  const int tiles_m = (M + F_BM - 1) / F_BM, tiles_n = (N + F_BN - 1) / F_BN;
  const int act = (ep.mode == 1 && ep.act > 0) ? ep.act : -1;
#ifdef CT2X_TUNING
  static const int var = [] {
    const char* e = tuning_env("CT2X_BIG5_VAR");
    return e ? std::atoi(e) : 0;
  }();
  if (var == 2) return launch_big5<-1, 2>(a, b, M, N, K, tiles_m, tiles_n, ep, stream);
#endif
  switch (act) {
    case 1: launch_big5<1>(a, b, M, N, K, tiles_m, tiles_n, ep, stream); break;
    case 2: launch_big5<2>(a, b, M, N, K, tiles_m, tiles_n, ep, stream); break;
    case 3: launch_big5<3>(a, b, M, N, K, tiles_m, tiles_n, ep, stream); break;
    case 4: launch_big5<4>(a, b, M, N, K, tiles_m, tiles_n, ep, stream); break;
    case 5: launch_big5<5>(a, b, M, N, K, tiles_m, tiles_n, ep, stream); break;
    case 6: launch_big5<6>(a, b, M, N, K, tiles_m, tiles_n, ep, stream); break;
    default: launch_big5<-1>(a, b, M, N, K, tiles_m, tiles_n, ep, stream); break;
  }
}

}  // namespace ct2x
