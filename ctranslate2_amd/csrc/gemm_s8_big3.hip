// MFMA-bound int8 projection GEMM, third form (round 4): the kernel behind launch_gemm_s8_big for
// M >= ~1024 rows (encoder, cross-attention K/V projection, decoder-only prefill). Same contract
// as gemm_s8_big.hip / gemm_s8.hip: C = A[m,k] . B[n,k]^T exact in int32, fused Dequantize epilogue
// in the reference's CPU operand order (src/cpu/kernels.cc:666-673), replacing
// src/cuda/primitives.cu:571-597 + src/ops/dequantize_gpu.cu:31-55.
//
// What changed against gemm_s8_big.hip (kept for A/B: CT2X_BIG_GEMM_V=1 in a tuning build):
//   * v_mfma_i32_32x32x32_i8 instead of 16x16x64: the same 16 operand bytes per lane feed twice
//     the multiply-adds, i.e. half the operand register reads and half the matrix instructions
//     per MAC (the 32x32 form also has the higher measured ceiling: 4404 vs 3944 TOP/s,
//     MI355X_MICROARCH.md, MFMA table). Per-wave sub-tile 128 x 64 = 4 x 2 accumulators of 16.
//   * The operands are SWAPPED at the matrix instruction (A-operand = weight rows, B-operand =
//     activation rows): D[i][j] then has i = output column n, j = output row m, and a lane holds
//     FOUR CONSECUTIVE n of one output row per register quad -> the epilogue stores 16 bytes per
//     lane (global_store_dwordx4) instead of 4, and the per-column operands (weight scale, its
//     reciprocal, bias) are one 16-byte LDS read per quad.
//   * PERSISTENT: one workgroup per CU walks a strided list of tiles; the K-tile stream never
//     drains between tiles - the first stage of tile i+1 and its epilogue operands (row scales,
//     column scales, bias: 3 x 256 floats by 4-byte LDS-DMA into a double-buffered parameter
//     block) are requested during the last K-tile of tile i, and the stores of tile i leave
//     through L2 while tile i+1 computes (gfx950 counts loads and stores on one in-order vmcnt,
//     so the first DMA wait of the next tile is also the point where tile i's stores are known
//     to have reached L2 - about three phases after they were issued).
//   * Interior tiles take an epilogue without bounds predicates; edge tiles keep them.
// Unchanged: 256 x 256 tile, K advanced 128 bytes per stage, L2 -> LDS by global_load_lds_dwordx4
// with the bank swizzle on the source address (chunk c of tile row r at slot c ^ ((r >> 1) & 7):
// SQ_LDS_BANK_CONFLICT = 0 for the 32-row fragment reads as well - the four 16-lane groups of a
// ds_read_b128 each touch 16 distinct 16-byte slots), two 64 KiB stages, raw s_barrier phases,
// the two waves of a SIMD one phase apart (LOAD | MFMA ping-pong), XCD-aware tile order.

#include "device_utils.h"
#include "kernels.h"
#include "runtime.h"

#include <atomic>
#include <cstdlib>

namespace ct2x {

namespace {

constexpr int T_BM = 256, T_BN = 256;
constexpr int T_BK = 128;                                 // K bytes per stage = four MFMA K-steps
constexpr int T_THREADS = 512;
constexpr int T_STAGE_BYTES = (T_BM + T_BN) * T_BK;       // 64 KiB
constexpr int T_RING_BYTES = 2 * T_STAGE_BYTES;           // 128 KiB
constexpr int T_PARAM_FLOATS = 4 * 256;                   // a_scale -> 1/a_scale | w_scale | bias | 1/w_scale
constexpr int T_PARAM_BYTES = T_PARAM_FLOATS * 4;
constexpr int T_FLAG_OFFSET = T_RING_BYTES + 2 * T_PARAM_BYTES;
constexpr int T_LDS_BYTES = T_FLAG_OFFSET + 16;           // 136 KiB + flag of the CU's 160 KiB

typedef int v16i __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__device__ __forceinline__ void glds16(const int8_t* src, int8_t* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((glb_void*)(src), (lds_void*)(lds_wave_base), 16, 0, 0);
}
__device__ __forceinline__ void glds4(const float* src, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((glb_void*)(src), (lds_void*)(lds_wave_base), 4, 0, 0);
}

template <int ACT>
__global__ __launch_bounds__(T_THREADS) void gemm_s8_big3_kernel(const int8_t* __restrict__ A,
                                                                 const int8_t* __restrict__ B,
                                                                 int M, int N, int K,
                                                                 int tiles_m, int tiles_n,
                                                                 GemmS8Epilogue ep) {
  extern __shared__ __attribute__((aligned(16))) int8_t lds[];  // the ONLY LDS object

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int group = wm;
  const int r32 = lane & 31, h = lane >> 5;

  // Persistent tile list: hardware block b sits on XCD b % 8 (observed, speed only); the slot
  // numbering gives every XCD a contiguous range of logical tile ids per round, and the band
  // order below makes a range an 8-row band walked column by column (shared A rows / B columns
  // in that XCD's L2).
  const int ntiles = tiles_m * tiles_n;
  const int G = gridDim.x;
  int slot;
  {
    const int b = blockIdx.x;
    const int xcd = b & 7, q = G >> 3, rem = G & 7;
    slot = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (b >> 3);
  }
  auto coords = [&](int L, int& tm, int& tn) {
    constexpr int BAND = 8;
    const int band = L / (BAND * tiles_n);
    const int band_rows = min(BAND, tiles_m - band * BAND);
    const int in_band = L - band * BAND * tiles_n;
    tm = band * BAND + in_band % band_rows;
    tn = in_band / band_rows;
  };

  // DMA map: one instruction moves 512 x 16 B = 64 tile rows of 128 B (full cache lines);
  // thread -> row tid / 8, slot tid % 8, which receives source chunk (tid % 8) ^ ((row >> 1) & 7).
  // Rows past M / N are clamped (their products are never stored).
  const int ld_row = tid >> 3;
  const uint32_t ld_chunk = ((tid & 7) ^ ((ld_row >> 1) & 7)) * 16;
  // Source of a DMA request = uniform tile base (scalar registers) + a 32-bit lane offset that
  // only changes at the M / N edge (launch_gemm_s8_big3 checks that a tile's 256 rows span
  // < 4 GiB).
  uint32_t a_rel[4], b_rel[4];
  const int8_t* a_base;
  const int8_t* b_base;
  auto set_sources = [&](int tm, int tn) {
    const int m_last = M - 1 - tm * T_BM, n_last = N - 1 - tn * T_BN;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a_rel[i] = static_cast<uint32_t>(min(i * 64 + ld_row, m_last)) * static_cast<uint32_t>(K) + ld_chunk;
      b_rel[i] = static_cast<uint32_t>(min(i * 64 + ld_row, n_last)) * static_cast<uint32_t>(K) + ld_chunk;
    }
    a_base = A + static_cast<int64_t>(tm) * T_BM * K;
    b_base = B + static_cast<int64_t>(tn) * T_BN * K;
  };
  auto stage = [&](int buf, int k0) {
    int8_t* base = lds + buf * T_STAGE_BYTES + wave * 1024;
    const int8_t* ak = a_base + k0;
    const int8_t* bk = b_base + k0;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(ak + a_rel[i], base + i * 8192);
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(bk + b_rel[i], base + T_BM * T_BK + i * 8192);
  };
  // Epilogue operands of a tile: waves 0-3 (group 0) fetch 64 floats each of the three arrays.
  auto stage_params = [&](int pbuf, int tm, int tn) {
    if (ep.mode != 1 || wave >= 4) return;
    float* p = reinterpret_cast<float*>(lds + T_RING_BYTES + pbuf * T_PARAM_BYTES) + wave * 64;
    const int row = min(tm * T_BM + wave * 64 + lane, M - 1);
    const int col = min(tn * T_BN + wave * 64 + lane, N - 1);
    glds4(ep.a_scale + row, p);
    glds4(ep.w_scale + col, p + 256);
    if (ep.bias) glds4(ep.bias + col, p + 512);
  };

  v16i acc[4][2];

  // Fragment addresses: tile row = base32 + r32; K chunk c (16 B) of the 128-byte row is stored
  // at slot c ^ ((row >> 1) & 7); lanes 32-63 hold the second 16 bytes of a 32-byte K-step.
  const int swz = (r32 >> 1) & 7;
  const int x_off = (wm * 128 + r32) * T_BK;
  const int w_off = T_BM * T_BK + (wn * 64 + r32) * T_BK;
  v4i xf[2][4], wf[2][2];
  auto load_frags = [&](int buf, int ks) {
    const int8_t* tile = lds + buf * T_STAGE_BYTES;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int slot16 = ((((ks * 2 + s) * 2 + h) ^ swz) << 4);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        wf[s][j] = *reinterpret_cast<const v4i*>(tile + w_off + j * 32 * T_BK + slot16);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        xf[s][i] = *reinterpret_cast<const v4i*>(tile + x_off + i * 32 * T_BK + slot16);
    }
  };
  auto mfmas = [&]() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[s][j], xf[s][i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  // Phase boundary: nothing may be scheduled across it by the compiler.
  auto phase_barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  const int ktiles = K / T_BK;  // >= 2 (gemm_s8_big_applicable)
  int* slow_div = reinterpret_cast<int*>(lds + T_FLAG_OFFSET);
  if (tid == 0) *slow_div = ep.plain_division ? 1 : 0;

  int L = slot;
  int tm, tn;
  coords(L, tm, tn);
  set_sources(tm, tn);
  stage(0, 0);
  stage_params(0, tm, tn);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  int gt = 0;      // K-tile counter across tiles: stage buffer = gt & 1
  int round = 0;   // parameter buffer = round & 1

  // Ping-pong schedule inside a tile (as gemm_s8_big.hip): the two waves of a SIMD (w, w + 4)
  // run one phase apart, kept there by one extra barrier in front of group 1 and one behind
  // group 0:
  //   group 0:      L(t,0) | M(t,0) | L(t,1) | M(t,1) | L(t+1,0) | ...
  //   group 1:  --  |  L(t,0) | M(t,0) | L(t,1) | M(t,1)   | ...
  // Stage gt+1 (the next K-tile, or the FIRST K-tile of this workgroup's next output tile) is
  // requested in L(t,0) into the buffer both groups finished reading a phase ago, and every
  // wave waits for its own share before the barrier in front of group 0's L(t+1,0).
  while (true) {
    const int Lnext = L + G;
    const bool has_next = Lnext < ntiles;
    int tm_n = 0, tn_n = 0;
    if (has_next) coords(Lnext, tm_n, tn_n);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;

    phase_barrier();  // this tile's first stage (and its parameters) are in LDS
    if (group == 1) phase_barrier();
    for (int t = 0; t < ktiles; ++t) {
      const int buf = gt & 1;
      load_frags(buf, 0);
      if (t + 1 < ktiles) {
        stage(buf ^ 1, (t + 1) * T_BK);
      } else if (has_next) {
        set_sources(tm_n, tn_n);
        stage(buf ^ 1, 0);
        stage_params((round + 1) & 1, tm_n, tn_n);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      phase_barrier();
      mfmas();
      phase_barrier();
      load_frags(buf, 1);
      if (group == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      phase_barrier();
      mfmas();
      if (group == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      phase_barrier();
      ++gt;
    }
    if (group == 0) phase_barrier();

    // ---- epilogue of tile (tm, tn); both groups are aligned here -------------------------
    // C/D layout of the 32x32 MFMA: col j = lane & 31, row i = (reg & 3) + 8 (reg >> 2) +
    // 4 (lane >> 5); with the operands swapped i is the output column and j the output row.
    // (The lane coordinates are re-derived behind an opaque move so that no per-lane address of
    // the epilogue is hoisted over the K loop, where every register is taken.)
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int r32e = lane_e & 31, he = lane_e >> 5;
    const int64_t m0 = static_cast<int64_t>(tm) * T_BM;
    const int64_t n0 = static_cast<int64_t>(tn) * T_BN;
    // 16-byte stores need the row pitch and the base addresses to allow them.
    const bool full = (m0 + T_BM <= M) && (n0 + T_BN <= N) && ((ep.ldy & 3) == 0) &&
                      (((reinterpret_cast<uintptr_t>(ep.y) | reinterpret_cast<uintptr_t>(ep.c_i32) |
                         reinterpret_cast<uintptr_t>(ep.residual)) & 15) == 0);
    if (ep.mode == 0) {
      int32_t* __restrict__ c = ep.c_i32;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int cl = wn * 64 + j * 32 + q * 8 + he * 4;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int64_t row = m0 + wm * 128 + i * 32 + r32e;
            int32_t* p = c + row * ep.ldy + n0 + cl;
            if (full) {
              *reinterpret_cast<v4i*>(p) = v4i{acc[i][j][q * 4], acc[i][j][q * 4 + 1],
                                               acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]};
            } else if (row < M) {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (n0 + cl + e < N) p[e] = acc[i][j][q * 4 + e];
            }
          }
        }
    } else {
      float* par = reinterpret_cast<float*>(lds + T_RING_BYTES + (round & 1) * T_PARAM_BYTES);
      // One reciprocal per row and per column for the whole tile, and the range check of
      // exact_div (device_utils.h): a scale outside it sends this workgroup to plain division.
      if (tid < 256) {
        const float f = 1.f / par[tid];
        par[tid] = f;
        const float w = par[256 + tid];
        par[768 + tid] = 1.f / w;
        if (!exact_div_factor_ok(f) || !exact_div_divisor_ok(w)) *slow_div = 1;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      phase_barrier();
      const bool fast_div = __builtin_amdgcn_readfirstlane(*slow_div) == 0;
      const bool has_bias = ep.bias != nullptr;
      const bool has_res = ep.residual != nullptr;
      float* __restrict__ y = ep.y;
      const float* __restrict__ res = ep.residual;
      float ra[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) ra[i] = par[wm * 128 + i * 32 + r32e];
#define CT2X_BIG3_EPILOGUE(FAST)                                                                  \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int q = 0; q < 4; ++q) {   \
    const int cl = wn * 64 + j * 32 + q * 8 + he * 4;                                              \
    const v4f ws = *reinterpret_cast<const v4f*>(par + 256 + cl);                                 \
    const v4f wi = *reinterpret_cast<const v4f*>(par + 768 + cl);                                 \
    const v4f bs = has_bias ? *reinterpret_cast<const v4f*>(par + 512 + cl) : v4f{0, 0, 0, 0};    \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                               \
      const int64_t row = m0 + wm * 128 + i * 32 + r32e;                                           \
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
        float v = static_cast<float>(acc[i][j][q * 4 + e]) * ra[i];                               \
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
        CT2X_BIG3_EPILOGUE(true)
      } else {
        CT2X_BIG3_EPILOGUE(false)
      }
#undef CT2X_BIG3_EPILOGUE
    }

    if (!has_next) break;
    L = Lnext;
    tm = tm_n;
    tn = tn_n;
    ++round;
  }
}

int device_cu_count() {
  static std::atomic<int> cached[64];
  int dev = 0;
  CT2X_HIP_CHECK(hipGetDevice(&dev));
  int v = cached[dev & 63].load(std::memory_order_acquire);
  if (v == 0) {
    CT2X_HIP_CHECK(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
    if (v <= 0) v = 256;
    cached[dev & 63].store(v, std::memory_order_release);
  }
  return v;
}

template <int ACT>
void launch_big3(const int8_t* a, const int8_t* b, int M, int N, int K, int tiles_m, int tiles_n,
                 const GemmS8Epilogue& ep, int grid, hipStream_t stream) {
  // > 64 KiB of dynamic LDS has to be opted into once per device.
  static std::atomic<bool> configured[64];
  int dev = 0;
  CT2X_HIP_CHECK(hipGetDevice(&dev));
  if (!configured[dev & 63].load(std::memory_order_acquire)) {
    CT2X_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_s8_big3_kernel<ACT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS_BYTES));
    configured[dev & 63].store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL((gemm_s8_big3_kernel<ACT>), dim3(grid), dim3(T_THREADS), T_LDS_BYTES, stream,
                     a, b, M, N, K, tiles_m, tiles_n, ep);
}

}  // namespace

void launch_gemm_s8_big3(const int8_t* a, const int8_t* b, int M, int N, int K,
                         const GemmS8Epilogue& ep_in, hipStream_t stream) {
  static const bool plain_division = [] {
    const char* e = tuning_env("CT2X_EXACT_DIV");
    return e != nullptr && e[0] == '0';
  }();
  // Tuning build: CT2X_BIG3_GRID=<n> workgroups (0 = one per tile, i.e. not persistent).
  static const int grid_override = [] {
    const char* e = tuning_env("CT2X_BIG3_GRID");
    return e ? std::atoi(e) : -1;
  }();
  GemmS8Epilogue ep = ep_in;
  ep.plain_division = plain_division;
  const int tiles_m = (M + T_BM - 1) / T_BM, tiles_n = (N + T_BN - 1) / T_BN;
  const int ntiles = tiles_m * tiles_n;
  int grid = std::min(ntiles, device_cu_count());
  if (grid_override == 0) grid = ntiles;
  else if (grid_override > 0) grid = std::min(ntiles, grid_override);
  const int act = (ep.mode == 1 && ep.act > 0) ? ep.act : -1;
  switch (act) {
    case 1: launch_big3<1>(a, b, M, N, K, tiles_m, tiles_n, ep, grid, stream); break;
    case 2: launch_big3<2>(a, b, M, N, K, tiles_m, tiles_n, ep, grid, stream); break;
    case 3: launch_big3<3>(a, b, M, N, K, tiles_m, tiles_n, ep, grid, stream); break;
    case 4: launch_big3<4>(a, b, M, N, K, tiles_m, tiles_n, ep, grid, stream); break;
    case 5: launch_big3<5>(a, b, M, N, K, tiles_m, tiles_n, ep, grid, stream); break;
    case 6: launch_big3<6>(a, b, M, N, K, tiles_m, tiles_n, ep, grid, stream); break;
    default: launch_big3<-1>(a, b, M, N, K, tiles_m, tiles_n, ep, grid, stream); break;
  }
}

}  // namespace ct2x
