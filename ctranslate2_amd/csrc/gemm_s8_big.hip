// MFMA-bound int8 projection GEMM for M >= ~1024 rows (encoder, cross-attention K/V projection,
// decoder-only prefill): the shapes where SURVEY 8(d) puts the int8-MFMA roofline within reach.
// Same contract as gemm_s8.hip (C = A[m,k] . B[n,k]^T, exact int32, fused Dequantize epilogue,
// replacing src/cuda/primitives.cu:571-597 + src/ops/dequantize_gpu.cu:31-55).
//
// Structure (gfx950, one workgroup of 8 waves per CU):
//   * 256 x 256 output tile, K advanced 128 bytes per stage (two MFMA K-steps); wave (wm, wn)
//     of a 2 x 4 grid owns a 128 x 64 sub-tile = 8 x 4 v_mfma_i32_16x16x64_i8 accumulators.
//   * Operands go L2 -> LDS by global_load_lds_dwordx4 (no staging registers, no ds_write pass),
//     one full 128-byte line per tile row. The DMA writes LDS lane-linearly, so the bank swizzle
//     is applied to the *source* address: 16-byte chunk c of tile row r is stored at chunk slot
//     c ^ ((r >> 1) & 7). The hardware services a ds_read_b128 in four groups of 16 lanes
//     ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...; MI355X_MICROARCH.md, LDS table): with this
//     key each group touches 16 distinct 16-byte slots of the 256-byte bank row (measured:
//     SQ_LDS_BANK_CONFLICT = 0).
//   * LDS holds two stages of (256 + 256) x 128 B = 64 KiB. Phases are separated by raw
//     s_barrier (never __syncthreads(), whose fence would drain the DMA queue at every phase);
//     the DMA is waited for once per tile, three phases after it was issued.
//   * The two waves of a SIMD ping-pong between a LOAD phase and an MFMA phase (see the K loop).
//   * Workgroup ids are remapped so that the 8 XCDs each walk a contiguous band of tiles (one L2
//     per XCD: neighbours share their A rows / B columns there).

#include "device_utils.h"
#include "kernels.h"
#include "runtime.h"

#include <atomic>
#include <cstdlib>
#include <type_traits>

namespace ct2x {

namespace {

constexpr int BIG_BM = 256, BIG_BN = 256;
constexpr int BIG_BK = 128;                                 // K bytes per stage = two MFMA K-steps
constexpr int BIG_THREADS = 512;
constexpr int BIG_STAGES = 2;
constexpr int BIG_TILE_BYTES = (BIG_BM + BIG_BN) * BIG_BK;  // 64 KiB per stage
constexpr int BIG_LDS_BYTES = BIG_STAGES * BIG_TILE_BYTES;  // 128 KiB of the CU's 160 KiB

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__device__ __forceinline__ void glds16(const int8_t* src, int8_t* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((glb_void*)(src), (lds_void*)(lds_wave_base), 16, 0, 0);
}

// ABLATE (timing experiments only, tools/gemm_bench.cpp): 1 = no DMA after the prologue,
// 2 = no MFMAs, 3 = no fragment reads, 4 = MFMAs only (no DMA, no fragment reads, no barriers),
// 5 = K loop only (no epilogue).
template <int ACT, int ABLATE = 0>
__global__ __launch_bounds__(BIG_THREADS) void gemm_s8_big_kernel(const int8_t* __restrict__ A,
                                                                  const int8_t* __restrict__ B,
                                                                  int M, int N, int K,
                                                                  int tiles_m, int tiles_n,
                                                                  GemmS8Epilogue ep) {
  extern __shared__ __attribute__((aligned(16))) int8_t lds[];  // the ONLY LDS object (see header)

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int r = lane & 15, g = lane >> 4;

  // XCD-aware, bijective remap of the hardware workgroup id (round-robin over the 8 XCDs) to a
  // logical id whose consecutive values sit on the same XCD, then bands of 8 row-tiles walked
  // column by column so neighbours share operands in that XCD's L2.
  const int nwg = tiles_m * tiles_n;
  int wg = blockIdx.x;
  {
    const int xcd = wg & 7, q = nwg >> 3, rem = nwg & 7;
    wg = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (wg >> 3);
  }
  constexpr int BAND = 8;
  const int band = wg / (BAND * tiles_n);
  const int band_rows = min(BAND, tiles_m - band * BAND);
  const int in_band = wg - band * BAND * tiles_n;
  const int tm = band * BAND + in_band % band_rows;
  const int tn = in_band / band_rows;
  const int64_t m0 = static_cast<int64_t>(tm) * BIG_BM;
  const int64_t n0 = static_cast<int64_t>(tn) * BIG_BN;

  // DMA map: one instruction moves 512 x 16 B = 64 tile rows of 128 B (full cache lines);
  // thread -> row tid/8, slot tid%8, which receives source chunk (tid%8) ^ ((row >> 1) & 7).
  // Rows past M / N are clamped (their products are never stored).
  const int ld_row = tid >> 3, ld_pos = tid & 7;
  const int8_t* a_src[4];
  const int8_t* b_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = i * 64 + ld_row;
    const int chunk = ld_pos ^ ((row >> 1) & 7);
    const int64_t ar = min<int64_t>(m0 + row, M - 1);
    const int64_t br = min<int64_t>(n0 + row, N - 1);
    a_src[i] = A + ar * K + chunk * 16;
    b_src[i] = B + br * K + chunk * 16;
  }
  auto stage = [&](int t) {
    if ((ABLATE == 1 || ABLATE == 4) && t >= 1) {
      asm volatile("" ::: "memory");
      return;
    }
    int8_t* base = lds + (t & (BIG_STAGES - 1)) * BIG_TILE_BYTES + wave * 1024;
    const int k0 = t * BIG_BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(a_src[i] + k0, base + i * 8192);
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(b_src[i] + k0, base + BIG_BM * BIG_BK + i * 8192);
  };

  v4i acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = v4i{0, 0, 0, 0};

  // Fragment addresses: tile row = base16 + r; K chunk c = ks * 4 + g of 8, stored at slot
  // c ^ ((r >> 1) & 7).
  const int swz = r >> 1;
  const int a_off = (wm * 128 + r) * BIG_BK;
  const int b_off = BIG_BM * BIG_BK + (wn * 64 + r) * BIG_BK;

  v4i af[8], bf[4];
  auto load_frags = [&](int t, int ks) {
    if ((ABLATE == 3 || ABLATE == 4) && t > 0) return;
    const int8_t* tile = lds + (t & (BIG_STAGES - 1)) * BIG_TILE_BYTES + (((ks * 4 + g) ^ swz) << 4);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      bf[j] = *reinterpret_cast<const v4i*>(tile + b_off + j * 16 * BIG_BK);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      af[i] = *reinterpret_cast<const v4i*>(tile + a_off + i * 16 * BIG_BK);
  };
  auto mfmas = [&]() {
    if (ABLATE == 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i][0] = acc[i][0] + af[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[0][j] = acc[0][j] + bf[j];
      return;
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[i], bf[j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  // Phase boundary: nothing may be scheduled across it by the compiler.
  auto phase_barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    if (ABLATE != 4) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // Ping-pong schedule. The two waves that share a SIMD (wave w and w + 4: group = wm) run one
  // phase apart, kept there by one extra barrier in front of group 1: between any two barriers
  // one of them is in a LOAD phase (12 fragment ds_read_b128 of one 64-byte K-step, its share of
  // the DMA of the next tile, the waits) while the other issues the 32 MFMAs of a K-step at
  // raised priority, so the matrix pipe of the SIMD never waits for LDS latency or a barrier.
  //   group 0:      L(t,0) | M(t,0) | L(t,1) | M(t,1) | L(t+1,0) | ...
  //   group 1:  --  |  L(t,0) | M(t,0) | L(t,1) | M(t,1)   | ...
  // Tile t+1 is requested from L(t,0) on (both groups are past their last read of tile t-1, whose
  // buffer it takes) and has to be in LDS when group 0 starts L(t+1,0): every wave waits for its
  // own share before that barrier - group 0 at the end of M(t,1), group 1 at the end of L(t,1) -
  // i.e. the DMA has three phases to land.
  const int ktiles = K / BIG_BK;  // >= 2 (gemm_s8_big_applicable)
  const int group = wm;
  stage(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  phase_barrier();  // tile 0 is in LDS
  if (group == 1) phase_barrier();
  for (int t = 0; t < ktiles; ++t) {
    load_frags(t, 0);
    if (t + 1 < ktiles) stage(t + 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    phase_barrier();
    mfmas();
    phase_barrier();
    load_frags(t, 1);
    if (group == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    phase_barrier();
    mfmas();
    if (group == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    phase_barrier();
  }
  if (group == 0) phase_barrier();

  // Epilogue. C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg.
  // No global load may follow a store to y in program order (it would have to assume aliasing
  // and serialise on memory latency): the per-row factors 1/a_scale go through LDS (free after
  // the K loop), the per-column operands sit in registers, and the residual of row group i+1 is
  // requested before the stores of row group i.
  if (ABLATE == 5) {  // K loop only: one store per lane keeps the accumulators alive
    int keep = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) keep ^= acc[i][j][0] ^ acc[i][j][1] ^ acc[i][j][2] ^ acc[i][j][3];
    if (keep == 0x7fffffff && ep.c_i32) ep.c_i32[0] = keep;
    return;
  }
  float* row_factor = reinterpret_cast<float*>(lds);
  int* slow_div = reinterpret_cast<int*>(lds + BIG_BM * sizeof(float));
  __syncthreads();  // every wave is done with its last tile
  if (tid == 0) *slow_div = ep.plain_division ? 1 : 0;
  __syncthreads();
  if (ep.mode == 1 && tid < BIG_BM) {
    const int64_t row = m0 + tid;
    const float f = row < M ? 1.f / ep.a_scale[row] : 0.f;
    row_factor[tid] = f;
    if (row < M && !exact_div_factor_ok(f)) *slow_div = 1;
  }
  // The 256 x 256 tile divides 128 values per lane by 4 column scales: the division runs as
  // exact_div (device_utils.h; same bits as `/`, 5 VALU operations instead of 11) unless a scale
  // of this tile is outside the range the identity is proven for.
  float ws[4], ws_inv[4], bias[4];
  bool col_ok[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t col = n0 + wn * 64 + j * 16 + r;
    col_ok[j] = col < N;
    ws[j] = (ep.mode == 1 && col_ok[j]) ? ep.w_scale[col] : 1.f;
    ws_inv[j] = 1.f / ws[j];
    if (!exact_div_divisor_ok(ws[j])) *slow_div = 1;
    bias[j] = (ep.mode == 1 && ep.bias && col_ok[j]) ? ep.bias[col] : 0.f;
  }
  __syncthreads();
  const bool fast_div = *slow_div == 0;  // workgroup-uniform
  const bool has_res = ep.mode == 1 && ep.residual != nullptr;
  const int64_t col0 = n0 + wn * 64 + r;
  auto load_residual = [&](int i, float (&res)[4][4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t row = m0 + wm * 128 + i * 16 + g * 4 + e;
      const float* p = ep.residual + row * ep.ldy + col0;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        res[e][j] = (has_res && row < M && col_ok[j]) ? p[j * 16] : 0.f;
    }
  };
  auto store_group = [&](int i, const float (&res)[4][4], auto fast_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int lrow = wm * 128 + i * 16 + g * 4 + e;
      const int64_t row = m0 + lrow;
      if (row >= M) continue;
      if (ep.mode == 0) {
        int32_t* p = ep.c_i32 + row * ep.ldy + col0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (col_ok[j]) p[j * 16] = acc[i][j][e];
      } else {
        const float r_a = row_factor[lrow];
        float* p = ep.y + row * ep.ldy + col0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // CPU formula order (src/cpu/kernels.cc:666-673): c * (1/a_scale) / b_scale + bias.
          float v = static_cast<float>(acc[i][j][e]) * r_a;
          v = FAST ? exact_div(v, ws[j], ws_inv[j]) : v / ws[j];
          if (ep.bias) v = v + bias[j];
          v = ACT > 0 ? apply_activation(v, ACT) : (ep.act == 0 /* ReLU */ ? fmaxf(v, 0.f) : v);
          if (has_res) v = res[e][j] + v;
          if (col_ok[j]) p[j * 16] = v;
        }
      }
    }
  };
  // (A macro, not a closure: register arrays reached through a nested closure are demoted to
  // scratch memory - a first version of this dispatch cost the Swish instantiation 1040 bytes of
  // stack per thread and 3.5x its run time, which the kernel metadata scan of
  // tests/test_host.py::test_hot_kernels_do_not_spill now guards.)
  float resA[4][4], resB[4][4];
#define CT2X_BIG_EPILOGUE(TAG)                     \
  load_residual(0, resA);                          \
  _Pragma("unroll") for (int i = 0; i < 8; i += 2) { \
    load_residual(i + 1, resB);                    \
    store_group(i, resA, TAG);                     \
    if (i + 2 < 8) load_residual(i + 2, resA);     \
    store_group(i + 1, resB, TAG);                 \
  }
  if (fast_div) {
    CT2X_BIG_EPILOGUE(std::true_type{})
  } else {
    CT2X_BIG_EPILOGUE(std::false_type{})
  }
#undef CT2X_BIG_EPILOGUE
}

template <int ACT, int ABLATE = 0>
void launch_big(const int8_t* a, const int8_t* b, int M, int N, int K, int tiles_m, int tiles_n,
                const GemmS8Epilogue& ep, hipStream_t stream) {
  // > 64 KiB of dynamic LDS has to be opted into once per device.
  static std::atomic<bool> configured[64];
  int dev = 0;
  CT2X_HIP_CHECK(hipGetDevice(&dev));
  if (!configured[dev & 63].load(std::memory_order_acquire)) {
    CT2X_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_s8_big_kernel<ACT, ABLATE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       BIG_LDS_BYTES));
    configured[dev & 63].store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL((gemm_s8_big_kernel<ACT, ABLATE>), dim3(tiles_m * tiles_n), dim3(BIG_THREADS),
                     BIG_LDS_BYTES, stream, a, b, M, N, K, tiles_m, tiles_n, ep);
}

}  // namespace

bool gemm_s8_big_applicable(int64_t m, int64_t n, int64_t k) {
  if (k % BIG_BK != 0 || k < 2 * BIG_BK) return false;
  const int64_t tiles = ((m + BIG_BM - 1) / BIG_BM) * ((n + BIG_BN - 1) / BIG_BN);
  return m >= 512 && n >= 512 && tiles >= 96;
}

void launch_gemm_s8_big(const int8_t* a, const int8_t* b, int M, int N, int K,
                        const GemmS8Epilogue& ep_in, hipStream_t stream) {
  static const bool plain_division = [] {
    const char* e = tuning_env("CT2X_EXACT_DIV");
    return e != nullptr && e[0] == '0';
  }();
  // Round 4: gemm_s8_big3.hip is the default; this kernel stays for A/B timing in a tuning build.
  static const bool use_v1 = [] {
    const char* e = tuning_env("CT2X_BIG_GEMM_V");
    return e != nullptr && e[0] == '1';
  }();
#ifdef CT2X_TUNING  // the two-workgroups-per-CU experiment is only part of a tuning build
  static const bool use_v5 = [] {
    const char* e = tuning_env("CT2X_BIG_GEMM_V");
    return e != nullptr && e[0] == '5';
  }();
  if (use_v5) return launch_gemm_s8_big5(a, b, M, N, K, ep_in, stream);
#endif
  // The fused forms run 4 - 9 % faster on gemm_s8_big3.hip (16-byte stores, operands by LDS-DMA,
  // no predicates inside a tile); the raw int32 result leaves 5 - 10 % faster from this kernel
  // (profiles/r04_gemm_bench_g.txt), so ops::Gemm's int32 form stays here.
  static const bool force_v3 = [] {
    const char* e = tuning_env("CT2X_BIG_GEMM_V");
    return e != nullptr && e[0] == '3';
  }();
  if (!use_v1 && (ep_in.mode == 1 || force_v3)) return launch_gemm_s8_big3(a, b, M, N, K, ep_in, stream);
  GemmS8Epilogue ep = ep_in;
  ep.plain_division = plain_division;
  const int tiles_m = (M + BIG_BM - 1) / BIG_BM, tiles_n = (N + BIG_BN - 1) / BIG_BN;
  const int act = (ep.mode == 1 && ep.act > 0) ? ep.act : -1;
  static const int ablate = [] {
    const char* e = tuning_env("CT2X_BIG_GEMM_ABLATE");
    return e ? std::atoi(e) : 0;
  }();
  if (ablate == 1) return launch_big<-1, 1>(a, b, M, N, K, tiles_m, tiles_n, ep, stream);
  if (ablate == 2) return launch_big<-1, 2>(a, b, M, N, K, tiles_m, tiles_n, ep, stream);
  if (ablate == 3) return launch_big<-1, 3>(a, b, M, N, K, tiles_m, tiles_n, ep, stream);
  if (ablate == 4) return launch_big<-1, 4>(a, b, M, N, K, tiles_m, tiles_n, ep, stream);
  if (ablate == 5) return launch_big<-1, 5>(a, b, M, N, K, tiles_m, tiles_n, ep, stream);
  switch (act) {
    case 1: launch_big<1>(a, b, M, N, K, tiles_m, tiles_n, ep, stream); break;
    case 2: launch_big<2>(a, b, M, N, K, tiles_m, tiles_n, ep, stream); break;
    case 3: launch_big<3>(a, b, M, N, K, tiles_m, tiles_n, ep, stream); break;
    case 4: launch_big<4>(a, b, M, N, K, tiles_m, tiles_n, ep, stream); break;
    case 5: launch_big<5>(a, b, M, N, K, tiles_m, tiles_n, ep, stream); break;
    case 6: launch_big<6>(a, b, M, N, K, tiles_m, tiles_n, ep, stream); break;
    default: launch_big<-1>(a, b, M, N, K, tiles_m, tiles_n, ep, stream); break;
  }
}

}  // namespace ct2x
