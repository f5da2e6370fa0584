// EXPERIMENT, NOT PART OF THE PRODUCT BUILD (round 3; kept as the record of a negative result).
// An int8 GEMM with ONE wave per SIMD and a 128 x 128 accumulator tile per wave (256 AGPRs), tried
// as a replacement for the 8-wave ping-pong kernel of ctranslate2_amd/csrc/gemm_s8_big.hip. It is
// exact (0 mismatches, the gemm_s8 tests pass with it) and SLOWER on the MI355X:
//   8192^3 int32 out:  546 us (register-staged operands) / 593 us (LDS-DMA)  vs 503 us (8 waves)
//   2048 x 14336 x 4096: 150 us vs 127 us                  (profiles/r03_gemm_bench_big2_{a,b}.txt)
// tools/big2_loop_probe.cpp rebuilds this K loop ingredient by ingredient
// (profiles/r03_big2_loop_probe.txt): bare matrix operations run 16.2 cycles each = 3.47 POP/s at
// the 1.68 GHz the chip sustains under this load (the ceiling of ANY int8 kernel here, 69 % of the
// nominal 5 POP/s); the whole loop with L2-shared operands reaches 2.6 (register staging) - 2.8
// (LDS-DMA) POP/s, and 1.4 POP/s when every operand byte comes from HBM. What remains between the
// probe and the kernels is outside the K loop's instruction stream (first-touch operand latency,
// prologue / epilogue per tile with one workgroup per CU), which this form does not improve.
// To build it again: add it to SRCS in ctranslate2_amd/csrc/Makefile and declare
// gemm_s8_big2_applicable / launch_gemm_s8_big2 in kernels.h.
//
// MFMA-bound int8 projection GEMM, second form (round 3): ONE wave per SIMD with a 128 x 128
// accumulator tile. Same contract and the same bits as gemm_s8_big_kernel (C = A[m,k] . B[n,k]^T,
// exact int32, fused Dequantize epilogue; replaces src/cuda/primitives.cu:571-597 +
// src/ops/dequantize_gpu.cu:31-55).
//
// Why a second form. The 8-wave kernel (gemm_s8_big.hip) gives each wave a 128 x 64 tile: per
// 64-byte K step a wave reads 12 fragments from LDS for 32 MFMAs, and its two waves per SIMD must
// alternate LOAD and MFMA phases around barriers - measured: the LOAD phases are longer than the
// MFMA phases, the matrix pipe is busy 47 % of the time (profiles/r01_pmc_gemm_big.json), 2.1 - 2.2
// POP/s. Here a workgroup is 4 waves, each alone on its SIMD with a 128 x 128 tile in 256
// accumulator registers (the register file is 512 per lane): 16 fragment reads feed 64 MFMAs (a
// third fewer LDS reads per MAC), and nothing alternates - a wave issues its MFMAs back to back and
// slips the next step's fragment reads and the DMA requests of a later stage into the issue slots
// the 16-cycle matrix operations leave free (sched_group_barrier pins the interleave).
//   * 256 x 256 workgroup tile, K advanced 64 bytes (one v_mfma_i32_16x16x64_i8 step) per stage,
//     a ring of FOUR 32 KB stages in LDS: stage t + 3 is requested (global_load_lds_dwordx4, full
//     64-byte rows) while step t computes, stage t + 1 is waited for two steps after its request;
//     one s_barrier per step.
//   * LDS rows are 64 bytes; chunk c of tile row r sits at slot c ^ ((r >> 1) & 2): each 16-lane
//     group of a ds_read_b128 then touches 16 distinct 16-byte slots of the 256-byte bank row
//     (derivation in DESIGN.md section 4); the DMA writes lane-linearly, so the swizzle is applied to
//     the source address.
//   * XCD-aware workgroup remap as in the 8-wave kernel.

#include "device_utils.h"
#include "kernels.h"
#include "runtime.h"

#include <atomic>
#include <cstdlib>
#include <type_traits>

namespace ct2x {

namespace {

constexpr int B2_BM = 256, B2_BN = 256, B2_BK = 64;
constexpr int B2_THREADS = 256;
constexpr int B2_STAGES = 4;
constexpr int B2_STAGE_BYTES = (B2_BM + B2_BN) * B2_BK;  // 32 KiB
constexpr int B2_LDS_BYTES = B2_STAGES * B2_STAGE_BYTES; // 128 KiB

typedef __attribute__((address_space(3))) void b2_lds_void;
typedef const __attribute__((address_space(1))) void b2_glb_void;

__device__ __forceinline__ void b2_glds16(const int8_t* src, int8_t* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((b2_glb_void*)(src), (b2_lds_void*)(lds_wave_base), 16, 0, 0);
}

template <int ACT>
__global__ __launch_bounds__(B2_THREADS) void gemm_s8_big2_kernel(const int8_t* __restrict__ A,
                                                                  const int8_t* __restrict__ B,
                                                                  int M, int N, int K, int tiles_m,
                                                                  int tiles_n, GemmS8Epilogue ep) {
  extern __shared__ __attribute__((aligned(16))) int8_t lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 15, g = lane >> 4;

  // XCD-aware bijective remap (see gemm_s8_big.hip)
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
  const int64_t m0 = static_cast<int64_t>(tm) * B2_BM;
  const int64_t n0 = static_cast<int64_t>(tn) * B2_BN;

  // DMA map: one instruction = 64 lanes x 16 B = 16 tile rows of 64 B; wave w moves the A rows and
  // the B rows [64 w, 64 w + 64) (4 + 4 instructions per stage). lane -> row lane >> 2, LDS slot
  // lane & 3, which receives source chunk slot ^ ((row >> 1) & 2).
  const int d_row = lane >> 2, d_slot = lane & 3;
  const int d_chunk = d_slot ^ ((d_row >> 1) & 2);
  const int8_t* a_src[4];
  const int8_t* b_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave * 64 + i * 16 + d_row;
    const int64_t ar = min<int64_t>(m0 + row, M - 1);  // rows past M / N: clamped, never stored
    const int64_t br = min<int64_t>(n0 + row, N - 1);
    a_src[i] = A + ar * K + d_chunk * 16;
    b_src[i] = B + br * K + d_chunk * 16;
  }
  v4i acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = v4i{0, 0, 0, 0};

  // fragment addresses: tile row base16 + r, chunk g at slot g ^ ((r >> 1) & 2)
  const int f_slot = (g ^ ((r >> 1) & 2)) << 4;
  const int a_off = (wm * 128 + r) * B2_BK + f_slot;
  const int b_off = B2_BM * B2_BK + (wn * 128 + r) * B2_BK + f_slot;
  // (macros, not closures: a register array that a closure reaches by reference can end up in
  // scratch memory - with 256 accumulators that is fatal)
  v4i fa0[8], fb0[8], fa1[8], fb1[8];
#define CT2X_B2_LOAD_FRAGS(T, AF, BF)                                                          \
  {                                                                                            \
    const int8_t* st_ = lds + ((T) & (B2_STAGES - 1)) * B2_STAGE_BYTES;                        \
    _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_)                                           \
      AF[i_] = *reinterpret_cast<const v4i*>(st_ + a_off + i_ * 16 * B2_BK);                   \
    _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_)                                           \
      BF[j_] = *reinterpret_cast<const v4i*>(st_ + b_off + j_ * 16 * B2_BK);                   \
  }
  // The matrix operation as inline assembly with the accumulator tied in place in an AGPR quad
  // ("+a"): with the builtin the register allocator shuffled the 256 loop-carried accumulators
  // between registers at the loop back-edge and spilled 150 of them. Volatile statements keep
  // their source order, so the interleave below IS the issue order: two matrix operations, then
  // one fragment read or DMA request of a later step, in the 16-cycle shadow of the matrix pipe.
#define CT2X_B2_MFMA(I, J, AF, BF)                                                             \
  asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(acc[I][J]) : "v"(AF[I]), "v"(BF[J]));

  // Operand staging: global -> registers -> LDS (ds_write_b128), NOT the LDS-DMA of the 8-wave
  // kernel: a wave that is alone on its SIMD pays the DMA's issue cost (60 - 185 cycles per 1 KB
  // piece, MI355X_MICROARCH.md) in matrix-pipe idle time - measured with DMA staging: 1.85 POP/s at
  // 8192^3, the pipe half idle (profiles/r03_gemm_bench_big2_a.txt). A global_load_dwordx4 and a
  // ds_write_b128 issue in the 16-cycle shadow of a matrix operation.
  // Stage s: loaded in step s - 3 (8 x 16 B per lane), written to LDS in step s - 2, visible behind
  // the barrier of step s - 1, its fragments read during step s - 1, multiplied in step s.
  const int ksteps = K / B2_BK;  // >= 4, even (gemm_s8_big2_applicable)
  v4i sg0[8], sg1[8];
#define CT2X_B2_GLOAD(S, SG)                                                                   \
  {                                                                                            \
    const int k0_ = (S) * B2_BK;                                                               \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                         \
      SG[q_] = *reinterpret_cast<const v4i*>(a_src[q_] + k0_);                                 \
      SG[4 + q_] = *reinterpret_cast<const v4i*>(b_src[q_] + k0_);                             \
    }                                                                                          \
  }
#define CT2X_B2_LWRITE(S, SG)                                                                  \
  {                                                                                            \
    int8_t* base_ = lds + ((S) & (B2_STAGES - 1)) * B2_STAGE_BYTES + wave * 4096 + lane * 16;  \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                         \
      *reinterpret_cast<v4i*>(base_ + q_ * 1024) = SG[q_];                                     \
      *reinterpret_cast<v4i*>(base_ + B2_BM * B2_BK + q_ * 1024) = SG[4 + q_];                 \
    }                                                                                          \
  }
  // prologue: stages 0 and 1 into LDS, stage 2 in flight in sg0 (written during step 0)
  CT2X_B2_GLOAD(0, sg0)
  CT2X_B2_GLOAD(1, sg1)
  CT2X_B2_LWRITE(0, sg0)
  CT2X_B2_LWRITE(1, sg1)
  CT2X_B2_GLOAD(2, sg1)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  CT2X_B2_LOAD_FRAGS(0, fa0, fb0)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  // Step t (fragments of step t in AF / BF): behind the barrier stage t + 1 is complete in LDS.
  // Issue order = source order (volatile asm): after every second matrix operation ONE of
  //   slots  0 .. 15: a fragment read of step t + 1,
  //   slots 16 .. 23: a global load of stage t + 3 into SGN (the staging set written two steps ago),
  //   slots 24 .. 31: an LDS write of stage t + 2 from SGW (loaded during step t - 1).
#define CT2X_B2_STEP(T, AF, BF, AFN, BFN, SGW, SGN)                                               \
  {                                                                                               \
    const int t_ = (T);                                                                           \
    __builtin_amdgcn_s_barrier();                                                                 \
    const int8_t* st_ = lds + (min(t_ + 1, ksteps - 1) & (B2_STAGES - 1)) * B2_STAGE_BYTES;       \
    int8_t* wbase_ = lds + ((t_ + 2) & (B2_STAGES - 1)) * B2_STAGE_BYTES + wave * 4096 + lane * 16; \
    const int lk0_ = min(t_ + 3, ksteps - 1) * B2_BK; /* past the end: a harmless re-load */                                                         \
    _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                                            \
      _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) {                                          \
        CT2X_B2_MFMA(i_, j_, AF, BF)                                                              \
        const int slot_ = i_ * 4 + (j_ >> 1);                                                     \
        if ((j_ & 1) == 1) {                                                                      \
          if (slot_ < 8)                                                                          \
            AFN[slot_] = *reinterpret_cast<const v4i*>(st_ + a_off + slot_ * 16 * B2_BK);         \
          else if (slot_ < 16)                                                                    \
            BFN[slot_ - 8] = *reinterpret_cast<const v4i*>(st_ + b_off + (slot_ - 8) * 16 * B2_BK); \
          else if (slot_ < 20) {                                                                  \
            SGN[slot_ - 16] = *reinterpret_cast<const v4i*>(a_src[slot_ - 16] + lk0_); \
          } else if (slot_ < 24) {                                                                \
            SGN[slot_ - 16] = *reinterpret_cast<const v4i*>(b_src[slot_ - 20] + lk0_); \
          } else if (slot_ < 28) {                                                                \
            *reinterpret_cast<v4i*>(wbase_ + (slot_ - 24) * 1024) = SGW[slot_ - 24];  \
          } else {                                                                                \
            *reinterpret_cast<v4i*>(wbase_ + B2_BM * B2_BK + (slot_ - 28) * 1024) = SGW[slot_ - 24]; \
          }                                                                                       \
        }                                                                                         \
      }                                                                                           \
    }                                                                                             \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                            \
  }
  for (int t = 0; t < ksteps; t += 2) {  // (ksteps is even: K % 128 == 0, see the launcher)
    // step t: writes stage t + 2 from sg1 (loaded in step t - 1), loads stage t + 3 into sg0
    CT2X_B2_STEP(t, fa0, fb0, fa1, fb1, sg1, sg0)
    CT2X_B2_STEP(t + 1, fa1, fb1, fa0, fb0, sg0, sg1)
  }
#undef CT2X_B2_STEP
#undef CT2X_B2_GLOAD
#undef CT2X_B2_LWRITE
#undef CT2X_B2_MFMA
#undef CT2X_B2_LOAD_FRAGS

  // ---- epilogue (the arithmetic of gemm_s8_big_kernel; C/D layout: col = lane & 15, row =
  // (lane >> 4) * 4 + reg) -----------------------------------------------------------------------
  float* row_factor = reinterpret_cast<float*>(lds);
  int* slow_div = reinterpret_cast<int*>(lds + B2_BM * sizeof(float));
  __syncthreads();
  if (tid == 0) *slow_div = ep.plain_division ? 1 : 0;
  __syncthreads();
  if (ep.mode == 1 && tid < B2_BM) {
    const int64_t row = m0 + tid;
    const float f = row < M ? 1.f / ep.a_scale[row] : 0.f;
    row_factor[tid] = f;
    if (row < M && !exact_div_factor_ok(f)) *slow_div = 1;
  }
  float ws[8], ws_inv[8], bias[8];
  bool col_ok[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int64_t col = n0 + wn * 128 + j * 16 + r;
    col_ok[j] = col < N;
    ws[j] = (ep.mode == 1 && col_ok[j]) ? ep.w_scale[col] : 1.f;
    ws_inv[j] = 1.f / ws[j];
    if (!exact_div_divisor_ok(ws[j])) *slow_div = 1;
    bias[j] = (ep.mode == 1 && ep.bias && col_ok[j]) ? ep.bias[col] : 0.f;
  }
  __syncthreads();
  const bool fast_div = *slow_div == 0;  // workgroup-uniform
  const bool has_res = ep.mode == 1 && ep.residual != nullptr;
  const int64_t col0 = n0 + wn * 128 + r;
  auto load_residual = [&](int i, float (&res)[4][8]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t row = m0 + wm * 128 + i * 16 + g * 4 + e;
      const float* p = ep.residual + row * ep.ldy + col0;
#pragma unroll
      for (int j = 0; j < 8; ++j) res[e][j] = (has_res && row < M && col_ok[j]) ? p[j * 16] : 0.f;
    }
  };
  auto store_group = [&](int i, const float (&res)[4][8], auto fast_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int lrow = wm * 128 + i * 16 + g * 4 + e;
      const int64_t row = m0 + lrow;
      if (row >= M) continue;
      if (ep.mode == 0) {
        int32_t* p = ep.c_i32 + row * ep.ldy + col0;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (col_ok[j]) p[j * 16] = acc[i][j][e];
      } else {
        const float r_a = row_factor[lrow];
        float* p = ep.y + row * ep.ldy + col0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
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
  float resA[4][8];
  // (spelled out, not a loop: a row-group index that is not a compile-time constant would demote
  // the 256 accumulator registers to scratch memory)
#define CT2X_B2_GROUP(I, TAG)              \
  __builtin_amdgcn_sched_barrier(0);       \
  load_residual(I, resA);                  \
  store_group(I, resA, TAG);               \
  __builtin_amdgcn_sched_barrier(0);
#define CT2X_B2_EPILOGUE(TAG)                                                          \
  CT2X_B2_GROUP(0, TAG) CT2X_B2_GROUP(1, TAG) CT2X_B2_GROUP(2, TAG) CT2X_B2_GROUP(3, TAG) \
  CT2X_B2_GROUP(4, TAG) CT2X_B2_GROUP(5, TAG) CT2X_B2_GROUP(6, TAG) CT2X_B2_GROUP(7, TAG)
  if (fast_div) {
    CT2X_B2_EPILOGUE(std::true_type{})
  } else {
    CT2X_B2_EPILOGUE(std::false_type{})
  }
#undef CT2X_B2_EPILOGUE
#undef CT2X_B2_GROUP
}

template <int ACT>
void launch_big2(const int8_t* a, const int8_t* b, int M, int N, int K, int tiles_m, int tiles_n,
                 const GemmS8Epilogue& ep, hipStream_t stream) {
  static std::atomic<bool> configured[64];
  int dev = 0;
  CT2X_HIP_CHECK(hipGetDevice(&dev));
  if (!configured[dev & 63].load(std::memory_order_acquire)) {
    CT2X_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_s8_big2_kernel<ACT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, B2_LDS_BYTES));
    configured[dev & 63].store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL((gemm_s8_big2_kernel<ACT>), dim3(tiles_m * tiles_n), dim3(B2_THREADS), B2_LDS_BYTES,
                     stream, a, b, M, N, K, tiles_m, tiles_n, ep);
}

}  // namespace

bool gemm_s8_big2_applicable(int64_t m, int64_t n, int64_t k) {
  if (k % (2 * B2_BK) != 0 || k < 4 * B2_BK) return false;  // (the K loop is unrolled by two steps)
  const int64_t tiles = ((m + B2_BM - 1) / B2_BM) * ((n + B2_BN - 1) / B2_BN);
  return m >= 512 && n >= 512 && tiles >= 96;
}

void launch_gemm_s8_big2(const int8_t* a, const int8_t* b, int M, int N, int K,
                         const GemmS8Epilogue& ep_in, hipStream_t stream) {
  GemmS8Epilogue ep = ep_in;
  const int tiles_m = (M + B2_BM - 1) / B2_BM, tiles_n = (N + B2_BN - 1) / B2_BN;
  const int act = (ep.mode == 1 && ep.act > 0) ? ep.act : -1;
  switch (act) {
    case 1: launch_big2<1>(a, b, M, N, K, tiles_m, tiles_n, ep, stream); break;
    case 2: launch_big2<2>(a, b, M, N, K, tiles_m, tiles_n, ep, stream); break;
    case 3: launch_big2<3>(a, b, M, N, K, tiles_m, tiles_n, ep, stream); break;
    case 4: launch_big2<4>(a, b, M, N, K, tiles_m, tiles_n, ep, stream); break;
    case 5: launch_big2<5>(a, b, M, N, K, tiles_m, tiles_n, ep, stream); break;
    case 6: launch_big2<6>(a, b, M, N, K, tiles_m, tiles_n, ep, stream); break;
    default: launch_big2<-1>(a, b, M, N, K, tiles_m, tiles_n, ep, stream); break;
  }
}

}  // namespace ct2x
