// EXPERIMENT (round 6, measured slower, not built): the K loop of gemm_s8_big3.hip with 64-byte
// stages in a ring of 4 / 5 (instead of two 128-byte stages), the four DMA requests of the stage
// ring - 1 ahead issued BETWEEN the matrix instructions of a stage (instead of eight in a fragment-
// read phase), and the first waits of a tile relaxed by the previous tile's store count. Hypothesis:
// the read phase that carries the eight requests is the long pole (the guide prices a request there
// at 100 - 185 clocks). Result on the Llama-3-8B prefill shape [2048, 14336] x 4096
// (profiles/r06_gemm_ab2_deep_ring_negative.txt): K loops only 104.2 -> 111.5 us (256-wide tiles),
// 103.4 -> 106.5 (224-wide); fused Swish form 145.0 -> 156.7 us. The 64-byte row requests cost more
// than the re-placed issue slots return; the kernel also failed [4200, 4404] x 256 with a residual
// (an edge case of the request cursor at K = 4 stages that was not chased). Kept for the record.
// MFMA-bound int8 projection GEMM, third form (round 4): the kernel behind launch_gemm_s8_big for
// M >= ~1024 rows (encoder, cross-attention K/V projection, decoder-only prefill). Same contract
// as gemm_s8_big.hip / gemm_s8.hip: C = A[m,k] . B[n,k]^T exact in int32, fused Dequantize epilogue
// in the reference's CPU operand order (src/cpu/kernels.cc:666-673), replacing
// src/cuda/primitives.cu:571-597 + src/ops/dequantize_gpu.cu:31-55.
//
// What changed against gemm_s8_big.hip (kept for A/B: CT2X_BIG_GEMM_V=1 in a tuning build):
//   * The operands are SWAPPED at the matrix instruction (A-operand = weight rows, B-operand =
//     activation rows): D[i][j] then has i = output column n, j = output row m, and a lane holds
//     FOUR CONSECUTIVE n of one output row per accumulator -> the epilogue stores 16 bytes per
//     lane (global_store_dwordx4) instead of 4, and the per-column operands (weight scale, its
//     reciprocal, bias) are one 16-byte LDS read per column tile.
//   * (v_mfma_i32_32x32x32_i8 was tried first - half the matrix instructions and operand register
//     reads per MAC, and the higher ceiling of the two on zero-filled operands - and measured
//     SLOWER on random operands: the chip clocks to its power budget, and the 32x32 form sustains
//     3.15 - 3.4 POP/s where 16x16x64 sustains 3.75 - 3.9 in a register-only loop
//     (tools/mfma_rate_probe.cpp, profiles/r04_mfma_rate_probe.txt); the same kernel ran 90 us
//     with it at 4096^3 against 63 for the round-3 kernel. Kept: v_mfma_i32_16x16x64_i8.)
//   * PERSISTENT: one workgroup per CU walks a strided list of tiles; the K-tile stream never
//     drains between tiles - the first stage of tile i+1 and its epilogue operands (row scales,
//     column scales, bias: 3 x 256 floats by 4-byte LDS-DMA into a double-buffered parameter
//     block) are requested during the last K-tile of tile i, and the stores of tile i leave
//     through L2 while tile i+1 computes (gfx950 counts loads and stores on one in-order vmcnt,
//     so the first DMA wait of the next tile is also the point where tile i's stores are known
//     to have reached L2 - about three phases after they were issued).
//   * (A STREAM-K work split - equal runs of K-tiles per workgroup, a tile shared by two
//     workgroups finished by the last arriver through a 256 KiB slab of raw accumulators - was
//     built on top of this kernel, bit-exact in all arms, and measured SLOWER: 160 - 165 us against
//     130 - 137 at the Llama-3-8B prefill shape (448 tiles on 256 CUs). The balance it buys (1.75
//     instead of 2 tile times) is lost twice over: the workgroups of an XCD no longer walk K in
//     lockstep, so an operand line is no longer in L2 when the next CU asks for it, and the slab
//     traffic comes on top. tools/experiments/gemm_s8_big3_streamk.hip, profiles/r04_gemm_bench_e / f.)
//   * Interior tiles take an epilogue without bounds predicates; edge tiles keep them.
// Unchanged: 256 x 256 tile, K advanced 128 bytes per stage, L2 -> LDS by global_load_lds_dwordx4
// with the bank swizzle on the source address (chunk c of tile row r at slot c ^ ((r >> 1) & 7):
// the four 16-lane groups of a ds_read_b128 each touch 16 distinct 16-byte slots,
// SQ_LDS_BANK_CONFLICT = 0), two 64 KiB stages, raw s_barrier phases,
// the two waves of a SIMD one phase apart (LOAD | MFMA ping-pong), XCD-aware tile order.

#include "device_utils.h"
#include "kernels.h"
#include "runtime.h"

#include <atomic>
#include <cstdlib>

namespace ct2x {

namespace {

constexpr int T_BM = 256;
constexpr int T_KS = 64;                                  // K bytes per stage = one MFMA K-step
constexpr int T_THREADS = 512;
constexpr int T_PARAM_FLOATS = 4 * 256;                   // a_scale -> 1/a_scale | w_scale | bias | 1/w_scale
constexpr int T_PARAM_BYTES = T_PARAM_FLOATS * 4;

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__device__ __forceinline__ void glds16(const int8_t* src, int8_t* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((glb_void*)(src), (lds_void*)(lds_wave_base), 16, 0, 0);
}
__device__ __forceinline__ void glds4(const float* src, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((glb_void*)(src), (lds_void*)(lds_wave_base), 4, 0, 0);
}

// IEEE division t / d as the compiler expands it on gfx950 (v_div_scale x 2, v_rcp, the Newton /
// residual FMA chain, v_div_fmas, v_div_fixup) MINUS the four instructions that are the identity
// when nothing has to be rescaled: for 2^-20 <= |t| <= 2^6 and 1 <= d < 2^93 (callers check) both
// v_div_scale return their operand unchanged with VCC = 0 (ISA: no denormal operand, reciprocal or
// quotient, exponent difference inside +-96, numerator exponent > 23), v_div_fmas is then a plain
// FMA and v_div_fixup passes a finite quotient of finite non-zero operands through. What is left is
// the SAME FMA chain on the same values, so the result is the correctly rounded quotient the plain
// `/` returns (checked on the device against `/`: ct2x_debug_fast_div_check, tests/test_ops_gpu.py).
__device__ __forceinline__ float div_noscale(float t, float d) {
  float y = __builtin_amdgcn_rcpf(d);
  const float e = __builtin_fmaf(-d, y, 1.f);
  y = __builtin_fmaf(e, y, y);
  const float q0 = t * y;
  const float r0 = __builtin_fmaf(-d, q0, t);
  const float q1 = __builtin_fmaf(r0, y, q0);
  const float r1 = __builtin_fmaf(-d, q1, t);
  return __builtin_fmaf(r1, y, q1);
}
__device__ __forceinline__ bool div_noscale_numerator_ok(float t) {
  const float a = __builtin_fabsf(t);
  return a >= 9.5367431640625e-07f /* 2^-20 */ && a <= 64.f;
}

// The exponential family of activations on four values at once (x / (1 + exp(-s x)): Swish,
// GELUSigmoid; 1 / (1 + exp(-x)): Sigmoid), the divisions on the reduced sequence when every
// numerator of the WAVE is inside its range (wave-uniform branch; real activations are), the plain
// division otherwise - bit-identical either way.
template <int ACT>
__device__ __forceinline__ v4f activation4(v4f v) {
  v4f out;
  if constexpr (ACT == 2 || ACT == 4 || ACT == 6) {
    v4f d, t;
    bool ok = true;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float arg = ACT == 4 ? -1.702f * v[e] : -v[e];
      d[e] = 1.f + exact_exp_avx2(arg);
      t[e] = ACT == 6 ? 1.f : v[e];
      // d < 2^93 <=> the exponential's argument <= 64 (|v| <= 37 covers GELUSigmoid's 1.702 v)
      ok = ok && (ACT == 6 ? __builtin_fabsf(v[e]) <= 64.f
                           : (div_noscale_numerator_ok(v[e]) && (ACT != 4 || __builtin_fabsf(v[e]) <= 37.f)));
    }
    if (__builtin_amdgcn_ballot_w64(!ok) == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) out[e] = div_noscale(t[e], d[e]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) out[e] = t[e] / d[e];
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      out[e] = apply_activation(v[e], ACT);
  }
  return out;
}

// Epilogue of an interior tile (every 16-byte store in bounds and aligned, scales inside the range
// of exact_div, no residual): y = act(c * (1 / a_scale) / w_scale + bias), CPU operand order.
template <int ACT, bool HAS_BIAS, int NI, int NJ>
__device__ __forceinline__ void big3_epilogue_interior(const v4i (&acc)[NI][NJ], const float* par, const float* ra,
                                                       int cl0, float* __restrict__ y0, int64_t ldy) {
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const v4f ws = *reinterpret_cast<const v4f*>(par + 256 + cl0 + j * 16);
    const v4f wi = *reinterpret_cast<const v4f*>(par + 768 + cl0 + j * 16);
    v4f bs = v4f{0, 0, 0, 0};
    if (HAS_BIAS) bs = *reinterpret_cast<const v4f*>(par + 512 + cl0 + j * 16);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const v4i c4 = acc[i][j];
      const float r = ra[i * 16];
      v4f v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = exact_div(static_cast<float>(c4[e]) * r, ws[e], wi[e]);
        if (HAS_BIAS) v[e] = v[e] + bs[e];
      }
      *reinterpret_cast<v4f*>(y0 + static_cast<int64_t>(i) * 16 * ldy + j * 16) = activation4<ACT>(v);
      // (one accumulator at a time: the four elements interleave, successive accumulators do not -
      // the cheap forms otherwise pull every operand of the tile forward and spill)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// ACT: ops::ActivationType 1..6 of the fused form as a constant; -1 = ReLU / none (ep.act) and the int32-out mode.
// VAR (timing experiments of a tuning build, ACT = -1 only): bit 0 = weight fragment in the outer
// MFMA loop, bit 1 = no epilogue (K loops only).
// SHAPE (round 6): the output tile and the wave grid over it -
//   0: 256 x 256, waves 2 (rows) x 4 (columns), 128 x 64 per wave   (8 x 4 accumulators)
//   1: 256 x 224, waves 4 x 2,                   64 x 112 per wave   (4 x 7)
//   2: 256 x 192, waves 4 x 2,                   64 x 96 per wave    (4 x 6)
// The narrower tiles exist for the LAST ROUND: one workgroup per CU walks ceil(tiles / 256)
// rounds of tiles, so [2048, 14336] in 256-wide tiles is 448 tiles = two rounds with 64 CUs idle
// in the second, and in 224-wide tiles 512 tiles = two full rounds of 7/8 the work each
// (launch_gemm_s8_big3 picks the width with the least rounds x width). The LDS image, the DMA map
// and the K-loop schedule are the same; the weight rows past the tile's width are not fetched.
template <int SHAPE>
struct Big3Shape {
  static constexpr bool kWide = SHAPE != 0;                 // 4 x 2 wave grid
  static constexpr int NI = kWide ? 4 : 8;                  // activation-row fragments per wave
  static constexpr int NJ = SHAPE == 0 ? 4 : (SHAPE == 1 ? 7 : 6);  // weight-row fragments per wave
  static constexpr int BN = (kWide ? 2 : 4) * NJ * 16;
  // LDS: a ring of stages (256 activation rows + BN weight rows of 64 K-bytes), two parameter
  // blocks, a flag - 136.0 / 158.0 / 148.0 KiB of the CU's 160
  static constexpr int kStageBytes = (T_BM + BN) * T_KS;    // 32 / 30 / 28 KiB
  static constexpr int kRing = SHAPE == 0 ? 4 : 5;
  static constexpr int kRingBytes = kRing * kStageBytes;
  static constexpr int kFlagOffset = kRingBytes + 2 * T_PARAM_BYTES;
  static constexpr int kLdsBytes = kFlagOffset + 16;
};
__device__ __forceinline__ int swz4(int x) { return (0x78 >> (2 * (x & 3))) & 3; }  // {0, 2, 3, 1}
constexpr int big3_tile_n(int shape) { return shape == 0 ? 256 : (shape == 1 ? 224 : 192); }

template <int ACT, int VAR = 0, int SHAPE = 0>
__global__ __launch_bounds__(T_THREADS) void gemm_s8_big3_kernel(const int8_t* __restrict__ A,
                                                                 const int8_t* __restrict__ B,
                                                                 int M, int N, int K,
                                                                 int tiles_m, int tiles_n,
                                                                 GemmS8Epilogue ep) {
  extern __shared__ __attribute__((aligned(16))) int8_t lds[];  // the ONLY LDS object

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  using S = Big3Shape<SHAPE>;
  constexpr int NI = S::NI, NJ = S::NJ, BN = S::BN;
  constexpr int WROWS = NI * 16, WCOLS = NJ * 16;  // the wave's part of the tile
  // (the two waves of a SIMD are w and w + 4: the ping-pong groups are the wave number's bit 2)
  const int group = wave >> 2;
  const int wm = S::kWide ? (wave & 3) : (wave >> 2), wn = S::kWide ? (wave >> 2) : (wave & 3);
  const int r = lane & 15, g = lane >> 4;

  // Persistent tile list: hardware block b sits on XCD b % 8 (observed, speed only). XCD x owns
  // the contiguous logical tile range [first, first + count) (band order below: an 8-row band
  // walked column by column - shared A rows / B columns in that XCD's L2); its blocks walk it
  // with a stride of the XCD's block count, all at the same K offset at the same time (an
  // operand line is fetched once per XCD and found in L2 by the other CUs), and a ragged last
  // round is spread over ALL XCDs (measured: a last round that left two XCDs idle and six full
  // ran 6 % slower than the same tiles spread over eight - the clocks follow the load per XCD).
  const int ntiles = tiles_m * tiles_n;
  const int G = gridDim.x;
  int L, Lend, Lstep;
  {
    const int b = blockIdx.x;
    const int xcd = b & 7;
    const int q = ntiles >> 3, rem = ntiles & 7;
    const int first = xcd * q + min(xcd, rem);
    Lstep = (G >> 3) + (xcd < (G & 7) ? 1 : 0);
    L = first + (b >> 3);
    Lend = first + q + (xcd < rem ? 1 : 0);
  }
  auto coords = [&](int L, int& tm, int& tn) {
    constexpr int BAND = 8;
    const int band = L / (BAND * tiles_n);
    const int band_rows = min(BAND, tiles_m - band * BAND);
    const int in_band = L - band * BAND * tiles_n;
    tm = band * BAND + in_band % band_rows;
    tn = in_band / band_rows;
  };

  // DMA map: one wave instruction moves 64 x 16 B = 16 tile rows of one stage (64 K-bytes per
  // row), the workgroup's eight 128 rows: thread -> row tid / 4 of the 128-row chunk, slot tid % 4,
  // which receives source chunk (tid % 4) ^ f((row >> 2) & 3), f = {0, 2, 3, 1} (the four 16-lane
  // groups of a ds_read_b128 fragment read then touch 16 distinct 16-byte slots of the 256-byte
  // bank row). A stage = four requests per wave: activation rows 0-127, 128-255, weight rows
  // 0-127, 128-BN. Rows past M / N are clamped (their products are never stored); in the narrower
  // tiles the waves whose rows of the last request lie past the tile's width repeat their third
  // request instead (same bytes to the same place: every wave has four requests per stage in
  // flight, and the matrix phase that carries them has no branch).
  const int ld_row = tid >> 2;
  const uint32_t ld_chunk = ((tid & 3) ^ swz4((ld_row >> 2) & 3)) * 16;
  // Source of a DMA request = uniform tile base (scalar registers) + a 32-bit lane offset that
  // only changes at the M / N edge (launch_gemm_s8_big3 checks that a tile's 256 rows span
  // < 4 GiB).
  uint32_t a_rel[2], b_rel[2];
  const int8_t* a_base;
  const int8_t* b_base;
  const bool b1_inside = BN == 256 || wave < (BN - 128) / 16;
  auto set_sources = [&](int tm, int tn) {
    const int m_last = M - 1 - tm * T_BM, n_last = N - 1 - tn * BN;
    int row0 = ld_row;  // opaque: the row numbers are not worth registers across the K loop
    asm volatile("" : "+v"(row0));
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      a_rel[c] = static_cast<uint32_t>(min(c * 128 + row0, m_last)) * static_cast<uint32_t>(K) + ld_chunk;
      b_rel[c] = static_cast<uint32_t>(min((c == 1 && !b1_inside ? 0 : c * 128) + row0, n_last)) *
                     static_cast<uint32_t>(K) + ld_chunk;
    }
    a_base = A + static_cast<int64_t>(tm) * T_BM * K;
    b_base = B + static_cast<int64_t>(tn) * BN * K;
  };
  const int b1_slot = b1_inside ? 16384 + 8192 : 16384;
  // Request p of 4 of the stage at K offset k0 into ring slot `slot` (byte offset).
  auto piece = [&](int p, int slot, int k0) {
    int8_t* base = lds + slot + wave * 1024;
    if (p == 0) glds16(a_base + k0 + a_rel[0], base);
    else if (p == 1) glds16(a_base + k0 + a_rel[1], base + 8192);
    else if (p == 2) glds16(b_base + k0 + b_rel[0], base + 16384);
    else glds16(b_base + k0 + b_rel[1], base + b1_slot);
  };
  // Epilogue operands of a tile: waves 0-3 (group 0) fetch 64 floats each of the three arrays.
  auto stage_params = [&](int pbuf, int tm, int tn) {
    if (ep.mode != 1 || wave >= 4) return;
    float* p = reinterpret_cast<float*>(lds + S::kRingBytes + pbuf * T_PARAM_BYTES) + wave * 64;
    const int row = min(tm * T_BM + wave * 64 + lane, M - 1);
    const int col = min(tn * BN + wave * 64 + lane, N - 1);
    glds4(ep.a_scale + row, p);
    glds4(ep.w_scale + col, p + 256);
    if (ep.bias) glds4(ep.bias + col, p + 512);
  };

  v4i acc[NI][NJ];

  // Fragment addresses: tile row = base16 + r; K chunk g of 4 (16 B each) of the stage row, stored
  // at slot g ^ f((r >> 2) & 3).
  const int frag_chunk = (g ^ swz4((r >> 2) & 3)) << 4;
  const int x_off = (wm * WROWS + r) * T_KS + frag_chunk;
  const int w_off = T_BM * T_KS + (wn * WCOLS + r) * T_KS + frag_chunk;
  v4i xf[NI], wf[NJ];
  auto load_frags = [&](int slot) {
    const int8_t* tile = lds + slot;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      wf[j] = *reinterpret_cast<const v4i*>(tile + w_off + j * 16 * T_KS);
#pragma unroll
    for (int i = 0; i < NI; ++i)
      xf[i] = *reinterpret_cast<const v4i*>(tile + x_off + i * 16 * T_KS);
  };
  // The matrix phase of a stage carries the four DMA requests of the stage kRing - 1 ahead, one
  // after every quarter of its matrix instructions: a request issued between matrix instructions
  // costs its issue slot (~60 clocks, hidden behind the matrix pipe), the same request in the
  // fragment-read phase 100 - 185 clocks on that phase's critical path (round 5 ran all eight
  // requests of a 128-byte stage there: its read phases took 2 x the matrix phases).
  auto mfmas = [&](int slot, int k0) {
    __builtin_amdgcn_s_setprio(1);
    constexpr int PER = NI / 4;  // fragment rows per request
#pragma unroll
    for (int i = 0; i < NI; ++i) {
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[j], xf[i], acc[i][j], 0, 0, 0);
      if ((i + 1) % PER == 0) piece((i + 1) / PER - 1, slot, k0);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      __builtin_amdgcn_sched_group_barrier(0x008, PER * NJ, 0);  // matrix instructions
      __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);         // one DMA request
    }
    __builtin_amdgcn_s_setprio(0);
  };
  // Phase boundary: nothing may be scheduled across it by the compiler.
  auto phase_barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  int* slow_div = reinterpret_cast<int*>(lds + S::kFlagOffset);
  if (tid == 0) *slow_div = ep.plain_division ? 1 : 0;

  if (L >= Lend) return;  // (fewer tiles than blocks on this XCD: never with the launcher's grid)
  int tm, tn;
  coords(L, tm, tn);
  const int nstages = K / T_KS;  // >= kRing (gemm_s8_big_applicable: K >= 512)
  constexpr int AHEAD = S::kRing - 1;
  // The request stream runs AHEAD stages in front of the matrix phases, across tile boundaries:
  // issue cursor = (tile whose stages are being requested, K offset, ring slot).
  set_sources(tm, tn);
  stage_params(0, tm, tn);
  int issue_k = 0, issue_slot = 0;
  int issue_L = L;             // tile of the issue cursor
  bool issue_open = true;      // false: nothing left to request (the last stage is requested again)
  int issue_round = 0;
#pragma unroll
  for (int a = 0; a < AHEAD; ++a) {
#pragma unroll
    for (int p = 0; p < 4; ++p) piece(p, issue_slot, issue_k);
    issue_k += T_KS;
    issue_slot += S::kStageBytes;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  int read_slot = 0;
  int round = 0;   // parameter buffer = round & 1

  // Ping-pong schedule inside a tile: the two waves of a SIMD (w, w + 4) run one phase apart, kept
  // there by one extra barrier in front of group 1 and one behind group 0:
  //   group 0:      L(s) | M(s) | L(s+1) | M(s+1) | ...
  //   group 1:  --  |  L(s) | M(s) | L(s+1) | ...
  // L(s) = the 12 (11, 10) fragment reads of stage s; M(s) = its matrix instructions with the four
  // requests of stage s + AHEAD between them, into the ring slot stage s - 1 left (both groups
  // finished reading it before group 0's L(s)). Every wave waits for ITS share of stage s + 1 in
  // the interval before group 0's L(s + 1): group 0 at the end of M(s) - AHEAD - 1 stages of
  // requests may stay in flight behind it -, group 1 at the end of its L(s), before it requests
  // stage s + AHEAD (AHEAD - 2 stages in flight). The stores of the previous tile's epilogue sit in
  // the same in-order queue (gfx950 counts loads and stores on one vmcnt): behind an interior tile -
  // exactly NI x NJ 16-byte stores per wave, all NEWER than the stages requested before the
  // epilogue - the first AHEAD - 1 waits of the next tile allow that many more operations in flight,
  // so the result has AHEAD - 1 stages of matrix work to leave before a wait drains it.
  int relax_left = 0;
  while (true) {
    const int Lnext = L + Lstep;
    const bool has_next = Lnext < Lend;
    int tm_n = 0, tn_n = 0;
    if (has_next) coords(Lnext, tm_n, tn_n);
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = v4i{0, 0, 0, 0};

    phase_barrier();  // this tile's first stage (and its parameters) are in LDS
    if (group == 1) phase_barrier();
    for (int s = 0; s < nstages; ++s) {
      load_frags(read_slot);
      if (group == 1) {
        if (relax_left > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 2) * 4 + NI * NJ) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 2) * 4) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      phase_barrier();
      mfmas(issue_slot, issue_k);
      __builtin_amdgcn_sched_barrier(0);
      // advance the issue cursor (scalar work; the next tile's sources when this one is requested)
      issue_slot += S::kStageBytes;
      if (issue_slot == S::kRingBytes) issue_slot = 0;
      read_slot += S::kStageBytes;
      if (read_slot == S::kRingBytes) read_slot = 0;
      if (issue_open) {
        issue_k += T_KS;
        if (issue_k == K) {
          const int Li = issue_L + Lstep;
          if (Li < Lend) {
            int tmi, tni;
            coords(Li, tmi, tni);
            set_sources(tmi, tni);
            ++issue_round;
            stage_params(issue_round & 1, tmi, tni);
            issue_L = Li;
            issue_k = 0;
          } else {
            issue_open = false;
            issue_k = K - T_KS;
          }
        }
      }
      if (group == 0) {
        if (relax_left > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * 4 + NI * NJ) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * 4) : "memory");
      }
      relax_left = max(relax_left - 1, 0);
      phase_barrier();
    }
    if (group == 0) phase_barrier();
    bool tile_full = false;

    // ---- epilogue of tile (tm, tn); both groups are aligned here -------------------------
    {
    // C/D layout of the 16x16 MFMA: col j = lane & 15, row i = (lane >> 4) * 4 + reg; with the
    // operands swapped i is the output column and j the output row.
    // (The lane coordinates are re-derived behind an opaque move so that no per-lane address of
    // the epilogue is hoisted over the K loop, where every register is taken.)
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int re = lane_e & 15, ge = lane_e >> 4;
    const int64_t m0 = static_cast<int64_t>(tm) * T_BM;
    const int64_t n0 = static_cast<int64_t>(tn) * BN;
    // 16-byte stores need the row pitch and the base addresses to allow them.
    tile_full = (m0 + T_BM <= M) && (n0 + BN <= N) && ((ep.ldy & 3) == 0) &&
                      (((reinterpret_cast<uintptr_t>(ep.y) | reinterpret_cast<uintptr_t>(ep.c_i32) |
                         reinterpret_cast<uintptr_t>(ep.residual)) & 15) == 0);
    const bool full = tile_full;
    if (VAR & 2) {  // K loops only: one conditional store keeps the accumulators alive
      int keep = 0;
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) keep ^= acc[i][j][0] ^ acc[i][j][1] ^ acc[i][j][2] ^ acc[i][j][3];
      if (keep == 0x7fffffff && ep.c_i32) ep.c_i32[0] = keep;
    } else if (ep.mode == 0) {
      int32_t* __restrict__ c = ep.c_i32;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int cl = wn * WCOLS + j * 16 + ge * 4;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int64_t row = m0 + wm * WROWS + i * 16 + re;
          int32_t* p = c + row * ep.ldy + n0 + cl;
          const v4i c4 = acc[i][j];
          if (full) {
            *reinterpret_cast<v4i*>(p) = c4;
          } else if (row < M) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (n0 + cl + e < N) p[e] = c4[e];
          }
        }
      }
    } else {
      float* par = reinterpret_cast<float*>(lds + S::kRingBytes + (round & 1) * T_PARAM_BYTES);
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
      const float* ra = par + wm * WROWS + re;  // 1 / a_scale of the lane's rows: ra[i * 16]
      // Interior tile with in-range scales and no residual (every tile of a Dense whose shape is a
      // multiple of the tile): straight-line code, nothing but the arithmetic and the stores.
      // (ACT > 0 only: the store-bound ReLU / plain forms keep the general code - with a second
      // copy of the epilogue the register allocator spills inside their K loop)
      if (ACT > 0 && fast_div && full && !has_res) {
        const int cl0 = wn * WCOLS + ge * 4;  // the lane's first column of column fragment 0
        float* y0 = ep.y + ((m0 + wm * WROWS + re) * ep.ldy + n0 + cl0);
        if (has_bias) big3_epilogue_interior<ACT, true, NI, NJ>(acc, par, ra, cl0, y0, ep.ldy);
        else big3_epilogue_interior<ACT, false, NI, NJ>(acc, par, ra, cl0, y0, ep.ldy);
      } else {
        float* __restrict__ y = ep.y;
        const float* __restrict__ res = ep.residual;
#define CT2X_BIG3_EPILOGUE(FAST)                                                                  \
  _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                \
    const int cl = wn * WCOLS + j * 16 + ge * 4;                                                  \
    const v4f ws = *reinterpret_cast<const v4f*>(par + 256 + cl);                                 \
    const v4f wi = *reinterpret_cast<const v4f*>(par + 768 + cl);                                 \
    const v4f bs = has_bias ? *reinterpret_cast<const v4f*>(par + 512 + cl) : v4f{0, 0, 0, 0};    \
    _Pragma("unroll") for (int i = 0; i < NI; ++i) {                                              \
      const v4i c4 = acc[i][j];                                                                   \
      const int64_t row = m0 + wm * WROWS + i * 16 + re;                                          \
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
        float v = static_cast<float>(c4[e]) * ra[i * 16];                                         \
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
      if (ACT > 0) __builtin_amdgcn_sched_barrier(0); /* (their edge / slow path: registers first) */ \
    }                                                                                             \
  }
        if (fast_div) {
          CT2X_BIG3_EPILOGUE(true)
        } else {
          CT2X_BIG3_EPILOGUE(false)
        }
#undef CT2X_BIG3_EPILOGUE
      }
    }

    }

    relax_left = (tile_full && !(VAR & 2)) ? AHEAD - 1 : 0;
    if (!has_next) break;
    L = Lnext;
    tm = tm_n;
    tn = tn_n;
    ++round;
  }
  // (the request stream ran past the last stage: nothing may land in the LDS of a finished workgroup)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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

template <int ACT, int VAR = 0, int SHAPE = 0>
void launch_big3_shape(const int8_t* a, const int8_t* b, int M, int N, int K, int tiles_m, int tiles_n,
                       const GemmS8Epilogue& ep, int grid, hipStream_t stream) {
  // > 64 KiB of dynamic LDS has to be opted into once per device.
  static std::atomic<bool> configured[64];
  int dev = 0;
  CT2X_HIP_CHECK(hipGetDevice(&dev));
  if (!configured[dev & 63].load(std::memory_order_acquire)) {
    CT2X_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_s8_big3_kernel<ACT, VAR, SHAPE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, Big3Shape<SHAPE>::kLdsBytes));
    configured[dev & 63].store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL((gemm_s8_big3_kernel<ACT, VAR, SHAPE>), dim3(grid), dim3(T_THREADS), Big3Shape<SHAPE>::kLdsBytes, stream,
                     a, b, M, N, K, tiles_m, tiles_n, ep);
}

template <int ACT, int VAR = 0>
void launch_big3(const int8_t* a, const int8_t* b, int M, int N, int K, int shape, int tiles_m, int tiles_n,
                 const GemmS8Epilogue& ep, int grid, hipStream_t stream) {
  if (shape == 1) launch_big3_shape<ACT, VAR, 1>(a, b, M, N, K, tiles_m, tiles_n, ep, grid, stream);
  else if (shape == 2) launch_big3_shape<ACT, VAR, 2>(a, b, M, N, K, tiles_m, tiles_n, ep, grid, stream);
  else launch_big3_shape<ACT, VAR, 0>(a, b, M, N, K, tiles_m, tiles_n, ep, grid, stream);
}

// Tile width for [M, N]: the one whose walk costs least, rounds x width (a workgroup per CU walks
// ceil(tiles / CUs) tiles; a tile's time follows its width). Ties go to the wider tile (fewer
// operand bytes per product). CT2X_BIG3_SHAPE=0|1|2 (tuning builds) forces one.
int big3_pick_shape(int M, int N, int cus) {
  const int tiles_m = (M + T_BM - 1) / T_BM;
  int best = 0;
  long best_cost = 0;
  for (int shape = 0; shape < 3; ++shape) {
    const int bn = big3_tile_n(shape);
    const long tiles = static_cast<long>(tiles_m) * ((N + bn - 1) / bn);
    const long cost = ((tiles + cus - 1) / cus) * bn;
    if (shape == 0 || cost < best_cost) {
      best = shape;
      best_cost = cost;
    }
  }
  return best;
}

}  // namespace

// div_noscale against the plain division on the device: numerators t = +-2^u (1 + f) over the whole
// admitted range (and its two ends exactly), denominators d = 1 + exp(-t) as Swish forms them for
// one half of the pairs, any d in [1, 2^93) for the other half. Counts differing bit patterns.
__global__ void fast_div_check_kernel(unsigned long long n, unsigned seed, unsigned long long* mismatches) {
  unsigned long long i = blockIdx.x * static_cast<unsigned long long>(blockDim.x) + threadIdx.x;
  const unsigned long long stride = static_cast<unsigned long long>(gridDim.x) * blockDim.x;
  unsigned long long bad = 0;
  for (; i < n; i += stride) {
    unsigned h = static_cast<unsigned>(i * 2654435761ull) ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    unsigned g = h * 747796405u + 2891336453u;
    g ^= g >> 16; g *= 2246822519u; g ^= g >> 13;
    // t: exponent -20 .. 5 (and 64.0 itself), random mantissa and sign
    const int ex = static_cast<int>(h % 26u) - 20;
    float t = __builtin_bit_cast(float, static_cast<unsigned>((ex + 127) << 23) | (g & 0x7fffffu));
    if ((h >> 8) % 1021u == 0) t = 64.f;
    if ((h >> 8) % 1021u == 1) t = 9.5367431640625e-07f;
    if (g & 0x800000u) t = -t;
    float d;
    if (h & 0x80000000u) {
      d = 1.f + exact_exp_avx2(-t);
    } else {
      const int ed = static_cast<int>((h >> 5) % 93u);
      d = __builtin_bit_cast(float, static_cast<unsigned>((ed + 127) << 23) | ((g >> 3) & 0x7fffffu));
    }
    const float a = div_noscale(t, d), b = t / d;
    const float c = div_noscale(1.f, d), e = 1.f / d;
    if (__builtin_bit_cast(unsigned, a) != __builtin_bit_cast(unsigned, b)) ++bad;
    if (__builtin_bit_cast(unsigned, c) != __builtin_bit_cast(unsigned, e)) ++bad;
  }
  if (bad) atomicAdd(mismatches, bad);
}

void launch_fast_div_check(unsigned long long n, unsigned seed, unsigned long long* d_mismatches,
                           hipStream_t stream) {
  hipLaunchKernelGGL(fast_div_check_kernel, dim3(2048), dim3(256), 0, stream, n, seed, d_mismatches);
}

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
  static const int shape_override = [] {
    const char* e = tuning_env("CT2X_BIG3_SHAPE");
    return e ? std::atoi(e) : -1;
  }();
  const int shape = (shape_override >= 0 && shape_override <= 2) ? shape_override
                                                                 : big3_pick_shape(M, N, device_cu_count());
  const int bn = big3_tile_n(shape);
  const int tiles_m = (M + T_BM - 1) / T_BM, tiles_n = (N + bn - 1) / bn;
  const int ntiles = tiles_m * tiles_n;
  int grid = std::min(ntiles, device_cu_count());
  if (grid_override == 0) grid = ntiles;
  else if (grid_override > 0) grid = std::min(ntiles, grid_override);
  const int act = (ep.mode == 1 && ep.act > 0) ? ep.act : -1;
#ifdef CT2X_TUNING
  static const int var = [] {
    const char* e = tuning_env("CT2X_BIG3_VAR");
    return e ? std::atoi(e) : 0;
  }();
  if (var == 1) return launch_big3<-1, 1>(a, b, M, N, K, shape, tiles_m, tiles_n, ep, grid, stream);
  if (var == 2) return launch_big3<-1, 2>(a, b, M, N, K, shape, tiles_m, tiles_n, ep, grid, stream);
  if (var == 3) return launch_big3<-1, 3>(a, b, M, N, K, shape, tiles_m, tiles_n, ep, grid, stream);
#endif
  switch (act) {
    case 1: launch_big3<1>(a, b, M, N, K, shape, tiles_m, tiles_n, ep, grid, stream); break;
    case 2: launch_big3<2>(a, b, M, N, K, shape, tiles_m, tiles_n, ep, grid, stream); break;
    case 3: launch_big3<3>(a, b, M, N, K, shape, tiles_m, tiles_n, ep, grid, stream); break;
    case 4: launch_big3<4>(a, b, M, N, K, shape, tiles_m, tiles_n, ep, grid, stream); break;
    case 5: launch_big3<5>(a, b, M, N, K, shape, tiles_m, tiles_n, ep, grid, stream); break;
    case 6: launch_big3<6>(a, b, M, N, K, shape, tiles_m, tiles_n, ep, grid, stream); break;
    default: launch_big3<-1>(a, b, M, N, K, shape, tiles_m, tiles_n, ep, grid, stream); break;
  }
}

}  // namespace ct2x
