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
//   * STREAM-K work split: the unit of work is one K-tile (128 bytes of K) of one output tile, and
//     every workgroup gets the same number of consecutive units (+-1). A launch whose tile count
//     is not a multiple of the CU count (Llama-3-8B prefill FFN-in: 448 tiles on 256 CUs = 1.75
//     rounds, which data-parallel tiles pay as 2) then ends on every CU at the same time, and -
//     as important - the tile boundaries of different CUs fall at DIFFERENT times, so the fp32
//     results no longer leave all 256 CUs in the same few microseconds (a burst the HBM absorbs at
//     ~5 TB/s with every matrix pipe idle). A tile split between two workgroups is finished by
//     whichever of the two arrives last (ticket), so nobody ever waits for a workgroup that is
//     not resident: the first arriver writes its raw int32 accumulators to a 256 KiB slab, waits
//     for the stores (vmcnt = the bytes are in the XCD's L2) and publishes a flag carrying its
//     XCC id; the second adds the slab with L1-bypassing (sc1) loads and runs the epilogue. The
//     split runs INSIDE an XCD (every XCD owns whole tiles, its workgroups share them), so both
//     contributors normally sit behind the same L2 and the slab needs no write-back: a first
//     version with write-through (sc1) slab stores for any placement stalled each workgroup
//     ~13 us on the drain and lost more than the balance gained (profiles/r04_gemm_*). The
//     placement is only observed, never promised, so the second arriver compares the XCC id in
//     the flag with its own (s_getreg HW_REG_XCC_ID) and, if they differ, ignores the slab and
//     computes the missing K-tiles itself (slower, always correct; CT2X_BIG3_SK=3 in a tuning
//     build forces that arm for the tests). int32 addition is exact and commutative: results are
//     bit-identical to the unsplit tile. Counters are zeroed by a memset node in front of every
//     launch (replay-safe).
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
#include <map>
#include <mutex>
#include <utility>

namespace ct2x {

namespace {

constexpr int T_BM = 256, T_BN = 256;
constexpr int T_BK = 128;                                 // K bytes per stage = two MFMA K-steps
constexpr int T_THREADS = 512;
constexpr int T_STAGE_BYTES = (T_BM + T_BN) * T_BK;       // 64 KiB
constexpr int T_RING_BYTES = 2 * T_STAGE_BYTES;           // 128 KiB
constexpr int T_PARAM_FLOATS = 4 * 256;                   // a_scale -> 1/a_scale | w_scale | bias | 1/w_scale
constexpr int T_PARAM_BYTES = T_PARAM_FLOATS * 4;
constexpr int T_FLAG_OFFSET = T_RING_BYTES + 2 * T_PARAM_BYTES;
constexpr int T_LDS_BYTES = T_FLAG_OFFSET + 16;           // 136 KiB + flag of the CU's 160 KiB

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__device__ __forceinline__ void glds16(const int8_t* src, int8_t* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((glb_void*)(src), (lds_void*)(lds_wave_base), 16, 0, 0);
}
__device__ __forceinline__ void glds4(const float* src, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((glb_void*)(src), (lds_void*)(lds_wave_base), 4, 0, 0);
}

// VAR (timing experiments of a tuning build, ACT = -1 only): bit 0 = weight fragment in the outer
// MFMA loop, bit 1 = no epilogue (K loops only).
template <int ACT, int VAR = 0>
__global__ __launch_bounds__(T_THREADS) void gemm_s8_big3_kernel(const int8_t* __restrict__ A,
                                                                 const int8_t* __restrict__ B,
                                                                 int M, int N, int K,
                                                                 int tiles_m, int tiles_n,
                                                                 GemmS8Epilogue ep,
                                                                 int unit_align, int sk_mode,
                                                                 int32_t* sk_slabs, unsigned* sk_counters) {
  extern __shared__ __attribute__((aligned(16))) int8_t lds[];  // the ONLY LDS object

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int group = wm;
  const int r = lane & 15, g = lane >> 4;

  // Work list: hardware block b sits on XCD b % 8 (observed, speed only). XCD x owns the
  // contiguous logical tiles [first, first + count) (band order: an 8-row band walked column by
  // column, shared A rows / B columns in that XCD's L2) = count * ktiles units of one K-tile each;
  // its P workgroups take consecutive, equally long unit ranges, rounded down to a multiple of
  // unit_align: 1 for the stream-K split, ktiles for whole tiles (no slab workspace, or nothing
  // to balance). count >= P (launcher), so a tile has at most two contributors: (x, idx) with its
  // first K-tiles and (x, idx + 1) = hardware block b + 8 with the rest.
  const int ntiles = tiles_m * tiles_n;
  const int ktiles = K / T_BK;  // >= 2 (gemm_s8_big_applicable)
  const int G = gridDim.x;
  const int blk = blockIdx.x;
  int u, u_end;
  {
    const int xcd = blk & 7, idx = blk >> 3;
    const int P = (G >> 3) + (xcd < (G & 7) ? 1 : 0);
    const int q = ntiles >> 3, rem = ntiles & 7;
    const int first = xcd * q + min(xcd, rem);
    const int64_t Ux = static_cast<int64_t>(q + (xcd < rem ? 1 : 0)) * ktiles;
    u = static_cast<int>(idx * Ux / P);
    u_end = static_cast<int>((idx + 1) * Ux / P);
    u = first * ktiles + u - u % unit_align;
    u_end = first * ktiles + u_end - u_end % unit_align;
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
    int row0 = ld_row;  // opaque: the four row numbers are not worth registers across the K loop
    asm volatile("" : "+v"(row0));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a_rel[i] = static_cast<uint32_t>(min(i * 64 + row0, m_last)) * static_cast<uint32_t>(K) + ld_chunk;
      b_rel[i] = static_cast<uint32_t>(min(i * 64 + row0, n_last)) * static_cast<uint32_t>(K) + ld_chunk;
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

  v4i acc[8][4];

  // Fragment addresses: tile row = base16 + r; K chunk c = ks * 4 + g of 8 (16 B each), stored at
  // slot c ^ ((r >> 1) & 7).
  const int swz = r >> 1;
  const int x_off = (wm * 128 + r) * T_BK;
  const int w_off = T_BM * T_BK + (wn * 64 + r) * T_BK;
  v4i xf[8], wf[4];
  auto load_frags = [&](int buf, int ks) {
    const int8_t* tile = lds + buf * T_STAGE_BYTES + (((ks * 4 + g) ^ swz) << 4);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      wf[j] = *reinterpret_cast<const v4i*>(tile + w_off + j * 16 * T_BK);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      xf[i] = *reinterpret_cast<const v4i*>(tile + x_off + i * 16 * T_BK);
  };
  auto mfmas = [&]() {
    __builtin_amdgcn_s_setprio(1);
    if (VAR & 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i)
          acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[j], xf[i], acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[j], xf[i], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
  };
  // Phase boundary: nothing may be scheduled across it by the compiler.
  auto phase_barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  int* slow_div = reinterpret_cast<int*>(lds + T_FLAG_OFFSET);
  if (tid == 0) *slow_div = ep.plain_division ? 1 : 0;

  if (u >= u_end) return;  // (never with the launcher's grid: G <= ntiles)
  int L = u / ktiles;
  int kb = u - L * ktiles;
  int ke = min(ktiles, kb + (u_end - u));
  int u_next = u + (ke - kb);
  bool keep_acc = false, whole = false;  // (set when a split tile is completed by recomputation)
  int tm, tn;
  coords(L, tm, tn);
  set_sources(tm, tn);
  stage(0, kb * T_BK);
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
    // this segment: K-tiles [kb, ke) of tile L; the next one starts at a tile boundary
    const bool has_next = u_next < u_end;
    const int Lnext = L + 1;
    int tm_n = 0, tn_n = 0;
    if (has_next) coords(Lnext, tm_n, tn_n);
    if (!keep_acc) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = v4i{0, 0, 0, 0};
    }
    keep_acc = false;

    phase_barrier();  // this tile's first stage (and its parameters) are in LDS
    if (group == 1) phase_barrier();
    for (int t = kb; t < ke; ++t) {
      const int buf = gt & 1;
      load_frags(buf, 0);
      if (t + 1 < ke) {
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

    // ---- a tile shared with the neighbouring workgroup: the last arriver finishes it ----------
    bool finish = true, merge = false;
    __amdgpu_buffer_rsrc_t slab = __builtin_amdgcn_make_buffer_rsrc(sk_slabs, 0, 0, 0x00020000);
    if (!whole && (kb > 0 || ke < ktiles)) {
      const int idx = kb > 0 ? blk - 8 : blk;  // split id = the block that holds K-tile 0
      unsigned* cnt = sk_counters + 2 * idx;
      slab = __builtin_amdgcn_make_buffer_rsrc(sk_slabs + static_cast<int64_t>(idx) * (T_BM * T_BN), 0,
                                               T_BM * T_BN * 4, 0x00020000);
      unsigned my_xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(my_xcc));
      unsigned* sk_lds = reinterpret_cast<unsigned*>(lds + T_FLAG_OFFSET) + 1;
      if (tid == 0)
        *sk_lds = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      phase_barrier();
      const unsigned ticket = __builtin_amdgcn_readfirstlane(*sk_lds);
      if (ticket == 0) {
        // first arriver: raw accumulators out; every wave waits until its stores are in L2, then
        // ONE lane raises the flag (agent-scope store) with this CU's XCC id in it
        int voff = tid;  // (opaque: not worth a register across the K loop)
        asm volatile("" : "+v"(voff));
        voff *= 16;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_raw_buffer_store_b128(acc[i][j], slab, voff + (i * 4 + j) * (T_THREADS * 16), 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        phase_barrier();
        if (tid == 0)
          __hip_atomic_store(cnt + 1, 0x100u | my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        finish = false;
      } else {
        // second arriver: the slab is complete once the flag is up (it usually has been for a
        // long time; otherwise the first arriver is resident and past its K loop)
        phase_barrier();  // (everyone has read the ticket before the word is reused)
        if (tid == 0) {
          unsigned spins = 0, f = 0;
          while ((f = __hip_atomic_load(cnt + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u &&
                 ++spins < (1u << 26))
            __builtin_amdgcn_s_sleep(8);
          *sk_lds = f;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        phase_barrier();
        const unsigned flag = __builtin_amdgcn_readfirstlane(*sk_lds);
        if ((flag & 0xffu) == my_xcc && flag != 0u && !(sk_mode & 2)) {
          merge = true;  // same L2: the epilogue reads accumulator + slab (L1 bypassed)
        } else {
          // the slab may sit in another XCD's L2: compute the partner's K-tiles here instead
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          phase_barrier();
          if (kb > 0) {
            ke = kb;
            kb = 0;
          } else {
            kb = ke;
            ke = ktiles;
          }
          set_sources(tm, tn);
          stage(gt & 1, kb * T_BK);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          keep_acc = true;
          whole = true;
          continue;
        }
      }
    }
    whole = false;

    // ---- epilogue of tile (tm, tn); both groups are aligned here -------------------------
    if (finish) {
    // C/D layout of the 16x16 MFMA: col j = lane & 15, row i = (lane >> 4) * 4 + reg; with the
    // operands swapped i is the output column and j the output row.
    // (The lane coordinates are re-derived behind an opaque move so that no per-lane address of
    // the epilogue is hoisted over the K loop, where every register is taken.)
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int re = lane_e & 15, ge = lane_e >> 4;
    const int voff = (lane_e + wave * 64) * 16;
    const int64_t m0 = static_cast<int64_t>(tm) * T_BM;
    const int64_t n0 = static_cast<int64_t>(tn) * T_BN;
    // 16-byte stores need the row pitch and the base addresses to allow them.
    const bool full = (m0 + T_BM <= M) && (n0 + T_BN <= N) && ((ep.ldy & 3) == 0) &&
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
          const int64_t row = m0 + wm * 128 + i * 16 + re;
          int32_t* p = c + row * ep.ldy + n0 + cl;
          v4i c4 = acc[i][j];
          if (merge) c4 += __builtin_amdgcn_raw_buffer_load_b128(slab, voff + (i * 4 + j) * (T_THREADS * 16), 0, 16);
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
      const float* ra = par + wm * 128 + re;  // 1 / a_scale of the lane's rows: ra[i * 16]
#define CT2X_BIG3_EPILOGUE(FAST, MERGE)                                                                  \
  _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                 \
    const int cl = wn * 64 + j * 16 + ge * 4;                                                     \
    const v4f ws = *reinterpret_cast<const v4f*>(par + 256 + cl);                                 \
    const v4f wi = *reinterpret_cast<const v4f*>(par + 768 + cl);                                 \
    const v4f bs = has_bias ? *reinterpret_cast<const v4f*>(par + 512 + cl) : v4f{0, 0, 0, 0};    \
    v4i part[4];                                                                                  \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                               \
      if (MERGE && (i & 3) == 0) {                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        _Pragma("unroll") for (int ii = 0; ii < 4; ++ii) part[ii] =                               \
            __builtin_amdgcn_raw_buffer_load_b128(slab, voff + ((i + ii) * 4 + j) * (T_THREADS * 16), 0, 16); \
      }                                                                                           \
      v4i c4 = acc[i][j];                                                                         \
      if (MERGE) c4 += part[i & 3];                                                               \
      const int64_t row = m0 + wm * 128 + i * 16 + re;                                            \
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
    }                                                                                             \
  }
      if (merge) {  // (rare: one tile per workgroup at most; plain division keeps the code small)
        CT2X_BIG3_EPILOGUE(false, true)
      } else if (fast_div) {
        CT2X_BIG3_EPILOGUE(true, false)
      } else {
        CT2X_BIG3_EPILOGUE(false, false)
      }
#undef CT2X_BIG3_EPILOGUE
    }

    }  // finish

    if (!has_next) break;
    u = u_next;
    L = Lnext;
    kb = 0;
    ke = min(ktiles, u_end - u);
    u_next = u + ke;
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

// Stream-K workspace of a stream: one 256 KiB slab of raw accumulators and two counters (ticket,
// ready flag) per workgroup slot. Allocated on first use, kept for the life of the process (a
// launch on a stream runs behind the previous one, so one set per stream is enough).
struct SkWorkspace {
  int32_t* slabs = nullptr;
  unsigned* counters = nullptr;
  int slots = 0;
};
SkWorkspace* sk_workspace(hipStream_t stream, int slots) {
  static std::mutex mu;
  static std::map<std::pair<int, hipStream_t>, SkWorkspace> table;
  int dev = 0;
  CT2X_HIP_CHECK(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  SkWorkspace& w = table[{dev, stream}];
  if (w.slots >= slots) return &w;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
    (void)hipGetLastError();
    return nullptr;  // no allocation inside a capture: this launch runs whole tiles
  }
  int32_t* slabs = nullptr;
  unsigned* counters = nullptr;
  if (hipMalloc(&slabs, static_cast<size_t>(slots) * T_BM * T_BN * 4) != hipSuccess ||
      hipMalloc(&counters, static_cast<size_t>(slots) * 2 * sizeof(unsigned)) != hipSuccess) {
    (void)hipGetLastError();
    if (slabs) (void)hipFree(slabs);
    return nullptr;
  }
  // (a smaller earlier set stays allocated: a captured graph may still hold its addresses)
  w.slabs = slabs;
  w.counters = counters;
  w.slots = slots;
  return &w;
}

template <int ACT, int VAR = 0>
void launch_big3(const int8_t* a, const int8_t* b, int M, int N, int K, int tiles_m, int tiles_n,
                 const GemmS8Epilogue& ep, int grid, int sk_mode, hipStream_t stream) {
  // > 64 KiB of dynamic LDS has to be opted into once per device.
  static std::atomic<bool> configured[64];
  int dev = 0;
  CT2X_HIP_CHECK(hipGetDevice(&dev));
  if (!configured[dev & 63].load(std::memory_order_acquire)) {
    CT2X_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_s8_big3_kernel<ACT, VAR>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS_BYTES));
    configured[dev & 63].store(true, std::memory_order_release);
  }
  int unit_align = K / T_BK;  // whole tiles
  int32_t* slabs = nullptr;
  unsigned* counters = nullptr;
  if (sk_mode) {
    if (SkWorkspace* w = sk_workspace(stream, grid)) {
      slabs = w->slabs;
      counters = w->counters;
      unit_align = 1;
      CT2X_HIP_CHECK(hipMemsetAsync(counters, 0, static_cast<size_t>(grid) * 2 * sizeof(unsigned), stream));
    }
  }
  hipLaunchKernelGGL((gemm_s8_big3_kernel<ACT, VAR>), dim3(grid), dim3(T_THREADS), T_LDS_BYTES, stream,
                     a, b, M, N, K, tiles_m, tiles_n, ep, unit_align, sk_mode, slabs, counters);
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
  // Stream-K only where there is something to balance: more than one tile per workgroup and a
  // ragged last round (CT2X_BIG3_SK=0 / 1 in a tuning build: never / whenever grid < tiles).
  static const int sk_override = [] {
    const char* e = tuning_env("CT2X_BIG3_SK");
    return e ? std::atoi(e) : -1;
  }();
  int stream_k = (grid < ntiles && ntiles % grid != 0) ? 1 : 0;  // kernel's sk_mode
  if (sk_override == 0) stream_k = 0;
  if (sk_override == 1 && grid < ntiles) stream_k = 1;
  if (sk_override == 3 && grid < ntiles) stream_k = 3;  // + treat every slab as foreign (tests)
  if (ntiles / 8 < (grid + 7) / 8) stream_k = 0;  // an XCD with fewer tiles than workgroups
  const int act = (ep.mode == 1 && ep.act > 0) ? ep.act : -1;
#ifdef CT2X_TUNING
  static const int var = [] {
    const char* e = tuning_env("CT2X_BIG3_VAR");
    return e ? std::atoi(e) : 0;
  }();
  if (var == 1) return launch_big3<-1, 1>(a, b, M, N, K, tiles_m, tiles_n, ep, grid, stream_k, stream);
  if (var == 2) return launch_big3<-1, 2>(a, b, M, N, K, tiles_m, tiles_n, ep, grid, stream_k, stream);
  if (var == 3) return launch_big3<-1, 3>(a, b, M, N, K, tiles_m, tiles_n, ep, grid, stream_k, stream);
#endif
  switch (act) {
    case 1: launch_big3<1>(a, b, M, N, K, tiles_m, tiles_n, ep, grid, stream_k, stream); break;
    case 2: launch_big3<2>(a, b, M, N, K, tiles_m, tiles_n, ep, grid, stream_k, stream); break;
    case 3: launch_big3<3>(a, b, M, N, K, tiles_m, tiles_n, ep, grid, stream_k, stream); break;
    case 4: launch_big3<4>(a, b, M, N, K, tiles_m, tiles_n, ep, grid, stream_k, stream); break;
    case 5: launch_big3<5>(a, b, M, N, K, tiles_m, tiles_n, ep, grid, stream_k, stream); break;
    case 6: launch_big3<6>(a, b, M, N, K, tiles_m, tiles_n, ep, grid, stream_k, stream); break;
    default: launch_big3<-1>(a, b, M, N, K, tiles_m, tiles_n, ep, grid, stream_k, stream); break;
  }
}

}  // namespace ct2x
