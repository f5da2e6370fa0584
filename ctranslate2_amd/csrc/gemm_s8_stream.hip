// Persistent weight-streaming form of the decode-step int8 Dense for models whose weights do not
// stay cache resident (the Llama-3-8B shape of BASELINE configs[2]: 7.5 GB of int8 weights per
// step, M = 16 rows). Same contract and the same bits as gemm_s8_skinny_kernel (gemm_s8.hip;
// replaces src/cuda/primitives.cu:571-597 + src/ops/dequantize_gpu.cu:31-55 + the residual Add of
// src/layers/common.cc:392-401): integer accumulation is exact and order independent, the epilogue
// is the same expression.
//
// Why a second form. The skinny kernel is one-shot: a workgroup requests its whole slab of the
// weight, waits, multiplies, reduces, stores and exits - with 200 - 2000 workgroups per launch the
// HBM stream is a sequence of bursts separated by drain / dispatch / ramp bubbles (measured on the
// Llama shape: 3.1 - 4.3 TB/s per launch, 3.35 TB/s over a layer, `profiles/r02_l8_kernel_trace_gaps.txt`).
// Here ONE workgroup per CU stays resident and walks over its slabs:
//   * the activation fragments (16 rows x this wave's K-steps) are loaded once and stay in registers;
//   * the weight fragments run through a ring of P register slots per wave (P K-steps ahead, the
//     ring continues ACROSS slab boundaries: while a slab is reduced and stored, the first K-steps
//     of the next slab are already in flight) - the request stream never drains inside a launch;
//   * the cross-wave reduction buffer in LDS is double buffered: one barrier per slab;
//   * slabs per workgroup are equal for every workgroup of the launch (grid = slabs / slabs_per_wg),
//     so no CU finishes a slab early or late.
// Fragment addressing is the skinny kernel's: v_mfma_i32_16x16x64_i8 takes 16 consecutive K bytes
// per lane for A (row = lane & 15) and B (column = lane & 15), K-block = lane >> 4; wave w owns the
// 64-byte K-steps w, w + WAVES, ...
//
// Round 4: the row-wide launches in front of these GEMMs (RMSNorm + Quantize: 10 us at [16, 4096];
// the element-wise Quantize behind the attention: 5 us - both with HBM idle) run as PROLOGUES of the
// weight stream (AMODE below): the first ring of weight requests leaves before the prologue starts,
// so the norm's dependent chain hides under the stream's own latency instead of standing in front
// of it. Every workgroup redoes the prologue for the 16 rows (256 KB of fp32 from L2).
//   AMODE 1 (quantise on load, ops::Quantize given the row absmax the producer left): every lane
//     converts the 16 floats of each of its A fragments itself (to_s8, src/cpu/kernels.cc:577-651).
//   AMODE 2 (RMSNorm + Quantize, K = 4096): the reference's statistics are eight lane accumulators
//     walking the row with stride 8 (row_norm.h) - 512 DEPENDENT steps per row that no layout can
//     shorten. The threads park the floats of their own fragments transposed in LDS, one 1024-column
//     chunk (= one K-step of all 16 waves) at a time through two buffers, and waves 0-3 (four rows
//     each, 16 lanes per row) run the chains one chunk behind the writers; then every thread
//     normalises, max-combines and quantises the 64 values it kept in registers.
//
// Round 6: LEADER forms. What made the prologues above cost as much as the launches they replaced
// was that every one of the ~200 workgroups redid the row-wide work on 256 KB of fp32 rows. Here
// the workgroups 0..15 - the first the dispatcher places - each own ONE row: they ask for their row
// first and for their weight ring second, run the row-wide work once (AMODE 3: RMSNorm + Quantize,
// the same functions as norm_vec_kernel, rowops.hip; AMODE 4: ops::Quantize given the producer's
// row absmax), store the int8 row and its scale write-through, drain, and publish a tagged flag word
// per row. Every workgroup (the leaders too) has its first weight ring in flight meanwhile - 128 KB
// per workgroup, 28 - 32 MB per launch: the whole weight of the QKV / out projections - and then
// waits for the sixteen tags, reads its int8 fragments (64 KB per workgroup from L2 as in AMODE 0;
// no workgroup of this launch has touched those lines before the flags, so no L2 holds a copy older
// than the write-through) and streams on. The tag is (device step + 1) * 1024 + the launch's index
// in the step: no counter to reset, nothing to clear, a stale flag never matches. Leaders never
// wait for anything, so the launch cannot deadlock whatever else holds CUs; the poll is bounded and
// traps when it runs out (a loud HIP error, never silent garbage).

#include "device_utils.h"
#include "grid_sync.h"
#include "kernels.h"
#include "row_norm.h"
#include "runtime.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>

namespace ct2x {

namespace {

// Geometry of the AMODE 2 staging (floats): a chain of a 1024-column chunk is 128 elements; the
// pitch 132 puts the eight chains' 16-byte reads on eight different bank quads, the row stride
// 8 * 132 + 4 spreads the rows' 8-byte writes over the banks (two-way conflicts at worst).
constexpr int kNormChainPitch = 132;
constexpr int kNormRowStride = 8 * kNormChainPitch + 4;
constexpr int kNormChunkFloats = 16 * kNormRowStride;

template <int NT, int WAVES, int KS, bool DUAL, int AMODE>
struct StreamLds {
  static constexpr int ND = DUAL ? 2 : 1;
  static constexpr int kPartialInts = 2 * ND * WAVES * NT * 256;
  // region 0: the reduction tiles; during the prologues the two chunk buffers (AMODE 2) and the
  // int8 activation tile
  static constexpr int kTileInts = 16 * (4096 + 16) / 4;  // the int8 activation tile (AMODE 1 / 2)
  // (leader forms: the transposed row of the statistics chain, norm_chain_pad(4096) = 516, + scratch)
  static constexpr int kLeadInts = 8 * 516 + 8 + 32 + 4;
  static constexpr int kProInts =
      AMODE == 2 ? 2 * kNormChunkFloats : (AMODE == 1 ? kTileInts : (AMODE >= 3 ? kLeadInts : 0));
  static constexpr int kRegion0 = kProInts > kPartialInts ? kProInts : kPartialInts;
  static constexpr int kTileAmax = kRegion0;            // [2][16] uint
  static constexpr int kRowScale = kTileAmax + 32;      // [16] float (AMODE != 0)
  static constexpr int kRowAmax = kRowScale + 16;       // [16] uint  (AMODE 2)
  static constexpr int kRowMul = kRowAmax + 16;         // [16] float (AMODE 2)
  static constexpr int kGamma = kRowMul + 16;           // [K] float  (AMODE 2)
  static constexpr size_t bytes(int k) {
    return sizeof(int) * static_cast<size_t>((AMODE == 0 || AMODE >= 3) ? kRowScale
                                                                         : (AMODE == 1 ? kRowAmax : kGamma + k));
  }
};

// Four floats -> four int8 (to_s8, device_utils.h) packed into one word. The x86 default (u8 shift
// + round to nearest even) is three VALU operations per element, the four low bytes gathered by two
// v_perm and "- 128" on every byte as one xor (as in the skinny kernel's prologues).
__device__ __forceinline__ int quantize_word_fast(const v4f v, float scale) {
  int b[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) b[e] = static_cast<int>(rintf(fmaf(v[e], scale, 128.f)));
  const unsigned lo = __builtin_amdgcn_perm(static_cast<unsigned>(b[1]), static_cast<unsigned>(b[0]),
                                            0x0c0c0400u);
  const unsigned hi = __builtin_amdgcn_perm(static_cast<unsigned>(b[3]), static_cast<unsigned>(b[2]),
                                            0x04000c0cu);
  return static_cast<int>((lo | hi) ^ 0x80808080u);
}
__device__ __forceinline__ int quantize_word_any(const v4f v, float scale, int quant_mode) {
  int word = 0;
#pragma unroll
  for (int e = 0; e < 4; ++e) word |= (static_cast<int>(to_s8(v[e], scale, quant_mode)) & 0xff) << (8 * e);
  return word;
}

template <int NT, int WAVES, int KS, int P, bool DUAL, bool PACKED, int AMODE = 0>
__global__ __launch_bounds__(WAVES * 64) void gemm_s8_stream_kernel(
    const int8_t* __restrict__ A, const int8_t* __restrict__ B, int M, int N, int K,
    int slabs_per_wg, int total_slabs, const int8_t* __restrict__ B2, const float* __restrict__ aux,
    int aux_n, GemmS8Epilogue ep) {
  // (A .. aux_n are preloaded into SGPRs; `ep` costs a kernarg round trip and is first touched after
  // the activation fragments and the first ring of weight requests have left. PACKED - the weight
  // is the fragment-major copy - is a template parameter: with both address forms live the long-K
  // ring spilled 15 registers. AMODE 1 / 2: A is the fp32 activation [M, K]; aux = the producer's
  // row absmax [M, aux_n] (AMODE 1) or the norm's gamma [K] (AMODE 2).)
  static_assert(KS % P == 0, "the ring slot of K-step i is i % P in every slab");
    constexpr int ND = DUAL ? 2 : 1;
  constexpr int THREADS = WAVES * 64;
  constexpr int OUTS = NT * 256;
  constexpr int OPT = (OUTS + THREADS - 1) / THREADS;
  // dynamic LDS (up to 152 KB: opted into by the launcher): the double-buffered reduction tiles
  // [2][ND][WAVES][NT][64][4] int32, the row-absmax words [2][16], then the prologue's words
  typedef StreamLds<NT, WAVES, KS, DUAL, AMODE> Lds;
  extern __shared__ __attribute__((aligned(16))) int stream_lds[];
  typedef int PartialT[ND][WAVES][NT][64][4];
  PartialT* partial = reinterpret_cast<PartialT*>(stream_lds);
  unsigned int (*tile_amax)[16] = reinterpret_cast<unsigned int (*)[16]>(stream_lds + Lds::kTileAmax);
  float* row_scale_lds = reinterpret_cast<float*>(stream_lds + Lds::kRowScale);

  const int tid = threadIdx.x;
  // (packed weights: the wave index as a scalar keeps the wave-uniform address parts in SGPRs; the
  // row-major forms allocate worse with it)
  const int lane = tid & 63, wave = PACKED ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6);
  const int r = lane & 15, g = lane >> 4;

  // activation fragments of this wave's K-steps: resident for the whole launch
  v4i af[KS];
  if (AMODE == 0) {
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      const int kk = (wave + i * WAVES) * 64 + g * 16;
      af[i] = (r < M) ? *reinterpret_cast<const v4i*>(A + static_cast<int64_t>(r) * K + kk)
                      : v4i{0, 0, 0, 0};
    }
  }
  // AMODE 1 / 2: the fp32 activation arrives COALESCED - thread t holds float4 (t & 255) of every
  // 1024-column chunk c of the rows 4 jj + (t >> 8): a wave's request is 1 KB of one row (the
  // per-lane fetch of the MFMA fragment layout - 16 rows x 64-byte pieces per request - kept the
  // CU's vector memory pipe busy for ~4 us, in front of the weight requests). The quantised rows go
  // through an int8 tile in LDS (row stride K + 16 as in the skinny kernel's prologues) from which
  // every lane reads its fragments. Rows past M repeat row 0: unconditional loads; what the MFMA
  // computes for those rows is never stored.
  constexpr bool kPro = AMODE == 1 || AMODE == 2;  // every workgroup redoes the row-wide work
  constexpr bool kLead = AMODE >= 3;               // workgroups 0..15 do it once (round 6)
  static_assert(!kPro || (KS == 4 && WAVES == 16), "prologue forms: K = 4096 on 16 waves");
  static_assert(AMODE != 3 || (KS == 4 && WAVES == 16), "leader norm: RMSNorm rows of 4096 on 16 waves");
  static_assert(!kLead || WAVES == 16, "leader forms: 1024 threads per row");
  constexpr int XP = kPro ? 16 : 1;  // float4 pieces per thread: piece 4 c + jj
  v4f xa[XP];
  const int xq = tid >> 8, xc4 = tid & 255;
  float q_amax[4] = {0.f, 0.f, 0.f, 0.f};
  v4f gpre = v4f{0.f, 0.f, 0.f, 0.f};
  if (kPro) {
    const float* X = reinterpret_cast<const float*>(A);
    const float* xbase[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int row = 4 * jj + xq;
      const int rowc = row < M ? row : 0;
      xbase[jj] = X + static_cast<int64_t>(rowc) * K + 4 * xc4;
      if (AMODE == 1)
        for (int p = 0; p < aux_n; ++p) q_amax[jj] = fmaxf(q_amax[jj], aux[rowc * aux_n + p]);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) xa[4 * c + jj] = *reinterpret_cast<const v4f*>(xbase[jj] + 1024 * c);
    if (AMODE == 2) gpre = reinterpret_cast<const v4f*>(aux)[tid];  // K / 4 == THREADS float4s
    __builtin_amdgcn_sched_barrier(0);
  }

  // weight ring: slot i % P holds K-step i of the current slab (all NT column tiles, both weights)
  v4i bf[P][ND][NT];
  auto request = [&](int slab, int i) {  // i: compile-time after unrolling
    const int64_t n0 = static_cast<int64_t>(slab) * (NT * 16);
    const int kk = (wave + i * WAVES) * 64 + g * 16;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int64_t off =
          PACKED ? ((((static_cast<int64_t>(slab) * WAVES + wave) * KS + i) * NT + j) << 10) + lane * 16
                      : (n0 + j * 16 + r) * K + kk;
      bf[i % P][0][j] = *reinterpret_cast<const v4i*>(B + off);
      if (DUAL) bf[i % P][ND - 1][j] = *reinterpret_cast<const v4i*>(B2 + off);
    }
  };

  const int first = blockIdx.x * slabs_per_wg;
  const int mine = min(slabs_per_wg, total_slabs - first);  // only the last workgroup has fewer
  // ---- leader forms: the row first, the ring second --------------------------------------------
  constexpr int LP = (KS + 3) / 4;  // float4 pieces of a row per thread (K / 4 over 1024 threads)
  v4f lx[kLead ? LP : 1];
  const bool leader = kLead && blockIdx.x < 16;
  const bool lead_work = leader && static_cast<int>(blockIdx.x) < M;
  if (kLead) {
    if (lead_work) {
      const v4f* xr = reinterpret_cast<const v4f*>(reinterpret_cast<const float*>(A) +
                                                   static_cast<int64_t>(blockIdx.x) * K);
#pragma unroll
      for (int p = 0; p < LP; ++p) {
        const int idx = tid + p * THREADS;
        lx[p] = (idx < KS * 256) ? xr[idx] : v4f{0.f, 0.f, 0.f, 0.f};
      }
      if (AMODE == 3) gpre = reinterpret_cast<const v4f*>(aux)[tid];  // gamma: K / 4 == THREADS
    }
    __builtin_amdgcn_sched_barrier(0);
    // The sixteen rows travel ahead of the launch's 28 - 32 MB of ring requests: a leader asks for its
    // ring once its row has landed, everyone else after a head start (units of 512 clocks) - the
    // rings still land long before the flags do, and a row request queued in the middle of that
    // burst arrives microseconds late, on the launch's critical path.
    if (lead_work) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (!leader) {
      for (int h = 0; h < ep.lead_headstart; ++h) __builtin_amdgcn_s_sleep(8);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (AMODE != 2) {
#pragma unroll
    for (int i = 0; i < P; ++i) request(first, i);
  }
  if (kLead) {
    __builtin_amdgcn_sched_barrier(0);
    const unsigned tag =
        (ep.lead_step != nullptr ? static_cast<unsigned>(*ep.lead_step + 1) * 1024u : 0u) + ep.lead_tag;
    if (leader) {
      const int lrow = blockIdx.x;
      if (lead_work) {
        float* lead_lds = reinterpret_cast<float*>(stream_lds);
        constexpr int kPad = 516;  // norm_chain_pad(4096)
        float* scratch = lead_lds + 8 * kPad + 8;  // [32] block_max, then [1] the row's multiplier
        float amax = 0.f;
        if (AMODE == 3) {
          // RMSNorm exactly as norm_vec_kernel<4> (rowops.hip): transposed chains, wave 0 walks them
          const int c0 = 4 * tid;
#pragma unroll
          for (int e = 0; e < 4; ++e) lead_lds[norm_chain_slot(c0 + e, K, kPad)] = lx[0][e];
          __syncthreads();
          if (wave == 0) {
            float sum, sumsq;
            ordered_row_stats_t<false, 4>(lead_lds, kPad, K, sum, sumsq);  // (128 registers: the ring stays)
            const NormStats s0 = norm_stats(sum, sumsq, K, ep.norm_eps, false);
            if (lane == 0) scratch[32] = s0.mul;
          }
          __syncthreads();
          NormStats st;
          st.mean = 0.f;
          st.mul = scratch[32];
          st.four_begin = st.four_end = K;  // K % 8 == 0: no 4-wide remainder columns
          const bool resid = ep.norm_rms_residual;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            lx[0][e] = norm_apply(lx[0][e], gpre[e], 0.f, st, false, resid, c0 + e);
            amax = fmaxf(amax, fabsf(lx[0][e]));
          }
          amax = block_max(amax, scratch);
        } else {
          // ops::Quantize given the producer's row absmax (aux: [M, aux_n] partial maxima)
          for (int p = 0; p < aux_n; ++p) amax = fmaxf(amax, aux[lrow * aux_n + p]);
        }
        const float scale = (amax != 0.f) ? 127.f / amax : 1.f;
        const bool fast = (ep.quant_mode & kQuantU8Shift) && (ep.quant_mode & kQuantRound);
        unsigned* qrow = reinterpret_cast<unsigned*>(ep.lead_q + static_cast<int64_t>(lrow) * K);
#pragma unroll
        for (int p = 0; p < LP; ++p) {
          const int idx = tid + p * THREADS;
          if (idx < KS * 256) {
            const int word = fast ? quantize_word_fast(lx[p], scale) : quantize_word_any(lx[p], scale, ep.quant_mode);
            grid_publish(qrow + idx, static_cast<unsigned>(word));
          }
        }
        if (tid == 0) grid_publish(reinterpret_cast<unsigned*>(ep.lead_scale) + lrow, __float_as_uint(scale));
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the row has left (and this ring has landed)
      __syncthreads();
      if (tid == 0) grid_publish(ep.lead_flag + lrow, tag);
    }
    // everyone: the sixteen tags, then this wave's int8 fragments
    if (wave == 0) {
      unsigned spins = 0;
      for (;;) {
        const unsigned f = (lane < 16) ? grid_observe(ep.lead_flag + lane) : tag;
        if (__all(f == tag)) break;
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1u << 24)) __builtin_trap();
      }
    }
    __syncthreads();
    const int8_t* Aq = ep.lead_q;
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      const int kk = (wave + i * WAVES) * 64 + g * 16;
      af[i] = (r < M) ? *reinterpret_cast<const v4i*>(Aq + static_cast<int64_t>(r) * K + kk)
                      : v4i{0, 0, 0, 0};
    }
  }

  int8_t* qtile = reinterpret_cast<int8_t*>(stream_lds);  // [16][K + 16] int8 (AMODE 1 / 2)
  const int qstride = K + 16;
  float q_scale[4] = {1.f, 1.f, 1.f, 1.f};
  if (AMODE == 1) {
    // ops::Quantize with the producer's absmax: scale = 127 / amax (1 for a zero row)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) q_scale[jj] = (q_amax[jj] != 0.f) ? 127.f / q_amax[jj] : 1.f;
  }
  if (AMODE == 2) {
    __builtin_amdgcn_sched_barrier(0);
    float* chunk_lds = reinterpret_cast<float*>(stream_lds);
    unsigned int* row_amax_lds = reinterpret_cast<unsigned int*>(stream_lds + Lds::kRowAmax);
    float* row_mul_lds = reinterpret_cast<float*>(stream_lds + Lds::kRowMul);
    float* gamma_lds = reinterpret_cast<float*>(stream_lds + Lds::kGamma);
    reinterpret_cast<v4f*>(gamma_lds)[tid] = gpre;
    if (tid < 16) row_amax_lds[tid] = 0u;
    // chunk-local column 4 xc4 + e sits in chain (4 xc4 + e) & 7 = 4 (xc4 & 1) + e at slot xc4 >> 1
    auto park = [&](int c) {
      float* dst = chunk_lds + (c & 1) * kNormChunkFloats + xq * kNormRowStride +
                   4 * (xc4 & 1) * kNormChainPitch + (xc4 >> 1);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          dst[jj * 4 * kNormRowStride + e * kNormChainPitch] = xa[4 * c + jj][e];
    };
    // Wave roles: waves 0-3 run the chains (four rows each, 16 lanes per row, lanes l and l + 8
    // both chain l) and ask for their share of the first weight ring only afterwards - the ring's
    // 32 registers are what lets eight 16-byte chain reads stay in flight ahead of the dependent
    // FMAs (two in flight: ~23 cycles per chain step instead of ~13). Waves 4-15 ask for their
    // weights right behind their fp32 loads and then only park their columns. Both arms execute
    // the same barriers.
    if (wave < 4) {
      park(0);
      __syncthreads();
      const int crow = wave * 4 + g;
      float ss = 0.f;
#pragma unroll
      for (int c = 0; c < KS; ++c) {
        if (c + 1 < KS) park(c + 1);
        const v4f* chain = reinterpret_cast<const v4f*>(chunk_lds + (c & 1) * kNormChunkFloats +
                                                        crow * kNormRowStride + (lane & 7) * kNormChainPitch);
#pragma unroll 8
        for (int q = 0; q < 32; ++q) {
          const v4f v = chain[q];
          ss = fmaf(v.x, v.x, ss);
          ss = fmaf(v.y, v.y, ss);
          ss = fmaf(v.z, v.z, ss);
          ss = fmaf(v.w, v.w, ss);
        }
        if (c + 1 == KS) {
          ss = dpp_xor4(ss) + ss;
          ss = dpp_move<0x4E, 0xf>(0.f, ss) + ss;
          ss = dpp_move<0xB1, 0xf>(0.f, ss) + ss;
          const NormStats st = norm_stats(0.f, ss, K, ep.norm_eps, false);
          if (r == 0) row_mul_lds[crow] = st.mul;
        }
        __syncthreads();
      }
#pragma unroll
      for (int i = 0; i < P; ++i) request(first, i);
    } else {
#pragma unroll
      for (int i = 0; i < P; ++i) request(first, i);
      park(0);
      __syncthreads();
#pragma unroll
      for (int c = 0; c < KS; ++c) {
        if (c + 1 < KS) park(c + 1);
        __syncthreads();
      }
    }
    // normalise (src/cpu/kernels.cc:543-573 in its compiled order: norm_apply, no 4-wide remainder
    // columns at K % 8 == 0): plain (x * gamma) * mul; use_residual (Gemma) (mul * x) * (gamma + 1) -
    // both are (x * p) * q, the uniform choice is two selects per element, not a branch per element
    const bool resid = ep.norm_rms_residual;
    const v4f* gamma4 = reinterpret_cast<const v4f*>(gamma_lds) + xc4;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const float mul = row_mul_lds[4 * jj + xq];
      float amax = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const v4f gv = gamma4[256 * c];
        v4f y;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float p = resid ? mul : gv[e];
          const float q = resid ? gv[e] + 1.f : mul;
          y[e] = (xa[4 * c + jj][e] * p) * q;
          amax = fmaxf(amax, fabsf(y[e]));
        }
        xa[4 * c + jj] = y;
      }
      // row absmax: the 64 lanes of a wave share their rows (|y| >= 0: bits order like uints)
      const float wm = wave_max(amax);
      if (lane == 0) atomicMax(&row_amax_lds[4 * jj + xq], __float_as_uint(wm));
    }
    __syncthreads();  // (also: every chain read of the chunk buffers is done - the tile may land)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const float ra = __uint_as_float(row_amax_lds[4 * jj + xq]);
      q_scale[jj] = (ra != 0.f) ? 127.f / ra : 1.f;
    }
  }
  if (kPro) {
    // Quantize (to_s8) into the LDS tile, then every lane's MFMA fragments out of it. The choice
    // between the x86 default (u8 shift + round to nearest even: three VALU operations per element,
    // bytes gathered by v_perm, "- 128" as one xor) and the general form is ONE uniform branch.
    const bool fast = (ep.quant_mode & kQuantU8Shift) && (ep.quant_mode & kQuantRound);
    if (fast) {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        int* qrow = reinterpret_cast<int*>(qtile + (4 * jj + xq) * qstride) + xc4;
#pragma unroll
        for (int c = 0; c < 4; ++c) qrow[256 * c] = quantize_word_fast(xa[4 * c + jj], q_scale[jj]);
      }
    } else {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        int* qrow = reinterpret_cast<int*>(qtile + (4 * jj + xq) * qstride) + xc4;
#pragma unroll
        for (int c = 0; c < 4; ++c) qrow[256 * c] = quantize_word_any(xa[4 * c + jj], q_scale[jj], ep.quant_mode);
      }
    }
    if (xc4 == 0) {  // the epilogue's 1 / a_scale
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) row_scale_lds[4 * jj + xq] = q_scale[jj];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < KS; ++i)
      af[i] = *reinterpret_cast<const v4i*>(qtile + r * qstride + (wave + i * WAVES) * 64 + g * 16);
    __syncthreads();  // the tile is dead before a reduction tile of slab 0 is written over it
  }

  for (int t = 0; t < mine; ++t) {
    const int slab = first + t;
    const int64_t n0 = static_cast<int64_t>(slab) * (NT * 16);
    const bool more = t + 1 < mine;

    // epilogue operands of this slab, requested before the K loop (thread -> output as in the
    // skinny kernel: 16 consecutive threads = 16 consecutive columns)
    float e_ra[OPT], e_ws[OPT], e_bias[OPT], e_res[OPT], e_ws2[OPT], e_bias2[OPT];
#pragma unroll
    for (int o = 0; o < OPT; ++o) {
      const int flat = tid + o * THREADS;
      const int tile = flat >> 8, within = flat & 255;
      const int row = within >> 4;
      const int64_t col = n0 + tile * 16 + (within & 15);
      e_ra[o] = e_ws[o] = e_ws2[o] = 1.f;
      e_bias[o] = e_res[o] = e_bias2[o] = 0.f;
      if (ep.mode == 1 && flat < OUTS && row < M) {
        if (AMODE == 0) e_ra[o] = ep.a_scale[row];
        if (kLead) e_ra[o] = __uint_as_float(grid_observe(reinterpret_cast<const unsigned*>(ep.lead_scale) + row));
        e_ws[o] = ep.w_scale[col];
        if (ep.bias) e_bias[o] = ep.bias[col];
        if (ep.residual) e_res[o] = ep.residual[static_cast<int64_t>(row) * ep.ldy + col];
        if (DUAL) {
          e_ws2[o] = ep.w_scale2[col];
          if (ep.bias2) e_bias2[o] = ep.bias2[col];
        }
      }
    }

    v4i acc[ND][NT];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[d][j] = v4i{0, 0, 0, 0};

#pragma unroll
    for (int i = 0; i < KS; ++i) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int d = 0; d < ND; ++d)
          acc[d][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[i], bf[i % P][d][j], acc[d][j], 0, 0, 0);
      // the slot is free again: K-step i + P of this slab, or K-step i + P - KS of the next one
      if (i + P < KS) request(slab, i + P);
      else if (more) request(slab + 1, i + P - KS);
    }

    const int pb = t & 1;
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        *reinterpret_cast<v4i*>(&partial[pb][d][wave][j][lane][0]) = acc[d][j];
    if (ep.row_amax_out != nullptr && tid < 16) tile_amax[pb][tid] = 0u;
    __syncthreads();  // the only barrier of a slab: buffer pb ^ 1 was read before the previous one

#pragma unroll
    for (int o = 0; o < OPT; ++o) {
      const int flat = tid + o * THREADS;
      if (flat >= OUTS) continue;
      const int j = flat >> 8, within = flat & 255;
      const int rt = within >> 4, ct = within & 15;
      const int64_t col = n0 + j * 16 + ct;
      const int src_lane = ((rt >> 2) << 4) | ct;  // MFMA C layout (see gemm_s8.hip)
      int sum = 0, sum2 = 0;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) {
        sum += partial[pb][0][w][j][src_lane][rt & 3];
        if (DUAL) sum2 += partial[pb][ND - 1][w][j][src_lane][rt & 3];
      }
      if (rt >= M) continue;
      const int64_t idx = static_cast<int64_t>(rt) * ep.ldy + col;
      if (ep.mode == 0) {
        ep.c_i32[idx] = sum;
      } else {
        // CPU formula order (src/cpu/kernels.cc:666-673): c * (1/a_scale) / b_scale + bias.
        const float r_a = 1.f / ((AMODE == 0 || kLead) ? e_ra[o] : row_scale_lds[rt]);
        float v = static_cast<float>(sum) * r_a;
        v = v / e_ws[o];
        if (ep.bias) v = v + e_bias[o];
        v = apply_activation(v, ep.act);
        if (DUAL) {
          float v2 = static_cast<float>(sum2) * r_a;
          v2 = v2 / e_ws2[o];
          if (ep.bias2) v2 = v2 + e_bias2[o];
          v = v2 * v;  // ops::Mul()(linear_0_noact_out, activated, out)
        }
        if (ep.residual) v = e_res[o] + v;
        ep.y[idx] = v;
        if (ep.row_amax_out != nullptr) atomicMax(&tile_amax[pb][rt], __float_as_uint(fabsf(v)));
      }
    }
    if (ep.row_amax_out != nullptr) {
      // |y| >= 0: the IEEE bit patterns order like unsigned integers (exact, order independent).
      // Safe without a second barrier: tile_amax[pb] is next cleared two slabs from now, after the
      // barrier of slab t + 1 - which every thread reaches only after this read.
      __syncthreads();
      if (tid < 16 && tid < M)
        atomicMax(reinterpret_cast<unsigned int*>(ep.row_amax_out) + tid, tile_amax[pb][tid]);
    }
  }
}

struct StreamPlan {
  int nt = 0, ks = 0, p = 0;
  int slabs_per_wg = 0, grid = 0, slabs = 0;
};

// Equal slabs per workgroup (but the last) and at least 3/4 of the CUs busy, as many bytes in
// flight per wave as the register file allows (P * NT * ND KB per wave, 16 waves per CU).
bool plan_stream(int64_t n, int64_t k, bool dual, int cus, StreamPlan* plan) {
  if (k % 1024 != 0) return false;  // 16 waves x 64-byte K-steps
  const int ks = static_cast<int>(k / 1024);
  const int nts[3] = {4, 2, 1};
  for (int nt : nts) {
    if (dual && nt > 2) continue;
    if (ks == 14 && (nt > 1 || dual)) continue;  // register ring of the long-K forms: one tile
    if (ks == 8 && nt > 2) continue;
    if (n % (16 * nt) != 0) continue;
    const int64_t slabs = n / (16 * nt);
    const int64_t spw = (slabs + cus - 1) / cus;
    const int64_t grid = (slabs + spw - 1) / spw;  // the last workgroup may hold fewer slabs
    if (grid * 4 < cus * 3) continue;
    int p = 0;
    if (ks == 4) p = (nt * (dual ? 2 : 1) >= 4) ? 2 : 4;
    else if (ks == 14) p = (nt * (dual ? 2 : 1) >= 4) ? 2 : 7;
    else if (ks == 8) p = (nt * (dual ? 2 : 1) >= 4) ? 2 : 4;
    else return false;
    plan->nt = nt;
    plan->ks = ks;
    plan->p = p;
    plan->slabs = static_cast<int>(slabs);
    plan->slabs_per_wg = static_cast<int>(spw);
    plan->grid = static_cast<int>(grid);
    return true;
  }
  return false;
}

// Row-major [n, k] -> fragment-major: 16-byte chunk `o` of the packed weight is lane o % 64 of the
// 1 KB block ((slab * WAVES + wave) * KS + i) * NT + j, i.e. the bytes the streaming kernel's wave
// `wave` requests for K-step i, column tile j of slab `slab` (request() above).
template <int WAVES>
__global__ void pack_stream_weight_kernel(const int8_t* __restrict__ src, int64_t k, int nt, int ks,
                                          int64_t chunks, int8_t* __restrict__ dst) {
  const int64_t o = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (o >= chunks) return;
  const int lane = static_cast<int>(o & 63);
  int64_t blk = o >> 6;
  const int j = static_cast<int>(blk % nt);
  blk /= nt;
  const int i = static_cast<int>(blk % ks);
  blk /= ks;
  const int wave = static_cast<int>(blk % WAVES);
  const int64_t slab = blk / WAVES;
  const int64_t row = slab * nt * 16 + j * 16 + (lane & 15);
  const int64_t kk = static_cast<int64_t>(wave + i * WAVES) * 64 + (lane >> 4) * 16;
  reinterpret_cast<uint4*>(dst)[o] = *reinterpret_cast<const uint4*>(src + row * k + kk);
}

int device_cus() {
  static const int cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
        n = prop.multiProcessorCount;
    }
    return n;
  }();
  return cus;
}

}  // namespace

// Decode rows of ONE 16-row tile, weights of at least 8 MB per launch (below that the one-shot
// kernel's single round trip wins and the weight may be cache resident anyway), 16-byte aligned
// operands, a (N, K) the planner can split evenly. CT2X_STREAM_GEMM=0 keeps the one-shot kernel.
bool gemm_s8_stream_applicable(int64_t m, int64_t n, int64_t k, bool dual) {
  static const bool enabled = [] {
    const char* e = std::getenv("CT2X_STREAM_GEMM");
    return e == nullptr || e[0] != '0';
  }();
  if (!enabled || m < 1 || m > 16) return false;
  if (n * k * (dual ? 2 : 1) < (int64_t(8) << 20)) return false;
  StreamPlan plan;
  return plan_stream(n, k, dual, device_cus(), &plan);
}

// Whether weights of this shape have a packed (fragment-major) streaming form at all - independent
// of the row count of a particular launch. CT2X_STREAM_PACKED=0 keeps the row-major weight.
bool gemm_s8_stream_pack_applicable(int64_t n, int64_t k, bool dual) {
  static const bool enabled = [] {
    const char* e = std::getenv("CT2X_STREAM_PACKED");
    return e == nullptr || e[0] != '0';
  }();
  return enabled && gemm_s8_stream_applicable(1, n, k, dual);
}

void launch_pack_stream_weight(const int8_t* src, int64_t n, int64_t k, bool dual, int8_t* dst,
                               hipStream_t stream) {
  StreamPlan plan;
  if (!plan_stream(n, k, dual, device_cus(), &plan))
    throw Error("launch_pack_stream_weight: shape outside the streaming form");
  const int64_t chunks = n * k / 16;
  hipLaunchKernelGGL(pack_stream_weight_kernel<16>, dim3(static_cast<unsigned>((chunks + 255) / 256)),
                     dim3(256), 0, stream, src, k, plan.nt, plan.ks, chunks, dst);
}

// `bytes` of the fragment-major weight [n, k] as a PrefetchRange: for a packed weight a unit is what
// one wave requests for one slab (KS * NT KB), so "the first take16 pieces of every unit" is the
// same leading fraction of EVERY wave's stream - a plain prefix would serve a few workgroups
// completely and the others not at all (whole weight when bytes >= n * k).
bool weight_prefetch_range(const int8_t* packed, int64_t n, int64_t k, bool dual, int64_t bytes,
                           PrefetchRange* out) {
  StreamPlan plan;
  if (bytes <= 0 || packed == nullptr || !plan_stream(n, k, dual, device_cus(), &plan)) return false;
  const int64_t unit16 = static_cast<int64_t>(plan.ks) * plan.nt * 64;
  const int64_t units = n * k / 16 / unit16;
  int64_t take16 = unit16;
  if (bytes < n * k) {
    take16 = (bytes / 16 / units) & ~int64_t(63);  // whole 1 KB requests
    if (take16 <= 0) return false;
  }
  out->p = packed;
  out->unit16 = unit16;
  out->take16 = take16;
  out->chunks = units * take16;
  return true;
}

// The prologue forms (AMODE 1 / 2 of the kernel) in the engine. CT2X_STREAM_PROLOGUE (read once):
// bit 0 = the RMSNorm + Quantize prologue, bit 1 = quantise on load. DEFAULT 0 - measured on the
// Llama-3-8B shape (MI355X, alternating runs of one call, profiles/r04_l8_stream_prologue_ab.txt):
// separate launches 3.12 - 3.14 ms per decode step, norm prologue 3.14, quantise on load 3.24, both
// 3.36 (first version, per-lane fragment fetch: 3.40 / 3.23 / 3.51). The launches that disappear
// (10 us and 5 us per use) are paid back in full by what replaces them: every one of the ~200
// workgroups pulls the 16 x 4096 fp32 rows itself, 32 CUs of an XCD read the same 256 KB through
// one L2 (8 MB per launch and XCD: ~4 us at the L2's bandwidth, against 2 MB for the int8 rows), and
// the norm's 512-step chain cannot start before its first chunk is there. Bit-identical either way.
// Round 6: bit 2 = RMSNorm + Quantize by sixteen leader workgroups (AMODE 3), bit 3 = ops::Quantize by
// the leaders (AMODE 4, any K). DEFAULT 12: both leader forms on (profiles/r06_l8_leader_ab.txt).
namespace {
std::atomic<int>& prologue_mask_word() {
  static std::atomic<int> mask([] {
    const char* e = std::getenv("CT2X_STREAM_PROLOGUE");
    return e != nullptr ? std::atoi(e) : 0;
  }());
  return mask;
}
}  // namespace
int gemm_s8_stream_prologue_mask() { return prologue_mask_word().load(std::memory_order_relaxed); }
int gemm_s8_stream_set_prologue_mask(int mask) {  // tests / A-B runs; returns the previous value
  return prologue_mask_word().exchange(mask, std::memory_order_relaxed);
}

// Leader forms: which launches they serve. Norm: the shapes of the AMODE 2 prologue (RMSNorm rows of
// 4096, plain or gated). Quantize: a plain Dense at <= 16 rows with a streaming form.
bool gemm_s8_stream_lead_norm_applicable(int64_t m, int64_t n, int64_t k, bool dual, bool has_beta) {
  StreamPlan plan;
  return gemm_s8_stream_norm_applicable(m, n, k, dual, has_beta) &&
         plan_stream(n, k, dual, device_cus(), &plan) && plan.grid >= 16;
}
bool gemm_s8_stream_lead_quant_applicable(int64_t m, int64_t n, int64_t k) {
  StreamPlan plan;
  return gemm_s8_stream_applicable(m, n, k, false) && plan_stream(n, k, false, device_cus(), &plan) &&
         plan.grid >= 16;
}

// RMSNorm (no beta) + Quantize as the prologue: K = 4096 (one compiled chunk geometry), any N
// with a streaming form. LayerNorm rows keep their own launch (gamma and beta do not fit LDS next
// to the two chunk buffers).
bool gemm_s8_stream_norm_applicable(int64_t m, int64_t n, int64_t k, bool dual, bool has_beta) {
  if (has_beta || k != 4096 || !gemm_s8_stream_applicable(m, n, k, dual)) return false;
  StreamPlan plan;
  return plan_stream(n, k, dual, device_cus(), &plan) && (!dual || plan.nt == 2);
}

// Quantise on load: K = 4096, plain Dense (at K = 14336 the fp32 fragments of a lane - 224 floats -
// would have to be pipelined through the K loop; that shape keeps the element-wise Quantize launch).
bool gemm_s8_stream_qol_applicable(int64_t m, int64_t n, int64_t k) {
  return k == 4096 && gemm_s8_stream_applicable(m, n, k, false);
}

void launch_gemm_s8_stream(const int8_t* a, const int8_t* b, int M, int N, int K,
                           const GemmS8Epilogue& ep_in, hipStream_t stream) {
  const bool dual = ep_in.b2 != nullptr;
  StreamPlan plan;
  if (!plan_stream(N, K, dual, device_cus(), &plan))
    throw Error("launch_gemm_s8_stream: shape outside the streaming form");
  const GemmS8Epilogue& ep = ep_in;
  const bool lead = ep.lead_flag != nullptr;
  const int amode = ep.norm_x != nullptr ? (lead ? 3 : 2) : (ep.a_f32 != nullptr ? (lead ? 4 : 1) : 0);
  if (lead && !(amode >= 3 && ep.lead_q != nullptr && ep.lead_scale != nullptr && ep.lead_tag >= 1 &&
                ep.lead_tag < 1024 && plan.grid >= 16 && ep.mode == 1 &&
                (reinterpret_cast<uintptr_t>(ep.lead_q) & 15) == 0))
    throw Error("launch_gemm_s8_stream: leader form without its row buffers / tag, or a grid below 16");
  if (amode == 3 && !(ep.norm_beta == nullptr && K == 4096 && ep.norm_gamma != nullptr &&
                      (reinterpret_cast<uintptr_t>(ep.norm_x) & 15) == 0 &&
                      (reinterpret_cast<uintptr_t>(ep.norm_gamma) & 15) == 0))
    throw Error("launch_gemm_s8_stream: the leader norm needs RMSNorm rows of 4096 aligned floats");
  if (amode == 4 && !(!dual && ep.a_amax != nullptr && (reinterpret_cast<uintptr_t>(ep.a_f32) & 15) == 0))
    throw Error("launch_gemm_s8_stream: the leader Quantize needs a plain Dense and the row absmax");
  if (amode == 2 && !(ep.norm_beta == nullptr && K == 4096 && ep.mode == 1 && ep.norm_gamma != nullptr &&
                      (reinterpret_cast<uintptr_t>(ep.norm_x) & 15) == 0 &&
                      (reinterpret_cast<uintptr_t>(ep.norm_gamma) & 15) == 0))
    throw Error("launch_gemm_s8_stream: the norm prologue needs RMSNorm rows of 4096 aligned floats");
  if (amode == 1 && !(K == 4096 && !dual && ep.mode == 1 && ep.a_amax != nullptr &&
                      (reinterpret_cast<uintptr_t>(ep.a_f32) & 15) == 0))
    throw Error("launch_gemm_s8_stream: the quantise-on-load form needs K = 4096, a plain Dense and the row absmax");
  const bool a_norm = amode == 2 || amode == 3, a_quant = amode == 1 || amode == 4;
  const int8_t* a_arg = a_norm ? reinterpret_cast<const int8_t*>(ep.norm_x)
                               : (a_quant ? reinterpret_cast<const int8_t*>(ep.a_f32) : a);
  const float* aux = a_norm ? ep.norm_gamma : (a_quant ? ep.a_amax : nullptr);
  const int aux_n = a_quant ? (ep.a_amax_parts > 0 ? ep.a_amax_parts : 1) : 0;
#define CT2X_STREAM_LAUNCH(NT, KS, P, DUAL, PACKED, AMODE)                                        \
  do {                                                                                            \
    auto fn = gemm_s8_stream_kernel<NT, 16, KS, P, DUAL, PACKED, AMODE>;                          \
    const size_t lds = StreamLds<NT, 16, KS, DUAL, AMODE>::bytes(K);                              \
    allow_big_lds(reinterpret_cast<const void*>(fn), lds);                                        \
    hipLaunchKernelGGL(fn, dim3(plan.grid), dim3(1024), lds, stream, a_arg, b, M, N, K,           \
                       plan.slabs_per_wg, plan.slabs, ep.b2, aux, aux_n, ep);                     \
  } while (0)
#define CT2X_STREAM_A(NT, KS, P, DUAL, AMODE)                                                     \
  if (plan.nt == NT && plan.ks == KS && plan.p == P && dual == DUAL && amode == AMODE) {          \
    if (ep.packed_b) CT2X_STREAM_LAUNCH(NT, KS, P, DUAL, true, AMODE);                            \
    else CT2X_STREAM_LAUNCH(NT, KS, P, DUAL, false, AMODE);                                       \
    return;                                                                                       \
  }
#define CT2X_STREAM(NT, KS, P, DUAL) CT2X_STREAM_A(NT, KS, P, DUAL, 0)
  // (forms whose register ring spills at 128 VGPRs - 16 waves per workgroup - are not compiled:
  // plan_stream never returns them)
  CT2X_STREAM(4, 4, 2, false) CT2X_STREAM(2, 4, 4, false) CT2X_STREAM(1, 4, 4, false)
  CT2X_STREAM(2, 8, 4, false) CT2X_STREAM(1, 8, 4, false)
  CT2X_STREAM(1, 14, 7, false)
  CT2X_STREAM(2, 4, 2, true) CT2X_STREAM(1, 4, 4, true)
  CT2X_STREAM(2, 8, 2, true) CT2X_STREAM(1, 8, 4, true)
  // K = 4096 with a prologue: quantise on load (plain Dense), RMSNorm + Quantize (plain and gated)
  CT2X_STREAM_A(4, 4, 2, false, 1) CT2X_STREAM_A(2, 4, 4, false, 1) CT2X_STREAM_A(1, 4, 4, false, 1)
  CT2X_STREAM_A(4, 4, 2, false, 2) CT2X_STREAM_A(2, 4, 4, false, 2) CT2X_STREAM_A(1, 4, 4, false, 2)
  CT2X_STREAM_A(2, 4, 2, true, 2)  // (the one-tile gated form would spill a register)
  // leader forms (round 6): RMSNorm + Quantize at K = 4096, ops::Quantize at any K
  CT2X_STREAM_A(4, 4, 2, false, 3) CT2X_STREAM_A(2, 4, 4, false, 3) CT2X_STREAM_A(1, 4, 4, false, 3)
  CT2X_STREAM_A(2, 4, 2, true, 3)
  CT2X_STREAM_A(4, 4, 2, false, 4) CT2X_STREAM_A(2, 4, 4, false, 4) CT2X_STREAM_A(1, 4, 4, false, 4)
  CT2X_STREAM_A(2, 8, 4, false, 4) CT2X_STREAM_A(1, 8, 4, false, 4) CT2X_STREAM_A(1, 14, 7, false, 4)
#undef CT2X_STREAM
#undef CT2X_STREAM_A
#undef CT2X_STREAM_LAUNCH
  throw Error("launch_gemm_s8_stream: no compiled form for this plan");
}

}  // namespace ct2x
