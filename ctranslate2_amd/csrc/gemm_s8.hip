// int8 x int8 -> int32 projection GEMM for gfx950 with the Dequantize(+bias,+activation,
// +residual) epilogue fused in, replacing the reference's cuBLAS GemmEx 8I->32I call
// (src/cuda/primitives.cu:571-597) + dequantize_gemm_output kernel
// (src/ops/dequantize_gpu.cu:31-55) + residual Add (src/layers/common.cc:392-401).
//
//   C[m,n] = A[m,k] . B[n,k]^T     A = quantised activations, B = int8 weight [N,K] row-major
//
// Both operands are K-contiguous, which is exactly the MFMA i8 fragment shape:
// v_mfma_i32_16x16x64_i8 takes 16 consecutive K bytes per lane for A (row = lane&15) and for B
// (col = lane&15), K-block = lane>>4. Tiles are staged through LDS with a 16-byte row pad
// (144-byte stride => conflict-free ds_read_b128), global->register->LDS with the next tile's
// global loads in flight during the MFMAs.
//
// Integer accumulation is exact, so any correct tiling is bit-identical to the oracle.

#include "device_utils.h"
#include "greedy_fuse.h"
#include "kernels.h"
#include "row_norm.h"
#include "runtime.h"

#include <cstdio>
#include <type_traits>
#include <cstdlib>

namespace ct2x {

namespace {

constexpr int BK = 128;            // K bytes per LDS tile (two MFMA K-steps of 64)
constexpr int LDS_STRIDE = BK + 16;  // bytes per tile row in LDS

__device__ __forceinline__ void store_epilogue(const GemmS8Epilogue& ep, int64_t row, int64_t col,
                                               int acc, float r_a) {
  const int64_t idx = row * ep.ldy + col;
  if (ep.mode == 0) {
    ep.c_i32[idx] = acc;
  } else {
    // CPU formula order (src/cpu/kernels.cc:666-673): c * (1/a_scale) / b_scale + bias.
    float v = static_cast<float>(acc) * r_a;
    v = v / ep.w_scale[col];
    if (ep.bias) v = v + ep.bias[col];
    v = apply_activation(v, ep.act);
    if (ep.residual) v = ep.residual[idx] + v;  // ops::Add()(residual, output, output)
    ep.y[idx] = v;
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void gemm_s8_mfma_kernel(const int8_t* __restrict__ A,
                                                           const int8_t* __restrict__ B,
                                                           int M, int N, int K,
                                                           GemmS8Epilogue ep) {
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int TM = WTM / 16, TN = WTN / 16;
  constexpr int A_LOADS = BM / 32, B_LOADS = BN / 32;  // 16-byte chunks per thread per tile
  static_assert(BM % 32 == 0 && BN % 32 == 0, "tile rows are loaded 32 at a time");

  __shared__ __attribute__((aligned(16))) int8_t lds[(BM + BN) * LDS_STRIDE];
  int8_t* As = lds;
  int8_t* Bs = lds + BM * LDS_STRIDE;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int64_t m0 = static_cast<int64_t>(blockIdx.y) * BM;
  const int64_t n0 = static_cast<int64_t>(blockIdx.x) * BN;

  // Staging map: 8 consecutive threads fetch one 128-byte row segment.
  const int ld_row = tid >> 3;  // 0..31
  const int ld_chunk = tid & 7;  // 16-byte chunk within the BK segment

  v4i acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = v4i{0, 0, 0, 0};

  uint4 ra[A_LOADS], rb[B_LOADS];

  auto load_tile = [&](int k0) {
    const int kk = k0 + ld_chunk * 16;
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
      const int64_t row = m0 + ld_row + i * 32;
      ra[i] = (row < M && kk < K) ? *reinterpret_cast<const uint4*>(A + row * K + kk)
                                  : uint4{0, 0, 0, 0};
    }
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) {
      const int64_t row = n0 + ld_row + i * 32;
      rb[i] = (row < N && kk < K) ? *reinterpret_cast<const uint4*>(B + row * K + kk)
                                  : uint4{0, 0, 0, 0};
    }
  };

  load_tile(0);
  for (int k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();  // previous tile fully consumed
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i)
      *reinterpret_cast<uint4*>(As + (ld_row + i * 32) * LDS_STRIDE + ld_chunk * 16) = ra[i];
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i)
      *reinterpret_cast<uint4*>(Bs + (ld_row + i * 32) * LDS_STRIDE + ld_chunk * 16) = rb[i];
    __syncthreads();
    if (k0 + BK < K) load_tile(k0 + BK);  // in flight while the MFMAs run

#pragma unroll
    for (int ks = 0; ks < BK / 64; ++ks) {
      v4i af[TM], bf[TN];
      const int koff = ks * 64 + (lane >> 4) * 16;
#pragma unroll
      for (int i = 0; i < TM; ++i)
        af[i] = *reinterpret_cast<const v4i*>(As + (wm * WTM + i * 16 + (lane & 15)) * LDS_STRIDE +
                                              koff);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bf[j] = *reinterpret_cast<const v4i*>(Bs + (wn * WTN + j * 16 + (lane & 15)) * LDS_STRIDE +
                                              koff);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  }

  // C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg.
  // Every epilogue operand is fetched BEFORE the first store: a load that follows a store to y
  // in program order has to assume aliasing, and the epilogue would serialise on memory latency
  // (measured: +23 us per launch at [1024, 512] x 512). The residual of row group i+1 is
  // requested ahead of the stores of row group i.
  const int r16 = lane & 15, g4 = lane >> 4;
  float ws[TN], bias[TN];
  bool col_ok[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int64_t col = n0 + wn * WTN + j * 16 + r16;
    col_ok[j] = col < N;
    ws[j] = (ep.mode == 1 && col_ok[j]) ? ep.w_scale[col] : 1.f;
    bias[j] = (ep.mode == 1 && ep.bias && col_ok[j]) ? ep.bias[col] : 0.f;
  }
  float row_scale[TM][4];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t row = m0 + wm * WTM + i * 16 + g4 * 4 + e;
      row_scale[i][e] = (ep.mode == 1 && row < M) ? ep.a_scale[row] : 1.f;
    }
  const bool has_res = ep.mode == 1 && ep.residual != nullptr;
  const int64_t col0 = n0 + wn * WTN + r16;
  auto load_residual = [&](int i, float (&res)[4][TN]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t row = m0 + wm * WTM + i * 16 + g4 * 4 + e;
      const float* p = ep.residual + row * ep.ldy + col0;
#pragma unroll
      for (int j = 0; j < TN; ++j)
        res[e][j] = (has_res && row < M && col_ok[j]) ? p[j * 16] : 0.f;
    }
  };
  // (ACT: no activation / ReLU / the runtime choice - decided ONCE per launch; with the eight-way choice
  // inside the 64 unrolled elements of a 128 x 128 tile the epilogue was several kilobytes of code
  // per element group and the fused form ran 29 us where the int32 form runs 12.5, [128, 32000] x 512)
  auto store_group = [&](int i, const float (&res)[4][TN], auto act_tag) {
    constexpr int ACT = decltype(act_tag)::value;  // -1 none, 0 ReLU, 1 the runtime choice
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t row = m0 + wm * WTM + i * 16 + g4 * 4 + e;
      if (row >= M) continue;
      if (ep.mode == 0) {
        int32_t* p = ep.c_i32 + row * ep.ldy + col0;
#pragma unroll
        for (int j = 0; j < TN; ++j)
          if (col_ok[j]) p[j * 16] = acc[i][j][e];
      } else {
        const float r_a = 1.f / row_scale[i][e];
        float* p = ep.y + row * ep.ldy + col0;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          // CPU formula order (src/cpu/kernels.cc:666-673): c * (1/a_scale) / b_scale + bias.
          float v = static_cast<float>(acc[i][j][e]) * r_a;
          v = v / ws[j];
          if (ep.bias) v = v + bias[j];
          if (ACT == 0) v = apply_activation(v, 0);
          else if (ACT > 0) v = apply_activation(v, ep.act);
          if (has_res) v = res[e][j] + v;
          if (col_ok[j]) p[j * 16] = v;
        }
      }
    }
  };
  float res_a[4][TN], res_b[4][TN];
  load_residual(0, res_a);
  // (three arms, no more: none, ReLU - the encoder's - and the runtime choice; with one arm per
  // activation - eight instantiations behind a switch - the accumulators went to scratch memory, 992
  // bytes per lane, and the launch ran 81 us instead of 17)
#define CT2X_STORE_ALL(TAG)                                       \
  _Pragma("unroll") for (int i = 0; i < TM; i += 2) {             \
    if (i + 1 < TM) load_residual(i + 1, res_b);                  \
    store_group(i, res_a, TAG);                                   \
    if (i + 2 < TM) load_residual(i + 2, res_a);                  \
    if (i + 1 < TM) store_group(i + 1, res_b, TAG);               \
  }
  if (ep.mode == 0 || ep.act < 0) {
    CT2X_STORE_ALL((std::integral_constant<int, -1>{}))
  } else if (ep.act == 0) {
    CT2X_STORE_ALL((std::integral_constant<int, 0>{}))
  } else {
    CT2X_STORE_ALL((std::integral_constant<int, 1>{}))
  }
#undef CT2X_STORE_ALL
}


// ---------------------------------------------------------------------------------------------
// Skinny-M kernel for decode steps (M = batch*beam <= 32): the GEMM is a weight stream, not a
// tile problem, and its wall time is ONE memory round trip if every load is in flight at once.
// A workgroup owns NT*16 output columns and all M rows; its waves split K (wave w takes the
// 64-byte K-steps w, w+WAVES, ...), load their A/B MFMA fragments straight from global memory
// into registers (no LDS staging: a weight byte is used by exactly one wave), and the partial
// int32 tiles are summed through LDS (exact, order independent) before the fused epilogue.
// ---------------------------------------------------------------------------------------------
// DUAL: the gated FFN input pair of FeedForwardNetwork::operator() (src/layers/transformer.cc:
// 21-51, ffn_glu) in one launch: y = act(x.W0^T + b0) * (x.W1^T + b1) - both weight streams share
// the activation fragments, the two fp32 intermediates and the ops::Mul launch disappear.
// NORM: the LayerNorm / RMSNorm + Quantize that feeds a pre-norm Dense runs as the prologue of
// this launch (every workgroup normalises and quantises the <= 32 input rows itself: 64 KB of
// fp32 from L2, one wave per row, DPP reductions) - one launch and one activation round trip
// less per Dense. The quantised rows live in LDS (row stride K + 16 bytes) and the MFMA A
// fragments are read from there.
// QOL ("quantise on load"): the ops::Quantize that feeds this Dense is folded into the A-fragment
// fetch. The producer of the fp32 activation left the row absmax (GemmS8Epilogue::row_amax_out of
// the previous launch), so scale = 127 / amax is known per row and every lane converts the 16 floats
// of its fragment itself (to_s8: src/cpu/kernels.cc:577-651) - the int8 copy of the activation, its
// scale vector and the Quantize launch between the two GEMMs of a feed-forward block disappear.
// NTW: the weight stream is requested with the non-temporal policy (`global_load_dwordx4 ... nt`):
// for models whose weights are far larger than the 256 MB MALL (the Llama shapes stream 7.5 GB per
// step) a weight byte is used once, and keeping it out of the cache hierarchy's retention policy
// shortens issue -> landed (MI355X_MICROARCH.md, "nt-weights"). Transformer-base (38 MB of decoder
// weights, MALL resident from step to step) keeps the default policy.
// ARGMAX: GreedyFuse (kernels.h) - the epilogue keeps, per row, the best (value, column) key of the
// workgroup's columns (thread-local over its column tiles, then across the 16 lanes that share
// the row), max-combines it into the row's global slot with one 64-bit atomic, and the workgroup
// that draws the last ticket finishes the step.
template <int MT, int NT, int WAVES, int U, bool DUAL = false, int NORM = 0, bool QOL = false,
          bool NTW = false, bool ARGMAX = false>
__global__ __launch_bounds__(WAVES * 64) void gemm_s8_skinny_kernel(
    const int8_t* __restrict__ A, const int8_t* __restrict__ B, int M, int N, int K,
    const float* __restrict__ aux, GemmS8Epilogue ep) {
  // aux (preloaded like A .. K): QOL = the row absmax of the fp32 activation; unused otherwise
  constexpr int ND = DUAL ? 2 : 1;
  __shared__ int partial[ND][WAVES][MT][NT][64][4];
  // NORM (dynamic LDS): transposed fp32 rows [16][8 * pad] (row_norm.h norm_chain_slot) | gamma [K]
  // | beta [K] | int8 A tile [16][K + 16] | row scales [16]
  extern __shared__ __attribute__((aligned(16))) int8_t norm_lds[];
  constexpr int NORM_CH = NORM > 0 ? NORM : 1;  // float4s per lane: K == 64 * NORM
  static_assert(!NORM || (MT == 1 && WAVES == 4), "NORM: 16 rows on four waves, 16 lanes per row");
  static_assert(!QOL || (MT == 1 && !NORM && !DUAL && (WAVES == 4 || WAVES == 8)),
                "QOL: one 16-row tile, plain Dense, K <= 256 * WAVES");
  const int qa_stride = K + 16;
  const int norm_pad = NORM ? norm_chain_pad(K) : 0;
  const int norm_x_stride = 8 * norm_pad;
  float* norm_x_lds = reinterpret_cast<float*>(norm_lds);
  float* norm_gamma_lds = norm_x_lds + 16 * norm_x_stride;
  float* norm_beta_lds = norm_gamma_lds + K;
  int8_t* norm_qa_lds = reinterpret_cast<int8_t*>(norm_beta_lds + K);
  float* norm_scales = reinterpret_cast<float*>(norm_qa_lds + 16 * qa_stride);
  // Only A, B, M, N, K arrive preloaded in SGPRs; every field of `ep` costs a kernarg round trip
  // (s_load + wait, ~1 us cold). The first round of weight and activation fragments needs nothing
  // but the preloaded arguments, so it is requested BEFORE the first access to `ep`: the kernarg
  // fetch, the epilogue operands and the weight stream share one memory round trip instead of
  // queueing behind each other.
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int64_t n0 = static_cast<int64_t>(blockIdx.x) * (NT * 16);
  const int ksteps = (K + 63) / 64;
  const int rb = blockIdx.y * (MT * 16);  // this workgroup's block of MT * 16 rows (grid.y)
  const int M_all = M;
  M -= rb;
  // NORM: the fp32 rows to normalise travel in the (otherwise unused) A slot, so that their request
  // does not wait for the argument block either
  const float* norm_x_pre = NORM ? reinterpret_cast<const float*>(A) + static_cast<int64_t>(rb) * K : nullptr;
  if (!NORM && !QOL) A += static_cast<int64_t>(rb) * K;
  v4f xv[NORM_CH];
  if (NORM) {
    // lane (rg, l16) of wave w: the float4s l16, l16 + 16, ... of row 4 w + rg (see the prologue)
    const int row_pre = wave * 4 + (lane >> 4);
    const v4f* xg = reinterpret_cast<const v4f*>(norm_x_pre + static_cast<int64_t>(row_pre < M ? row_pre : 0) * K) + (lane & 15);
#pragma unroll
    for (int j = 0; j < NORM_CH; ++j) xv[j] = xg[16 * j];
  }
  // QOL: thread t owns float4 column t of every row of the 16-row tile (K / 4 <= THREADS): sixteen
  // coalesced row segments, requested here, quantised into the LDS A tile once the row scales are
  // known (the strided per-lane fetch of the first version cost more than the Quantize launch)
  v4f qx[QOL ? 16 : 1];
  if (QOL) {
    const v4f* xsrc = reinterpret_cast<const v4f*>(reinterpret_cast<const float*>(A) +
                                                   static_cast<int64_t>(rb) * K);
    const int k4 = K >> 2;
#pragma unroll
    for (int row = 0; row < 16; ++row)
      qx[row] = (tid < k4 && row < M) ? xsrc[static_cast<int64_t>(row) * k4 + tid] : v4f{0.f, 0.f, 0.f, 0.f};
  }
  v4i bpre[U][NT], apre[U][MT];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int ks = wave + u * WAVES;
    const int kk = ks * 64 + g * 16;
    const bool k_ok = (ks < ksteps) && (kk < K);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int64_t col = n0 + j * 16 + r;
      bpre[u][j] = (k_ok && col < N)
                       ? (NTW ? __builtin_nontemporal_load(reinterpret_cast<const v4i*>(B + col * K + kk))
                              : *reinterpret_cast<const v4i*>(B + col * K + kk))
                       : v4i{0, 0, 0, 0};
    }
    if (!NORM && !QOL) {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int row = i * 16 + r;
        apre[u][i] = (k_ok && row < M)
                         ? *reinterpret_cast<const v4i*>(A + static_cast<int64_t>(row) * K + kk)
                         : v4i{0, 0, 0, 0};
      }
    }
  }
  if (blockIdx.y != 0) {
    // rebase the row-indexed operands of the epilogue. Many short row blocks beat one tall
    // workgroup per column tile: [128, 512] x 512 runs 6.9 us as 32 workgroups of 8 row tiles and
    // 4.7 us as 256 workgroups of one (the weight tile is re-read per block, from L2).
    if (ep.a_scale) ep.a_scale += rb;
    if (ep.c_i32) ep.c_i32 += static_cast<int64_t>(rb) * ep.ldy;
    if (ep.y) ep.y += static_cast<int64_t>(rb) * ep.ldy;
    if (ep.residual) ep.residual += static_cast<int64_t>(rb) * ep.ldy;
    if (ep.row_amax_out) ep.row_amax_out += rb;
  }
  constexpr int THREADS = WAVES * 64;
  constexpr int OUTS = MT * NT * 256;                       // outputs of the workgroup tile
  constexpr int OPT = (OUTS + THREADS - 1) / THREADS;       // outputs per thread in the epilogue
  const int8_t* __restrict__ B2 = DUAL ? ep.b2 : nullptr;

  // Epilogue operands are fetched FIRST so their latency hides under the weight stream: thread t
  // finalises outputs t, t+THREADS, ... of the [MT*16, NT*16] tile, 16 consecutive threads = 16
  // consecutive columns (coalesced 64-byte stores).
  float e_ra[OPT], e_ws[OPT], e_bias[OPT], e_res[OPT], e_ws2[OPT], e_bias2[OPT];
#pragma unroll
  for (int o = 0; o < OPT; ++o) {
    const int flat = tid + o * THREADS;
    const int tile = flat >> 8, within = flat & 255;
    const int row = (tile / NT) * 16 + (within >> 4);
    const int64_t col = n0 + (tile % NT) * 16 + (within & 15);
    e_ra[o] = e_ws[o] = e_ws2[o] = 1.f;
    e_bias[o] = e_res[o] = e_bias2[o] = 0.f;
    if (ep.mode == 1 && flat < OUTS && row < M && col < N) {
      if (!NORM && !QOL) e_ra[o] = ep.a_scale[row];
      e_ws[o] = ep.w_scale[col];
      if (ep.bias) e_bias[o] = ep.bias[col];
      if (ep.residual) e_res[o] = ep.residual[static_cast<int64_t>(row) * ep.ldy + col];
      if (DUAL) {
        e_ws2[o] = ep.w_scale2[col];
        if (ep.bias2) e_bias2[o] = ep.bias2[col];
      }
    }
  }

  // ARGMAX: the step and the row's minimum length decide which column is masked (the end token
  // below the minimum length, apply_min_length src/decoding.cc:384-408); thread t finalises row
  // i * 16 + (t >> 4) of every row tile i (THREADS == 256: `within` == t for every output).
  static_assert(!ARGMAX || (WAVES == 4 && !DUAL && !NORM && !QOL), "ARGMAX: plain Dense on 4 waves");
  // (requested here, consumed after the K loop: the requests ride along with the weight stream)
  int gf_step = 0, gf_plen[MT];
  if (ARGMAX) {
    gf_step = *ep.gf.step_ptr;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int row = i * 16 + (tid >> 4);
      gf_plen[i] = 0;
      if (ep.gf.prefix_ids != nullptr && ep.gf.prefix_extends_min && row < M)
        gf_plen[i] = ep.gf.prefix_lens[rb + row];
    }
  }

  if (NORM) {
    // FOUR ROWS PER WAVE, 16 lanes per row, 16 rows per workgroup (grid.y walks the row blocks;
    // the weight tile is re-read per block from L2). The reference's compiled statistics are
    // eight lane accumulators walking the row with stride 8 followed by a 4/2/1 xor tree
    // (row_norm.h): lanes l and l + 8 of a row both run chain l, so the normalisation of a whole
    // step is ONE pass of depth/8 dependent steps on every wave. Lane (rg, l16) loads the float4s
    // l16, l16 + 16, ... of its row (16 lanes = 256 contiguous bytes), parks them transposed in
    // LDS for the contiguous chain reads, keeps them in registers for the normalisation, and
    // writes the quantised bytes to the LDS A tile.
    // Formulas: src/cpu/kernels.cc:461-496 (layer_norm), :543-573 (rms_norm), :577-651
    // (quantize), in the compiled order restated by oracle/ct2_oracle.c row_stats.
    // The code below is branch-free per element on purpose (uniform choices are hoisted or turned
    // into selects): a first version with per-element branches was 6000 instructions long and
    // slower than the launch it replaced. Per-lane work is what bounds the prologue (a wave64
    // VALU instruction occupies its SIMD for 4 cycles), hence 16 rows per workgroup, not 32.
    const int rg = lane >> 4, l16 = lane & 15, l = lane & 7;
    const int row = wave * 4 + rg;
    const bool valid = row < M;
    const int n4 = K / 4;
    const v4f* g4 = reinterpret_cast<const v4f*>(ep.norm_gamma);
    const v4f* b4 = reinterpret_cast<const v4f*>(ep.norm_beta);
    const bool ln = ep.norm_beta != nullptr;
    // (xv: the row's float4s, requested at the top of the kernel)
    for (int i = tid; i < n4; i += THREADS) {
      reinterpret_cast<v4f*>(norm_gamma_lds)[i] = g4[i];
      if (ln) reinterpret_cast<v4f*>(norm_beta_lds)[i] = b4[i];
    }
    float* my_row = norm_x_lds + row * norm_x_stride;
    {
      // column c = 4 * (l16 + 16 j) + e sits at (c & 7) * pad + (c >> 3)
      float* park = my_row + ((4 * l16) & 7) * norm_pad + (l16 >> 1);
#pragma unroll
      for (int j = 0; j < NORM_CH; ++j) {
#pragma unroll
        for (int e = 0; e < 4; ++e) park[e * norm_pad + 8 * j] = xv[j][e];
      }
    }
    __syncthreads();
    float sum = 0.f, sumsq = 0.f;
    {
      const v4f* chain = reinterpret_cast<const v4f*>(my_row + l * norm_pad);
      constexpr int STEPS4 = NORM_CH * 2;  // K / 8 chain steps, four per 16-byte read
#pragma unroll
      for (int i = 0; i < STEPS4; ++i) {
        const v4f v = chain[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          sumsq = fmaf(v[e], v[e], sumsq);
          sum = sum + v[e];
        }
      }
      sumsq = dpp_xor4(sumsq) + sumsq;
      sum = dpp_xor4(sum) + sum;
      sumsq = dpp_move<0x4E, 0xf>(0.f, sumsq) + sumsq;
      sum = dpp_move<0x4E, 0xf>(0.f, sum) + sum;
      sumsq = dpp_move<0xB1, 0xf>(0.f, sumsq) + sumsq;
      sum = dpp_move<0xB1, 0xf>(0.f, sum) + sum;
    }
    const NormStats st = norm_stats(sum, sumsq, K, ep.norm_eps, ln);
    const v4f* gl = reinterpret_cast<const v4f*>(norm_gamma_lds) + l16;
    const v4f* bl = reinterpret_cast<const v4f*>(norm_beta_lds) + l16;
    // K % 8 == 0: no column takes the 4-wide remainder form of norm_apply (row_norm.h)
    if (ln) {
#pragma unroll
      for (int j = 0; j < NORM_CH; ++j) {
        const v4f gv = gl[16 * j], bv = bl[16 * j];
#pragma unroll
        for (int e = 0; e < 4; ++e) xv[j][e] = fmaf(xv[j][e] - st.mean, st.mul * gv[e], bv[e]);
      }
    } else if (ep.norm_rms_residual) {
#pragma unroll
      for (int j = 0; j < NORM_CH; ++j) {
        const v4f gv = gl[16 * j];
#pragma unroll
        for (int e = 0; e < 4; ++e) xv[j][e] = (st.mul * xv[j][e]) * (gv[e] + 1.f);
      }
    } else {
#pragma unroll
      for (int j = 0; j < NORM_CH; ++j) {
        const v4f gv = gl[16 * j];
#pragma unroll
        for (int e = 0; e < 4; ++e) xv[j][e] = (xv[j][e] * gv[e]) * st.mul;
      }
    }
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < NORM_CH; ++j)
      amax = fmaxf(amax, fmaxf(fmaxf(fabsf(xv[j][0]), fabsf(xv[j][1])),
                               fmaxf(fabsf(xv[j][2]), fabsf(xv[j][3]))));
    amax = fmaxf(amax, dpp_move<0x140, 0xf>(0.f, amax));  // row_mirror: lane i <- lane 15 - i
    amax = fmaxf(amax, dpp_xor4(amax));
    amax = fmaxf(amax, dpp_move<0x4E, 0xf>(0.f, amax));
    amax = fmaxf(amax, dpp_move<0xB1, 0xf>(0.f, amax));
    const float scale = (amax != 0.f) ? 127.f / amax : 1.f;
    const bool q_shift = (ep.quant_mode & kQuantU8Shift) != 0;
    const bool q_round = (ep.quant_mode & kQuantRound) != 0;
    if (valid) {
      if (l16 == 0) norm_scales[row] = scale;
      int* qr = reinterpret_cast<int*>(norm_qa_lds + row * qa_stride) + l16;
      if (q_shift && q_round) {
        // the x86 default (u8 shift + round to nearest even): three VALU ops per element, the four
        // low bytes gathered by two v_perm, and "- 128" on every byte as one xor
#pragma unroll
        for (int j = 0; j < NORM_CH; ++j) {
          int b[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) b[e] = static_cast<int>(rintf(fmaf(xv[j][e], scale, 128.f)));
          const unsigned lo = __builtin_amdgcn_perm(static_cast<unsigned>(b[1]),
                                                    static_cast<unsigned>(b[0]), 0x0c0c0400u);
          const unsigned hi = __builtin_amdgcn_perm(static_cast<unsigned>(b[3]),
                                                    static_cast<unsigned>(b[2]), 0x04000c0cu);
          qr[16 * j] = static_cast<int>((lo | hi) ^ 0x80808080u);
        }
      } else {
#pragma unroll
        for (int j = 0; j < NORM_CH; ++j) {
          int packed = 0;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            // to_s8 (device_utils.h) with the two uniform mode bits as selects
            float v = q_shift ? fmaf(xv[j][e], scale, 128.f) : xv[j][e] * scale;
            v = q_round ? rintf(v) : v;
            const int i32 = static_cast<int>(v);
            const int b = q_shift ? ((i32 & 0xff) - 128) : i32;
            packed |= (b & 0xff) << (8 * e);
          }
          qr[16 * j] = packed;
        }
      }
    }
    __syncthreads();
  }

  v4i acc[ND][MT][NT];
#pragma unroll
  for (int d = 0; d < ND; ++d)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[d][i][j] = v4i{0, 0, 0, 0};

  // QOL prologue: row scales (127 / absmax, 1 for a zero row: what ops::Quantize would have stored),
  // then every thread quantises its column of the tile into LDS (to_s8, device_utils.h).
  float* qol_scales = reinterpret_cast<float*>(norm_lds);          // [16]
  int8_t* qol_qa = norm_lds + 64;                                  // [16][K + 16]
  if (QOL) {
    if (tid < 16) {
      const float am = (tid < M) ? aux[rb + tid] : 0.f;
      qol_scales[tid] = (am != 0.f) ? 127.f / am : 1.f;
    }
    __syncthreads();
    const bool q_shift = (ep.quant_mode & kQuantU8Shift) != 0;
    const bool q_round = (ep.quant_mode & kQuantRound) != 0;
    if (tid < (K >> 2)) {
#pragma unroll
      for (int row = 0; row < 16; ++row) {
        const float scale = qol_scales[row];
        int word;
        if (q_shift && q_round) {
          int b[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) b[e] = static_cast<int>(rintf(fmaf(qx[row][e], scale, 128.f)));
          const unsigned lo = __builtin_amdgcn_perm(static_cast<unsigned>(b[1]),
                                                    static_cast<unsigned>(b[0]), 0x0c0c0400u);
          const unsigned hi = __builtin_amdgcn_perm(static_cast<unsigned>(b[3]),
                                                    static_cast<unsigned>(b[2]), 0x04000c0cu);
          word = static_cast<int>((lo | hi) ^ 0x80808080u);
        } else {
          word = 0;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            word |= (static_cast<int>(to_s8(qx[row][e], scale, ep.quant_mode)) & 0xff) << (8 * e);
        }
        // rows >= M hold zeros, whose image under the u8-shift arm is 0 as well (128 - 128)
        reinterpret_cast<int*>(qol_qa + row * qa_stride)[tid] = word;
      }
    }
    __syncthreads();
  }

  for (int ks0 = wave; ks0 < ksteps; ks0 += WAVES * U) {
    const bool first_round = ks0 == wave;  // its fragments were requested at the top of the kernel
    v4i af[U][MT], bf[ND][U][NT];
    // Issue every load of this round before the first MFMA consumes one.
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ks = ks0 + u * WAVES;
      const int kk = ks * 64 + g * 16;
      const bool k_ok = (ks < ksteps) && (kk < K);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int row = i * 16 + r;
        if (QOL) {
          af[u][i] = (k_ok && row < M) ? *reinterpret_cast<const v4i*>(qol_qa + row * qa_stride + kk)
                                       : v4i{0, 0, 0, 0};
        } else if (NORM)
          af[u][i] = (k_ok && row < M)
                         ? *reinterpret_cast<const v4i*>(norm_qa_lds + row * qa_stride + kk)
                         : v4i{0, 0, 0, 0};
        else
          af[u][i] = first_round ? apre[u][i]
                     : (k_ok && row < M)
                         ? *reinterpret_cast<const v4i*>(A + static_cast<int64_t>(row) * K + kk)
                         : v4i{0, 0, 0, 0};
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int64_t col = n0 + j * 16 + r;
        if (NORM || first_round)
          bf[0][u][j] = bpre[u][j];
        else
          bf[0][u][j] = (k_ok && col < N)
                            ? (NTW ? __builtin_nontemporal_load(reinterpret_cast<const v4i*>(B + col * K + kk))
                                   : *reinterpret_cast<const v4i*>(B + col * K + kk))
                            : v4i{0, 0, 0, 0};
        if (DUAL)
          bf[ND - 1][u][j] = (k_ok && col < N)
                                 ? (NTW ? __builtin_nontemporal_load(reinterpret_cast<const v4i*>(B2 + col * K + kk))
                                        : *reinterpret_cast<const v4i*>(B2 + col * K + kk))
                                 : v4i{0, 0, 0, 0};
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int d = 0; d < ND; ++d)
            acc[d][i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[u][i], bf[d][u][j],
                                                                 acc[d][i][j], 0, 0, 0);
  }

#pragma unroll
  for (int d = 0; d < ND; ++d)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        *reinterpret_cast<v4i*>(&partial[d][wave][i][j][lane][0]) = acc[d][i][j];
  __syncthreads();

  // Optional per-row absmax of the stored values (first half of the next Quantize): LDS max per
  // row of the tile, then one global max per row and workgroup. |y| >= 0, so the IEEE bit
  // patterns order like unsigned integers: the max is exact and order independent.
  __shared__ unsigned int tile_amax[MT * 16];
  if (ep.row_amax_out != nullptr) {
    if (tid < MT * 16) tile_amax[tid] = 0u;
    __syncthreads();
  }

  unsigned long long gf_best[MT];
  int gf_masked[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    gf_best[i] = 0ull;
    gf_masked[i] = (ARGMAX && gf_step < ep.gf.min_length + gf_plen[i]) ? ep.gf.disable_id : -1;
  }

  // Cross-wave sum (exact int32) + fused epilogue, spread over ALL threads of the workgroup.
#pragma unroll
  for (int o = 0; o < OPT; ++o) {
    const int flat = tid + o * THREADS;
    if (flat >= OUTS) continue;
    // (256 threads: output o of a thread lies in tile o - compile-time tile indices)
    const int tile = THREADS == 256 ? o : (flat >> 8), within = THREADS == 256 ? tid : (flat & 255);
    const int i = tile / NT, j = tile % NT;
    const int rt = within >> 4, ct = within & 15;
    const int row = i * 16 + rt;
    const int64_t col = n0 + j * 16 + ct;
    // MFMA C layout: element (row rt, col ct) lives in lane (rt/4)*16 + ct, register rt%4.
    const int src_lane = ((rt >> 2) << 4) | ct;
    int sum = 0, sum2 = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
      sum += partial[0][w][i][j][src_lane][rt & 3];
      if (DUAL) sum2 += partial[ND - 1][w][i][j][src_lane][rt & 3];
    }
    if (row >= M || col >= N) continue;
    const int64_t idx = static_cast<int64_t>(row) * ep.ldy + col;
    if (ep.mode == 0) {
      ep.c_i32[idx] = sum;
    } else {
      // CPU formula order (src/cpu/kernels.cc:666-673): c * (1/a_scale) / b_scale + bias.
      const float r_a = 1.f / (NORM ? norm_scales[row] : (QOL ? qol_scales[row] : e_ra[o]));
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
      if (ARGMAX) {
        // key order = (value, then LOWER column): the arg-max tie rule of the sampler (greedy_fuse.h)
        if (static_cast<int>(col) != gf_masked[i] && v == v) {
          const unsigned long long key = greedy_fuse_key(v, static_cast<unsigned>(col));
          gf_best[i] = key > gf_best[i] ? key : gf_best[i];
        }
      } else {
        ep.y[idx] = v;
      }
      if (ep.row_amax_out != nullptr) atomicMax(&tile_amax[row], __float_as_uint(fabsf(v)));
    }
  }
  if (ARGMAX) {
    static_assert(!ARGMAX || (WAVES * 64 == 256 && (MT * NT * 256) % 256 == 0), "ARGMAX: tile o == output o");
    __shared__ unsigned int gf_ticket;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      unsigned long long k = gf_best[i];
#pragma unroll
      for (int off = 8; off >= 1; off >>= 1) {
        const unsigned long long other = __shfl_xor(k, off, 64);
        k = other > k ? other : k;
      }
      const int row = i * 16 + (tid >> 4);
      if ((tid & 15) == 0 && row < M && k != 0ull) {
        // (a returning atomic: it has been performed when its value arrives, i.e. before the ticket)
        const unsigned long long old =
            atomicMax(ep.gf.slots + static_cast<int64_t>(rb + row) * GreedyFuse::kSubSlots +
                          (blockIdx.x % GreedyFuse::kSubSlots), k);
        asm volatile("" ::"v"(static_cast<unsigned>(old)));
      }
    }
    __syncthreads();
    if (tid == 0) gf_ticket = atomicAdd(ep.gf.ticket, 1u);
    __syncthreads();
    if (gf_ticket == gridDim.x * gridDim.y - 1) {
      // every workgroup of the launch has combined its rows: finish the step (greedy_fuse.h)
      greedy_fuse_finish<THREADS>(ep.gf, M_all, gf_step, tid);
    }
  }
  if (ep.row_amax_out != nullptr) {
    __syncthreads();
    if (tid < MT * 16 && tid < M)
      atomicMax(reinterpret_cast<unsigned int*>(ep.row_amax_out) + tid, tile_amax[tid]);
  }
}

// Shapes the MFMA path cannot take (K not a multiple of 16 or unaligned bases): one thread
// per output element. Only tiny fixture models hit this.
__global__ void gemm_s8_naive_kernel(const int8_t* __restrict__ A, const int8_t* __restrict__ B,
                                     int M, int N, int K, GemmS8Epilogue ep) {
  const int64_t col = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  const int64_t row = blockIdx.y;
  if (col >= N || row >= M) return;
  const int8_t* a = A + row * K;
  const int8_t* b = B + col * K;
  int acc = 0;
  for (int k = 0; k < K; ++k) acc += static_cast<int>(a[k]) * static_cast<int>(b[k]);
  const float r_a = (ep.mode == 1) ? 1.f / ep.a_scale[row] : 0.f;
  store_epilogue(ep, row, col, acc, r_a);
}

}  // namespace

// Shapes the norm-prologue form serves: decode rows (<= 128, in blocks of 16 rows per workgroup), a
// row that fits the float4 registers a lane keeps (K = 64, 128, 256 or 512: one compiled
// form per depth, no per-element bounds) and a
// projection narrow enough that redoing the norm in every workgroup is cheaper than a launch
// (N < 8192: the vocabulary projection keeps its own norm launch).
bool gemm_s8_norm_applicable(int64_t m, int64_t n, int64_t k) {
  return m >= 1 && m <= 128 && (k == 64 || k == 128 || k == 256 || k == 512) && n >= 1 && n < 8192;
}

// Shapes the fused greedy arg-max serves: the vocabulary-projection forms of the skinny kernel
// (<= 32 rows as one or two 16-row tiles per workgroup, K short of the long-K forms).
bool gemm_s8_argmax_applicable(int64_t m, int64_t n, int64_t k) {
  static const bool plain = tuning_env("CT2X_GEMM_TILE") == nullptr &&
                            tuning_env("CT2X_NO_SKINNY_GEMM") == nullptr &&
                            tuning_env("CT2X_SKINNY_MT") == nullptr &&
                            tuning_env("CT2X_SKINNY_CFG") == nullptr;
  return plain && m >= 1 && m <= 32 && n >= 8192 && k % 16 == 0 && k > 0 && k < 4096;
}

// Shapes the quantise-on-load form serves: decode rows and a K that is whole MFMA K-steps and at
// most 2048 (one float4 column of the 16-row tile per thread of an 8-wave workgroup). The
// fp32 activation is 4x the bytes of its int8 copy and every column workgroup re-reads its 16 rows
// from L2, so the form only pays where that stays small next to a launch: m * k * 4 <= 1 MiB
// (Transformer-base FFN: 32 x 2048) - the Llama shapes (16 x 14336 into 256 workgroups) keep
// the element-wise Quantize launch.
bool gemm_s8_qol_applicable(int64_t m, int64_t n, int64_t k) {
  return m >= 1 && m <= 128 && k % 64 == 0 && k >= 64 && k <= 2048 && n >= 1 && n <= 4096 &&
         m * k * 4 <= (1 << 20);
}

// Measured on Transformer-base (batch 32, same box, alternating runs). First version (every lane
// fetched and converted the 16 floats of its own fragment, 64 bytes per lane and 16 rows apart):
// 91.3k tokens/s against 94.4k with the Quantize launch - a loss. Second version (coalesced row
// segments, one conversion per tile through LDS): 97.9 - 98.2k against 97.6 - 99.7k - break-even:
// what a decode kernel costs is its chain of dependent memory round trips, and the folded form
// trades the Quantize launch's round trip for 64 conversions per thread on the consumer's critical
// path. On by default for <= 32 rows (six launches less per Transformer-base step at the same
// speed: less host-side graph work); CT2X_QUANT_ON_LOAD=0 restores the launch (bit-identical).
bool gemm_s8_qol_enabled() {
  static const bool enabled = [] {
    const char* e = tuning_env("CT2X_QUANT_ON_LOAD");
    return e == nullptr || e[0] != '0';
  }();
  return enabled;
}

// Round 1 measured a one-wave-per-row prologue at ~10 us (eight dependent row reductions per
// wave) and switched it off. The eight-rows-per-wave form runs the same arithmetic in one pass
// (see the kernel); CT2X_NORM_FUSION=0 restores the separate norm launch (bit-identical).
bool gemm_s8_norm_fusion_enabled() {
  static const bool enabled = [] {
    const char* e = tuning_env("CT2X_NORM_FUSION");
    return e == nullptr || e[0] != '0';
  }();
  return enabled;
}

// Non-temporal weight stream for the long-K decode forms (CT2X_NT_WEIGHTS=1). Measured on the
// Llama-3-8B shape, same box, alternating runs: 3.81 - 3.84 ms per decode step with `nt` against
// 3.65 - 3.66 without - the guide's gain belongs to LDS-DMA streams, register loads lose. Off.
static bool nt_weights(int64_t, int64_t) {
  static const bool on = [] {
    const char* e = tuning_env("CT2X_NT_WEIGHTS");
    return e != nullptr && e[0] == '1';
  }();
  return on;
}

void launch_gemm_s8(const int8_t* a, const int8_t* b, int64_t m, int64_t n, int64_t k,
                    const GemmS8Epilogue& ep_in, hipStream_t stream) {
  if (m == 0 || n == 0) return;
  GemmS8Epilogue ep = ep_in;
  if (ep.ldy == 0) ep.ldy = n;
  const bool mfma_ok = (k % 16 == 0) && k > 0 &&
                       ((reinterpret_cast<uintptr_t>(a) & 15) == 0) &&
                       ((reinterpret_cast<uintptr_t>(b) & 15) == 0);
  const int M = static_cast<int>(m), N = static_cast<int>(n), K = static_cast<int>(k);
  if (ep.gf.slots != nullptr &&
      !(mfma_ok && gemm_s8_argmax_applicable(m, n, k) && ep.mode == 1 && ep.norm_x == nullptr &&
        ep.a_f32 == nullptr && ep.b2 == nullptr && !ep.packed_b))
    throw Error("launch_gemm_s8: the fused arg-max form needs a plain int8 vocabulary projection of <= 32 rows");
  if (!mfma_ok) {
    hipLaunchKernelGGL(gemm_s8_naive_kernel, dim3((n + 63) / 64, m), dim3(64), 0, stream, a, b,
                       M, N, K, ep);
    return;
  }
  // Debug override of the tile choice (CT2X_GEMM_TILE=32|64|128).
  static const int forced = [] {
    const char* e = tuning_env("CT2X_GEMM_TILE");
    return e ? std::atoi(e) : 0;
  }();
  const int64_t sel = forced ? (forced <= 32 ? 1 : (forced <= 64 ? 33 : 65)) : m;
  static const bool no_skinny = tuning_env("CT2X_NO_SKINNY_GEMM") != nullptr;
  if ((ep.norm_x != nullptr &&
       gemm_s8_stream_norm_applicable(m, n, k, ep.b2 != nullptr, ep.norm_beta != nullptr)) ||
      (ep.a_f32 != nullptr && ep.b2 == nullptr && gemm_s8_stream_qol_applicable(m, n, k)) ||
      (ep.a_f32 != nullptr && ep.b2 == nullptr && ep.lead_flag != nullptr &&
       gemm_s8_stream_lead_quant_applicable(m, n, k))) {
    // the prologue forms of the persistent streaming kernel (K = 4096, <= 16 rows)
    launch_gemm_s8_stream(a, b, M, N, K, ep, stream);
    return;
  }
  if (ep.norm_x != nullptr) {
    // Norm + Quantize prologue (gemm_s8_norm_applicable): decode rows of a pre-norm Dense.
    if (!(mfma_ok && gemm_s8_norm_applicable(m, n, k) && ep.mode == 1 && ep.b2 == nullptr))
      throw Error("launch_gemm_s8: the norm-prologue form needs M <= 128, K in {64, 128, 256, 512}, N < 8192");
    const size_t lds = 16 * 8 * static_cast<size_t>(norm_chain_pad(k)) * sizeof(float) +
                       2 * k * sizeof(float) + 16 * (k + 16) + 16 * sizeof(float);
#define CT2X_NORMG(CH)                                                                            \
  do {                                                                                            \
    auto fn = gemm_s8_skinny_kernel<1, 1, 4, 4, false, CH>;                                       \
    allow_big_lds(reinterpret_cast<const void*>(fn), lds);                                        \
    hipLaunchKernelGGL(fn, dim3((n + 15) / 16, (m + 15) / 16), dim3(256), lds, stream,            \
                       reinterpret_cast<const int8_t*>(ep.norm_x), b, M, N, K,                    \
                       static_cast<const float*>(nullptr), ep);                                   \
  } while (0)
    if (k == 512) CT2X_NORMG(8);
    else if (k == 256) CT2X_NORMG(4);
    else if (k == 128) CT2X_NORMG(2);
    else CT2X_NORMG(1);
#undef CT2X_NORMG
    return;
  }
  if (ep.norm_x == nullptr && ep.a_f32 == nullptr &&
      (ep.b2 == nullptr || (reinterpret_cast<uintptr_t>(ep.b2) & 15) == 0) &&
      gemm_s8_stream_applicable(m, n, k, ep.b2 != nullptr)) {
    launch_gemm_s8_stream(a, b, M, N, K, ep, stream);
    return;
  }
  if (ep.packed_b)
    throw Error("launch_gemm_s8: a packed (fragment-major) weight only runs in the streaming form (m <= 16)");
  if (ep.a_f32 != nullptr) {
    // Quantize folded into the A-fragment fetch (gemm_s8_qol_applicable)
    if (!(gemm_s8_qol_applicable(m, n, k) && ep.mode == 1 && ep.a_amax != nullptr &&
          ep.b2 == nullptr && (reinterpret_cast<uintptr_t>(ep.a_f32) & 15) == 0 &&
          (reinterpret_cast<uintptr_t>(b) & 15) == 0))
      throw Error("launch_gemm_s8: the quantise-on-load form needs M <= 128, K % 64 == 0, the row absmax and 16-byte aligned operands");
#define CT2X_QOL(WAVES, U)                                                                        \
  do {                                                                                            \
    auto fn = gemm_s8_skinny_kernel<1, 1, WAVES, U, false, 0, true>;                              \
    const size_t lds = 64 + 16 * static_cast<size_t>(k + 16);                                     \
    allow_big_lds(reinterpret_cast<const void*>(fn), lds);                                        \
    hipLaunchKernelGGL(fn, dim3((n + 15) / 16, (m + 15) / 16), dim3(WAVES * 64), lds, stream,     \
                       reinterpret_cast<const int8_t*>(ep.a_f32), b, M, N, K, ep.a_amax, ep);     \
  } while (0)
    if (k > 1024) CT2X_QOL(8, 4);
    else CT2X_QOL(4, 4);
#undef CT2X_QOL
    return;
  }
  if (ep.b2 != nullptr) {
    // Gated pair (see gemm_s8_skinny_kernel<..., DUAL>): only the decode shapes take this form
    // (gemm_s8_dual_applicable); column tiles per workgroup are halved, both streams count.
    if (!(mfma_ok && m <= 32 && ep.mode == 1 && (reinterpret_cast<uintptr_t>(ep.b2) & 15) == 0))
      throw Error("launch_gemm_s8: the gated (dual) form needs M <= 32 and 16-byte aligned operands");
#define CT2X_DUAL(MT, NT, WAVES, U)                                                              \
  hipLaunchKernelGGL((gemm_s8_skinny_kernel<MT, NT, WAVES, U, true>),                            \
                     dim3((n + 16 * NT - 1) / (16 * NT)), dim3(WAVES * 64), 0, stream, a, b, M, N, \
                     K, static_cast<const float*>(nullptr), ep)
    if (m <= 16) {
      if (k >= 4096 && nt_weights(n, k))
        hipLaunchKernelGGL((gemm_s8_skinny_kernel<1, 2, 16, 2, true, 0, false, true>),
                           dim3((n + 31) / 32), dim3(1024), 0, stream, a, b, M, N, K,
                           static_cast<const float*>(nullptr), ep);
      else if (k >= 4096) CT2X_DUAL(1, 2, 16, 2);
      else if (k >= 2048) CT2X_DUAL(1, 1, 8, 4);
      else CT2X_DUAL(1, 1, 4, 4);
    } else {
      if (k >= 4096) CT2X_DUAL(2, 2, 8, 2);
      else if (k >= 2048) CT2X_DUAL(2, 1, 8, 2);
      else CT2X_DUAL(2, 1, 4, 2);
    }
#undef CT2X_DUAL
    return;
  }
  // 65 .. 128 rows against a wide, shallow weight (the vocabulary projection of beam search,
  // [128, 32000] x 512): the 128 x 128 tile kernel reads the weight once and runs 17.1 us where the
  // skinny form takes 24.0 (four row tiles per workgroup; 28.2 with one) - since its epilogue decides
  // "no activation" once per launch (profiles/r06_proj128_row_tiles.txt). 64 rows: skinny 14.3 / tile 16.0.
  const bool tile_wins = m > 64 && n >= 8192 && k <= 1024 && ep.mode == 1 && ep.gf.slots == nullptr &&
                         ep.b2 == nullptr && ep.a_f32 == nullptr && ep.norm_x == nullptr && !ep.packed_b;
  if (m <= 128 && !forced && !no_skinny && !tile_wins) {
    // Decode-step shapes (rows = batch * beam): one memory round trip per launch
    // (see gemm_s8_skinny_kernel). MT = row tiles of 16, NT = column tiles per workgroup.
#define CT2X_SKINNY(MT, NT, WAVES, U)                                                          \
  hipLaunchKernelGGL((gemm_s8_skinny_kernel<MT, NT, WAVES, U>),                                \
                     dim3((n + 16 * NT - 1) / (16 * NT), (m + 16 * MT - 1) / (16 * MT)),       \
                     dim3(WAVES * 64), 0, stream, a, b, M, N, K, static_cast<const float*>(nullptr), ep)
    // Rows beyond 16 are split into row blocks (grid.y) of `mt` 16-row tiles; one tile per block
    // measured best at every decode shape of Transformer-base ([32|64|128, 512] x 512: 2.7 us
    // against 3.2 / 4.5 / 7.0 us with 2 / 4 / 8 tiles per workgroup; tools/kernel_bench.cpp).
    static const int mt_env = [] {
      const char* e = tuning_env("CT2X_SKINNY_MT");
      const int v = e ? std::atoi(e) : 0;
      return (v == 1 || v == 2 || v == 4 || v == 8) ? v : 0;
    }();
    int mt = 1;
    if (m > 16) {
      // (the vocabulary projection streams 16 MB of weights: read them once per 32 / 64 rows, not
      // once per 16 - [128, 32000] x 512 fused 28.2 -> 24.0 us with four row tiles per workgroup,
      // [64, 32000] 15.7 -> 14.4 with two; profiles/r06_proj128_row_tiles.txt)
      mt = mt_env ? mt_env : (n >= 8192 ? (m <= 64 ? 2 : 4) : 1);
      while (mt > 1 && (mt / 2) * 16 >= m) mt /= 2;  // never more tiles than rows
    }
    // Tuning override (tools/gemm_bench.cpp): CT2X_SKINNY_CFG="NT,WAVES,U" for m <= 32.
    static const int cfg = [] {
      const char* e = tuning_env("CT2X_SKINNY_CFG");
      int nt = 0, wv = 0, u = 0;
      if (e && std::sscanf(e, "%d,%d,%d", &nt, &wv, &u) == 3) return nt * 10000 + wv * 100 + u;
      return 0;
    }();
    if (cfg != 0 && mt <= 2) {
#define CT2X_TRY(NT, WAVES, U)                                         \
  if (cfg == NT * 10000 + WAVES * 100 + U) {                           \
    if (mt == 1) CT2X_SKINNY(1, NT, WAVES, U);                         \
    else CT2X_SKINNY(2, NT, WAVES, U);                                 \
    return;                                                            \
  }
      CT2X_TRY(1, 4, 4) CT2X_TRY(1, 4, 8) CT2X_TRY(1, 8, 4) CT2X_TRY(1, 8, 8) CT2X_TRY(1, 16, 4)
      CT2X_TRY(2, 4, 4) CT2X_TRY(2, 4, 8) CT2X_TRY(2, 8, 2) CT2X_TRY(2, 8, 4) CT2X_TRY(2, 8, 8)
      CT2X_TRY(2, 16, 2) CT2X_TRY(2, 16, 4)
      CT2X_TRY(4, 4, 2) CT2X_TRY(4, 4, 4) CT2X_TRY(4, 8, 2) CT2X_TRY(4, 8, 4) CT2X_TRY(4, 16, 2)
#undef CT2X_TRY
    }
    // Shapes measured with tools/skinny_sweep.sh (weights rotated through 600 MB, MI355X):
    // long weight streams want 16 waves per workgroup (more bytes in flight per CU); the
    // Transformer-base shapes (K = 512 / 2048) keep the 4-8 wave forms.
    if (mt == 1 && k >= 4096 && nt_weights(n, k)) {
#define CT2X_SKINNY_NT(NT, WAVES, U)                                                            \
  hipLaunchKernelGGL((gemm_s8_skinny_kernel<1, NT, WAVES, U, false, 0, false, true>),            \
                     dim3((n + 16 * NT - 1) / (16 * NT), (m + 15) / 16), dim3(WAVES * 64), 0,    \
                     stream, a, b, M, N, K, static_cast<const float*>(nullptr), ep)
      if (n >= 8192) CT2X_SKINNY_NT(4, 16, 2);
      else if (k >= 8192) CT2X_SKINNY_NT(1, 16, 4);
      else if (n >= 6144) CT2X_SKINNY_NT(2, 8, 8);
      else CT2X_SKINNY_NT(1, 16, 4);
#undef CT2X_SKINNY_NT
    } else if (mt == 1 && k >= 4096) {
      if (n >= 8192) CT2X_SKINNY(1, 4, 16, 2);        // [16, 14336, 4096] 19.2 -> 15.5 us
      else if (k >= 8192) CT2X_SKINNY(1, 1, 16, 4);   // [16, 4096, 14336] 20.5 -> 18.3 us
      else if (n >= 6144) CT2X_SKINNY(1, 2, 8, 8);    // [16, 6144, 4096]  10.4 ->  9.3 us
      else CT2X_SKINNY(1, 1, 16, 4);                  // [16, 4096, 4096]   7.6 ->  6.7 us
    } else if (n >= 8192 && ep.gf.slots != nullptr) {
      if (!(gemm_s8_argmax_applicable(m, n, k) && ep.mode == 1 && mt <= 2))
        throw Error("launch_gemm_s8: the fused arg-max form needs a vocabulary projection of <= 32 rows");
#define CT2X_SKINNY_ARGMAX(MT)                                                                      \
  hipLaunchKernelGGL((gemm_s8_skinny_kernel<MT, 4, 4, 2, false, 0, false, false, true>),           \
                     dim3((n + 63) / 64, (m + 16 * MT - 1) / (16 * MT)), dim3(256), 0, stream, a,   \
                     b, M, N, K, static_cast<const float*>(nullptr), ep)
      if (mt == 1) CT2X_SKINNY_ARGMAX(1);
      else CT2X_SKINNY_ARGMAX(2);
#undef CT2X_SKINNY_ARGMAX
    } else if (n >= 8192) {
      if (mt == 1) CT2X_SKINNY(1, 4, 4, 2);
      else if (mt == 2) CT2X_SKINNY(2, 4, 4, 2);
      else if (mt == 4) CT2X_SKINNY(4, 2, 4, 2);
      else CT2X_SKINNY(8, 2, 4, 1);
    } else if (k >= 2048) {
      if (mt == 1) CT2X_SKINNY(1, 1, 8, 4);
      else if (mt == 2) CT2X_SKINNY(2, 1, 8, 4);
      else if (mt == 4) CT2X_SKINNY(4, 1, 8, 2);
      else CT2X_SKINNY(8, 1, 8, 2);
    } else {
      if (mt == 1) CT2X_SKINNY(1, 1, 4, 4);
      else if (mt == 2) CT2X_SKINNY(2, 1, 4, 4);
      else if (mt == 4) CT2X_SKINNY(4, 1, 4, 2);
      else CT2X_SKINNY(8, 1, 4, 2);
    }
#undef CT2X_SKINNY
    return;
  }
  static const bool no_big = tuning_env("CT2X_NO_BIG_GEMM") != nullptr;
  if (!forced && !no_big && (reinterpret_cast<uintptr_t>(a) & 15) == 0 &&
      gemm_s8_big_applicable(m, n, k)) {
    launch_gemm_s8_big(a, b, M, N, K, ep, stream);
    return;
  }
  // Tile choice: the largest tile that still gives every CU work (>= 192 workgroups on the 256
  // CUs); these launches are microseconds long, so parallelism beats per-tile reuse.
  int tile = 0;
  if (forced) {
    tile = sel <= 32 ? 32 : (sel <= 64 ? 64 : 128);
  } else {
    const int64_t wg128 = ((m + 127) / 128) * ((n + 127) / 128);
    const int64_t wg64 = ((m + 63) / 64) * ((n + 63) / 64);
    tile = wg128 >= 192 ? 128 : (wg64 >= 192 ? 64 : 32);
  }
  if (tile == 32) {
    hipLaunchKernelGGL((gemm_s8_mfma_kernel<32, 64, 1, 4>), dim3((n + 63) / 64, (m + 31) / 32),
                       dim3(256), 0, stream, a, b, M, N, K, ep);
  } else if (tile == 64) {
    hipLaunchKernelGGL((gemm_s8_mfma_kernel<64, 64, 2, 2>), dim3((n + 63) / 64, (m + 63) / 64),
                       dim3(256), 0, stream, a, b, M, N, K, ep);
  } else {
    hipLaunchKernelGGL((gemm_s8_mfma_kernel<128, 128, 2, 2>),
                       dim3((n + 127) / 128, (m + 127) / 128), dim3(256), 0, stream, a, b, M, N,
                       K, ep);
  }
}

}  // namespace ct2x
