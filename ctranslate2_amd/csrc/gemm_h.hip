// float16 / bfloat16 projection GEMM for gfx950: the half-precision arm of ops::Gemm
// (src/ops/gemm.cc:45-107 -> primitives<CUDA>::gemm<float16_t / bfloat16_t>,
// src/cuda/primitives.cu:519-569: cublasGemmEx, fp32 accumulate for bf16 and, unless
// CT2_CUDA_TRUE_FP16_GEMM, for fp16) with apply_bias_and_activation + residual
// (src/ops/gemm.cc:11-25, src/ops/bias_add_gpu.cu:46-102) fused in.
//
//   Y[m,n] = act( A[m,k] . W[n,k]^T + bias[n] ) + residual[m,n]
//
// W is stored in the 16-bit type (half the weight stream of an fp32 model); A arrives in fp32 and
// is rounded to the 16-bit type when it enters the matrix pipe - exactly the rounding the
// reference applies to its 16-bit activations - and every product is accumulated in fp32 by
// v_mfma_f32_16x16x32_{f16,bf16}. Everything outside the GEMM (norms, softmax, attention,
// residual stream) stays in fp32, i.e. is at least as accurate as the reference's 16-bit pipeline.
//
// Two kernels, as for int8 (gemm_s8.hip): a skinny split-K form for decode rows (M <= 128: a
// weight stream, one memory round trip) and an LDS-tiled 128x128x64 form for encoder / prefill.

#include "device_utils.h"
#include "grid_sync.h"
#include "kernels.h"
#include "runtime.h"

#include <cstdlib>

namespace ct2x {

namespace {

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef __bf16 v8b __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <bool kBF16>
struct HalfOps;
template <>
struct HalfOps<false> {
  using vec = v8h;
  __device__ static __forceinline__ v4f mfma(vec a, vec b, v4f c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  __device__ static __forceinline__ vec from_f32(const float4& lo, const float4& hi) {
    vec v;
    v[0] = static_cast<_Float16>(lo.x); v[1] = static_cast<_Float16>(lo.y);
    v[2] = static_cast<_Float16>(lo.z); v[3] = static_cast<_Float16>(lo.w);
    v[4] = static_cast<_Float16>(hi.x); v[5] = static_cast<_Float16>(hi.y);
    v[6] = static_cast<_Float16>(hi.z); v[7] = static_cast<_Float16>(hi.w);
    return v;
  }
};
template <>
struct HalfOps<true> {
  using vec = v8b;
  __device__ static __forceinline__ v4f mfma(vec a, vec b, v4f c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  __device__ static __forceinline__ vec from_f32(const float4& lo, const float4& hi) {
    vec v;
    v[0] = static_cast<__bf16>(lo.x); v[1] = static_cast<__bf16>(lo.y);
    v[2] = static_cast<__bf16>(lo.z); v[3] = static_cast<__bf16>(lo.w);
    v[4] = static_cast<__bf16>(hi.x); v[5] = static_cast<__bf16>(hi.y);
    v[6] = static_cast<__bf16>(hi.z); v[7] = static_cast<__bf16>(hi.w);
    return v;
  }
};

// float16 only: the fp32 activation enters the matrix pipe as TWO float16 values, a = hi + lo with
// hi = RN_f16(a), lo = RN_f16(a - hi) (22 significant bits), and every weight fragment is
// multiplied by both - the weights are the model's own float16 values and the products accumulate
// in fp32, so the layer is a float32 evaluation of the stored model up to 2^-22 relative: the
// north-star's "fp16 logits within 1e-3 of the CPU reference" holds with three orders of magnitude
// to spare instead of by 20 % (1.2e-3 measured at the NLLB geometry with one rounding to 2^-11).
// Decode rows are bandwidth bound (the second MFMA is free); sequence-mode rows pay 2x MFMAs.
__device__ __forceinline__ void split_f16(const float4& lo4, const float4& hi4, v8h& hi, v8h& lo) {
  const float a[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    hi[e] = static_cast<_Float16>(a[e]);
    lo[e] = static_cast<_Float16>(a[e] - static_cast<float>(hi[e]));
  }
}

struct HalfEpilogue {
  const float* bias;
  const float* residual;
  float* y;
  int act;
};

__device__ __forceinline__ float finish(float v, float bias, bool has_bias, int act, float res,
                                        bool has_res) {
  if (has_bias) v = v + bias;
  v = apply_activation(v, act);
  if (has_res) v = res + v;
  return v;
}

// ---- skinny split-K form (M <= 16 * MT) -------------------------------------------------------
// A workgroup owns NT*16 output columns and all rows; wave w takes the 32-element K-steps
// w, w + WAVES, ...; A fragments are read as fp32 (2 x 16 bytes per lane) and converted, W
// fragments are 16 bytes per lane straight from memory; fp32 partial tiles are summed through LDS
// in wave order (deterministic).
// Split K over workgroups (gridDim.y > 1): slice z = blockIdx.y covers K elements
// [z * k_slice, (z + 1) * k_slice) and stores its fp32 partial tile to partial[z][M][N];
// gemm_h_splitk_reduce_kernel adds the slices in z order and applies the epilogue. A 40-row decode
// step of the Whisper-large-v3 shape (N = K = 1280) is otherwise 80 workgroups whose waves walk
// their K-steps through five dependent memory round trips (25.6 us per launch = 0.27 TB/s,
// profiles/r03_whisper_kernel_stats_a.csv).
template <bool kBF16, int MT, int NT, int WAVES, int U>
__global__ __launch_bounds__(WAVES * 64) void gemm_h_skinny_kernel(const float* __restrict__ A,
                                                                   const uint16_t* __restrict__ W,
                                                                   int M, int N, int K,
                                                                   HalfEpilogue ep, int k_slice,
                                                                   float* __restrict__ partial_out) {
  using Ops = HalfOps<kBF16>;
  using vec = typename Ops::vec;
  __shared__ float partial[WAVES][MT][NT][64][4];
  constexpr int THREADS = WAVES * 64;
  constexpr int OUTS = MT * NT * 256;
  constexpr int OPT = (OUTS + THREADS - 1) / THREADS;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int64_t n0 = static_cast<int64_t>(blockIdx.x) * (NT * 16);
  // this workgroup's K range, in 32-element steps (k_slice is a multiple of 32)
  const int ks_begin = static_cast<int>(blockIdx.y) * (k_slice / 32);
  const int ksteps = min((K + 31) / 32, ks_begin + k_slice / 32);

  float e_bias[OPT], e_res[OPT];
#pragma unroll
  for (int o = 0; o < OPT; ++o) {
    const int flat = tid + o * THREADS;
    const int tile = flat >> 8, within = flat & 255;
    const int row = (tile / NT) * 16 + (within >> 4);
    const int64_t col = n0 + (tile % NT) * 16 + (within & 15);
    e_bias[o] = e_res[o] = 0.f;
    if (partial_out == nullptr && flat < OUTS && row < M && col < N) {
      if (ep.bias) e_bias[o] = ep.bias[col];
      if (ep.residual) e_res[o] = ep.residual[static_cast<int64_t>(row) * N + col];
    }
  }

  v4f acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};

  for (int ks0 = ks_begin + wave; ks0 < ksteps; ks0 += WAVES * U) {
    float4 alo[U][MT], ahi[U][MT];
    vec bf[U][NT];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ks = ks0 + u * WAVES;
      const int kk = ks * 32 + g * 8;
      const bool k_ok = (ks < ksteps) && (kk < K);  // K % 8 == 0 is a launch precondition
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int row = i * 16 + r;
        alo[u][i] = ahi[u][i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k_ok && row < M) {
          const float4* p = reinterpret_cast<const float4*>(A + static_cast<int64_t>(row) * K + kk);
          alo[u][i] = p[0];
          ahi[u][i] = p[1];
        }
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int64_t col = n0 + j * 16 + r;
        vec z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = 0;
        bf[u][j] = (k_ok && col < N) ? *reinterpret_cast<const vec*>(W + col * K + kk) : z;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        if constexpr (!kBF16) {
          v8h a_hi, a_lo;
          split_f16(alo[u][i], ahi[u][i], a_hi, a_lo);
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            acc[i][j] = Ops::mfma(a_lo, bf[u][j], acc[i][j]);
            acc[i][j] = Ops::mfma(a_hi, bf[u][j], acc[i][j]);
          }
        } else {
          const vec af = Ops::from_f32(alo[u][i], ahi[u][i]);
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = Ops::mfma(af, bf[u][j], acc[i][j]);
        }
      }
  }

#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
      *reinterpret_cast<v4f*>(&partial[wave][i][j][lane][0]) = acc[i][j];
  __syncthreads();

#pragma unroll
  for (int o = 0; o < OPT; ++o) {
    const int flat = tid + o * THREADS;
    if (flat >= OUTS) continue;
    const int tile = flat >> 8, within = flat & 255;
    const int i = tile / NT, j = tile % NT;
    const int rt = within >> 4, ct = within & 15;
    const int row = i * 16 + rt;
    const int64_t col = n0 + j * 16 + ct;
    const int src_lane = ((rt >> 2) << 4) | ct;  // MFMA C layout: col = lane & 15, row = 4*(lane>>4)+reg
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) sum += partial[w][i][j][src_lane][rt & 3];
    if (row >= M || col >= N) continue;
    if (partial_out != nullptr)
      partial_out[(static_cast<int64_t>(blockIdx.y) * M + row) * N + col] = sum;
    else
      ep.y[static_cast<int64_t>(row) * N + col] =
          finish(sum, e_bias[o], ep.bias != nullptr, ep.act, e_res[o], ep.residual != nullptr);
  }
}

// ---- row-staged form (decode rows, M <= 64) -----------------------------------------------------
// The skinny form above re-reads the whole fp32 activation from L2 for every 16 output columns
// (Whisper-large-v3 decode, 40 rows: 16 MB of activation reads for a 3.3 MB weight) and sums its
// waves' K-step partials through LDS. Here a workgroup owns 64 output columns x one K slice of at
// most 512 elements: the activation slice is converted to the 16-bit type ONCE on its way into
// LDS (float16: hi and lo halves, see split_f16), every wave owns 16 columns for the whole slice -
// all its weight fragments (16 x 16 bytes per lane) are requested before the first LDS access, so
// the launch is one memory round trip deep - and no cross-wave reduction is needed. K slices
// (gridDim.y) go through partial[z][M][N] and gemm_h_splitk_reduce_kernel as above.
//
// Round 5: at the Whisper-large-v3 decode step (40 rows, d = 1280) a launch of the form above was
// 60 - 240 workgroups of 64 KB of weights each, staged its activation slice through ten dependent
// load rounds, and was followed by a second launch that summed the K slices: 11 + 5 us per Dense
// for 3 - 13 MB of weights (0.3 - 0.8 TB/s; six of them per layer = more than half of the step,
// profiles/r05_kernel_stats_bench_whisper_attn16.csv). Now
//   * the K slice is a template parameter (256 when 512 leaves the chip half empty) and the
//     activation staging is fully unrolled: every load of the slice is in flight at once, next to
//     the weight fragments - one memory round trip before the MFMAs;
//   * the slices are summed IN the launch: a workgroup publishes its [M, 64] partial tile
//     (write-through stores into a region only it writes), takes a ticket of its column tile, and
//     the last of the column tile's workgroups to arrive adds the slices in z order - the same
//     sums as the separate reduce launch, deterministic - and applies the epilogue
//     (grid_sync.h: nobody waits, so no residency requirement; the tickets end at zero).
//     The partial tiles are read with plain 16-byte loads: every line has exactly one writer, was
//     written through to memory before its writer's ticket, and cannot be in the reader's L2
//     (invalidated at the kernel boundary, untouched by this launch before the tickets are complete).
// TUNING builds: shader-clock stamps of workgroup (0, 0) and of column tile 0's last arrival
// (tools/gemm_rows_probe.py)
#ifdef CT2X_TUNING
__device__ unsigned long long g_rows_stamps[16];
#define CT2X_ROWS_STAMP(i)                                                                        \
  do {                                                                                            \
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_rows_stamps[i] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#define CT2X_ROWS_STAMP_LAST(i)                                                                   \
  do {                                                                                            \
    if (blockIdx.x == 0 && threadIdx.x == 0) g_rows_stamps[i] = __builtin_amdgcn_s_memtime();     \
  } while (0)
#else
#define CT2X_ROWS_STAMP(i)
#define CT2X_ROWS_STAMP_LAST(i)
#endif
constexpr int kRowsKSlice = 512;  // K elements per workgroup (the unfused form; the fused one: 256 or 512)
constexpr int kRowsMaxFusedSlices = 20;  // 3 x 20 partial loads in flight per thread of the last arrival
template <bool kBF16, int MT, int KSL, bool kFused>
__global__ __launch_bounds__(256) void gemm_h_rows_kernel(const float* __restrict__ A,
                                                          const uint16_t* __restrict__ W, int M,
                                                          int N, int K, HalfEpilogue ep,
                                                          float* __restrict__ partial_out,
                                                          unsigned* __restrict__ tickets) {
  using Ops = HalfOps<kBF16>;
  using vec = typename Ops::vec;
  extern __shared__ __attribute__((aligned(16))) uint8_t rows_lds[];
  constexpr int ROWS = MT * 16;
  constexpr int KSTEPS = KSL / 32;
  constexpr int STRIDE = KSL * 2 + 16;  // bytes per LDS row: rows 4 banks apart
  uint8_t* a_hi = rows_lds;
  uint8_t* a_lo = rows_lds + ROWS * STRIDE;  // float16 only
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int64_t n0 = static_cast<int64_t>(blockIdx.x) * 64 + wave * 16;
  const int k0 = static_cast<int>(blockIdx.y) * KSL;
  const int kn = min(KSL, K - k0);  // a multiple of 8 (K % 8 == 0)

  CT2X_ROWS_STAMP(0);
  // weight fragments of this wave's 16 columns over the whole slice: requested first
  vec wf[KSTEPS];
  {
    const int64_t col = n0 + r;
    const uint16_t* wrow = W + col * K + k0;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int kk = ks * 32 + g * 8;
      vec z;
#pragma unroll
      for (int e = 0; e < 8; ++e) z[e] = 0;
      wf[ks] = (kk < kn && col < N) ? *reinterpret_cast<const vec*>(wrow + kk) : z;
    }
  }
  // activation slice -> LDS in the 16-bit type (rows >= M and columns >= kn are zero); the loads of
  // a batch of (up to) eight pieces per thread are all requested before the first conversion
  constexpr int CHUNKS = KSL / 8;                 // 8-element pieces per row
  constexpr int ITEMS = ROWS * CHUNKS / 256;      // pieces per thread
  constexpr int BATCH = ITEMS <= 8 ? ITEMS : (ITEMS % 8 == 0 ? 8 : (ITEMS % 6 == 0 ? 6 : 4));
  static_assert(ITEMS % BATCH == 0, "staging batches");
#pragma unroll
  for (int it0 = 0; it0 < ITEMS; it0 += BATCH) {
    float4 x0[BATCH], x1[BATCH];
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const int c = tid + (it0 + u) * 256;
      const int row = c / CHUNKS, kk = (c % CHUNKS) * 8;
      x0[u] = x1[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < M && kk < kn) {
        const float4* p = reinterpret_cast<const float4*>(A + static_cast<int64_t>(row) * K + k0 + kk);
        x0[u] = p[0];
        x1[u] = p[1];
      }
    }
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const int c = tid + (it0 + u) * 256;
      const int row = c / CHUNKS, kk = (c % CHUNKS) * 8;
      if constexpr (!kBF16) {
        v8h hi, lo;
        split_f16(x0[u], x1[u], hi, lo);
        *reinterpret_cast<v8h*>(a_hi + row * STRIDE + kk * 2) = hi;
        *reinterpret_cast<v8h*>(a_lo + row * STRIDE + kk * 2) = lo;
      } else {
        *reinterpret_cast<vec*>(a_hi + row * STRIDE + kk * 2) = Ops::from_f32(x0[u], x1[u]);
      }
    }
  }
  CT2X_ROWS_STAMP(1);
  __syncthreads();
  CT2X_ROWS_STAMP(2);

  v4f acc[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) acc[i] = v4f{0.f, 0.f, 0.f, 0.f};
  // (steps beyond the slice's end multiply zeros - both operands were zero-filled - so the loop is
  // branch-free; the activation fragments of step ks + 1 are read from LDS before the MFMAs of
  // step ks: read one after the other behind their uses they were 2.2k of the launch's shader clocks)
  vec fa[2][MT], fl[2][MT];
  auto read_step = [&](int ks, int buf) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int off = (i * 16 + r) * STRIDE + (ks * 32 + g * 8) * 2;
      fa[buf][i] = *reinterpret_cast<const vec*>(a_hi + off);
      if constexpr (!kBF16) fl[buf][i] = *reinterpret_cast<const vec*>(a_lo + off);
    }
  };
  read_step(0, 0);
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
    if (ks + 1 < KSTEPS) read_step(ks + 1, (ks + 1) & 1);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      if constexpr (!kBF16) acc[i] = Ops::mfma(fl[ks & 1][i], wf[ks], acc[i]);
      acc[i] = Ops::mfma(fa[ks & 1][i], wf[ks], acc[i]);
    }
  }
  // MFMA C layout: column = lane & 15, row = 4 * (lane >> 4) + register
  const int64_t col = n0 + r;
  CT2X_ROWS_STAMP(3);
  if constexpr (kFused) {
    // (N % 64 == 0 here: every column of the tile exists)
    // partial tile of (slice z, column tile x): [M][64] floats in a region of its own
    const int slices = static_cast<int>(gridDim.y);
    const int64_t tile_floats = static_cast<int64_t>(M) * 64;
    float* mine = partial_out + (static_cast<int64_t>(blockIdx.x) * slices + blockIdx.y) * tile_floats;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = i * 16 + g * 4 + e;
        if (row < M) grid_publish(mine + row * 64 + wave * 16 + r, acc[i][e]);
      }
    // the epilogue's operands of the pieces this thread would finish do not depend on the other
    // slices: requested BEFORE the ticket, they arrive during its round trip
    const int quads = M * 16;  // float4 pieces of the [M][64] tile; thread t takes t, t + 256, ..
    constexpr int QIT = MT;    // ROWS * 16 / 256 pieces per thread
    v4f bv[QIT], rv[QIT];
#pragma unroll
    for (int i = 0; i < QIT; ++i) {
      const int q = min(tid + i * 256, quads - 1);
      const int64_t c0 = static_cast<int64_t>(blockIdx.x) * 64 + (q & 15) * 4;
      bv[i] = rv[i] = v4f{0.f, 0.f, 0.f, 0.f};
      if (ep.bias) bv[i] = *reinterpret_cast<const v4f*>(ep.bias + c0);
      if (ep.residual) rv[i] = *reinterpret_cast<const v4f*>(ep.residual + static_cast<int64_t>(q >> 4) * N + c0);
    }
    CT2X_ROWS_STAMP(4);
    const bool last_here = grid_last_arrival(tickets + blockIdx.x, static_cast<unsigned>(slices));
    CT2X_ROWS_STAMP(5);
    if (!last_here) return;
    CT2X_ROWS_STAMP_LAST(6);
    asm volatile("" ::: "memory");
    // all partial tiles of a batch of slices in flight at once (the first form walked its pieces one
    // after the other, two dependent round trips each: 12 - 22k shader clocks of a 14.5 us launch,
    // profiles/r05_gemm_rows_probe.txt). Loads are unconditional (clamped indices), so the batch is
    // straight-line code behind ONE wait; the sums keep the z order.
    const float* tile = partial_out + static_cast<int64_t>(blockIdx.x) * slices * tile_floats;
    constexpr int ZB = MT == 1 ? 20 : (MT <= 3 ? 10 : 8);
    v4f v[QIT];
#pragma unroll
    for (int i = 0; i < QIT; ++i) v[i] = v4f{0.f, 0.f, 0.f, 0.f};
    for (int z0 = 0; z0 < slices; z0 += ZB) {
      v4f sl[QIT][ZB];
#pragma unroll
      for (int zz = 0; zz < ZB; ++zz) {
        const int z = min(z0 + zz, slices - 1);
#pragma unroll
        for (int i = 0; i < QIT; ++i) {
          const int q = min(tid + i * 256, quads - 1);
          sl[i][zz] = *reinterpret_cast<const v4f*>(tile + z * tile_floats + q * 4);
        }
      }
#pragma unroll
      for (int zz = 0; zz < ZB; ++zz)
        if (z0 + zz < slices) {
#pragma unroll
          for (int i = 0; i < QIT; ++i) v[i] = (z0 + zz == 0) ? sl[i][zz] : v[i] + sl[i][zz];
        }
    }
#pragma unroll
    for (int i = 0; i < QIT; ++i) {
      const int q = tid + i * 256;
      if (q < quads) {
        const int64_t c0 = static_cast<int64_t>(blockIdx.x) * 64 + (q & 15) * 4;
        const int64_t idx = static_cast<int64_t>(q >> 4) * N + c0;
        v4f o;
#pragma unroll
        for (int c = 0; c < 4; ++c)
          o[c] = finish(v[i][c], bv[i][c], ep.bias != nullptr, ep.act, rv[i][c], ep.residual != nullptr);
        *reinterpret_cast<v4f*>(ep.y + idx) = o;
      }
    }
    CT2X_ROWS_STAMP_LAST(7);
    if (tid == 0) grid_publish(tickets + blockIdx.x, 0u);
    return;
  }
  if (col >= N) return;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = i * 16 + g * 4 + e;
      if (row >= M) continue;
      const int64_t idx = static_cast<int64_t>(row) * N + col;
      if (partial_out != nullptr)
        partial_out[static_cast<int64_t>(blockIdx.y) * M * N + idx] = acc[i][e];
      else
        ep.y[idx] = finish(acc[i][e], ep.bias ? ep.bias[col] : 0.f, ep.bias != nullptr, ep.act,
                           ep.residual ? ep.residual[idx] : 0.f, ep.residual != nullptr);
    }
}

// ---- LDS-tiled form (encoder / prefill rows) ------------------------------------------------
// 128 x 128 output tile, 64 K elements (128 bytes of the 16-bit type) per stage, 4 waves as
// 2 x 2, each 64 x 64 = 4 x 4 MFMA tiles. Rows are padded to 144 bytes in LDS (conflict-free
// ds_read_b128 fragments); the A tile is converted fp32 -> 16 bit on its way into LDS; the next
// stage's global loads are in flight during the MFMAs.
// Split K (gridDim.z > 1): slice z covers K elements [z * k_slice, (z + 1) * k_slice) and writes its
// fp32 partial tile to partial[z][M][N]; gemm_h_splitk_reduce_kernel adds the slices in z order
// (deterministic) and applies the epilogue. Without it a [256 x 2048] x [2048 x 2048]^T launch is 32
// workgroups on a 256-CU chip (the NLLB decode step at batch 64 x beam 4: 22.9 ms per step,
// profiles/r03_nllb_bench_a.json).
template <bool kBF16>
__global__ __launch_bounds__(256) void gemm_h_tiled_kernel(const float* __restrict__ A,
                                                           const uint16_t* __restrict__ W, int M,
                                                           int N, int K, HalfEpilogue ep,
                                                           int k_slice, float* __restrict__ partial) {
  using Ops = HalfOps<kBF16>;
  using vec = typename Ops::vec;
  constexpr int BM = 128, BN = 128, BK = 64, STRIDE = BK * 2 + 16;  // bytes per LDS row
  // (float16: a third tile holds the low halves of the split activation, see split_f16)
  __shared__ __attribute__((aligned(16))) uint8_t lds[(BM + BN + (kBF16 ? 0 : BM)) * STRIDE];
  uint8_t* As = lds;
  uint8_t* Bs = lds + BM * STRIDE;
  uint8_t* Al = lds + (BM + BN) * STRIDE;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 15, g = lane >> 4;
  const int64_t m0 = static_cast<int64_t>(blockIdx.y) * BM;
  const int64_t n0 = static_cast<int64_t>(blockIdx.x) * BN;
  // staging map: 8 threads per row (8 elements = 16 bytes of the 16-bit type each), 32 rows per pass
  const int ld_row = tid >> 3, ld_c = tid & 7;

  v4f acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};

  float4 ra_lo[4], ra_hi[4];
  vec rb[4];
  auto load_tile = [&](int k0) {
    const int kk = k0 + ld_c * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t row = m0 + ld_row + i * 32;
      ra_lo[i] = ra_hi[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < M && kk < K) {
        const float4* p = reinterpret_cast<const float4*>(A + row * K + kk);
        ra_lo[i] = p[0];
        ra_hi[i] = p[1];
      }
      const int64_t col = n0 + ld_row + i * 32;
      vec z;
#pragma unroll
      for (int e = 0; e < 8; ++e) z[e] = 0;
      rb[i] = (col < N && kk < K) ? *reinterpret_cast<const vec*>(W + col * K + kk) : z;
    }
  };

  const int k_begin = blockIdx.z * k_slice;
  const int k_end = min(K, k_begin + k_slice);
  load_tile(k_begin);
  for (int k0 = k_begin; k0 < k_end; k0 += BK) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (!kBF16) {
        v8h a_hi, a_lo;
        split_f16(ra_lo[i], ra_hi[i], a_hi, a_lo);
        *reinterpret_cast<v8h*>(As + (ld_row + i * 32) * STRIDE + ld_c * 16) = a_hi;
        *reinterpret_cast<v8h*>(Al + (ld_row + i * 32) * STRIDE + ld_c * 16) = a_lo;
      } else {
        *reinterpret_cast<vec*>(As + (ld_row + i * 32) * STRIDE + ld_c * 16) =
            Ops::from_f32(ra_lo[i], ra_hi[i]);
      }
      *reinterpret_cast<vec*>(Bs + (ld_row + i * 32) * STRIDE + ld_c * 16) = rb[i];
    }
    __syncthreads();
    if (k0 + BK < k_end) load_tile(k0 + BK);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      vec af[4], bfr[4];
      const int koff = ks * 64 + g * 16;  // bytes: 32 elements per MFMA K-step, 8 per lane group
#pragma unroll
      for (int i = 0; i < 4; ++i)
        af[i] = *reinterpret_cast<const vec*>(As + (wm * 64 + i * 16 + r) * STRIDE + koff);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        bfr[j] = *reinterpret_cast<const vec*>(Bs + (wn * 64 + j * 16 + r) * STRIDE + koff);
      if constexpr (!kBF16) {
        vec al[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          al[i] = *reinterpret_cast<const vec*>(Al + (wm * 64 + i * 16 + r) * STRIDE + koff);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = Ops::mfma(al[i], bfr[j], acc[i][j]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = Ops::mfma(af[i], bfr[j], acc[i][j]);
    }
  }

  if (partial != nullptr) {  // split K: raw partial sums, the reduce kernel finishes
    float* pz = partial + static_cast<int64_t>(blockIdx.z) * M * N;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int64_t row = m0 + wm * 64 + i * 16 + g * 4 + e;
        if (row >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int64_t col = n0 + wn * 64 + j * 16 + r;
          if (col < N) pz[row * N + col] = acc[i][j][e];
        }
      }
    return;
  }
  // Epilogue: operands first, then stores (see gemm_s8.hip).
  float bias[4];
  bool col_ok[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t col = n0 + wn * 64 + j * 16 + r;
    col_ok[j] = col < N;
    bias[j] = (ep.bias && col_ok[j]) ? ep.bias[col] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float res[4][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t row = m0 + wm * 64 + i * 16 + g * 4 + e;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t col = n0 + wn * 64 + j * 16 + r;
        res[e][j] = (ep.residual && row < M && col_ok[j]) ? ep.residual[row * N + col] : 0.f;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t row = m0 + wm * 64 + i * 16 + g * 4 + e;
      if (row >= M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t col = n0 + wn * 64 + j * 16 + r;
        if (col_ok[j])
          ep.y[row * N + col] = finish(acc[i][j][e], bias[j], ep.bias != nullptr, ep.act, res[e][j],
                                       ep.residual != nullptr);
      }
    }
  }
}

// ---- 129 .. 256 rows (a beam-search decode step of 64 x 4 hypotheses, round 6) -----------------
// The LDS-tiled form above at 256 rows was 128 x 128 tiles whose weight tile is fetched by two
// workgroups, one stage of register prefetch (a stage computes for ~1 us, a first-touch weight
// line takes longer) and two barriers per stage: 29 us + a 5 us slice sum per Dense of the
// NLLB-3.3B geometry, 0.7 TB/s. Here a workgroup of eight waves owns ALL rows x 128 columns x one
// K slice: every weight byte is fetched once, stages are 32 K elements (one MFMA k-step), the
// operands of stage s + 2 and s + 3 are in flight in registers while stage s is multiplied, LDS is
// double-buffered with ONE barrier per stage, and both operand tiles live in LDS in MFMA fragment
// order (1 KB per 16 rows x 32 K elements, a lane's 16 bytes at position r * 4 + (g ^ ((r >> 1) &
// 3)): conflict-free for the coalesced staging writes AND the fragment reads).
// float16: hi + lo split of the activation as in the other forms (split_f16), lo first.
constexpr int R256_BN = 128, R256_BK = 32;
__device__ __forceinline__ float4 as_float4(const v4f& v) { return make_float4(v.x, v.y, v.z, v.w); }
// ROWS = 128 (second half of round 6): the same stage machinery on 128 rows x 128 columns - waves
// 2 x 4 instead of 4 x 2, three requests per stage instead of five, 48 KB of LDS (24 KB bfloat16)
// so that two or three workgroups share a CU. Why: a stage is LDS traffic (~1100 clocks), 32 matrix
// instructions per wave (2 x 512 clocks on a SIMD) and the float16 split's VALU IN LOCKSTEP behind
// one barrier - nothing overlaps inside a workgroup, and the 256-row form puts only
// column tiles x K slices workgroups on the chip (64 at [256, 2048] x 2048: three of the six Dense
// layers of an NLLB decoder layer ran on a quarter of the CUs, 31 us for 8 MB of weights). Half the
// rows per workgroup = twice the workgroups: idle CUs fill, and where the chip was already full two
// workgroups per CU interleave their phases. The weight tile is fetched by two workgroups (the
// second read comes from the memory-side cache).
template <bool kBF16, int ROWS>
__global__ __launch_bounds__(512) void gemm_h_rows256_kernel(const float* __restrict__ A,
                                                             const uint16_t* __restrict__ W, int M, int N, int K,
                                                             HalfEpilogue ep, int k_slice,
                                                             float* __restrict__ partial) {
  using Ops = HalfOps<kBF16>;
  using vec = typename Ops::vec;
  static_assert(ROWS == 256 || ROWS == 128, "row tiles of 256 or 128");
  constexpr int WN = ROWS == 256 ? 2 : 4;  // waves along the columns; 8 / WN along the rows (64 rows each)
  constexpr int NJ = 8 / WN;               // 16-column tiles of a wave
  constexpr int A_BYTES = ROWS * 64, W_BYTES = 8 * 1024;
  constexpr int STAGE = (kBF16 ? 1 : 2) * A_BYTES + W_BYTES;
  extern __shared__ __attribute__((aligned(16))) uint8_t r256_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = blockIdx.y * ROWS;
  const int64_t n0 = static_cast<int64_t>(blockIdx.x) * R256_BN;
  const int k_begin = blockIdx.z * k_slice;
  const int k_end = min(K, k_begin + k_slice);
  const int stages = (k_end - k_begin + R256_BK - 1) / R256_BK;
  // staging map: four consecutive threads cover the 32 K elements of a row (8 each)
  const int s_g = tid & 3, s_row = tid >> 2;  // rows s_row and s_row + 128 of A, column s_row of W
  const int s_pos = ((s_row & 15) * 4 + (s_g ^ (((s_row & 15) >> 1) & 3))) * 16;
  const int sa_off0 = (s_row >> 4) * 1024 + s_pos, sa_off1 = ((s_row + 128) >> 4) * 1024 + s_pos;
  const bool a_ok0 = m0 + s_row < M, a_ok1 = ROWS == 256 && m0 + s_row + 128 < M;
  const bool w_ok = n0 + s_row < N;
  const float* a_src0 = A + static_cast<int64_t>(a_ok0 ? m0 + s_row : 0) * K + s_g * 8;
  const float* a_src1 = A + static_cast<int64_t>(a_ok1 ? m0 + s_row + 128 : 0) * K + s_g * 8;
  const uint16_t* w_src = W + (w_ok ? n0 + s_row : 0) * K + s_g * 8;
  // fragment read map
  const int r = lane & 15, g = lane >> 4;
  const int f_pos = (r * 4 + (g ^ ((r >> 1) & 3))) * 16;

  struct Regs {
    v4f a[4];
    vec w;
  };
  // The requests are inline assembly and so is the wait for them: the compiler's own counter
  // bookkeeping gave up at the loop's back edge and waited for EVERYTHING in flight at the top of
  // every stage (vmcnt(0): a prefetch distance of one stage instead of two, and with predicated loads
  // - the first build - of none: 8.4 instead of 7.3 ms per NLLB step). Five requests per stage, always
  // (clamped addresses: rows past M and columns past N compute on the clamped row's data - MFMA
  // rows / columns are independent and those outputs are never stored; stages past the slice
  // re-request its last stage and are never multiplied), nothing else touches vector memory inside
  // the loop, so "all but the newest five have landed" is s_waitcnt vmcnt(5).
  auto load = [&](Regs& R, int st) {
    const int kk = k_begin + min(st, stages - 1) * R256_BK;
    const float* p0 = a_src0 + kk;
    const float* p1 = a_src1 + kk;
    const uint16_t* pw = w_src + kk;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(R.a[0]) : "v"(p0) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(R.a[1]) : "v"(p0) : "memory");
    if constexpr (ROWS == 256) {
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(R.a[2]) : "v"(p1) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(R.a[3]) : "v"(p1) : "memory");
    }
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(R.w) : "v"(pw) : "memory");
  };
  // the requests before the newest `newer` loads have landed (the registers are operands so that no
  // use of them can be scheduled in front of the wait)
  auto landed5 = [&](Regs& R) {  // (three requests per stage in the 128-row form)
    if constexpr (ROWS == 256)
      asm volatile("s_waitcnt vmcnt(5)" : "+v"(R.a[0]), "+v"(R.a[1]), "+v"(R.a[2]), "+v"(R.a[3]), "+v"(R.w)::"memory");
    else
      asm volatile("s_waitcnt vmcnt(3)" : "+v"(R.a[0]), "+v"(R.a[1]), "+v"(R.w)::"memory");
  };
  auto stash = [&](const Regs& R, int buf) {
    uint8_t* base = r256_lds + buf * STAGE;
    if constexpr (!kBF16) {
      v8h h0, l0;
      split_f16(as_float4(R.a[0]), as_float4(R.a[1]), h0, l0);
      *reinterpret_cast<v8h*>(base + sa_off0) = h0;
      *reinterpret_cast<v8h*>(base + A_BYTES + sa_off0) = l0;
      if constexpr (ROWS == 256) {
        v8h h1, l1;
        split_f16(as_float4(R.a[2]), as_float4(R.a[3]), h1, l1);
        *reinterpret_cast<v8h*>(base + sa_off1) = h1;
        *reinterpret_cast<v8h*>(base + A_BYTES + sa_off1) = l1;
      }
    } else {
      *reinterpret_cast<vec*>(base + sa_off0) = Ops::from_f32(as_float4(R.a[0]), as_float4(R.a[1]));
      if constexpr (ROWS == 256)
        *reinterpret_cast<vec*>(base + sa_off1) = Ops::from_f32(as_float4(R.a[2]), as_float4(R.a[3]));
    }
    *reinterpret_cast<vec*>(base + (kBF16 ? 1 : 2) * A_BYTES + (s_row >> 4) * 1024 + s_pos) = R.w;
  };

  v4f acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
  auto multiply = [&](int buf) {
    const uint8_t* base = r256_lds + buf * STAGE;
    vec af[4], bfr[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      bfr[j] = *reinterpret_cast<const vec*>(base + (kBF16 ? 1 : 2) * A_BYTES + (wn * NJ + j) * 1024 + f_pos);
    if constexpr (!kBF16) {
      vec al[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        al[i] = *reinterpret_cast<const vec*>(base + A_BYTES + (wm * 4 + i) * 1024 + f_pos);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = Ops::mfma(al[i], bfr[j], acc[i][j]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const vec*>(base + (wm * 4 + i) * 1024 + f_pos);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = Ops::mfma(af[i], bfr[j], acc[i][j]);
  };

  // prologue: stages 0, 1, 2 requested; stage 0 parked in LDS
  Regs Ra, Rb;
  load(Ra, 0);
  load(Rb, 1);
  landed5(Ra);
  stash(Ra, 0);
  load(Ra, 2);
  // stage s is multiplied from buffer s & 1 while stage s + 1 is parked (its registers were
  // requested two stages ago) and stage s + 3 is requested into the registers that just emptied
  for (int s = 0; s < stages; s += 2) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    landed5(Rb);
    stash(Rb, 1);
    load(Rb, s + 3);
    multiply(0);
    if (s + 1 >= stages) break;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    landed5(Ra);
    stash(Ra, 0);
    load(Ra, s + 4);
    multiply(1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (requests past the slice: their registers are dead only now)

  if (partial != nullptr) {  // split K: raw partial sums, the reduce kernel finishes
    float* pz = partial + static_cast<int64_t>(blockIdx.z) * M * N;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int64_t row = m0 + (wm * 4 + i) * 16 + g * 4 + e;
        if (row >= M) continue;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int64_t col = n0 + (wn * NJ + j) * 16 + r;
          if (col < N) pz[row * N + col] = acc[i][j][e];
        }
      }
    return;
  }
  // Epilogue: operands first, then stores (as in the 128 x 128 form).
  float bias[NJ];
  bool col_ok[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int64_t col = n0 + (wn * NJ + j) * 16 + r;
    col_ok[j] = col < N;
    bias[j] = (ep.bias && col_ok[j]) ? ep.bias[col] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float res[4][NJ];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t row = m0 + (wm * 4 + i) * 16 + g * 4 + e;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int64_t col = n0 + (wn * NJ + j) * 16 + r;
        res[e][j] = (ep.residual && row < M && col_ok[j]) ? ep.residual[row * N + col] : 0.f;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t row = m0 + (wm * 4 + i) * 16 + g * 4 + e;
      if (row >= M) continue;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int64_t col = n0 + (wn * NJ + j) * 16 + r;
        if (col_ok[j])
          ep.y[row * N + col] = finish(acc[i][j][e], bias[j], ep.bias != nullptr, ep.act, res[e][j],
                                       ep.residual != nullptr);
      }
    }
  }
}

__global__ void gemm_h_splitk_reduce_kernel(const float* __restrict__ partial, int splits, int64_t n,
                                           int64_t total, HalfEpilogue ep) {
  const int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (e >= total) return;
  float v = partial[e];
  for (int z = 1; z < splits; ++z) v = v + partial[z * total + e];  // fixed order: deterministic
  const int64_t col = e % n;
  ep.y[e] = finish(v, ep.bias ? ep.bias[col] : 0.f, ep.bias != nullptr, ep.act,
                   ep.residual ? ep.residual[e] : 0.f, ep.residual != nullptr);
}

// The same sums, four columns per thread (n % 4 == 0, 16-byte aligned operands): every slice is
// requested before the first add - the scalar form above is 5.9 us per use at [256, 2048] x 8 slices
// (144 uses per NLLB decode step: 11 % of the step), bound by its dependent 4-byte loads.
template <int SPLITS>
__global__ void gemm_h_splitk_reduce4_kernel(const float* __restrict__ partial, int64_t n,
                                            int64_t total4, HalfEpilogue ep) {
  const int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (e >= total4) return;
  const v4f* p = reinterpret_cast<const v4f*>(partial) + e;
  v4f s[SPLITS];
#pragma unroll
  for (int z = 0; z < SPLITS; ++z) s[z] = p[z * total4];
  const int64_t col = (e * 4) % n;
  v4f b = v4f{0.f, 0.f, 0.f, 0.f}, r = v4f{0.f, 0.f, 0.f, 0.f};
  if (ep.bias) b = *reinterpret_cast<const v4f*>(ep.bias + col);
  if (ep.residual) r = reinterpret_cast<const v4f*>(ep.residual)[e];
  v4f v = s[0];
#pragma unroll
  for (int z = 1; z < SPLITS; ++z) v = v + s[z];  // fixed order: deterministic, same sums as above
  v4f o;
#pragma unroll
  for (int c = 0; c < 4; ++c)
    o[c] = finish(v[c], b[c], ep.bias != nullptr, ep.act, r[c], ep.residual != nullptr);
  reinterpret_cast<v4f*>(ep.y)[e] = o;
}

static void launch_splitk_reduce(const float* partial, int splits, int64_t n, int64_t total,
                                 const HalfEpilogue& ep, hipStream_t stream) {
  const bool vec = n % 4 == 0 &&
                   ((reinterpret_cast<uintptr_t>(partial) | reinterpret_cast<uintptr_t>(ep.y) |
                     reinterpret_cast<uintptr_t>(ep.bias) | reinterpret_cast<uintptr_t>(ep.residual)) & 15) == 0;
  const int64_t total4 = total / 4;
  const dim3 grid4(static_cast<unsigned>((total4 + 255) / 256));
#define CT2X_RED4(SP)                                                                             \
  hipLaunchKernelGGL(gemm_h_splitk_reduce4_kernel<SP>, grid4, dim3(256), 0, stream, partial, n, total4, ep)
  if (vec && splits == 2) CT2X_RED4(2);
  else if (vec && splits == 3) CT2X_RED4(3);
  else if (vec && splits == 4) CT2X_RED4(4);
  else if (vec && splits == 5) CT2X_RED4(5);
  else if (vec && splits == 8) CT2X_RED4(8);
  else if (vec && splits == 10) CT2X_RED4(10);
  else if (vec && splits == 16) CT2X_RED4(16);
  else
    hipLaunchKernelGGL(gemm_h_splitk_reduce_kernel, dim3(static_cast<unsigned>((total + 255) / 256)),
                       dim3(256), 0, stream, partial, splits, n, total, ep);
#undef CT2X_RED4
}

template <bool kBF16>
void launch_typed(const float* a, const uint16_t* w, int M, int N, int K, const HalfEpilogue& ep,
                  hipStream_t stream, float* splitk_ws, size_t splitk_floats, unsigned* tickets,
                  size_t n_tickets) {
#define CT2X_HSKINNY(MT, NT, WAVES, U)                                                       \
  hipLaunchKernelGGL((gemm_h_skinny_kernel<kBF16, MT, NT, WAVES, U>),                        \
                     dim3((N + 16 * NT - 1) / (16 * NT), sk), dim3(WAVES * 64), 0, stream, a, w, M, \
                     N, K, ep, sk > 1 ? K / sk : ((K + 31) / 32) * 32,                             \
                     sk > 1 ? splitk_ws : static_cast<float*>(nullptr))
  if (M <= 128) {
    const int mt = M <= 16 ? 1 : (M <= 32 ? 2 : (M <= 64 ? 4 : 8));
    const bool wide = N >= 8192;
    // K slices over workgroups: enough of them to cover the chip twice, at most two rounds of
    // K-steps per wave (8 waves x U = 2 x 32 elements = 512 per round), whole 32-element steps
    static const int variant = [] {
      const char* e = tuning_env("CT2X_HSKINNY_VARIANT");  // (make TUNING=1 builds only)
      return e ? std::atoi(e) : 2;
    }();
    if (variant >= 2 && M <= 64 && K >= 256) {
      // row-staged form: 64 columns x one K slice per workgroup
      const int mt3 = (M > 32 && M <= 48) ? 3 : mt;
      auto aligned16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
      // fused K-slice sum: whole 64-column tiles, vector epilogue operands, a ticket per column tile
      // (256 when 512 leaves the chip half empty - but never more than 10 slices: the last arrival
      // reads them in one batch, and 20 tickets on one counter were 6k shader clocks of waiting)
      const int ksl = ((N / 64) * ((K + 511) / 512) >= 256 || K % 256 != 0 || K > 2560) ? 512 : 256;
      const int fslices = (K + ksl - 1) / ksl;
      const bool fused = fslices > 1 && fslices <= kRowsMaxFusedSlices && tickets != nullptr && N % 64 == 0 &&
                         static_cast<size_t>(N / 64) <= n_tickets && splitk_ws != nullptr &&
                         static_cast<size_t>(fslices) * M * N <= splitk_floats && aligned16(ep.y) &&
                         aligned16(ep.bias) && aligned16(ep.residual) && aligned16(splitk_ws);
      if (fused) {
        const dim3 grid(N / 64, fslices);
#define CT2X_HROWS_F(MT, KSL)                                                                    \
  do {                                                                                           \
    const size_t lds = static_cast<size_t>(MT) * 16 * (KSL * 2 + 16) * (kBF16 ? 1 : 2);           \
    allow_big_lds(reinterpret_cast<const void*>(gemm_h_rows_kernel<kBF16, MT, KSL, true>), lds); \
    hipLaunchKernelGGL((gemm_h_rows_kernel<kBF16, MT, KSL, true>), grid, dim3(256), lds, stream,  \
                       a, w, M, N, K, ep, splitk_ws, tickets);                                   \
  } while (0)
        if (ksl == 256) {
          if (mt3 == 1) CT2X_HROWS_F(1, 256);
          else if (mt3 == 2) CT2X_HROWS_F(2, 256);
          else if (mt3 == 3) CT2X_HROWS_F(3, 256);
          else CT2X_HROWS_F(4, 256);
        } else {
          if (mt3 == 1) CT2X_HROWS_F(1, 512);
          else if (mt3 == 2) CT2X_HROWS_F(2, 512);
          else if (mt3 == 3) CT2X_HROWS_F(3, 512);
          else CT2X_HROWS_F(4, 512);
        }
#undef CT2X_HROWS_F
        return;
      }
      const int slices = (K + kRowsKSlice - 1) / kRowsKSlice;
      if (slices == 1 || (splitk_ws != nullptr &&
                          static_cast<size_t>(slices) * M * N <= splitk_floats)) {
        float* part = slices > 1 ? splitk_ws : nullptr;
        const dim3 grid((N + 63) / 64, slices);
#define CT2X_HROWS(MT)                                                                          \
  do {                                                                                          \
    const size_t lds = static_cast<size_t>(MT) * 16 * (kRowsKSlice * 2 + 16) * (kBF16 ? 1 : 2);  \
    allow_big_lds(reinterpret_cast<const void*>(gemm_h_rows_kernel<kBF16, MT, kRowsKSlice, false>), lds); \
    hipLaunchKernelGGL((gemm_h_rows_kernel<kBF16, MT, kRowsKSlice, false>), grid, dim3(256), lds, stream, a, w, M, N, \
                       K, ep, part, static_cast<unsigned*>(nullptr));                           \
  } while (0)
        if (mt == 1) CT2X_HROWS(1);
        else if (mt == 2) CT2X_HROWS(2);
        else CT2X_HROWS(4);
#undef CT2X_HROWS
        if (slices > 1) launch_splitk_reduce(splitk_ws, slices, N, static_cast<int64_t>(M) * N, ep, stream);
        return;
      }
    }
    int sk = 1;
    if (variant != 0 && splitk_ws != nullptr && !wide) {
      const int col_tiles = (N + 15) / 16;
      while (sk < 8 && col_tiles * sk < 512 && K / (sk * 2) >= 512 && (K / (sk * 2)) % 32 == 0 &&
             static_cast<size_t>(sk) * 2 * M * N <= splitk_floats)
        sk *= 2;
    }
    if (variant == 0) {
      if (mt == 1) { if (wide) CT2X_HSKINNY(1, 4, 4, 2); else CT2X_HSKINNY(1, 1, 4, 4); }
      else if (mt == 2) { if (wide) CT2X_HSKINNY(2, 4, 4, 2); else CT2X_HSKINNY(2, 1, 4, 4); }
      else if (mt == 4) { if (wide) CT2X_HSKINNY(4, 2, 4, 2); else CT2X_HSKINNY(4, 1, 4, 2); }
      else { if (wide) CT2X_HSKINNY(8, 2, 4, 1); else CT2X_HSKINNY(8, 1, 4, 2); }
    } else {
      if (mt == 1) { if (wide) CT2X_HSKINNY(1, 4, 4, 2); else CT2X_HSKINNY(1, 1, 8, 4); }
      else if (mt == 2) { if (wide) CT2X_HSKINNY(2, 4, 4, 2); else CT2X_HSKINNY(2, 1, 8, 4); }
      else if (mt == 4) { if (wide) CT2X_HSKINNY(4, 2, 4, 2); else CT2X_HSKINNY(4, 1, 8, 2); }
      else { if (wide) CT2X_HSKINNY(8, 2, 4, 1); else CT2X_HSKINNY(8, 1, 4, 2); }
    }
    if (sk > 1) launch_splitk_reduce(splitk_ws, sk, N, static_cast<int64_t>(M) * N, ep, stream);
    return;
  }
#undef CT2X_HSKINNY
  // 129 .. 256 rows: row tile x 128 columns x a K slice per workgroup (gemm_h_rows256_kernel).
  // CT2X_H_ROWS256: 0 = the 128 x 128 tiles below, 1 = all rows per workgroup, 2 (default) = 128-row
  // tiles; CT2X_H_ROWS_TARGET = workgroups a launch should reach before it stops slicing K (A/B runs).
  static const int rows_mode = [] {
    const char* e = std::getenv("CT2X_H_ROWS256");
    return e == nullptr ? 2 : std::atoi(e);
  }();
  static const int rows_target = [] {
    const char* e = std::getenv("CT2X_H_ROWS_TARGET");
    return e == nullptr ? 256 : std::atoi(e);
  }();
  if (rows_mode != 0 && M > 128 && M <= 256 && K % 32 == 0 && K >= 256 &&
      ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(w)) & 15) == 0) {
    const int col_tiles = (N + R256_BN - 1) / R256_BN;
    const int row_tiles = rows_mode == 1 ? 1 : (M + 127) / 128;
    const int target = rows_mode == 1 ? 192 : rows_target;
    int splits = 1;
    if (splitk_ws != nullptr)  // at least eight stages per workgroup
      while (col_tiles * row_tiles * splits < target && splits < 16 && K / (splits * 2) >= 256 &&
             (K / (splits * 2)) % 32 == 0 &&
             static_cast<size_t>(splits) * 2 * M * N <= splitk_floats)  // (splits * 2 slices after the doubling)
        splits *= 2;
    float* part = splits > 1 ? splitk_ws : static_cast<float*>(nullptr);
    if (row_tiles == 1) {
      const size_t lds = 2 * ((kBF16 ? 1 : 2) * 16 * 1024 + 8 * 1024);
      allow_big_lds(reinterpret_cast<const void*>(gemm_h_rows256_kernel<kBF16, 256>), lds);
      hipLaunchKernelGGL((gemm_h_rows256_kernel<kBF16, 256>), dim3(col_tiles, 1, splits), dim3(512), lds, stream,
                         a, w, M, N, K, ep, K / splits, part);
    } else {
      const size_t lds = 2 * ((kBF16 ? 1 : 2) * 8 * 1024 + 8 * 1024);
      hipLaunchKernelGGL((gemm_h_rows256_kernel<kBF16, 128>), dim3(col_tiles, row_tiles, splits), dim3(512), lds,
                         stream, a, w, M, N, K, ep, K / splits, part);
    }
    if (splits > 1) launch_splitk_reduce(splitk_ws, splits, N, static_cast<int64_t>(M) * N, ep, stream);
    return;
  }
  // More than 256 rows (sequence mode: the Whisper encoder's 12000): the same 128-row stage machinery,
  // no K slices - two workgroups per CU interleave their phases where the 128 x 128 form below keeps
  // one stage of register prefetch and two barriers per stage. CT2X_H_ROWS128_BIG=0 keeps that form.
  static const bool rows128_big = [] {
    const char* e = std::getenv("CT2X_H_ROWS128_BIG");
    return e == nullptr || e[0] != '0';
  }();
  if (rows128_big && M > 256 && K % 32 == 0 && K >= 256 && (M + 127) / 128 <= 65535 &&
      static_cast<int64_t>((N + 127) / 128) * ((M + 127) / 128) >= 256 &&
      ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(w)) & 15) == 0) {
    const size_t lds = 2 * ((kBF16 ? 1 : 2) * 8 * 1024 + 8 * 1024);
    hipLaunchKernelGGL((gemm_h_rows256_kernel<kBF16, 128>), dim3((N + 127) / 128, (M + 127) / 128, 1), dim3(512),
                       lds, stream, a, w, M, N, K, ep, K, static_cast<float*>(nullptr));
    return;
  }
  const int tiles = ((N + 127) / 128) * ((M + 127) / 128);
  int splits = 1;
  if (splitk_ws != nullptr) {
    // enough workgroups for the chip, slices of whole 64-element stages, at least 4 stages each
    while (tiles * splits < 256 && splits < 16 && K / (splits * 2) >= 256 && (K / (splits * 2)) % 64 == 0 &&
           static_cast<size_t>(splits) * 2 * M * N <= splitk_floats)
      splits *= 2;
  }
  if (splits == 1) {
    hipLaunchKernelGGL((gemm_h_tiled_kernel<kBF16>), dim3((N + 127) / 128, (M + 127) / 128, 1), dim3(256),
                       0, stream, a, w, M, N, K, ep, K, static_cast<float*>(nullptr));
    return;
  }
  hipLaunchKernelGGL((gemm_h_tiled_kernel<kBF16>), dim3((N + 127) / 128, (M + 127) / 128, splits),
                     dim3(256), 0, stream, a, w, M, N, K, ep, K / splits, splitk_ws);
  launch_splitk_reduce(splitk_ws, splits, N, static_cast<int64_t>(M) * N, ep, stream);
}

}  // namespace

#ifdef CT2X_TUNING
void gemm_h_rows_read_stamps(unsigned long long* host16) {
  CT2X_HIP_CHECK(hipDeviceSynchronize());
  CT2X_HIP_CHECK(hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_rows_stamps), sizeof(unsigned long long) * 16));
}
#endif

void launch_gemm_h(const float* a, const void* w, bool bf16, int64_t m, int64_t n, int64_t k,
                   const float* bias, const float* residual, int act, float* y,
                   hipStream_t stream, float* splitk_ws, size_t splitk_floats, unsigned* tickets,
                   size_t n_tickets) {
  if (m == 0 || n == 0) return;
  if (k % 8 != 0 || (reinterpret_cast<uintptr_t>(a) & 15) || (reinterpret_cast<uintptr_t>(w) & 15))
    throw Error("16-bit GEMM: K must be a multiple of 8 and the operands 16-byte aligned");
  HalfEpilogue ep{bias, residual, y, act};
  const auto* wp = static_cast<const uint16_t*>(w);
  if (bf16)
    launch_typed<true>(a, wp, static_cast<int>(m), static_cast<int>(n), static_cast<int>(k), ep, stream,
                       splitk_ws, splitk_floats, tickets, n_tickets);
  else
    launch_typed<false>(a, wp, static_cast<int>(m), static_cast<int>(n), static_cast<int>(k), ep, stream,
                        splitk_ws, splitk_floats, tickets, n_tickets);
}

}  // namespace ct2x
