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

#include "device_utils.h"
#include "kernels.h"
#include "runtime.h"

#include <algorithm>
#include <cstdlib>

namespace ct2x {

namespace {

template <int NT, int WAVES, int KS, int P, bool DUAL, bool PACKED>
__global__ __launch_bounds__(WAVES * 64) void gemm_s8_stream_kernel(
    const int8_t* __restrict__ A, const int8_t* __restrict__ B, int M, int N, int K,
    int slabs_per_wg, int total_slabs, const int8_t* __restrict__ B2, GemmS8Epilogue ep) {
  // (A .. B2 are preloaded into SGPRs; `ep` costs a kernarg round trip and is first touched after
  // the activation fragments and the first ring of weight requests have left. PACKED - the weight
  // is the fragment-major copy - is a template parameter: with both address forms live the long-K
  // ring spilled 15 registers.)
  static_assert(KS % P == 0, "the ring slot of K-step i is i % P in every slab");
  constexpr int ND = DUAL ? 2 : 1;
  constexpr int THREADS = WAVES * 64;
  constexpr int OUTS = NT * 256;
  constexpr int OPT = (OUTS + THREADS - 1) / THREADS;
  // dynamic LDS (up to 128 KB: opted into by the launcher): the double-buffered reduction tiles
  // [2][ND][WAVES][NT][64][4] int32, then the row-absmax words [2][16]
  extern __shared__ __attribute__((aligned(16))) int stream_lds[];
  typedef int PartialT[ND][WAVES][NT][64][4];
  PartialT* partial = reinterpret_cast<PartialT*>(stream_lds);
  unsigned int (*tile_amax)[16] =
      reinterpret_cast<unsigned int (*)[16]>(stream_lds + 2 * ND * WAVES * NT * 256);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;

  // activation fragments of this wave's K-steps: resident for the whole launch
  v4i af[KS];
#pragma unroll
  for (int i = 0; i < KS; ++i) {
    const int kk = (wave + i * WAVES) * 64 + g * 16;
    af[i] = (r < M) ? *reinterpret_cast<const v4i*>(A + static_cast<int64_t>(r) * K + kk)
                    : v4i{0, 0, 0, 0};
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
#pragma unroll
  for (int i = 0; i < P; ++i) request(first, i);

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
        e_ra[o] = ep.a_scale[row];
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
        const float r_a = 1.f / e_ra[o];
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

void launch_gemm_s8_stream(const int8_t* a, const int8_t* b, int M, int N, int K,
                           const GemmS8Epilogue& ep_in, hipStream_t stream) {
  const bool dual = ep_in.b2 != nullptr;
  StreamPlan plan;
  if (!plan_stream(N, K, dual, device_cus(), &plan))
    throw Error("launch_gemm_s8_stream: shape outside the streaming form");
  const GemmS8Epilogue& ep = ep_in;
#define CT2X_STREAM_LAUNCH(NT, KS, P, DUAL, PACKED)                                               \
  do {                                                                                            \
    auto fn = gemm_s8_stream_kernel<NT, 16, KS, P, DUAL, PACKED>;                                 \
    const size_t lds = sizeof(int) * 2 * (DUAL ? 2 : 1) * 16 * NT * 256 + 2 * 16 * sizeof(int);   \
    allow_big_lds(reinterpret_cast<const void*>(fn), lds);                                        \
    hipLaunchKernelGGL(fn, dim3(plan.grid), dim3(1024), lds, stream, a, b, M, N, K,               \
                       plan.slabs_per_wg, plan.slabs, ep.b2, ep);                                 \
  } while (0)
#define CT2X_STREAM(NT, KS, P, DUAL)                                                              \
  if (plan.nt == NT && plan.ks == KS && plan.p == P && dual == DUAL) {                            \
    if (ep.packed_b) CT2X_STREAM_LAUNCH(NT, KS, P, DUAL, true);                                   \
    else CT2X_STREAM_LAUNCH(NT, KS, P, DUAL, false);                                              \
    return;                                                                                       \
  }
  // (forms whose register ring spills at 128 VGPRs - 16 waves per workgroup - are not compiled:
  // plan_stream never returns them)
  CT2X_STREAM(4, 4, 2, false) CT2X_STREAM(2, 4, 4, false) CT2X_STREAM(1, 4, 4, false)
  CT2X_STREAM(2, 8, 4, false) CT2X_STREAM(1, 8, 4, false)
  CT2X_STREAM(1, 14, 7, false)
  CT2X_STREAM(2, 4, 2, true) CT2X_STREAM(1, 4, 4, true)
  CT2X_STREAM(2, 8, 2, true) CT2X_STREAM(1, 8, 4, true)
#undef CT2X_STREAM
#undef CT2X_STREAM_LAUNCH
  throw Error("launch_gemm_s8_stream: no compiled form for this plan");
}

}  // namespace ct2x
