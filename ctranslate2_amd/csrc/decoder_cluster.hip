// Cluster-resident decoder stack: ALL layers of one greedy decode step of an int8 encoder-decoder
// Transformer (TransformerDecoder::operator() one step, src/layers/transformer.cc:621-871, the
// layer body :196-366) as ONE launch, for gfx950.
//
// Why: at decode rows the step was 48 dependent launches of 5 - 8 us each, none of them bound by
// bytes or flops: a launch boundary costs ~1.7 us plus the kernel's own chain of
// dependent memory round trips. Measured this round (profiles/r05_row_resident_probe.txt): eight
// workgroups hand a 256-byte line to each other through memory-side (sc1) atomics in 1.2 - 1.4 us,
// and one CU pulls ~120 GB/s through its L2 whoever else reads the same lines. So:
//
//   * a batch ROW is owned by a CLUSTER of eight workgroups (one per attention head) for the whole
//     step; member h computes head h's slice of every projection (an int8 GEMV over its own 1/8 of
//     the weight rows: v_dot4_i32_i8, the row-wide MFMA tile would be 31/32 empty), runs head h's
//     self- and encoder-decoder attention itself (QKV -> attention needs no hand-over at all), and
//     every member redoes the row-wide LayerNorm / Quantize statistics for itself (redundant work
//     is free here, a hand-over is not);
//   * the members meet SIX times per layer (attention context x 2, residual stream x 3, the
//     feed-forward inner row), never across clusters: no device-wide barrier exists in the launch;
//   * the weights are re-packed once per model so that member h's slice of a Dense is contiguous
//     and a wave's 64 lanes read 1 KB lines (chunk-major: 16 K-bytes of every row of the slice,
//     then the next 16); blockIdx % 8 = member = the XCD the dispatcher deals the workgroup to, so
//     an XCD's L2 only ever holds slice h of every weight (2.7 MB for Transformer-base);
//   * the requests for the NEXT projection's weights leave before the member waits for its peers.
//
// Arithmetic: the int32 sums are exact in any order; every fp32 formula (Dequantize epilogue,
// LayerNorm statistics, Quantize, attention) is the one the launch-per-op path runs (gemm_s8.hip,
// row_norm.h, attention_exact_core.h) - the results are the same bits, which
// tests/test_decoder_cluster_gpu.py checks against that path and the reference goldens.
//
// Hand-over rules (grid_sync.h): data published with agent-scope atomic stores, read with
// agent-scope atomic loads, arrival = one relaxed atomicAdd after the stores have drained, bounded
// polls - a lost peer raises the error flag, it never hangs the device.

#include <atomic>
#include <cstdlib>

#include "attention_exact_core.h"
#include "device_utils.h"
#include "greedy_fuse.h"
#include "grid_sync.h"
#include "kernels.h"
#include "row_norm.h"
#include "runtime.h"

namespace ct2x {

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int kMembers = kClusterMembers;  // workgroups per row = attention heads
constexpr int kThreads = 512;
constexpr int kD = 512;      // d_model
constexpr int kDH = 64;      // d_head
constexpr int kFfn = 2048;   // feed-forward inner size
constexpr int kXKS = kDH + 4;
constexpr unsigned kSpinLimit = 1u << 21;
#ifndef CT2X_PACE_PARTS
#define CT2X_PACE_PARTS 4
#endif
constexpr int kPaceParts = CT2X_PACE_PARTS;
static_assert(8 % kPaceParts == 0, "the projection tail's eight DMA requests per wave leave in kPaceParts parts");  // parts a 16-chunk weight request is issued in between a meeting's polls

// exchange areas of a row (floats): attention context, residual stream, feed-forward inner row
constexpr int kXchCtx = 0, kXchX = kD, kXchInner = 2 * kD, kXchFloats = 2 * kD + kFfn;

// projection tail (ClusterProjArgs, kernels.h)
constexpr int kProjRowSlots = kClusterProjRowSlots;
constexpr int kProjCols = kClusterProjCols;
constexpr int kProjWBytes = kProjCols * kD;   // 64 KB: the weight rows of a pass, fragment-major
constexpr int kProjAStride = kD + 16;         // the quantised rows of all clusters [32][528]
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

struct Shared {
  float x_row[kD];                 // residual stream of the row
  float stage[8 * 68 + 8];         // the row parked for the statistics chains (norm_chain_pad(512) = 68)
  alignas(16) int8_t xq[kFfn];     // the quantised input row of the next GEMV
  int part[kThreads];              // GEMV partial sums [kparts][rows of the slice]
  float outv[256];                 // GEMV outputs of the slice (q | k | v, or the inner columns)
  float red[16];
  float tq[kDH];
  float ks[64 * kXKS];             // keys 0 .. 63 of the (row, head), staged by the whole workgroup
  float vs[64 * kDH];              // values likewise
  int dead;                        // a poll ran out: stop waiting, results are void
};

// Workgroup barrier for LDS hand-overs ONLY: waits for this wave's LDS traffic, not for its global
// loads. __syncthreads() carries a workgroup-scope fence that drains vmcnt as well - and with it
// the weight requests of the NEXT phase that are deliberately left in flight across the barriers of
// this one (the workgroups of a launch never exchange global data except through Handover, whose
// atomics carry their own ordering).
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// A per-lane index behind an opaque move: the 64-bit address built from it can not be hoisted out of
// the layer loop (hoisted addresses were the values the allocator spilled, and a spill RELOAD is a
// vmcnt(0) wait that drains the weight requests deliberately left in flight).
__device__ __forceinline__ int opaque_index(int v) {
  asm volatile("" : "+v"(v));
  return v;
}
// ... and a wave-uniform one (stays in a scalar register)
__device__ __forceinline__ int opaque_uniform(int v) {
  asm volatile("" : "+s"(v));
  return v;
}

__device__ __forceinline__ int dot16(const v4i w, const v4i x, int acc) {
  acc = __builtin_amdgcn_sdot4(w.x, x.x, acc, false);
  acc = __builtin_amdgcn_sdot4(w.y, x.y, acc, false);
  acc = __builtin_amdgcn_sdot4(w.z, x.z, acc, false);
  acc = __builtin_amdgcn_sdot4(w.w, x.w, acc, false);
  return acc;
}

// Member slice of a packed Dense: [chunks = K / 16][NS rows][16 bytes]. Thread t serves row
// t % NS and the chunks [kp * CPT, (kp + 1) * CPT) with kp = t / NS (idle when kp >= KP).
// The shape is a template argument of the CALLS, not of the register array: ONE array serves the
// input projections of all three sub-layers, so nothing but the array that is really in flight is
// live across the attention core; every call sits in a branch that is uniform for the workgroup and
// issues its loads back to back.
template <int CAP>  // chunks per thread the register array can hold
struct Gemv {
  v4i w[CAP];
  template <int NS, int KP, int CPT>
  __device__ __forceinline__ void request(const int8_t* __restrict__ slice) {
    static_assert(CPT <= CAP && NS * KP <= kThreads, "Gemv shape");
    const int t = opaque_index(threadIdx.x), j = t % NS, kp = t / NS;
    const v4i* base = reinterpret_cast<const v4i*>(slice) + static_cast<size_t>(kp) * CPT * NS + j;
#pragma unroll
    for (int i = 0; i < CAP; ++i) w[i] = v4i{0, 0, 0, 0};
    if (kp < KP) {
#pragma unroll
      for (int i = 0; i < CPT; ++i) w[i] = __builtin_nontemporal_load(base + static_cast<size_t>(i) * NS);
    }
  }
  // the same request in PIECES parts (part p = chunks [p * CPT / PIECES, (p + 1) * CPT / PIECES)): issued
  // between the polls of a meeting (Handover::collect_paced) - a poll behind the whole request
  // returns only after all of it has landed (loads return in order), which made a meeting with a
  // 16-chunk request 0.8 us longer than one without (round 6 timeline)
  template <int NS, int KP, int CPT, int PIECES>
  __device__ __forceinline__ void request_piece(const int8_t* __restrict__ slice, int p) {
    static_assert(CPT <= CAP && NS * KP <= kThreads && CPT % PIECES == 0, "Gemv shape");
    constexpr int PER = CPT / PIECES;
    const int t = opaque_index(threadIdx.x), j = t % NS, kp = t / NS;
    const v4i* base = reinterpret_cast<const v4i*>(slice) + static_cast<size_t>(kp) * CPT * NS + j;
    if (p == 0) {
#pragma unroll
      for (int i = 0; i < CAP; ++i) w[i] = v4i{0, 0, 0, 0};
    }
    if (kp < KP) {
#pragma unroll
      for (int i = 0; i < CPT; ++i)
        if (i / PER == p) w[i] = __builtin_nontemporal_load(base + static_cast<size_t>(i) * NS);
    }
  }
  // returns the int32 sum of row t of the slice to the threads t < NS (after a barrier)
  template <int NS, int KP, int CPT>
  __device__ __forceinline__ int finish(Shared& sh) {
    const int t = opaque_index(threadIdx.x), kp = t / NS;
    const v4i* xq4 = reinterpret_cast<const v4i*>(sh.xq) + (kp < KP ? kp : 0) * CPT;
    int acc = 0;  // (ONE chain here: two measured 0.2 % slower in this launch, profiles/r06_gemv_two_chains_ab.txt)
#pragma unroll
    for (int i = 0; i < CPT; ++i) acc = dot16(w[i], xq4[i], acc);
    if (kp < KP) sh.part[t] = acc;
    lds_barrier();
    int sum = 0;
    if (t < NS) {
#pragma unroll
      for (int p = 0; p < KP; ++p) sum += sh.part[p * NS + t];
    }
    return sum;
  }
};

// Dequantize epilogue of layers::Dense (gemm_s8.hip: c * (1 / a_scale) / w_scale + bias, then the
// activation, then the residual), one output column
__device__ __forceinline__ float dense_epilogue(int sum, float r_a, float w_scale, bool has_bias, float bias,
                                                int act) {
  float v = static_cast<float>(sum) * r_a;
  v = v / w_scale;
  if (has_bias) v = v + bias;
  return apply_activation(v, act);
}

// `red`: two sets of eight partial maxima used alternately (`flip` toggles per call): a wave that
// writes set A for call n + 2 can not meet a slow wave still reading set A of call n - the barrier
// of call n + 1 lies between them - so the barrier that used to guard the reuse is gone (six calls
// per layer)
__device__ __forceinline__ float block_absmax(float v, Shared& sh, int& flip) {
  float m = wave_max_dpp(fabsf(v));
  float* red = sh.red + flip * 8;
  flip ^= 1;
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  lds_barrier();
  m = red[0];
#pragma unroll
  for (int w = 1; w < kThreads / 64; ++w) m = fmaxf(m, red[w]);
  return m;
}

// ---- hand-over between the members of a cluster ---------------------------------------------
// Two forms (CT2X_CLUSTER_HANDOVER, A/B in profiles/r05_cluster_handover_ab.txt):
//   counter (TAGGED = false): the lines are plain floats; a member drains its stores, arrives with one
//     relaxed atomicAdd on the cluster's counter, polls the counter, then reads the lines: four
//     dependent trips to the memory side per meeting.
//   tagged  (TAGGED = true): every float travels as a 64-bit slot {tag, value} written by ONE atomic
//     store; the reader polls the slot it needs until it carries the meeting's tag: the data IS the
//     flag - no drain, no counter, two trips. Tags never repeat within a session (launch epoch x 64
//     + meeting index), so a stale slot can not be mistaken; a slot is only rewritten two meetings
//     later, when every member has provably consumed it (each meeting needs all members' lines of
//     the previous one).
template <bool TAGGED>
struct Handover {
  float* lines;            // counter form: [kXchFloats] floats; tagged form: as many 64-bit slots
  unsigned* counter;       // counter form: arrivals (monotonic)
  unsigned arrivals;
  unsigned tag;            // tagged form: tag of the NEXT meeting
  int* error;
#if CT2X_POLL_STAMPS
  unsigned npoll = 0;      // polls of this thread since the last CT2X_CL_POLLS (timeline only)
#endif

  __device__ __forceinline__ void publish(int idx, float v) {
    idx = opaque_index(idx);
    if (TAGGED) {
      const unsigned long long u = (static_cast<unsigned long long>(tag) << 32) | __float_as_uint(v);
      grid_publish(reinterpret_cast<unsigned long long*>(lines) + idx, u);
    } else {
      grid_publish(reinterpret_cast<unsigned*>(lines) + idx, __float_as_uint(v));
    }
  }
  // after the publishing code of a meeting (all threads)
  __device__ __forceinline__ void arrive() {
    if (!TAGGED) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this member's published stores have left
      lds_barrier();
      if (threadIdx.x == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      arrivals += kMembers;
    }
  }
  // before the collecting code of a meeting (all threads); also the workgroup barrier between the
  // LDS reads of the phase behind and the LDS writes of the phase ahead
  template <typename S>
  __device__ __forceinline__ void await(S& sh) {
    if (!TAGGED) {
      if (threadIdx.x == 0 && !sh.dead) {
        unsigned spins = 0;
        while (static_cast<int>(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - arrivals) < 0) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > kSpinLimit) {
            sh.dead = 1;
            atomicExch(error, 1);
            break;
          }
        }
      }
    }
    lds_barrier();
  }
  // N lines of this meeting, idx[i] each; call done() once per meeting after the last collect
  template <int N, typename S>
  __device__ __forceinline__ void collect(const int (&idx)[N], float (&out)[N], S& sh) {
    if (TAGGED) {
      const unsigned long long* slots = reinterpret_cast<const unsigned long long*>(lines);
      unsigned long long u[N];
      unsigned spins = 0;
      bool ok;
      int at[N];
#pragma unroll
      for (int i = 0; i < N; ++i) at[i] = opaque_index(idx[i]);
      do {
#if CT2X_POLL_STAMPS
        ++npoll;
#endif
#pragma unroll
        for (int i = 0; i < N; ++i) u[i] = grid_observe(slots + at[i]);
        ok = true;
#pragma unroll
        for (int i = 0; i < N; ++i) ok = ok && static_cast<unsigned>(u[i] >> 32) == tag;
        if (!ok) {
          if (++spins > kSpinLimit || sh.dead) {
            sh.dead = 1;
            atomicExch(error, 1);
            break;
          }
          __builtin_amdgcn_s_sleep(1);  // (a slot that is not there yet: do not hammer the fabric)
        }
      } while (!ok);
#pragma unroll
      for (int i = 0; i < N; ++i) out[i] = __uint_as_float(static_cast<unsigned>(u[i]));
    } else {
#pragma unroll
      for (int i = 0; i < N; ++i)
        out[i] = __uint_as_float(grid_observe(reinterpret_cast<const unsigned*>(lines) + opaque_index(idx[i])));
    }
  }
  // collect, with `issue(p)`, p = 0 .. PIECES - 1 (parts of the next phase's weight request) issued
  // one before each of the first polls; whatever is left after the lines have come is issued then
  template <int N, int PIECES, typename S, typename F>
  __device__ __forceinline__ void collect_paced(const int (&idx)[N], float (&out)[N], S& sh, F&& issue) {
    if (TAGGED) {
      const unsigned long long* slots = reinterpret_cast<const unsigned long long*>(lines);
      unsigned long long u[N];
      int at[N];
#pragma unroll
      for (int i = 0; i < N; ++i) at[i] = opaque_index(idx[i]);
      bool ok = false;
#pragma unroll
      for (int p = 0; p < PIECES; ++p) {
        issue(p);
        if (!ok) {
  #if CT2X_POLL_STAMPS
        ++npoll;
#endif
#pragma unroll
          for (int i = 0; i < N; ++i) u[i] = grid_observe(slots + at[i]);
          ok = true;
#pragma unroll
          for (int i = 0; i < N; ++i) ok = ok && static_cast<unsigned>(u[i] >> 32) == tag;
        }
      }
      unsigned spins = 0;
      while (!ok) {
        if (++spins > kSpinLimit || sh.dead) {
          sh.dead = 1;
          atomicExch(error, 1);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
#if CT2X_POLL_STAMPS
        ++npoll;
#endif
#pragma unroll
        for (int i = 0; i < N; ++i) u[i] = grid_observe(slots + at[i]);
        ok = true;
#pragma unroll
        for (int i = 0; i < N; ++i) ok = ok && static_cast<unsigned>(u[i] >> 32) == tag;
      }
#pragma unroll
      for (int i = 0; i < N; ++i) out[i] = __uint_as_float(static_cast<unsigned>(u[i]));
    } else {  // (counter form: the caller issued the whole request before it waited)
      collect<N>(idx, out, sh);
    }
  }
  template <typename S>
  __device__ __forceinline__ float collect1(int idx, S& sh) {
    const int i1[1] = {idx};
    float o1[1];
    collect<1>(i1, o1, sh);
    return o1[0];
  }
  __device__ __forceinline__ void done() { ++tag; }
};

}  // namespace

// per-thread operands of a phase, requested together with the phase's weights (nothing on the
// dependent chain of a phase should be a first-touch global load)
struct ColRegs {  // Dequantize operands of this thread's output column
  float wscale, bias;
  bool has_bias;
  __device__ __forceinline__ void request(const ClusterDenseW& w, int col, bool active) {
    has_bias = w.bias != nullptr;
    wscale = 1.f;
    bias = 0.f;
    if (active) {
      col = opaque_index(col);
      wscale = w.wscale[col];
      if (has_bias) bias = w.bias[col];
    }
  }
};
struct NormRegs {  // element t of the norm's parameters
  float gamma, beta, eps;
  bool layer_norm, rms_use_residual;
  __device__ __forceinline__ void request(const ClusterNormW& n) {
    layer_norm = n.beta != nullptr;
    rms_use_residual = n.rms_use_residual != 0;
    eps = n.eps;
    const int t = opaque_index(threadIdx.x);
    gamma = n.gamma[t];
    beta = layer_norm ? n.beta[t] : 0.f;
  }
};

// LayerNorm / RMSNorm of the row (element t in `v`) followed by ops::Quantize into sh.xq; returns
// the quantisation scale. Every wave runs the statistics chain itself.
struct NoRequests {
  __device__ __forceinline__ void operator()() const {}
};
// `parked()`: called once the row is parked in LDS and the statistics chains may start - requests
// that do not depend on the row (the attention's cache tile) leave here, under the chains, instead
// of in front of the store + barrier every wave's chain waits for
template <typename F = NoRequests>
__device__ __forceinline__ float norm_quantize_row(float v, const NormRegs& ln, int quant_mode, Shared& sh, int& flip,
                                                   unsigned long long* dbg = nullptr, F&& parked = F()) {
  const int t = threadIdx.x;
  constexpr int pad = 68;
  sh.stage[norm_chain_slot(t, kD, pad)] = v;
  lds_barrier();
  parked();
  float sum, sumsq;
  ordered_row_stats_t<true, 0>(sh.stage, pad, kD, sum, sumsq);
  if (dbg != nullptr && t == 0) dbg[0] = wall_clock64();
  const NormStats st = norm_stats(sum, sumsq, kD, ln.eps, ln.layer_norm);
  const float y = norm_apply(v, ln.gamma, ln.beta, st, ln.layer_norm, ln.rms_use_residual, t);
  const float amax = block_absmax(y, sh, flip);
  if (dbg != nullptr && t == 0) dbg[1] = wall_clock64();
  const float scale = (amax != 0.f) ? 127.f / amax : 1.f;
  sh.xq[t] = to_s8(y, scale, quant_mode);
  lds_barrier();
  if (dbg != nullptr && t == 0) dbg[2] = wall_clock64();
  return scale;
}

// maximum over the row of 16 lanes a lane belongs to, to every lane of the row (row_ror 8, 4, 2, 1)
__device__ __forceinline__ unsigned row16_max_u32(unsigned v) {
  v = max(v, static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x128, 0xf, 0xf, false)));
  v = max(v, static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x124, 0xf, 0xf, false)));
  v = max(v, static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x122, 0xf, 0xf, false)));
  v = max(v, static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x121, 0xf, 0xf, false)));
  return v;
}

// Projection tail: wave w's share of a pass's weight rows, global -> LDS by DMA (1 KB per
// instruction, no registers): chunk (w, ks) = the B fragment of MFMA k-step ks for the wave's 16
// columns - lane l supplies 16 bytes of column l % 16 at K offset ks * 64 + (l / 16) * 16, the LDS
// image is lane-linear, so the fragment is read back with one conflict-free ds_read_b128. Columns
// past the vocabulary read its last row (masked in the epilogue).
__device__ __forceinline__ int8_t* proj_wbuf(unsigned char* smem_raw, int t_pad) {  // behind Shared and the score row
  return reinterpret_cast<int8_t*>(smem_raw) + ((sizeof(Shared) + static_cast<size_t>(t_pad) * sizeof(float) + 1023) & ~size_t(1023));
}
// chunk c = (N tile c / 8, k-step c % 8) of a pass; the policy is non-temporal: a weight row is read
// once per step and should not push the XCD's slice of the decoder weights out of its L2
__device__ __forceinline__ void proj_request_chunk(const ClusterProjArgs& P, int8_t* wbuf, int col_base, int c, int lane) {
  const int col = min(col_base + (c >> 3) * 16 + (lane & 15), P.vocab - 1);
  const int8_t* src = P.w + static_cast<int64_t>(col) * kD + (c & 7) * 64 + (lane >> 4) * 16;
  __builtin_amdgcn_global_load_lds((glb_void*)(src), (lds_void*)(wbuf + c * 1024), 16, 0, 2);
}
// a whole pass, by all eight waves (the passes after the first)
__device__ __forceinline__ void proj_request(const ClusterProjArgs& P, int8_t* wbuf, int col_base) {
  const int t = opaque_index(threadIdx.x), lane = t & 63, wave = t >> 6;
  for (int ks = 0; ks < 8; ++ks) proj_request_chunk(P, wbuf, col_base, wave * 8 + ks, lane);
}
// the FIRST pass, by the seven waves that sit idle while wave 0 runs the last layer's
// encoder-decoder attention core: the requests miss the L2 (16 MB of projection weights do not
// live there), and loads return in order - in front of a meeting's polls they cost the meeting their
// whole latency (+1.5 us measured), here they land while their waves have nothing to wait for
__device__ __forceinline__ void proj_request_first(const ClusterProjArgs& P, int8_t* wbuf, int col_base) {
  const int t = opaque_index(threadIdx.x), lane = t & 63, wave = t >> 6;
  for (int c = wave - 1; c < 64; c += 7) proj_request_chunk(P, wbuf, col_base, c, lane);
}

#define CT2X_CL_STAMP(i)                                                                     \
  if (a.stamps != nullptr && blockIdx.x == 0 && t == 0) a.stamps[stamp_base + (i)] = wall_clock64();
// polls the slowest lane of wave 0 needed in the meeting behind (shown as N.00). A build option
// (-DCT2X_POLL_STAMPS=1): the counter costs a register the launch does not have (256 of 256; the
// counted build ran 10 % slower). What it showed (round 6): EVERY meeting's first poll succeeds - a
// meeting costs one store-to-visible plus one load round trip (0.96 us on an idle chip,
// tools/xcd_local_handover_probe.cpp), there is nothing to gain from keeping several polls in flight
#ifndef CT2X_POLL_STAMPS
#define CT2X_POLL_STAMPS 0
#endif
#if CT2X_POLL_STAMPS
#define CT2X_CL_POLLS(i)                                                                     \
  if (a.stamps != nullptr && blockIdx.x == 0 && wave == 0) {                                 \
    const float np_ = wave_max_dpp(static_cast<float>(ho.npoll));                            \
    if (t == 0) a.stamps[stamp_base + (i)] = a.stamps[0] + static_cast<unsigned long long>(np_) * 100u; \
    ho.npoll = 0;                                                                            \
  }
#else
#define CT2X_CL_POLLS(i)
#endif
// the extra stamps of a layer: 14 .. 16 inside the self-attention core (QK^T, SoftMax, P.V), 17 .. 19
// inside the self sub-layer's LayerNorm + Quantize (statistics, absmax, quantised), 20 the QKV sums
// are there, 21 the K / V tile is staged

// grid = rows * 8 workgroups of 512 threads; blockIdx = row * 8 + member
template <bool TAGGED>
__global__ __launch_bounds__(kThreads) void decoder_cluster_kernel(ClusterStackArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Shared& sh = *reinterpret_cast<Shared*>(smem_raw);
  float* tp = reinterpret_cast<float*>(smem_raw + sizeof(Shared));  // attention scores [t_pad]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int member = blockIdx.x % kMembers, row = blockIdx.x / kMembers;
  Handover<TAGGED> ho;
  ho.lines = a.xch + static_cast<size_t>(row) * kXchFloats * (TAGGED ? 2 : 1);
  ho.counter = a.counters + static_cast<size_t>(row) * 16;
  ho.error = a.error;
  // The launch's first memory round trips, merged (round 6: the prologue was four dependent trips -
  // error flag, epoch / step / id, embedding row behind the 96 KB of QKV weights - 3.5 us before
  // the first layer began): everything that does not depend on anything leaves together, the
  // embedding row leaves as soon as the id is here and BEFORE the weight requests (loads return in
  // order, and the row is what the first LayerNorm waits for).
  int stamp_base = 0;
  CT2X_CL_STAMP(0)
  const int err = __hip_atomic_load(a.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // counter form: where the counter stood when this launch began - the members of the previous
  // launch all left at a multiple of kMembers, and at most kMembers - 1 early arrivals of THIS
  // launch's first meeting can have been added since. Tagged form: the launch epoch of this row
  // (advanced by member 0 when the launch ends; zero at session start, so is every slot's tag).
  const unsigned arrivals0 = TAGGED ? 0u : __hip_atomic_load(ho.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned epoch = TAGGED ? ho.counter[1] : 0u;
  const int step = *a.step_ptr;
  const int mem_len = a.mem_lengths[row];
  const bool proj = TAGGED && a.proj.w != nullptr;
  const unsigned pseq = proj ? *a.proj.seq : 0u;
  const int64_t id = a.ids[row];
  // ---- embedding row (embed_kernel, rowops.hip), every member for itself ----------------------
  float e_f = 0.f, e_scale = 1.f, e_pos = 0.f;
  int e_q = 0;
  {
    const int te = opaque_index(threadIdx.x);
    if (a.emb_int8) {
      e_q = (static_cast<const int8_t*>(a.emb_table) + id * kD)[te];
      e_scale = a.emb_scale[id];
    } else {
      e_f = static_cast<const float*>(a.emb_table)[id * kD + te];
    }
    if (a.pos_table) e_pos = (a.pos_table + static_cast<int64_t>(step) * kD)[te];
  }
  // (a session whose earlier launch lost a member is void: its remaining replays return at once
  // instead of each waiting out the bounded polls again)
  if (err != 0) return;
  if (t == 0) sh.dead = 0;
  ho.arrivals = arrivals0 / kMembers * kMembers;
  ho.tag = epoch * 64u + 1u;

  // operands of the NEXT input projection (QKV / cross-attention query / feed-forward input) and
  // of the next output projection: requested one meeting ahead
  Gemv<16> g_in;
  Gemv<4> g_out;  // (the one array that is live across the attention core)
  ColRegs c_in, c_out;
  NormRegs n_in;
  const int qkv_col = (t >> 6) * kD + member * kDH + (t & 63);  // Q | K | V sections of linear_0
  const int out_col = member * kDH + (t & 63);                  // this thread's column of a 64-wide slice
  g_in.request<192, 2, 16>(a.layers[0].qkv.w + static_cast<size_t>(member) * 192 * kD);
  c_in.request(a.layers[0].qkv, qkv_col, t < 192);
  n_in.request(a.layers[0].ln_self);

  float xv;
  if (a.emb_int8) {
    const float r_s = 1.f / e_scale;
    xv = static_cast<float>(e_q) * r_s;
  } else {
    xv = e_f;
  }
  if (a.emb_apply_mul) xv = xv * a.emb_mul;
  if (a.pos_table) xv = xv + e_pos;
  sh.x_row[t] = xv;
  stamp_base = 1;
  float out_scale = 1.f;  // quantisation scale of the row the output norm leaves (projection tail)
  int red_flip = 0;       // which set of sh.red the next block_absmax writes

  for (int l = 0; l < a.n_layers; ++l) {
    const ClusterLayerArgs& L = a.layers[l];  // (kernel arguments: scalar loads)
    CT2X_CL_STAMP(0)
    // ====== masked self-attention (sub 0), encoder-decoder attention (sub 1): one body, so that
    // ====== the attention core is instantiated once and keeps the whole register file
#pragma unroll 1
    for (int sub = 0; sub < 2; ++sub) {
      const bool self = sub == 0;
      const ClusterDenseW& W_out = self ? L.self_out : L.cross_out;
      // self: keys = the cache + this step's; cross: the PADDED memory, SoftMax over its length
      const int t_cached = self ? max(0, min(step, a.t_max)) : max(0, min(mem_len, a.s_max));
      const int n_keys = self ? t_cached + 1 : a.s_max;
      float* kc_self = L.self_k + (static_cast<int64_t>(row) * kMembers + member) * a.t_max * kDH;
      float* vc_self = L.self_v + (static_cast<int64_t>(row) * kMembers + member) * a.t_max * kDH;
      const float* kc_cross = L.cross_kv + static_cast<int64_t>(row) * a.s_max * (2 * kD) + member * kDH;
      const float* kc = self ? kc_self : kc_cross;
      const float* vc = self ? vc_self : kc_cross + kD;
      const int kvs = self ? kDH : 2 * kD;
      // the cache rows of past steps do not depend on this step: their loads ride under the norm
      // and the projection
      AttnStage64 stage;
      {
        const float scale = norm_quantize_row(xv, n_in, a.quant_mode, sh, red_flip,
                                              (a.stamps != nullptr && blockIdx.x == 0 && self)
                                                  ? a.stamps + stamp_base + 17 : nullptr,
                                              [&]() { stage.request(kc, vc, kvs, t_cached, n_keys); });
        const int sum = self ? g_in.finish<192, 2, 16>(sh) : g_in.finish<64, 8, 4>(sh);
        if (self) { CT2X_CL_STAMP(20) }
        if (t < (self ? 192 : 64))
          sh.outv[t] = dense_epilogue(sum, 1.f / scale, c_in.wscale, c_in.has_bias, c_in.bias, -1);
      }
      stage.store(sh.ks, sh.vs);
      if (self) { CT2X_CL_STAMP(21) }
      // (behind the tiles' stores: a store waits for every request ahead of it in the queue, and the
      // next phase's weights are meant to arrive under the attention)
      g_out.request<64, 8, 4>(W_out.w + static_cast<size_t>(member) * 64 * kD);
      c_out.request(W_out, out_col, t < 64);
      lds_barrier();
      CT2X_CL_STAMP(1 + 5 * sub)
      if (wave == 0) {
        const float qd = sh.outv[lane];
        const float knew = self ? sh.outv[64 + lane] : 0.f, vnew = self ? sh.outv[128 + lane] : 0.f;
        if (self && t_cached < a.t_max) {  // cache append in place of ops::Concat (attention.cc:541-546)
          kc_self[t_cached * kDH + lane] = knew;
          vc_self[t_cached * kDH + lane] = vnew;
        }
        const float ctx = attn_exact_wave64(qd, self, knew, vnew, kc, vc, kvs, t_cached, n_keys,
                                            self ? t_cached + 1 : t_cached, self ? L.self_scale : L.cross_scale,
                                            sh.tq, tp, a.t_pad, sh.ks, sh.vs, 1, nullptr, 0, 0, 0,
                                            (a.stamps != nullptr && blockIdx.x == 0 && self)
                                                ? a.stamps + stamp_base + 14 : nullptr);
        ho.publish(kXchCtx + member * kDH + lane, ctx);
      } else if (proj && !self && l + 1 == a.n_layers) {
        proj_request_first(a.proj, proj_wbuf(smem_raw, a.t_pad), static_cast<int>(blockIdx.x) * a.proj.cols_per_wg);
      }
      CT2X_CL_STAMP(2 + 5 * sub)
      ho.arrive();
      ho.await(sh);
      CT2X_CL_STAMP(3 + 5 * sub)
      // ---- attention output projection + residual --------------------------------------------
      {
        const float c = ho.collect1(kXchCtx + t, sh);
        ho.done();
        CT2X_CL_POLLS(22 + 2 * sub)
        const float amax = block_absmax(c, sh, red_flip);
        const float scale = (amax != 0.f) ? 127.f / amax : 1.f;
        sh.xq[t] = to_s8(c, scale, a.quant_mode);
        lds_barrier();
        const int sum = g_out.finish<64, 8, 4>(sh);
        if (t < 64) {
          float v = dense_epilogue(sum, 1.f / scale, c_out.wscale, c_out.has_bias, c_out.bias, -1);
          v = sh.x_row[out_col] + v;
          ho.publish(kXchX + out_col, v);
        }
      }
      CT2X_CL_STAMP(4 + 5 * sub)
      ho.arrive();
      if (self) {
        g_in.request<64, 8, 4>(L.cross_q.w + static_cast<size_t>(member) * 64 * kD);
        c_in.request(L.cross_q, out_col, t < 64);
        n_in.request(L.ln_cross);
        ho.await(sh);
        CT2X_CL_STAMP(5 + 5 * sub)
        xv = ho.collect1(kXchX + t, sh);
      } else {
        const int8_t* w_ffn0 = L.ffn0.w + static_cast<size_t>(member) * 256 * kD;
        if (!TAGGED) g_in.request<256, 2, 16>(w_ffn0);
        c_in.request(L.ffn0, member * 256 + (t & 255), t < 256);
        n_in.request(L.ln_ffn);
        ho.await(sh);
        CT2X_CL_STAMP(5 + 5 * sub)
        const int xi[1] = {kXchX + t};
        float xo[1];
        ho.template collect_paced<1, kPaceParts>(xi, xo, sh, [&](int p) { g_in.request_piece<256, 2, 16, kPaceParts>(w_ffn0, p); });
        xv = xo[0];
      }
      ho.done();
      CT2X_CL_POLLS(23 + 2 * sub)
      sh.x_row[t] = xv;
    }
    // ================= feed forward ===========================================================
    {
      const float scale = norm_quantize_row(xv, n_in, a.quant_mode, sh, red_flip);
      const int sum = g_in.finish<256, 2, 16>(sh);
      if (t < 256) {
        const float v = dense_epilogue(sum, 1.f / scale, c_in.wscale, c_in.has_bias, c_in.bias, a.act);
        ho.publish(kXchInner + member * 256 + t, v);
      }
    }
    CT2X_CL_STAMP(11)
    ho.arrive();
    Gemv<16> g_f1;
    const int8_t* w_ffn1 = L.ffn1.w + static_cast<size_t>(member) * 64 * kFfn;
    if (!TAGGED) g_f1.request<64, 8, 16>(w_ffn1);
    c_out.request(L.ffn1, out_col, t < 64);
    ho.await(sh);
    CT2X_CL_STAMP(12)
    {
      float iv[4];
      const int inner_idx[4] = {kXchInner + t, kXchInner + kThreads + t, kXchInner + 2 * kThreads + t,
                                kXchInner + 3 * kThreads + t};
      ho.template collect_paced<4, kPaceParts>(inner_idx, iv, sh,
                                               [&](int p) { g_f1.request_piece<64, 8, 16, kPaceParts>(w_ffn1, p); });
      ho.done();
      CT2X_CL_POLLS(26)
      const float amax =
          block_absmax(fmaxf(fmaxf(fabsf(iv[0]), fabsf(iv[1])), fmaxf(fabsf(iv[2]), fabsf(iv[3]))), sh, red_flip);
      const float scale = (amax != 0.f) ? 127.f / amax : 1.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) sh.xq[i * kThreads + t] = to_s8(iv[i], scale, a.quant_mode);
      lds_barrier();
      const int sum = g_f1.finish<64, 8, 16>(sh);
      if (t < 64) {
        float v = dense_epilogue(sum, 1.f / scale, c_out.wscale, c_out.has_bias, c_out.bias, -1);
        v = sh.x_row[out_col] + v;
        const bool last = l + 1 == a.n_layers;
        if (!last || a.q_out != nullptr || proj) ho.publish(kXchX + out_col, v);
        // (read by the next launch; the address behind opaque moves - hoisted, it was spilled)
        if (last)
          a.x_out[static_cast<int64_t>(opaque_uniform(static_cast<int>(blockIdx.x)) / kMembers) * kD + opaque_index(out_col)] = v;
      }
    }
    CT2X_CL_STAMP(13)
    if (l + 1 < a.n_layers) {
      ho.arrive();
      const ClusterLayerArgs& N = a.layers[l + 1];
      const int8_t* w_qkv = N.qkv.w + static_cast<size_t>(member) * 192 * kD;
      if (!TAGGED) g_in.request<192, 2, 16>(w_qkv);
      {  // (recomputed from fresh indices: kept across the layer loop it was the register that got spilled)
        const int tt = opaque_index(threadIdx.x);
        c_in.request(N.qkv, (tt >> 6) * kD + member * kDH + (tt & 63), tt < 192);
      }
      n_in.request(N.ln_self);
      ho.await(sh);
      const int xi[1] = {kXchX + t};
      float xo[1];
      ho.template collect_paced<1, kPaceParts>(xi, xo, sh, [&](int p) { g_in.request_piece<192, 2, 16, kPaceParts>(w_qkv, p); });
      xv = xo[0];
      ho.done();
      CT2X_CL_POLLS(27)
      sh.x_row[t] = xv;
    } else if (a.q_out != nullptr || proj) {
      // ---- the decoder's output norm + Quantize: the input row of the vocabulary projection ----
      ho.arrive();
      n_in.request(a.out_norm);
      ho.await(sh);
      xv = ho.collect1(kXchX + t, sh);
      ho.done();
      out_scale = norm_quantize_row(xv, n_in, a.quant_mode, sh, red_flip);
      if (!proj) {
        // (every member holds the whole quantised row: each stores its own 64 columns)
        // (the address behind opaque moves: hoisted out of the layer loop it was a spilled register pair)
        if (t < kDH) {
          const int oc = opaque_index(out_col);
          a.q_out[static_cast<int64_t>(opaque_uniform(static_cast<int>(blockIdx.x)) / kMembers) * kD + oc] = sh.xq[oc];
        }
        if (member == 0 && t == 0) a.qscale_out[row] = out_scale;
      }
    }
    stamp_base += kClusterStampsPerLayer;
  }
  if (proj) {
    {
      {
        // ================= vocabulary projection + greedy arg-max (ClusterProjArgs) =============
        // Dense (gemm_s8.hip epilogue: c * (1 / a_scale) / w_scale + bias) of ALL rows against this
        // workgroup's columns on the MFMA, the row's best (value, column) key max-combined per row,
        // the last workgroup to arrive finishes the step (greedy_fuse.h).
        const ClusterProjArgs& P = a.proj;
        // (this block's own thread indices: nothing of it is live across the layer loop)
        const int t = opaque_index(threadIdx.x), lane = t & 63, wave = t >> 6;
        const int row = opaque_uniform(static_cast<int>(blockIdx.x)) / kMembers, member = static_cast<int>(blockIdx.x) % kMembers;
        int8_t* wbuf = proj_wbuf(smem_raw, a.t_pad);
        const int col0 = static_cast<int>(blockIdx.x) * P.cols_per_wg;
        float ws0 = 1.f, bias0 = 0.f;
        {  // (the first pass's column operands: requested ahead of the polls)
          const int c = min(col0 + wave * 16 + (lane & 15), P.vocab - 1);
          ws0 = P.wscale[opaque_index(c)];
          if (P.bias != nullptr) bias0 = P.bias[opaque_index(c)];
        }
        int8_t* a_tile = reinterpret_cast<int8_t*>(sh.ks);   // [32][kProjAStride] (the tiles are dead)
        float* r_a = sh.outv;                                // 1 / a_scale of every row
        int* masked = sh.part;                               // the column DisableTokens masks, per row
        unsigned long long* keys = reinterpret_cast<unsigned long long*>(sh.stage);
        const unsigned ptag = pseq + 1u;
        CT2X_CL_STAMP(1)  // (tail stamps: 1 = the output norm is done, 2 = all rows are here, 3 = the weights too,
                          // 4 = the passes, 5 = the row maxima have left, 6 = the ticket is drawn, 0 = end)
        if (member == 0) {  // (every member holds the row: one publishes it)
          unsigned long long* rq = P.rows_q + static_cast<size_t>(row) * kProjRowSlots;
          if (t < kD / 4)
            grid_publish(rq + t, (static_cast<unsigned long long>(ptag) << 32) |
                                     *reinterpret_cast<const unsigned*>(sh.xq + 4 * t));
          else if (t == kD / 4)
            grid_publish(rq + t, (static_cast<unsigned long long>(ptag) << 32) | __float_as_uint(out_scale));
        }
        if (t < 32) {
          keys[t] = 0ull;
          int plen = 0;
          if (t < a.rows && P.gf.prefix_ids != nullptr && P.gf.prefix_extends_min) plen = P.gf.prefix_lens[t];
          masked[t] = (t < a.rows && step < P.gf.min_length + plen) ? P.gf.disable_id : -1;
        }
        {
          constexpr int NP = (32 * kProjRowSlots + kThreads - 1) / kThreads;
          const int total = a.rows * kProjRowSlots;
          unsigned pending = 0u;
#pragma unroll
          for (int i = 0; i < NP; ++i)
            if (t + kThreads * i < total) pending |= 1u << i;
          unsigned spins = 0;
          while (pending != 0u) {
            unsigned long long u[NP];
#pragma unroll
            for (int i = 0; i < NP; ++i)
              if ((pending >> i) & 1u) u[i] = grid_observe(P.rows_q + opaque_index(t + kThreads * i));
#pragma unroll
            for (int i = 0; i < NP; ++i) {
              if (((pending >> i) & 1u) && static_cast<unsigned>(u[i] >> 32) == ptag) {
                const int sidx = t + kThreads * i, rr = sidx / kProjRowSlots, j = sidx - rr * kProjRowSlots;
                if (j < kD / 4)
                  *reinterpret_cast<unsigned*>(a_tile + rr * kProjAStride + 4 * j) = static_cast<unsigned>(u[i]);
                else
                  r_a[rr] = 1.f / __uint_as_float(static_cast<unsigned>(u[i]));
                pending &= ~(1u << i);
              }
            }
            if (pending != 0u) {
              if (++spins > kSpinLimit || sh.dead) {
                sh.dead = 1;
                atomicExch(a.error, 1);
                break;
              }
              __builtin_amdgcn_s_sleep(1);
            }
          }
        }
        CT2X_CL_STAMP(2)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the DMA of the first pass has landed)
        lds_barrier();
        CT2X_CL_STAMP(3)
        unsigned long long best[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) best[i][q] = 0ull;
        const int passes = P.cols_per_wg / kProjCols;
        for (int p = 0; p < passes; ++p) {
          float ws = ws0, bs = bias0;
          const int col = col0 + p * kProjCols + wave * 16 + (lane & 15);
          if (p > 0) {
            lds_barrier();  // (the previous pass's fragments have been read)
            proj_request(P, wbuf, col0 + p * kProjCols);
            const int c = min(col, P.vocab - 1);
            ws = P.wscale[c];
            if (P.bias != nullptr) bs = P.bias[c];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();
          }
          v4i acc[2] = {v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}};
#pragma unroll
          for (int ks = 0; ks < 8; ++ks) {
            const v4i bf = *reinterpret_cast<const v4i*>(wbuf + (wave * 8 + ks) * 1024 + lane * 16);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const v4i af = *reinterpret_cast<const v4i*>(a_tile + (i * 16 + (lane & 15)) * kProjAStride + ks * 64 +
                                                           (lane >> 4) * 16);
              acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af, bf, acc[i], 0, 0, 0);
            }
          }
          if (col < P.vocab) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int rr = i * 16 + (lane >> 4) * 4 + q;
                float v = static_cast<float>(acc[i][q]) * r_a[rr];
                v = v / ws;
                if (P.bias != nullptr) v = v + bs;
                if (col != masked[rr] && v == v) {
                  const unsigned long long key = greedy_fuse_key(v, static_cast<unsigned>(col));
                  best[i][q] = key > best[i][q] ? key : best[i][q];
                }
              }
          }
        }
        CT2X_CL_STAMP(4)
        // the 16 lanes that share a row (a DPP row rotation is an all-reduce of a row of 16 lanes: the
        // order word first, then the column word among the lanes that hold the maximum - the 64-bit
        // shuffles of the first form were 2 us of this tail), then the eight waves (LDS), then the launch
        {
          unsigned long long* wave_keys = reinterpret_cast<unsigned long long*>(sh.vs);  // [8 waves][32 rows] (the value tile is dead)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const unsigned hi = static_cast<unsigned>(best[i][q] >> 32);
              const unsigned m_hi = row16_max_u32(hi);
              const unsigned lo = (hi == m_hi) ? static_cast<unsigned>(best[i][q]) : 0u;
              const unsigned m_lo = row16_max_u32(lo);
              const int rr = i * 16 + (lane >> 4) * 4 + q;
              if ((lane & 15) == 0) wave_keys[wave * 32 + rr] = (static_cast<unsigned long long>(m_hi) << 32) | m_lo;
            }
          lds_barrier();
          if (t < 32) {
            unsigned long long m = wave_keys[t];
#pragma unroll
            for (int w = 1; w < kThreads / 64; ++w) m = wave_keys[w * 32 + t] > m ? wave_keys[w * 32 + t] : m;
            keys[t] = m;
          }
        }
        __syncthreads();
        if (P.board != nullptr) {
          // ---- the board form: every workgroup posts its 32 row maxima, workgroup r finishes row r
          const unsigned btag = pseq % 65535u + 1u;
          const int wgs = static_cast<int>(gridDim.x);
          if (t < a.rows) {
            const unsigned long long k = keys[t];  // (ord << 32) | (0xFFFFFFFF - column), 0 = none
            const unsigned long long k48 = (k == 0ull) ? 0ull : (((k >> 32) << 16) | (k & 0xFFFFull));
            grid_publish(P.board + (static_cast<size_t>(t) * wgs + blockIdx.x), (k48 << 16) | btag);
          }
          CT2X_CL_STAMP(5)
          if (static_cast<int>(blockIdx.x) < a.rows) {
            const int r = static_cast<int>(blockIdx.x);
            unsigned long long u = 0ull;
            if (t < wgs) {
              unsigned spins = 0;
              const unsigned long long* slot = P.board + (static_cast<size_t>(r) * wgs + t);
              for (;;) {
                u = grid_observe(slot);
                if (static_cast<unsigned>(u & 0xFFFFull) == btag) break;
                if (++spins > kSpinLimit || sh.dead) {
                  sh.dead = 1;
                  atomicExch(a.error, 1);
                  u = 0ull;
                  break;
                }
                __builtin_amdgcn_s_sleep(1);
              }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
              const unsigned long long other = __shfl_xor(u, off, 64);
              u = other > u ? other : u;
            }
            lds_barrier();  // (keys[] of the phase above have been read)
            if (lane == 0) keys[wave] = u;
            lds_barrier();
            CT2X_CL_STAMP(6)
            if (t == 0) {
              unsigned long long m = keys[0];
#pragma unroll
              for (int w = 1; w < kThreads / 64; ++w) m = keys[w] > m ? keys[w] : m;
              const unsigned long long k48 = m >> 16;
              int id = 0;
              float score = 0.f;
              if (k48 != 0ull) {
                id = static_cast<int>(0xFFFFu - static_cast<unsigned>(k48 & 0xFFFFull));
                const unsigned ord = static_cast<unsigned>(k48 >> 16);
                score = __uint_as_float((ord & 0x80000000u) ? (ord & 0x7FFFFFFFu) : ~ord);
              }
              greedy_fuse_finish_row(P.gf, a.rows, step, r, id, score);
              if (r == 0 && P.gf.advance_step_ptr != nullptr) *P.gf.advance_step_ptr = step + 1;
            }
          }
        } else {
        if (t < a.rows && keys[t] != 0ull) {
          // (a returning atomic: it has been performed when its value arrives, i.e. before the ticket)
          const unsigned long long old =
              atomicMax(P.gf.slots + static_cast<int64_t>(t) * GreedyFuse::kSubSlots + (blockIdx.x % GreedyFuse::kSubSlots), keys[t]);
          asm volatile("" ::"v"(static_cast<unsigned>(old)));
        }
        __syncthreads();
        CT2X_CL_STAMP(5)
        if (t == 0) masked[32] = static_cast<int>(atomicAdd(P.gf.ticket, 1u));
        __syncthreads();
        CT2X_CL_STAMP(6)
        if (static_cast<unsigned>(masked[32]) == gridDim.x - 1) greedy_fuse_finish<kThreads>(P.gf, a.rows, step, t);
        }
        if (blockIdx.x == 0 && t == 0) *P.seq = pseq + 1u;  // (read by the NEXT launch)
      }
    }
  }
  // (read by the NEXT launch's members; re-read here: only this thread writes it. Fresh indices: the
  // test's operands would otherwise be one more value kept alive - spilled - across the layer loop)
  if (TAGGED && opaque_uniform(static_cast<int>(blockIdx.x)) % kMembers == 0 && opaque_index(threadIdx.x) == 0) {
    unsigned* ep = a.counters + static_cast<size_t>(opaque_uniform(static_cast<int>(blockIdx.x)) / kMembers) * 16 + 1;
    *ep = *ep + 1u;
  }
  CT2X_CL_STAMP(0)
}

// ================================================================================================
// Beam search: the cluster owns a batch ENTRY - its `beam` hypothesis rows (<= 4) travel through
// the launch together. The weights of a phase are requested once and serve all rows (acc[r]); the
// rows' LayerNorm chains interleave; the encoder-decoder attention of the entry's rows shares ONE
// staged memory tile and takes MKL's several-row forms (m = beam query rows on the same keys,
// except at the search's first step, which the reference still runs on one row per entry:
// src/decoding.cc:453-454); the self-attention follows the beam search's ancestry table
// (position t of row r lives in cache row anc[t][r]) and runs every row on a wave and a tile of its own.
// Same arithmetic per row as the one-row launch above and as the launch-per-op step.
// ================================================================================================
constexpr int kBeamRows = 4;

// Round 6 (last session): FOUR K / V tiles, one per row of the entry, so that the self-attention of
// the entry's rows runs in ONE round on four waves (it was two rows at a time on two tiles: 12.2 of
// a layer's 47 us, profiles/r05_cluster_beam_timeline_c.txt). Four tiles are 132 KB; they fit the
// CU's 160 KB because the fourth lies over what only the row-wide phases use (the parked rows of
// the statistics chains, the quantised rows, the GEMV partial sums: all dead while the attention
// runs; a barrier separates their last readers from the tile's store, Handover::await's barrier the
// tile's last readers from their next writers).
struct SharedBeam {
  // (the row-wide phases' arrays FIRST: within reach of the LDS instructions' 16-bit immediate
  // offsets - behind 100 KB of tiles every access needed an address register of its own and the
  // launch kept 30 more values in scratch memory)
  union {
    struct {
      float stage[kBeamRows][8 * 68 + 8];  // the rows parked for their statistics chains
      alignas(16) int8_t xq[kBeamRows][kFfn];
      int part[kBeamRows][kThreads];
    };
    struct {                               // tile 3
      float ks3[64 * kXKS];
      float vs3[64 * kDH];
    };
  };
  float x_row[kBeamRows][kD];       // residual stream of the entry's rows (the source of truth)
  float outv[kBeamRows][256];
  float red[kBeamRows][4];
  float rs[kBeamRows];              // quantisation scale of every row's current input
  float tq[kBeamRows][kDH];
  int dead;
  int pad_[3];
  float ks[3][64 * kXKS];           // tiles 0 .. 2 (the shared memory tile of beam search is [0])
  float vs[3][64 * kDH];
  __device__ __forceinline__ float* k_tile(int i) { return i < 3 ? ks[i] : ks3; }
  __device__ __forceinline__ float* v_tile(int i) { return i < 3 ? vs[i] : vs3; }
};
static_assert(sizeof(SharedBeam) % 16 == 0 && offsetof(SharedBeam, ks) % 16 == 0, "SharedBeam: 16-byte tiles");

// Every loop over the entry's rows is ROLLED (one copy of the per-row code; the launch must stay
// inside the instruction cache: an unrolled first version was 94 KB of code and ran at a third of
// the speed); what is shared between the rows - the weights of a phase - stays in registers.
template <int CAP>
struct GemvRows {
  v4i w[CAP];
  template <int NS, int KP, int CPT>
  __device__ __forceinline__ void request(const int8_t* __restrict__ slice) {
    static_assert(CPT <= CAP && NS * KP <= kThreads, "Gemv shape");
    const int t = opaque_index(threadIdx.x), j = t % NS, kp = t / NS;
    const v4i* base = reinterpret_cast<const v4i*>(slice) + static_cast<size_t>(kp) * CPT * NS + j;
#pragma unroll
    for (int i = 0; i < CAP; ++i) w[i] = v4i{0, 0, 0, 0};
    if (kp < KP) {
#pragma unroll
      for (int i = 0; i < CPT; ++i) w[i] = __builtin_nontemporal_load(base + static_cast<size_t>(i) * NS);
    }
  }
  // partial sums of every row into sh.part, then a barrier; row_sum() gives t < NS its column
  template <int NS, int KP, int CPT>
  __device__ __forceinline__ void finish(SharedBeam& sh, int nr) {
    const int t = opaque_index(threadIdx.x), kp = t / NS;
#pragma unroll 1
    for (int r = 0; r < nr; ++r) {
      const v4i* xq4 = reinterpret_cast<const v4i*>(sh.xq[r]) + (kp < KP ? kp : 0) * CPT;
      // (two dependent chains instead of one - integer sums are exact in any order: beam 4 90.5k ->
      // 91.2k tokens/s, and the allocator keeps 4 values in scratch memory instead of 13)
      int acc = 0, acc1 = 0;
#pragma unroll
      for (int i = 0; i < CPT; ++i) {
        if (i & 1) acc1 = dot16(w[i], xq4[i], acc1);
        else acc = dot16(w[i], xq4[i], acc);
      }
      acc += acc1;
      if (kp < KP) sh.part[r][t] = acc;
    }
    lds_barrier();
  }
};

// (column `col` of row r for a thread that is not thread `col`: the rows' epilogues in parallel)
template <int NS, int KP>
__device__ __forceinline__ int row_sum_col(const SharedBeam& sh, int r, int col) {
  int sum = 0;
#pragma unroll
  for (int p = 0; p < KP; ++p) sum += sh.part[r][p * NS + col];
  return sum;
}

template <int NS, int KP>
__device__ __forceinline__ int row_sum(const SharedBeam& sh, int r) {
  int sum = 0;
  const int t = opaque_index(threadIdx.x);
#pragma unroll
  for (int p = 0; p < KP; ++p) sum += sh.part[r][p * NS + t];
  return sum;
}

// The row-wide phases run the entry's rows IN PARALLEL: row rw = wave / 2 is served by two waves (128
// threads, elements tt + 128 i of the row); waves of absent rows only keep the barriers company.
struct RowLane {
  int rw, tt;
  bool on;
  __device__ __forceinline__ RowLane(int nr) : rw(threadIdx.x >> 7), tt(threadIdx.x & 127), on((threadIdx.x >> 7) < nr) {}
};

// absmax over a row's 128 threads (one value per thread); every thread of the row returns it
__device__ __forceinline__ float row_absmax(float v, const RowLane& rl, SharedBeam& sh) {
  const float m = wave_max_dpp(fabsf(v));
  lds_barrier();  // (red is reused)
  if ((threadIdx.x & 63) == 0) sh.red[rl.rw][(threadIdx.x >> 6) & 1] = m;
  lds_barrier();
  return fmaxf(sh.red[rl.rw][0], sh.red[rl.rw][1]);
}

// LayerNorm / RMSNorm + Quantize of every row of the entry: sh.x_row[r] -> sh.xq[r], sh.rs[r].
// gamma / beta of a thread's four elements come from LDS-free registers: the caller passes NormRegs4.
struct NormRegs4 {
  float gamma[4], beta[4], eps;
  bool layer_norm, rms_use_residual;
  __device__ __forceinline__ void request(const ClusterNormW& n) {
    layer_norm = n.beta != nullptr;
    rms_use_residual = n.rms_use_residual != 0;
    eps = n.eps;
    const int tt = opaque_index(threadIdx.x) & 127;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      gamma[i] = n.gamma[tt + 128 * i];
      beta[i] = layer_norm ? n.beta[tt + 128 * i] : 0.f;
    }
  }
};

__device__ __forceinline__ void norm_quantize_rows(int nr, const NormRegs4& ln, int quant_mode, SharedBeam& sh) {
  const RowLane rl(nr);
  constexpr int pad = 68;
  float v[4], y[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[i] = rl.on ? sh.x_row[rl.rw][rl.tt + 128 * i] : 0.f;
    if (rl.on) sh.stage[rl.rw][norm_chain_slot(rl.tt + 128 * i, kD, pad)] = v[i];
  }
  lds_barrier();
  float sum = 0.f, sumsq = 0.f;
  if (rl.on) ordered_row_stats_t<true, 0>(sh.stage[rl.rw], pad, kD, sum, sumsq);
  const NormStats st = norm_stats(sum, sumsq, kD, ln.eps, ln.layer_norm);
  float m = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    y[i] = norm_apply(v[i], ln.gamma[i], ln.beta[i], st, ln.layer_norm, ln.rms_use_residual, rl.tt + 128 * i);
    m = fmaxf(m, fabsf(y[i]));
  }
  const float amax = row_absmax(m, rl, sh);
  const float scale = (amax != 0.f) ? 127.f / amax : 1.f;
  if (rl.on) {
#pragma unroll
    for (int i = 0; i < 4; ++i) sh.xq[rl.rw][rl.tt + 128 * i] = to_s8(y[i], scale, quant_mode);
    if (rl.tt == 0) sh.rs[rl.rw] = scale;
  }
  lds_barrier();
}

#define CT2X_CL_STAMP(i)                                                                     \
  if (a.stamps != nullptr && blockIdx.x == 0 && t == 0) a.stamps[stamp_base + (i)] = wall_clock64();

// grid = batch * 8 workgroups of 512 threads; blockIdx = entry * 8 + member; rows entry * beam + r
__global__ __launch_bounds__(kThreads) void decoder_cluster_beam_kernel(ClusterStackArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  SharedBeam& sh = *reinterpret_cast<SharedBeam*>(smem_raw);
  float* tp_base = reinterpret_cast<float*>(smem_raw + sizeof(SharedBeam));  // [kBeamRows][t_pad]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int member = blockIdx.x % kMembers, entry = blockIdx.x / kMembers;
  // beam search: the cluster's rows are the hypotheses of batch entry `entry` (they share its
  // encoder memory); greedy batches of more than 32 rows: rows_per_cluster independent rows
  const bool shared = a.beam > 1;
  const int row0 = entry * a.rows_per_cluster;
  const int nr = min(a.rows_per_cluster, a.rows - row0);  // rows of this cluster (<= kBeamRows)
  Handover<true> ho;
  ho.lines = a.xch + static_cast<size_t>(entry) * kBeamRows * kXchFloats * 2;
  ho.counter = a.counters + static_cast<size_t>(entry) * 16;
  ho.error = a.error;
  ho.arrivals = 0u;
  if (__hip_atomic_load(a.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
  if (t == 0) sh.dead = 0;
  const unsigned epoch = ho.counter[1];
  ho.tag = epoch * 64u + 1u;
  const int step = *a.step_ptr;
  const int mem_base = shared ? entry : row0;  // first memory (length, K / V) of the cluster's rows
  int stamp_base = 0;
  CT2X_CL_STAMP(0)
  stamp_base = 1;
  // the reference's beam search has not expanded the batch at its first step: one query row per
  // entry, the one-row forms of the encoder-decoder attention (src/decoding.cc:453-454, 700-712)
  const int cross_m_rows = (!shared || step == a.one_row_step) ? 1 : nr;

  GemvRows<16> g_in;
  GemvRows<4> g_out;
  ColRegs c_in, c_out;
  NormRegs4 n_in;
  const RowLane rl(nr);
  const int qkv_col = (t >> 6) * kD + member * kDH + (t & 63);
  const int out_col = member * kDH + (t & 63);
  g_in.request<192, 2, 16>(a.layers[0].qkv.w + static_cast<size_t>(member) * 192 * kD);
  c_in.request(a.layers[0].qkv, qkv_col, t < 192);
  n_in.request(a.layers[0].ln_self);

  if (rl.on) {  // embedding rows (embed_kernel, rowops.hip)
    const int64_t id = a.ids[row0 + rl.rw];
    const float* p = a.pos_table ? a.pos_table + static_cast<int64_t>(step) * kD : nullptr;
    const float r_s = a.emb_int8 ? 1.f / a.emb_scale[id] : 1.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = rl.tt + 128 * i;
      float v = a.emb_int8 ? static_cast<float>(static_cast<const int8_t*>(a.emb_table)[id * kD + e]) * r_s
                           : static_cast<const float*>(a.emb_table)[id * kD + e];
      if (a.emb_apply_mul) v = v * a.emb_mul;
      if (p) v = v + p[e];
      sh.x_row[rl.rw][e] = v;
    }
  }
  // (a thread reads back only what it wrote: x_row[rw][tt + 128 i])

  for (int l = 0; l < a.n_layers; ++l) {
    const ClusterLayerArgs& L = a.layers[l];
    CT2X_CL_STAMP(0)
#pragma unroll 1
    for (int sub = 0; sub < 2; ++sub) {
      const bool self = sub == 0;
      const ClusterDenseW& W_out = self ? L.self_out : L.cross_out;
      const int n_keys = self ? max(0, min(step, a.t_max)) + 1 : a.s_max;
      const int64_t kv_row_stride = static_cast<int64_t>(kMembers) * a.t_max * kDH;  // self caches
      const int64_t mem_stride = shared ? 0 : static_cast<int64_t>(a.s_max) * (2 * kD);  // memories of the rows
      const float* kc_cross = L.cross_kv + static_cast<int64_t>(mem_base) * a.s_max * (2 * kD) + member * kDH;
      float* self_k0 = L.self_k + (static_cast<int64_t>(row0) * kMembers + member) * a.t_max * kDH;
      float* self_v0 = L.self_v + (static_cast<int64_t>(row0) * kMembers + member) * a.t_max * kDH;
      // keys before this step (self) / valid memory length (cross) of the cluster's row r
      auto cached_of = [&](int r) {
        return self ? max(0, min(step, a.t_max)) : max(0, min(a.mem_lengths[mem_base + (shared ? 0 : r)], a.s_max));
      };
      // the tile of row r. Self and independent memories: rows 0, 1 are requested here and stored
      // after the projection, rows 2, 3 then travel in the same registers (one exposed trip to the
      // L2 instead of a second attention round); a shared memory: one tile. (The first pair is
      // DEFINED on every path - an absent row re-reads the first: a conditional definition would
      // keep 32 registers alive through the whole launch.)
      // Measured (profiles/r06_beam_four_tiles_ab.txt, beam 4): the second pair requested right
      // behind the statistics chains 85.3 - 86.9k tokens/s; requested here with the next phase's
      // weight request AHEAD of its stores 86.9 - 87.1k; this order 87.7 - 87.9k (two tiles, two
      // rounds: 83.8k).
      auto request_tile = [&](AttnStage64& st, int r) {
        if (self)
          st.request(self_k0 + r * kv_row_stride, self_v0 + r * kv_row_stride, kDH, cached_of(r), n_keys,
                     a.ancestry ? a.ancestry + row0 + r : nullptr, a.rows, kv_row_stride, row0 + r);
        else
          st.request(kc_cross + r * mem_stride, kc_cross + r * mem_stride + kD, 2 * kD, cached_of(r), n_keys);
      };
      const bool per_row_tiles = self || !shared;
      AttnStage64 stage_a[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) request_tile(stage_a[i], (i < nr) ? i : 0);
      norm_quantize_rows(nr, n_in, a.quant_mode, sh);
      if (self) g_in.finish<192, 2, 16>(sh, nr);
      else g_in.finish<64, 8, 4>(sh, nr);
      if (t < (self ? 192 : 64)) {
#pragma unroll 1
        for (int r = 0; r < nr; ++r) {
          const int sum = self ? row_sum<192, 2>(sh, r) : row_sum<64, 8>(sh, r);
          sh.outv[r][t] = dense_epilogue(sum, 1.f / sh.rs[r], c_in.wscale, c_in.has_bias, c_in.bias, -1);
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) stage_a[i].store(sh.ks[i], sh.vs[i]);
      const bool more_tiles = per_row_tiles && nr > 2;  // (uniform over the workgroup)
      if (more_tiles) {
#pragma unroll
        for (int i = 0; i < 2; ++i) request_tile(stage_a[i], (2 + i < nr) ? 2 + i : 2);
      }
      if (more_tiles) {
        lds_barrier();  // (the partial sums have been read: tile 3 lies over them)
        stage_a[0].store(sh.ks[2], sh.vs[2]);
        stage_a[1].store(sh.ks3, sh.vs3);
      }
      g_out.request<64, 8, 4>(W_out.w + static_cast<size_t>(member) * 64 * kD);
      c_out.request(W_out, out_col, t < 64 * kBeamRows);
      lds_barrier();
      CT2X_CL_STAMP(1 + 5 * sub)
      // ---- attention: ONE call site (the core keeps the register file), ONE round: wave w = row w,
      // on its own tile (self, independent memories) or on the shared memory tile (beam search).
      const float att_scale = self ? L.self_scale : L.cross_scale;
      // (The loop runs ONCE - an entry has at most four rows - and its rd > 0 arm never: they are
      // what round 5's two-round form looked like, and with them the register allocator finds the
      // assignment it found then (13 values in scratch memory, none reloaded on a hot path); as
      // straight-line code, or as a one-trip loop without the arm, the same kernel kept 42 - 73 values
      // there and every phase of the launch slowed down: 84.1k -> 80.7k tokens/s instead of a gain.)
      const int rounds = per_row_tiles ? (nr + 3) / 4 : 1;
#pragma unroll 1
      for (int rd = 0; rd < rounds; ++rd) {
        if (rd > 0) {  // (never: an entry has at most four rows; the loop keeps the core a region of its own)
          lds_barrier();
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            AttnStage64 next;
            request_tile(next, (2 + i < nr) ? 2 + i : 2);
            next.store(sh.ks[i], sh.vs[i]);
          }
          lds_barrier();
        }
        const int r = per_row_tiles ? rd * 4 + wave : wave;
        const bool mine = per_row_tiles ? (wave < 4 && r < nr) : (wave < nr);
        if (mine) {
          float* tq = sh.tq[r];
          float* tp = tp_base + static_cast<size_t>(r) * a.t_pad;
          const int tile = per_row_tiles ? wave : 0;
          const int t_cached = cached_of(r);
          const float qd = sh.outv[r][lane];
          const float knew = self ? sh.outv[r][64 + lane] : 0.f, vnew = self ? sh.outv[r][128 + lane] : 0.f;
          float* kc_w = self_k0 + r * kv_row_stride;
          float* vc_w = self_v0 + r * kv_row_stride;
          if (self && t_cached < a.t_max) {  // the row appends to its OWN cache row
            kc_w[t_cached * kDH + lane] = knew;
            vc_w[t_cached * kDH + lane] = vnew;
          }
          const float* kc_m = kc_cross + r * mem_stride;
          const float ctx = attn_exact_wave64(
              qd, self, knew, vnew, self ? kc_w : kc_m, self ? vc_w : kc_m + kD, self ? kDH : 2 * kD, t_cached, n_keys,
              self ? t_cached + 1 : t_cached, att_scale, tq, tp, a.t_pad, sh.k_tile(tile), sh.v_tile(tile),
              self ? 1 : cross_m_rows, (self && a.ancestry) ? a.ancestry + row0 + r : nullptr, a.rows, kv_row_stride,
              row0 + r);
          ho.publish(r * kXchFloats + kXchCtx + member * kDH + lane, ctx);
        }
      }
      CT2X_CL_STAMP(2 + 5 * sub)
      ho.await(sh);
      CT2X_CL_STAMP(3 + 5 * sub)
      // ---- attention output projection + residual, every row -----------------------------------
      {
        float c[4] = {0.f, 0.f, 0.f, 0.f};
        if (rl.on) {
          const int base = rl.rw * kXchFloats + kXchCtx + rl.tt;
          const int idx[4] = {base, base + 128, base + 256, base + 384};
          ho.template collect<4>(idx, c, sh);
        }
        ho.done();
        const float amax = row_absmax(fmaxf(fmaxf(fabsf(c[0]), fabsf(c[1])), fmaxf(fabsf(c[2]), fabsf(c[3]))), rl, sh);
        const float scale = (amax != 0.f) ? 127.f / amax : 1.f;
        if (rl.on) {
#pragma unroll
          for (int i = 0; i < 4; ++i) sh.xq[rl.rw][rl.tt + 128 * i] = to_s8(c[i], scale, a.quant_mode);
          if (rl.tt == 0) sh.rs[rl.rw] = scale;
        }
      }
      lds_barrier();
      g_out.finish<64, 8, 4>(sh, nr);
      if (t < 64 * nr) {  // the rows' epilogues in parallel: thread t = (row t / 64, column t % 64)
        const int r = t >> 6;
        float v = dense_epilogue(row_sum_col<64, 8>(sh, r, t & 63), 1.f / sh.rs[r], c_out.wscale, c_out.has_bias,
                                 c_out.bias, -1);
        v = sh.x_row[r][out_col] + v;
        ho.publish(r * kXchFloats + kXchX + out_col, v);
      }
      CT2X_CL_STAMP(4 + 5 * sub)
      if (self) {
        g_in.request<64, 8, 4>(L.cross_q.w + static_cast<size_t>(member) * 64 * kD);
        c_in.request(L.cross_q, out_col, t < 64);
        n_in.request(L.ln_cross);
      } else {
        g_in.request<256, 2, 16>(L.ffn0.w + static_cast<size_t>(member) * 256 * kD);
        c_in.request(L.ffn0, member * 256 + (t & 255), t < 256);
        n_in.request(L.ln_ffn);
      }
      ho.await(sh);  // (also: the residual reads of x_row above are done)
      CT2X_CL_STAMP(5 + 5 * sub)
      if (rl.on) {
        float xr[4];
        const int base = rl.rw * kXchFloats + kXchX + rl.tt;
        const int idx[4] = {base, base + 128, base + 256, base + 384};
        ho.template collect<4>(idx, xr, sh);
#pragma unroll
        for (int i = 0; i < 4; ++i) sh.x_row[rl.rw][rl.tt + 128 * i] = xr[i];
      }
      ho.done();
    }
    // ================= feed forward ===========================================================
    norm_quantize_rows(nr, n_in, a.quant_mode, sh);
    g_in.finish<256, 2, 16>(sh, nr);
    if (t < 256) {
#pragma unroll 1
      for (int r = 0; r < nr; ++r)
        ho.publish(r * kXchFloats + kXchInner + member * 256 + t,
                   dense_epilogue(row_sum<256, 2>(sh, r), 1.f / sh.rs[r], c_in.wscale, c_in.has_bias, c_in.bias, a.act));
    }
    CT2X_CL_STAMP(11)
    GemvRows<16> g_f1;
    g_f1.request<64, 8, 16>(L.ffn1.w + static_cast<size_t>(member) * 64 * kFfn);
    c_out.request(L.ffn1, out_col, t < 64 * kBeamRows);
    ho.await(sh);
    CT2X_CL_STAMP(12)
    {
      float iv[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) iv[i] = 0.f;
      if (rl.on) {
        const int base = rl.rw * kXchFloats + kXchInner + rl.tt;
#pragma unroll
        for (int h = 0; h < 2; ++h) {  // two batches of eight polls
          int idx[8];
          float got[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) idx[i] = base + 128 * (h * 8 + i);
          ho.template collect<8>(idx, got, sh);
#pragma unroll
          for (int i = 0; i < 8; ++i) iv[h * 8 + i] = got[i];
        }
      }
      float m = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) m = fmaxf(m, fabsf(iv[i]));
      const float amax = row_absmax(m, rl, sh);
      const float scale = (amax != 0.f) ? 127.f / amax : 1.f;
      if (rl.on) {
#pragma unroll
        for (int i = 0; i < 16; ++i) sh.xq[rl.rw][rl.tt + 128 * i] = to_s8(iv[i], scale, a.quant_mode);
        if (rl.tt == 0) sh.rs[rl.rw] = scale;
      }
    }
    ho.done();
    lds_barrier();
    g_f1.finish<64, 8, 16>(sh, nr);
    const bool more = l + 1 < a.n_layers;
    if (t < 64 * nr) {  // (rows in parallel, as behind the attention)
      const int r = t >> 6;
      float v = dense_epilogue(row_sum_col<64, 8>(sh, r, t & 63), 1.f / sh.rs[r], c_out.wscale, c_out.has_bias,
                               c_out.bias, -1);
      v = sh.x_row[r][out_col] + v;
      if (more || a.q_out != nullptr) ho.publish(r * kXchFloats + kXchX + out_col, v);
      if (!more) a.x_out[static_cast<int64_t>(row0 + r) * kD + out_col] = v;
    }
    CT2X_CL_STAMP(13)
    stamp_base += kClusterStampsPerLayer;
    if (more || a.q_out != nullptr) {
      if (more) {
        const ClusterLayerArgs& N = a.layers[l + 1];
        g_in.request<192, 2, 16>(N.qkv.w + static_cast<size_t>(member) * 192 * kD);
        c_in.request(N.qkv, qkv_col, t < 192);
        n_in.request(N.ln_self);
      } else {
        n_in.request(a.out_norm);
      }
      ho.await(sh);
      if (rl.on) {
        float xr[4];
        const int base = rl.rw * kXchFloats + kXchX + rl.tt;
        const int idx[4] = {base, base + 128, base + 256, base + 384};
        ho.template collect<4>(idx, xr, sh);
#pragma unroll
        for (int i = 0; i < 4; ++i) sh.x_row[rl.rw][rl.tt + 128 * i] = xr[i];
      }
      ho.done();
      if (!more) {
        norm_quantize_rows(nr, n_in, a.quant_mode, sh);
#pragma unroll 1
        for (int r = 0; r < nr; ++r) {
          if (t < kDH) a.q_out[static_cast<int64_t>(row0 + r) * kD + out_col] = sh.xq[r][out_col];
          if (member == 0 && t == 0) a.qscale_out[row0 + r] = sh.rs[r];
        }
      }
    }
  }
  if (member == 0 && t == 0) ho.counter[1] = epoch + 1u;
  CT2X_CL_STAMP(0)
}
#undef CT2X_CL_STAMP

// ---- weight re-pack (once per model) ----------------------------------------------------------
// src [n, k] row-major int8 -> dst: member m's slice [k / 16][ns][16], ns = n / 8 rows; `sections`
// > 1: the Dense is the concatenation of `sections` sub-matrices of n / sections rows each (Q | K
// | V) and a member takes rows [m * ns / sections, ...) of EVERY section.
__global__ void pack_cluster_weight_kernel(const int8_t* __restrict__ src, int n, int k, int sections,
                                           int8_t* __restrict__ dst) {
  const int ns = n / kMembers, per = ns / sections, section_rows = n / sections, chunks = k / 16;
  const int64_t total = static_cast<int64_t>(n) * chunks;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int j = static_cast<int>(i % ns);
    const int c = static_cast<int>((i / ns) % chunks);
    const int m = static_cast<int>(i / (static_cast<int64_t>(ns) * chunks));
    const int srow = (j / per) * section_rows + m * per + (j % per);
    reinterpret_cast<uint4*>(dst)[i] =
        *reinterpret_cast<const uint4*>(src + static_cast<int64_t>(srow) * k + c * 16);
  }
}

void launch_pack_cluster_weight(const int8_t* w, int64_t n, int64_t k, int sections, int8_t* packed,
                                hipStream_t stream) {
  CT2X_CHECK(n % (kMembers * sections) == 0 && k % 16 == 0, "pack_cluster_weight: shape");
  hipLaunchKernelGGL(pack_cluster_weight_kernel, dim3(256), dim3(256), 0, stream, w, static_cast<int>(n),
                     static_cast<int>(k), sections, packed);
}

bool decoder_cluster_geometry_ok(int64_t d_model, int64_t heads, int64_t dhead, int64_t ffn) {
  return d_model == kD && heads == kMembers && dhead == kDH && ffn == kFfn;
}

// (sized for the tagged form: a 64-bit slot per line)
size_t decoder_cluster_xch_floats(int64_t rows) { return static_cast<size_t>(rows) * kXchFloats * 2; }
// several rows per cluster (beam search: a batch entry; greedy batches of more than 32 rows):
// kBeamRows line sets per cluster
size_t decoder_cluster_beam_xch_floats(int64_t clusters) { return static_cast<size_t>(clusters) * kBeamRows * kXchFloats * 2; }

bool decoder_cluster_lds_ok(int t_pad) { return sizeof(Shared) + static_cast<size_t>(t_pad) * sizeof(float) <= 64 * 1024; }

int decoder_cluster_t_pad(int64_t t_max, int64_t s_max) {
  const int64_t keys = (t_max > s_max ? t_max : s_max) + 1;
  return static_cast<int>((keys + 15) / 16 * 16 + 16);
}

namespace {
// CT2X_CLUSTER_HANDOVER=counter: the arrival-counter form of the hand-over (A/B runs)
bool cluster_handover_tagged() {
  static const bool tagged = [] {
    const char* e = std::getenv("CT2X_CLUSTER_HANDOVER");
    return e == nullptr || e[0] != 'c';
  }();
  return tagged;
}
}  // namespace

int decoder_cluster_proj_cols(int64_t rows, int64_t vocab, int64_t k) {
  static const bool enabled = [] {  // CT2X_CLUSTER_PROJ=0: the projection keeps its own launch (A/B runs)
    const char* e = std::getenv("CT2X_CLUSTER_PROJ");
    return e == nullptr || e[0] != '0';
  }();
  if (!enabled || !cluster_handover_tagged() || k != kD || rows < 1 || rows > 32 || vocab < 1) return 0;
  const int64_t wgs = rows * kMembers;
  const int64_t cols = ((vocab + wgs - 1) / wgs + kProjCols - 1) / kProjCols * kProjCols;
  return cols / kProjCols <= kClusterProjMaxPasses ? static_cast<int>(cols) : 0;
}

void launch_decoder_cluster(const ClusterStackArgs& a, hipStream_t stream) {
  CT2X_CHECK(a.n_layers >= 1 && a.n_layers <= kClusterMaxLayers && a.n_layers * 6 + 3 < 64, "decoder_cluster: layers");
  if (a.rows_per_cluster > 0) {  // beam search, or a greedy batch of more than 32 rows
    const int R = a.rows_per_cluster;
    CT2X_CHECK(R <= kBeamRows && (a.beam == 1 || (a.beam == R && a.ancestry != nullptr && a.rows % R == 0)),
               "decoder_cluster: rows per cluster");
    const int clusters = (a.rows + R - 1) / R;
    CT2X_CHECK(clusters <= 32, "decoder_cluster: more clusters than the device holds");
    const size_t lds = sizeof(SharedBeam) + static_cast<size_t>(kBeamRows) * a.t_pad * sizeof(float);
    CT2X_CHECK(lds <= 160 * 1024, "decoder_cluster: the score rows do not fit the LDS");
    // (the attribute belongs to the CURRENT device's function object: one flag per device, so that
    // every Engine of a multi-GPU process gets it, not only the first to launch)
    static std::atomic<bool> opted[64];
    int dev = 0;
    CT2X_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !opted[dev].load(std::memory_order_acquire)) {
      CT2X_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_cluster_beam_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      if (dev >= 0 && dev < 64) opted[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(decoder_cluster_beam_kernel, dim3(static_cast<unsigned>(clusters * kMembers)),
                       dim3(kThreads), lds, stream, a);
    return;
  }
  size_t lds = sizeof(Shared) + static_cast<size_t>(a.t_pad) * sizeof(float);
  CT2X_CHECK(lds <= 64 * 1024, "decoder_cluster: the score row does not fit the LDS");
  const bool tagged = cluster_handover_tagged();
  if (a.proj.w != nullptr) {
    CT2X_CHECK(tagged && a.proj.cols_per_wg > 0 && a.proj.cols_per_wg % kProjCols == 0 &&
                   a.proj.cols_per_wg / kProjCols <= kClusterProjMaxPasses &&
                   static_cast<int64_t>(a.rows) * kMembers * a.proj.cols_per_wg >= a.proj.vocab && a.rows <= 32 &&
                   a.proj.rows_q != nullptr && a.proj.seq != nullptr && a.proj.gf.slots != nullptr,
               "decoder_cluster: projection tail");
    lds = ((lds + 1023) & ~size_t(1023)) + kProjWBytes;
    static std::atomic<bool> opted[64];
    int dev = 0;
    CT2X_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !opted[dev].load(std::memory_order_acquire)) {
      CT2X_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_cluster_kernel<true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      if (dev >= 0 && dev < 64) opted[dev].store(true, std::memory_order_release);
    }
  }
  const dim3 grid(static_cast<unsigned>(a.rows * kMembers)), block(kThreads);
  if (tagged) hipLaunchKernelGGL(decoder_cluster_kernel<true>, grid, block, lds, stream, a);
  else hipLaunchKernelGGL(decoder_cluster_kernel<false>, grid, block, lds, stream, a);
}

bool decoder_cluster_beam_lds_ok(int t_pad) {
  return sizeof(SharedBeam) + static_cast<size_t>(kBeamRows) * t_pad * sizeof(float) <= 160 * 1024;
}

}  // namespace ct2x
