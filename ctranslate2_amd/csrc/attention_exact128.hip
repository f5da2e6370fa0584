// Decode attention for d_head = 128 that reproduces the reference CPU path BIT FOR BIT
// (Llama-3-8B geometry: 32 query heads on 8 KV heads, RoPE, caches of hundreds of positions).
//
// What has to be reproduced (src/layers/attention.cc:178-287 on Device::CPU; restated and pinned
// bit for bit against the reference library in oracle/ct2_oracle.c: mkl_dot_main / mkl_dot_tail /
// mkl_pv_m1_col / orc_softmax, tests/test_oracle.py::test_exact_matmul_decode_*):
//   * RotaryEmbeddings::apply on q and the new key (src/ops/rotary_cpu.cc:9-62: two products and a
//     sum, no FMA), cache append in place of ops::Concat (attention.cc:541-546);
//   * QK^T = MKL sgemm with ONE query row per (row, head): every score is a dot product over the
//     128 channels with SIXTEEN lane accumulators (lane (t - 1) % 16 for channel t >= 1, channel 0
//     apart), folded by a 8 / 4 / 2 / 1 tree; the keys in the groups of four take the "main" form
//     (7 chunks accumulate, the 8th goes through a second tree seeded with the first tree's sum),
//     the n % 4 last keys the "tail" forms (type A: thirty-two accumulators over the first 96
//     channels; type B: sixteen throughout; channel 0 added last);
//   * SoftMax = cpu::softmax<AVX2> (exp256_ps, eight strided partial sums, e * (1 / sum));
//   * P.V = MKL's one-row form: blocks of eight keys (C + D) + (A + B), then one FMA per key.
//
// MI355X mapping - the 16 accumulators of MKL's dot product ARE the 16 lanes of a DPP row:
//   * one workgroup (8 waves) per (row, KV head): the G query heads of a GQA group share every K
//     and V byte the workgroup reads (the reference replicates the KV heads, attention.cc:291-295);
//   * K and V travel global -> LDS in tiles of 64 rows by global_load_lds_dwordx4 (1 KB per wave
//     instruction, double buffered, the next tile in flight under the arithmetic of this one);
//   * QK^T: a 16-lane row owns one key at a time - lane l holds channels l + 1, l + 17, ... of the
//     key in 8 registers (conflict-free 4-byte LDS reads), q of all G heads in G x 8 registers; the
//     16-accumulator tree is four row_shl DPP adds. A wave works on 4 keys per pass;
//   * SoftMax: one wave per head on the scores in LDS;
//   * P.V: lane = (head of the group, channel), values from the LDS tile.
// No matrix cores: the order of every addition is prescribed; the kernel is bound by the K / V
// stream (2 * T * 512 B per (row, KV head)) and by the dependent add chains of the order.

#include "device_utils.h"
#include "kernels.h"
#include "runtime.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <float.h>

namespace ct2x {

namespace {

constexpr int X128_DH = 128;
constexpr int X128_WAVES = 8;
constexpr int X128_VB = 16;  // V rows per batch

// tools/attn128_probe.cpp defines this to stamp the phases with s_memtime
#ifndef CT2X_X128_STAMP
#define CT2X_X128_STAMP(i)
#define CT2X_X128_TICK(i)
#endif
// tools/attn128_probe.cpp: timing experiments that end the kernel after phase N (results are wrong)
#ifndef CT2X_X128_STOP_AFTER
#define CT2X_X128_STOP_AFTER 99
#endif

struct AttnX128Args {
  AttnDecodeParams p;
  int group;   // query heads per KV head
  int t_pad;   // score slots per head (multiple of 16, >= keys + 16)
  int n_buf;   // K / V tiles in the LDS ring
};

// a[l] + a[l + w] for the lanes l < w of every 16-lane row (w = 8, 4, 2, 1): the sum lands in lane
// 0 of the row; the other lanes hold partial garbage that is never used.
__device__ __forceinline__ float row_tree16(float a) {
  a = a + dpp_move<0x108, 0xf>(0.f, a);  // row_shl:8  lane l <- lane l + 8
  a = a + dpp_move<0x104, 0xf>(0.f, a);
  a = a + dpp_move<0x102, 0xf>(0.f, a);
  a = a + dpp_move<0x101, 0xf>(0.f, a);
  return a;
}

// One score of MKL's one-row product in a 16-lane row. kr[i] / qr[i]: channel (16 i + l + 1) & 127
// of the key / the query in lane l - so (i = 7, l = 15) holds channel 0. form: 0 main, 1 tail type
// A, 2 tail type B (oracle/ct2_oracle.c mkl_dot_main / mkl_dot_tail). Result in lane 0 of the row.
__device__ __forceinline__ float x128_dot(const float (&qr)[8], const float (&kr)[8], int l, int form) {
  const float last = qr[7] * kr[7];             // lanes 0..14: channel 113 + l; lane 15: channel 0
  const float p0 = dpp_move<0x121, 0xf>(0.f, last);  // row_ror:1 - lane 0 <- lane 15 (q0 * k0)
  if (form == 0) {
    float acc = (l == 0) ? p0 : 0.f;
#pragma unroll
    for (int i = 0; i < 7; ++i) acc = fmaf(qr[i], kr[i], acc);
    const float s = row_tree16(acc);
    float r = (l == 0) ? fmaf(qr[7], kr[7], s) : (l < 15 ? last : 0.f);
    return row_tree16(r);
  }
  float acc;
  if (form == 1) {
    float lo = qr[0] * kr[0], hi = qr[1] * kr[1];
    lo = fmaf(qr[2], kr[2], lo);
    hi = fmaf(qr[3], kr[3], hi);
    lo = fmaf(qr[4], kr[4], lo);
    hi = fmaf(qr[5], kr[5], hi);
    acc = lo + hi;
    acc = fmaf(qr[6], kr[6], acc);
  } else {
    acc = qr[0] * kr[0];
#pragma unroll
    for (int i = 1; i < 7; ++i) acc = fmaf(qr[i], kr[i], acc);
  }
  if (l < 15) acc = fmaf(qr[7], kr[7], acc);
  const float s = row_tree16(acc);
  return s + p0;
}

// The main form for the G heads of the group at once (branch free; the heads' chains interleave).
template <int G>
__device__ __forceinline__ void x128_dot_main(const float (&qr)[G][8], const float (&kr)[8], int l,
                                              float (&out)[G]) {
  float acc[G], last[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    last[g] = qr[g][7] * kr[7];                              // lane 15: q0 * k0
    const float p0 = dpp_move<0x121, 0xf>(0.f, last[g]);     // row_ror:1 - lane 0 <- lane 15
    acc[g] = (l == 0) ? p0 : 0.f;
  }
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = fmaf(qr[g][i], kr[i], acc[g]);
#pragma unroll
  for (int g = 0; g < G; ++g) acc[g] = acc[g] + dpp_move<0x108, 0xf>(0.f, acc[g]);
#pragma unroll
  for (int g = 0; g < G; ++g) acc[g] = acc[g] + dpp_move<0x104, 0xf>(0.f, acc[g]);
#pragma unroll
  for (int g = 0; g < G; ++g) acc[g] = acc[g] + dpp_move<0x102, 0xf>(0.f, acc[g]);
#pragma unroll
  for (int g = 0; g < G; ++g) acc[g] = acc[g] + dpp_move<0x101, 0xf>(0.f, acc[g]);
  // second tree: lane 0 fuses its product onto the first tree's sum, lanes 1 .. 14 hold rounded
  // products, lane 15 holds 0. Branch free: fma(q, k, -0) == RN(q * k) for every q * k (also for a
  // zero product of either sign), so ONE fma with a per-lane addend serves lanes 0 .. 14.
  float r[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const float addend = (l == 0) ? acc[g] : -0.f;
    const float f = fmaf(qr[g][7], kr[7], addend);
    r[g] = (l < 15) ? f : 0.f;
  }
#pragma unroll
  for (int g = 0; g < G; ++g) r[g] = r[g] + dpp_move<0x108, 0xf>(0.f, r[g]);
#pragma unroll
  for (int g = 0; g < G; ++g) r[g] = r[g] + dpp_move<0x104, 0xf>(0.f, r[g]);
#pragma unroll
  for (int g = 0; g < G; ++g) r[g] = r[g] + dpp_move<0x102, 0xf>(0.f, r[g]);
#pragma unroll
  for (int g = 0; g < G; ++g) out[g] = r[g] + dpp_move<0x101, 0xf>(0.f, r[g]);
}

// MKL's "main" one-row dot product with FOUR lanes per key (round 3, second form: the 16-lane form
// above spends most of its instructions on the two 16-way trees; here a lane owns four of MKL's
// sixteen accumulators, the trees are two quad permutes and three in-lane adds, and a wave works
// on 16 keys per pass). Lane j of a quad holds channels 16 c + 4 j + e (c = 0 .. 7, e = 0 .. 3) of
// the key (kq[c][e]) and of the query (qq[c][e]); channel t >= 1 belongs to MKL accumulator
// (t - 1) % 16 = (4 j + e - 1) % 16:  lane 0: {15, 0, 1, 2}, lane 1: {3 .. 6}, lane 2: {7 .. 10},
// lane 3: {11 .. 14}. Channel 0 seeds accumulator 0; channels 1 .. 112 accumulate; the tree
// 8 / 4 / 2 / 1 gives s; the 15 channels 113 .. 127 go through a second tree whose element 0 fuses
// its product onto s (oracle/ct2_oracle.c mkl_dot_main). Result in lane 0 of the quad.
__device__ __forceinline__ float x128_quad_tree(const float (&x)[4], int j) {
  // level 1: a[l] += a[l + 8] = quad lane ^ 2 (a7 + a15 lands in lane 0 slot 0); level 2:
  // a[l] += a[l + 4] = quad lane ^ 1; levels 3, 4 in lane 0: (e1 + e3) + (e2 + e0)
  float y[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) y[e] = x[e] + dpp_move<0x4E, 0xf>(0.f, x[e]);   // quad_perm [2,3,0,1]
#pragma unroll
  for (int e = 0; e < 4; ++e) y[e] = y[e] + dpp_move<0xB1, 0xf>(0.f, y[e]);   // quad_perm [1,0,3,2]
  (void)j;
  return (y[1] + y[3]) + (y[2] + y[0]);
}

template <int H>
__device__ __forceinline__ void x128_dot_main_quad(const float (&qq)[H][8][4], const v4f (&kq)[8], int j,
                                                   float (&out)[H]) {
  float acc[H][4];
#pragma unroll
  for (int h = 0; h < H; ++h) {
    // chunk 0: channel 0 (lane 0, e = 0) is the seed of accumulator 0 (lane 0, e = 1); accumulator
    // 15 (lane 0, e = 0) starts at channel 16
    // (accumulators start at +0 and take an FMA - fma(q, k, +0), not the bare product: the two
    // differ in the sign of a zero product)
    const float p0 = qq[h][0][0] * kq[0][0];
    const float f0 = fmaf(qq[h][0][0], kq[0][0], 0.f);
    acc[h][0] = (j == 0) ? 0.f : f0;
    acc[h][1] = fmaf(qq[h][0][1], kq[0][1], (j == 0) ? p0 : 0.f);
    acc[h][2] = fmaf(qq[h][0][2], kq[0][2], 0.f);
    acc[h][3] = fmaf(qq[h][0][3], kq[0][3], 0.f);
  }
#pragma unroll
  for (int c = 1; c < 7; ++c)
#pragma unroll
    for (int h = 0; h < H; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[h][e] = fmaf(qq[h][c][e], kq[c][e], acc[h][e]);
#pragma unroll
  for (int h = 0; h < H; ++h) {
    // channel 112 (chunk 7, lane 0, e = 0) is the last element of accumulator 15
    const float f = fmaf(qq[h][7][0], kq[7][0], acc[h][0]);
    acc[h][0] = (j == 0) ? f : acc[h][0];
  }
#pragma unroll
  for (int h = 0; h < H; ++h) {
    const float s = x128_quad_tree(acc[h], j);             // lane 0
    // second tree: r[4 j + e - 1] = q k of channel 112 + 4 j + e; r[0] (lane 0, e = 1) fuses onto s;
    // r[15] (lane 0, e = 0) = 0. fma(q, k, -0) == RN(q k) bit for bit.
    float r[4];
    r[0] = (j == 0) ? 0.f : qq[h][7][0] * kq[7][0];
    r[1] = fmaf(qq[h][7][1], kq[7][1], (j == 0) ? s : -0.f);
    r[2] = qq[h][7][2] * kq[7][2];
    r[3] = qq[h][7][3] * kq[7][3];
    out[h] = x128_quad_tree(r, j);
  }
}

// vectorized_reduce_all<AVX2> over e[0 .. size): eight strided partial sums, 4 / 2 / 1 tree, scalar
// tail (device_utils.h ordered_sum8), with the addends of a chain requested ahead of its adds.
__device__ __forceinline__ float x128_ordered_sum8(const float* e, int size) {
  if (size <= 8) return ordered_sum8(e, size);
  const int l = threadIdx.x & 7;
  const int vsize = size & ~7;
  float acc = 0.f;
  int j = l;
  for (; j + 8 * 15 < vsize; j += 8 * 16) {
    float a[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = e[j + 8 * k];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc = acc + a[k];
  }
  for (; j < vsize; j += 8) acc = acc + e[j];
  acc = acc + dpp_xor4(acc);
  acc = acc + dpp_move<0x4E, 0xf>(0.f, acc);
  acc = acc + dpp_move<0xB1, 0xf>(0.f, acc);
  for (int t = vsize; t < size; ++t) acc = acc + e[t];
  return acc;
}

typedef __attribute__((address_space(3))) void x128_lds_void;
typedef const __attribute__((address_space(1))) void x128_glb_void;

constexpr int X128_TILE = 64;                    // keys per staged K / V tile
constexpr int X128_TILE_FLOATS = X128_TILE * X128_DH;  // 32 KB

// One K or V tile of 64 cache rows, global -> LDS by global_load_lds_dwordx4 (1 KB per wave
// instruction, no staging registers; measured: with per-lane dword loads this kernel was bound by
// the ISSUE of its ~2600 vector-memory instructions per workgroup - 19k cycles for QK^T and 29k for
// P.V at 255 keys, profiles/r03_attn128_probe_a.txt - not by bytes or arithmetic). Wave w moves the
// rows 8 w .. 8 w + 7, two per instruction: lane -> (row parity, 16-byte unit). The DMA writes LDS
// lane-linearly, so the bank swizzle is applied to the SOURCE address: row r is stored rotated by
// 16 (r & 3) floats, which puts the four key rows that one 16-lane group of a ds_read_b128 touches
// (QK^T: four lanes per key) on four different 64-byte bank quarters.
// Rows past the cache are clamped to the allocation (t_max); they never enter a result.
__device__ __forceinline__ void x128_stage(const float* base, int kvs, int t0, int t_max, float* tile,
                                           int wave, int lane) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = 2 * (wave * 4 + j) + (lane >> 5);
    const int u = ((lane & 31) + 4 * (r & 3)) & 31;  // source unit of LDS unit (lane & 31)
    const int t = min(t0 + r, t_max - 1);
    __builtin_amdgcn_global_load_lds((x128_glb_void*)(base + static_cast<int64_t>(t) * kvs + 4 * u),
                                     (x128_lds_void*)(tile + (wave * 4 + j) * 256), 16, 0, 0);
  }
}
// Waits until at most `later` tiles (4 DMA instructions per wave each) are still in flight.
__device__ __forceinline__ void x128_wait_tiles(int later) {
  if (later <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if (later == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (later == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
}
// Workgroup barrier that does NOT drain the vector-memory queue (__syncthreads() carries a fence
// that waits for every outstanding DMA): LDS traffic of this wave is complete, then s_barrier.
__device__ __forceinline__ void x128_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}
// float index of channel ch of tile row r
__device__ __forceinline__ int x128_slot(int r, int ch) { return r * X128_DH + ((ch - 16 * (r & 3)) & 127); }

// The leading scalars are what the launch's FIRST memory requests need (they are preloaded into
// SGPRs at wave launch, Makefile -amdgpu-kernarg-preload-count; the AttnX128Args block costs an
// s_load round trip): the cache length and the first K tile leave before anything else is known.
//   geom = t_max | heads_kv << 24
template <int G>
__global__ __launch_bounds__(X128_WAVES * 64) void attn_x128_decode_kernel(
    const float* __restrict__ k_cache_e, const float* __restrict__ v_cache_e,
    const float* __restrict__ q_e, const int32_t* __restrict__ step_ptr_e, int kv_row_stride_e,
    int kv_head_stride_e, int kvs_e, int q_row_stride_e, unsigned geom_e, int length_e,
    AttnX128Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int DH = X128_DH;
  constexpr int THREADS = X128_WAVES * 64;
  constexpr int CPW = 64 / G;        // P.V: channels per wave item
  constexpr int ITEMS = DH / CPW;    // 2 G
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (in an SGPR: wave-uniform branches)
  const int t_max_e = static_cast<int>(geom_e & 0xffffffu), heads_kv_e = static_cast<int>(geom_e >> 24);
  const int row = blockIdx.x / heads_kv_e;
  const int hk = blockIdx.x % heads_kv_e;
  const int l = lane & 15, grp = lane >> 4;
  const int NB = a.n_buf;                               // tiles in the LDS ring (2 .. 4)
  float* stage = smem;                                  // [NB][64][128] K / V tiles (DMA targets)
  float* qs = stage + NB * X128_TILE_FLOATS;            // [G][128] rotated queries
  float* knew_s = qs + G * DH;                          // [128] rotated new key
  float* vnew_s = knew_s + DH;                          // [128]
  float* sc = vnew_s + DH;                              // [G][t_pad] scores, then probabilities

  CT2X_X128_STAMP(0)
  // ---- requests that need nothing but the preloaded arguments ---------------------------------
  int t_cached = length_e;
  if (step_ptr_e != nullptr) t_cached = *step_ptr_e;
  const float* kc_e = k_cache_e + static_cast<int64_t>(row) * kv_row_stride_e + static_cast<int64_t>(hk) * kv_head_stride_e;
  const float* vc_e = v_cache_e + static_cast<int64_t>(row) * kv_row_stride_e + static_cast<int64_t>(hk) * kv_head_stride_e;
  x128_stage(kc_e, kvs_e, 0, t_max_e, stage, wave, lane);
  __builtin_amdgcn_sched_barrier(0);  // the argument block's s_loads stay below the requests
  CT2X_X128_STAMP(1)

  const AttnDecodeParams& p = a.p;
  const int t_pad = a.t_pad;
  t_cached = max(0, min(t_cached, static_cast<int>(p.t_max)));
  const bool has_new = p.k_new != nullptr;
  const int pos = t_cached;
  const int n_keys = t_cached + (has_new ? 1 : 0);
  const int n_tiles = (n_keys + X128_TILE - 1) / X128_TILE;
  const int kvs = kvs_e;
  float* kc = const_cast<float*>(kc_e);
  float* vc = const_cast<float*>(vc_e);

  // ---- prologue: RoPE on q (G heads) and the new key, cache append --------------------------
  {
    const int rd = static_cast<int>(p.rot_dim), middle = rd / 2;
    const bool has_rope = p.rope_sin != nullptr;
    for (int idx = tid; idx < (G + 2) * DH; idx += THREADS) {
      const int which = idx / DH, c = idx % DH;  // 0 .. G-1: query heads, G: new key, G + 1: new value
      if (which > G) {
        if (has_new) {
          const float v = p.v_new[static_cast<int64_t>(row) * p.new_row_stride + hk * DH + c];
          vnew_s[c] = v;
          vc[pos * kvs + c] = v;
        }
        continue;
      }
      if (which == G && !has_new) continue;
      const float* src = which < G
                             ? q_e + static_cast<int64_t>(row) * q_row_stride_e + (hk * G + which) * DH
                             : p.k_new + static_cast<int64_t>(row) * p.new_row_stride + hk * DH;
      float x = src[c];
      if (has_rope && c < rd) {
        int partner;
        bool neg;
        if (p.rope_interleave) {
          partner = c ^ 1;
          neg = (c & 1) == 0;
        } else {
          partner = c < middle ? c + middle : c - middle;
          neg = c < middle;
        }
        const float xo = src[partner];
        const float cs = p.rope_cos[static_cast<int64_t>(pos) * rd + c];
        const float sv = p.rope_sin[static_cast<int64_t>(pos) * rd + c];
        const float sn = neg ? -sv : sv;  // the sign of rot(x) folded into sin: exact
        x = x * cs + xo * sn;
      }
      if (which < G) {
        qs[which * DH + c] = x;
      } else {
        knew_s[c] = x;
        kc[pos * kvs + c] = x;  // ops::Concat of the cache (attention.cc:541-546), in place
      }
    }
  }
  CT2X_X128_STAMP(2)
  if (CT2X_X128_STOP_AFTER <= 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  // Tile pipeline (K tiles, then V tiles, ONE stream of 2 n_tiles tiles through a ring of NB LDS
  // buffers): tile s sits in buffer s % NB. Before tile s is used its own DMA is waited for (counted
  // vmcnt: the NB - 2 younger tiles stay in flight), a barrier makes everybody's share visible - and
  // says tile s - 1 has been consumed, whose buffer the request of tile s + NB - 1 then takes.
  // (The prologue's cache-append stores share the vmcnt queue: they are drained once, below, before
  // the counted waits start.)
  const int n_stream = 2 * n_tiles;
#define CT2X_X128_REQUEST(S)                                                                      \
  {                                                                                               \
    const int rq_ = (S);                                                                           \
    if (rq_ < n_stream) {                                                                          \
      const bool rv_ = rq_ >= n_tiles;                                                            \
      x128_stage(rv_ ? vc : kc, kvs, (rv_ ? rq_ - n_tiles : rq_) * X128_TILE, t_max_e,            \
                 stage + (rq_ % NB) * X128_TILE_FLOATS, wave, lane);                               \
    }                                                                                             \
  }
#define CT2X_X128_ACQUIRE(S)                                                                      \
  {                                                                                               \
    const int s_ = (S);                                                                           \
    if (s_ == 0) {                                                                                \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                            \
      x128_barrier();                                                                             \
      for (int n_ = 1; n_ < NB; ++n_) CT2X_X128_REQUEST(n_)                                       \
    } else {                                                                                      \
      x128_wait_tiles(min(NB - 2, n_stream - 1 - s_));                                            \
      x128_barrier();                                                                             \
      CT2X_X128_REQUEST(s_ + NB - 1)                                                              \
    }                                                                                             \
  }

  // ---- QK^T ------------------------------------------------------------------------------------
  // Main keys (the groups of four): four lanes per key, 16 keys per wave pass; wave w takes the rows
  // 16 (w & 3) .. + 15 of the tile for the heads of its half of the group (w >> 2). The n % 4 tail
  // keys (at most three, in the last tile) run the 16-lane forms on wave 0.
  {
    constexpr int HPW = G >= 2 ? G / 2 : 1;   // heads per wave
    const int hg = wave >> 2;                 // which half of the group's heads
    const bool works = G >= 2 || hg == 0;
    const int jq = lane & 3, kslot = lane >> 2;
    float qq[HPW][8][4];
    float qr[G][8];                           // 16-lane layout, tail keys only (wave 0)
    const int main_n = (n_keys / 4) * 4;
    for (int ti = 0; ti < n_tiles; ++ti) {
      CT2X_X128_TICK(0)
      CT2X_X128_ACQUIRE(ti)
      CT2X_X128_TICK(1)
      if (ti == 0) {
        CT2X_X128_STAMP(3)
#pragma unroll
        for (int h = 0; h < HPW; ++h)
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            const v4f v = *reinterpret_cast<const v4f*>(qs + (hg * HPW + h) * DH + 16 * c + 4 * jq);
            qq[h][c][0] = v.x; qq[h][c][1] = v.y; qq[h][c][2] = v.z; qq[h][c][3] = v.w;
          }
      }
      const float* tile = stage + (ti % NB) * X128_TILE_FLOATS;
      const int t_pass = ti * X128_TILE + (wave & 3) * 16;   // first key of this wave's pass
      if (works && t_pass < main_n) {
        const int r = (wave & 3) * 16 + kslot;
        const int t = ti * X128_TILE + r;
        v4f kq[8];
#pragma unroll
        for (int c = 0; c < 8; ++c)
          kq[c] = *reinterpret_cast<const v4f*>(tile + x128_slot(r, 16 * c + 4 * jq));
        if (has_new && pos >= t_pass && pos < t_pass + 16) {  // (uniform: the pass that holds pos)
          if (t == pos) {
#pragma unroll
            for (int c = 0; c < 8; ++c) kq[c] = *reinterpret_cast<const v4f*>(knew_s + 16 * c + 4 * jq);
          }
        }
        float sv[HPW];
        x128_dot_main_quad<HPW>(qq, kq, jq, sv);
        if (jq == 0 && t < main_n) {
#pragma unroll
          for (int h = 0; h < HPW; ++h) sc[(hg * HPW + h) * t_pad + t] = sv[h] * p.scale;
        }
      }
      // tail keys main_n .. n_keys - 1 that live in this tile: 16 lanes per key, wave 0
      if (wave == 0 && main_n < n_keys && main_n < (ti + 1) * X128_TILE && n_keys > ti * X128_TILE) {
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
          for (int i = 0; i < 8; ++i) qr[g][i] = qs[g * DH + ((16 * i + l + 1) & 127)];
        const int t = main_n + grp;                       // grp 0 .. 2 hold keys, 3 idles
        const int r = min(t - ti * X128_TILE, X128_TILE - 1);
        float kk[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) kk[i] = tile[x128_slot(max(r, 0), (16 * i + l + 1) & 127)];
        if (has_new && t == pos) {
#pragma unroll
          for (int i = 0; i < 8; ++i) kk[i] = knew_s[(16 * i + l + 1) & 127];
        }
        const int form = grp != 1 ? 1 : 2;
        const bool here = t < n_keys && t >= ti * X128_TILE && t < (ti + 1) * X128_TILE;
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const float s = x128_dot(qr[g], kk, l, form) * p.scale;
          if (l == 0 && here) sc[g * t_pad + t] = s;
        }
      }
    }
  }
  CT2X_X128_STAMP(4)
  if (CT2X_X128_STOP_AFTER <= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  if (n_tiles == 0) {  // (no keys at all: drain the speculative first tile, publish the prologue)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    x128_barrier();
  }
  x128_barrier();  // scores complete
  CT2X_X128_STAMP(5)

  // ---- SoftMax: one wave per head --------------------------------------------------------------
  for (int g = wave; g < G; g += X128_WAVES) {
    float* e = sc + g * t_pad;
    float m = -FLT_MAX;
    for (int t = lane; t < n_keys; t += 64) m = fmaxf(m, e[t]);
    m = wave_max_dpp(m);
    for (int t = lane; t < t_pad; t += 64) e[t] = (t < n_keys) ? exp_avx2(e[t] - m) : 0.f;
    __builtin_amdgcn_wave_barrier();
    const float esum = x128_ordered_sum8(e, n_keys);
    const float rcp = 1.f / esum;
    __builtin_amdgcn_wave_barrier();
    for (int t = lane; t < n_keys; t += 64) e[t] = e[t] * rcp;
  }
  CT2X_X128_STAMP(6)
  if (CT2X_X128_STOP_AFTER <= 2) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  // ---- P.V: lane = (head of the group, channel) -----------------------------------------------
  // (ITEMS = 2 G wave items; with G = 8 every wave carries two - their chains interleave)
  constexpr int IPW = (ITEMS + X128_WAVES - 1) / X128_WAVES;
  float y[IPW];
  bool has[IPW];
#pragma unroll
  for (int it = 0; it < IPW; ++it) {
    y[it] = 0.f;
    has[it] = false;
  }
  const int n_blocks = n_keys / 8;
  for (int ti = 0; ti < n_tiles; ++ti) {
    CT2X_X128_ACQUIRE(n_tiles + ti)  // (first pass: the barrier also publishes the probabilities)
    if (ti == 0) {
      CT2X_X128_STAMP(7)
    }
    const float* tile = stage + ((n_tiles + ti) % NB) * X128_TILE_FLOATS;
#pragma unroll
    for (int it = 0; it < IPW; ++it) {
      const int item = wave + it * X128_WAVES;
      if (item >= ITEMS) continue;
      const int g = lane / CPW, c = item * CPW + lane % CPW;
      const float* pr = sc + g * t_pad;
      const float vnew = has_new ? vnew_s[c] : 0.f;
#pragma unroll
      for (int b = 0; b < X128_TILE / X128_VB; ++b) {
        const int tb = ti * X128_TILE + b * X128_VB;
        if (tb >= n_keys) break;
        float w[X128_VB];
#pragma unroll
        for (int jv = 0; jv < X128_VB; ++jv) w[jv] = tile[x128_slot(b * X128_VB + jv, c)];
        if (has_new && pos >= tb && pos < tb + X128_VB) {  // (uniform: the batch that holds pos)
#pragma unroll
          for (int jv = 0; jv < X128_VB; ++jv)
            if (tb + jv == pos) w[jv] = vnew;
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int t8 = tb + 8 * half;
          if (t8 >= n_keys) break;
          const v4f pa = *reinterpret_cast<const v4f*>(pr + t8);
          const v4f pb = *reinterpret_cast<const v4f*>(pr + t8 + 4);
          const float* wv = w + 8 * half;
          if (t8 + 8 <= n_blocks * 8) {
            const float A_ = fmaf(pa.x, wv[0], pa.z * wv[2]);
            const float B_ = fmaf(pa.y, wv[1], pa.w * wv[3]);
            const float c6 = has[it] ? fmaf(pb.z, wv[6], y[it]) : pb.z * wv[6];
            const float C_ = fmaf(pb.x, wv[4], c6);
            const float D_ = fmaf(pb.y, wv[5], pb.w * wv[7]);
            y[it] = (C_ + D_) + (A_ + B_);
            has[it] = true;
          } else {
            const float pj[8] = {pa.x, pa.y, pa.z, pa.w, pb.x, pb.y, pb.z, pb.w};
#pragma unroll
            for (int jv = 0; jv < 8; ++jv) {
              if (t8 + jv < n_keys) {
                y[it] = has[it] ? fmaf(pj[jv], wv[jv], y[it]) : pj[jv] * wv[jv];
                has[it] = true;
              }
            }
          }
        }
      }
    }
  }
#undef CT2X_X128_ACQUIRE
#undef CT2X_X128_REQUEST
  float row_max = 0.f;
#pragma unroll
  for (int it = 0; it < IPW; ++it) {
    const int item = wave + it * X128_WAVES;
    if (item >= ITEMS) continue;
    const int g = lane / CPW, c = item * CPW + lane % CPW;
    p.out[static_cast<int64_t>(row) * (p.heads * DH) + (hk * G + g) * DH + c] = y[it];
    row_max = fmaxf(row_max, fabsf(y[it]));
  }
  CT2X_X128_STAMP(8)
  if (p.row_amax_out != nullptr) {
    // producer-side absmax of the context row for the Quantize that follows: the workgroup's maximum
    // through LDS, then ONE plain store at [row, kv head] (row_amax_parts) - or, for callers that
    // want the folded vector, one atomicMax per workgroup (|y| >= 0: bit patterns order like
    // unsigned integers; exact and order independent)
    row_max = wave_max_dpp(row_max);
    x128_barrier();  // (every wave is past its last read of the score buffer)
    if (lane == 0) sc[wave] = row_max;
    x128_barrier();
    if (tid == 0) {
      float m = sc[0];
#pragma unroll
      for (int w = 1; w < X128_WAVES; ++w) m = fmaxf(m, sc[w]);
      if (p.row_amax_parts > 0)
        p.row_amax_out[static_cast<int64_t>(row) * p.row_amax_parts + hk] = m;
      else
        atomicMax(reinterpret_cast<unsigned int*>(p.row_amax_out) + row, __float_as_uint(m));
    }
  }
}

int x128_t_pad(const AttnDecodeParams& p) {
  return static_cast<int>(((p.t_max + 1 + 15) / 16) * 16 + 16);
}

size_t x128_lds_bytes(const AttnDecodeParams& p, int n_buf) {
  const size_t group = static_cast<size_t>(p.heads / p.heads_kv);
  return (n_buf * static_cast<size_t>(X128_TILE_FLOATS) + (group + 2) * X128_DH +
          group * static_cast<size_t>(x128_t_pad(p))) * sizeof(float);
}
// As many tiles in the ring as fit (4: three tiles = 96 KB in flight per workgroup; a cold tile
// takes ~1.7 us to arrive, profiles/r03_attn128_probe_b.txt).
int x128_n_buf(const AttnDecodeParams& p) {
#ifdef CT2X_X128_FORCE_NB
  return CT2X_X128_FORCE_NB;
#endif
  for (int nb = 4; nb >= 2; --nb)
    if (x128_lds_bytes(p, nb) <= 160 * 1024) return nb;
  return 0;
}

bool x128_disabled() {
  static const bool off = [] {
    const char* e = std::getenv("CT2X_FAST_ATTENTION");
    return e != nullptr && e[0] == '1';
  }();
  return off;
}

template <int G>
void launch_x128(const AttnX128Args& a, size_t lds, hipStream_t stream) {
  static std::atomic<bool> configured[64];  // > 64 KiB of dynamic LDS: opted into once per device
  int dev = 0;
  CT2X_HIP_CHECK(hipGetDevice(&dev));
  if (!configured[dev & 63].load(std::memory_order_acquire)) {
    CT2X_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_x128_decode_kernel<G>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    configured[dev & 63].store(true, std::memory_order_release);
  }
  (void)lds;
  const AttnDecodeParams& p = a.p;
  const unsigned geom = static_cast<unsigned>(p.t_max) | static_cast<unsigned>(p.heads_kv) << 24;
  hipLaunchKernelGGL(attn_x128_decode_kernel<G>, dim3(static_cast<unsigned>(p.rows * p.heads_kv)),
                     dim3(X128_WAVES * 64), lds, stream, static_cast<const float*>(p.k_cache),
                     static_cast<const float*>(p.v_cache), p.q, p.step_ptr,
                     static_cast<int>(p.kv_row_stride), static_cast<int>(p.kv_head_stride),
                     static_cast<int>(p.kv_time_stride), static_cast<int>(p.q_row_stride), geom,
                     static_cast<int>(p.length), a);
}

}  // namespace

// Self-attention decode at d_head 128: one query row per cache row (no encoder-memory beams), no
// per-row length vector, no ancestry table, GQA group 1 / 2 / 4 / 8, scores of the group within
// the LDS budget (T up to ~9000 at group 4).
bool attention_exact128_decode_applicable(const AttnDecodeParams& p) {
  if (x128_disabled()) return false;
  if (p.dhead != X128_DH || p.heads < 1 || p.heads_kv < 1 || p.heads % p.heads_kv != 0) return false;
  const int64_t group = p.heads / p.heads_kv;
  if (group != 1 && group != 2 && group != 4 && group != 8) return false;
  if (p.kv_group != 1 || p.lengths != nullptr || p.ancestry != nullptr) return false;
  if (p.rope_sin != nullptr && (p.rope_cos == nullptr || p.rot_dim < 2 || p.rot_dim > p.dhead ||
                                p.rot_dim % 2 != 0 || p.k_new == nullptr))
    return false;
  if (p.k_new != nullptr && p.v_new == nullptr) return false;
  if (p.kv_time_stride < p.dhead || p.t_max < 1) return false;
  const int64_t big = int64_t(1) << 31;
  if (p.kv_row_stride >= big || p.kv_head_stride >= big || p.kv_time_stride >= big ||
      p.q_row_stride >= big || p.t_max >= (int64_t(1) << 24) || p.heads_kv > 255 || p.length >= big)
    return false;
  return x128_n_buf(p) >= 2;
}

void launch_attention_exact128_decode(const AttnDecodeParams& p, hipStream_t stream) {
  AttnX128Args a;
  a.p = p;
  a.group = static_cast<int>(p.heads / p.heads_kv);
  a.t_pad = x128_t_pad(p);
  a.n_buf = x128_n_buf(p);
  const size_t lds = x128_lds_bytes(p, a.n_buf);
  switch (a.group) {
    case 1: launch_x128<1>(a, lds, stream); break;
    case 2: launch_x128<2>(a, lds, stream); break;
    case 4: launch_x128<4>(a, lds, stream); break;
    default: launch_x128<8>(a, lds, stream); break;
  }
}

}  // namespace ct2x
