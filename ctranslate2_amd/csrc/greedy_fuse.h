// The tail of the fused greedy sampler (GreedyFuse, kernels.h): what the LAST workgroup of a launch
// that max-combined packed (value, column) keys per row does once every workgroup has arrived -
// greedy_sample_kernel's tail (decode_ops.hip; src/decoding.cc:874-880): fold the row's sub-slots,
// force a prefix token, hand the id to the next step, append to the history, advance the step.
// Shared by the vocabulary projection's ARGMAX form (gemm_s8.hip) and the decoder-stack launch's
// projection tail (decoder_cluster.hip).
#pragma once

#include "device_utils.h"
#include "kernels.h"

namespace ct2x {

// key order = (value, then LOWER column): the arg-max tie rule of the sampler (better()); -0.0 and
// +0.0 compare equal in the sampler's `>`: both signs pack as +0.0 so that the lower column wins
// between them as well. NaN never becomes a key (callers test v == v).
__device__ __forceinline__ unsigned long long greedy_fuse_key(float v, unsigned col) {
  const unsigned u = (v == 0.f) ? 0u : __float_as_uint(v);
  const unsigned ord = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return (static_cast<unsigned long long>(ord) << 32) | (0xFFFFFFFFu - col);
}

// One row's share of the tail: what the sampler does with the row's best (id, score) - force a
// prefix token, hand the id to the next step, append to the history (one thread per row calls).
__device__ __forceinline__ void greedy_fuse_finish_row(const GreedyFuse& gf, int m_all, int gf_step, int row, int id,
                                                       float score) {
  if (gf.prefix_ids != nullptr) {
    const int local = gf_step - gf.prefix_first_step;
    if (local >= 0 && local < gf.prefix_lens[row]) {
      id = gf.prefix_ids[static_cast<int64_t>(row) * gf.prefix_stride + local];
      score = 0.f;
    }
  }
  gf.next_ids[row] = id;
  gf.hist_ids[static_cast<int64_t>(gf_step) * m_all + row] = id;
  if (gf.host_hist_ids != nullptr) gf.host_hist_ids[static_cast<int64_t>(gf_step) * m_all + row] = id;
  gf.hist_scores[static_cast<int64_t>(gf_step) * m_all + row] = score;
}

// All THREADS threads of the finishing workgroup call. m_all <= 32 rows.
template <int THREADS>
__device__ __forceinline__ void greedy_fuse_finish(const GreedyFuse& gf, int m_all, int gf_step, int tid) {
  // (thread t reads - and clears - sub-slot t % 16 of row t / 16; 16 lanes fold a row)
  constexpr int RPP = THREADS / GreedyFuse::kSubSlots;  // rows per pass
  constexpr int PASSES = (32 + RPP - 1) / RPP;
  unsigned long long keys[PASSES];
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {  // every exchange of the workgroup in one round trip
    const int row = ps * RPP + tid / GreedyFuse::kSubSlots;
    keys[ps] = 0ull;
    if (row < m_all)
      keys[ps] = atomicExch(gf.slots + static_cast<int64_t>(row) * GreedyFuse::kSubSlots +
                                (tid % GreedyFuse::kSubSlots), 0ull);
  }
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
    const int row = ps * RPP + tid / GreedyFuse::kSubSlots;
    unsigned long long key = keys[ps];
#pragma unroll
    for (int off = GreedyFuse::kSubSlots / 2; off >= 1; off >>= 1) {
      const unsigned long long other = __shfl_xor(key, off, 64);
      key = other > key ? other : key;
    }
    if (row >= m_all || (tid % GreedyFuse::kSubSlots) != 0) continue;
    int id = 0;
    float score = 0.f;
    if (key != 0ull) {
      id = static_cast<int>(0xFFFFFFFFu - static_cast<unsigned>(key));
      const unsigned ord = static_cast<unsigned>(key >> 32);
      score = __uint_as_float((ord & 0x80000000u) ? (ord & 0x7FFFFFFFu) : ~ord);
    }
    greedy_fuse_finish_row(gf, m_all, gf_step, row, id, score);
  }
  if (tid == 0) {
    *gf.ticket = 0u;
    if (gf.advance_step_ptr != nullptr) *gf.advance_step_ptr = gf_step + 1;
  }
}

}  // namespace ct2x
