// What does each ingredient of the one-wave-per-SIMD int8 GEMM K loop (gemm_s8_big2.hip) cost?
// The loop body - 64 v_mfma_i32_16x16x64_i8 per step on 64 accumulator quads held in AGPRs - is
// rebuilt here from bare matrix operations, adding one ingredient per MODE bit:
//   1: 16 ds_read_b128 fragment reads per step (one after every second matrix operation)
//   2: 8 global_load_dwordx4 per step into staging registers
//   4: 8 ds_write_b128 per step from the staging registers
//   8: s_barrier + s_waitcnt lgkmcnt(0) per step
//  16: LDS-DMA (global_load_lds_dwordx4) instead of bits 2 + 4 (8 per step)
// Output: shader cycles (s_memtime) per step, per matrix operation, and the wall time.
// Build: hipcc --offload-arch=gfx950 -O3 -o build/tools/big2_loop_probe tools/big2_loop_probe.cpp
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

#define CHECK(x)                                                        \
  do {                                                                  \
    hipError_t e_ = (x);                                                \
    if (e_ != hipSuccess) {                                             \
      printf("%s: %s\n", #x, hipGetErrorString(e_));                    \
      exit(1);                                                          \
    }                                                                   \
  } while (0)

constexpr int STAGE = 32768;

template <int MODE, bool SHARE>
__global__ __launch_bounds__(256) void probe(const int8_t* __restrict__ src, int64_t row_stride,
                                             int steps, int* out, uint64_t* cycles) {
  extern __shared__ __attribute__((aligned(16))) int8_t lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int f_slot = (g ^ ((r >> 1) & 2)) << 4;
  const int a_off = ((wave >> 1) * 128 + r) * 64 + f_slot;
  const int b_off = 16384 + ((wave & 1) * 128 + r) * 64 + f_slot;
  const int8_t* gsrc[8];
  // operand sharing of the real kernel: the 32 workgroups of an XCD (blockIdx & 7) hold an 8 x 4
  // band of output tiles = 8 A tiles and 4 B tiles of 256 rows (SHARE); or no sharing at all
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int tile_a = SHARE ? (idx & 7) + 8 * (xcd & 3) : blockIdx.x * 2;
  const int tile_b = SHARE ? 32 + (idx >> 3) + 4 * xcd : blockIdx.x * 2 + 1;
  for (int q = 0; q < 8; ++q)
    gsrc[q] = src + (static_cast<int64_t>(q < 4 ? tile_a : tile_b) * 256 + wave * 64 + (q & 3) * 16 + (lane >> 2)) * row_stride +
              (lane & 3) * 16;
  for (int i = threadIdx.x; i < 4 * STAGE / 4; i += 256) reinterpret_cast<int*>(lds)[i] = i * 2654435761u;
  __syncthreads();
  v4i acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = v4i{0, 0, 0, 0};
  v4i fa0[8], fb0[8], fa1[8], fb1[8], sg0[8], sg1[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    fa0[i] = *reinterpret_cast<const v4i*>(lds + a_off + i * 1024);
    fb0[i] = *reinterpret_cast<const v4i*>(lds + b_off + i * 1024);
    fa1[i] = fa0[i];
    fb1[i] = fb0[i];
    sg0[i] = fa0[i];
    sg1[i] = fb0[i];
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const uint64_t t0 = __builtin_readcyclecounter();

#define MFMA(I, J, AF, BF) \
  asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(acc[I][J]) : "v"(AF[I]), "v"(BF[J]));
#define STEP(T, AF, BF, AFN, BFN, SGW, SGN)                                                        \
  {                                                                                                \
    const int t_ = (T);                                                                            \
    if (MODE & 8) __builtin_amdgcn_s_barrier();                                                    \
    const int8_t* st_ = lds + ((t_ + 1) & 3) * STAGE;                                              \
    int8_t* wbase_ = lds + ((t_ + 2) & 3) * STAGE + wave * 4096 + lane * 16;                       \
    const int lk0_ = (t_ & 63) * 64;                                                               \
    _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                                             \
      _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) {                                           \
        MFMA(i_, j_, AF, BF)                                                                       \
        const int slot_ = i_ * 4 + (j_ >> 1);                                                      \
        if ((j_ & 1) == 1) {                                                                       \
          if (slot_ < 8) {                                                                         \
            if (MODE & 1) AFN[slot_] = *reinterpret_cast<const v4i*>(st_ + a_off + slot_ * 1024);  \
          } else if (slot_ < 16) {                                                                 \
            if (MODE & 1) BFN[slot_ - 8] = *reinterpret_cast<const v4i*>(st_ + b_off + (slot_ - 8) * 1024); \
          } else if (slot_ < 24) {                                                                 \
            if (MODE & 2) SGN[slot_ - 16] = *reinterpret_cast<const v4i*>(gsrc[slot_ - 16] + lk0_); \
            if (MODE & 16)                                                                         \
              __builtin_amdgcn_global_load_lds((glb_void*)(gsrc[slot_ - 16] + lk0_),               \
                                               (lds_void*)(wbase_ - lane * 16 + (slot_ - 16) * 1024), 16, 0, 0); \
          } else {                                                                                 \
            if (MODE & 4) *reinterpret_cast<v4i*>(wbase_ + (slot_ - 24) * 1024) = SGW[slot_ - 24]; \
          }                                                                                        \
        }                                                                                          \
      }                                                                                            \
    }                                                                                              \
    if (MODE & 8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                               \
    if (MODE & 16) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                \
  }
  for (int t = 0; t < steps; t += 2) {
    STEP(t, fa0, fb0, fa1, fb1, sg1, sg0)
    STEP(t + 1, fa1, fb1, fa0, fb0, sg0, sg1)
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const uint64_t t1 = __builtin_readcyclecounter();
  if (lane == 0) cycles[blockIdx.x * 4 + wave] = t1 - t0;
  int s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += sg0[i][0] ^ sg1[i][3];
  if (s == 0x7fffffff) out[threadIdx.x] = s;
}

template <int MODE, bool SHARE>
void run(const int8_t* src, int64_t stride, int steps, int* out, uint64_t* cyc, const char* what) {
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE, SHARE>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 4 * STAGE));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int it = 0; it < 4; ++it) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe<MODE, SHARE>), dim3(256), dim3(256), 4 * STAGE, 0, src, stride, steps, out, cyc);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  uint64_t h[1024];
  CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
  double sum = 0;
  for (int i = 0; i < 1024; ++i) sum += static_cast<double>(h[i]);
  const double per_step = sum / 1024 / steps;
  const double tops = 256.0 * 4 * steps * 64 * (16.0 * 16 * 64 * 2) / (best * 1e-3) / 1e12;
  printf("mode %2d %s %-58s %8.1f cyc/step %6.2f cyc/mfma  %8.3f ms  %7.1f TOPS  (%.2f GHz)\n", MODE, SHARE ? "shared " : "private", what,
         per_step, per_step / 64, best, tops, per_step * steps / (best * 1e-3) / 1e9);
}

int main(int argc, char** argv) {
  const int steps = argc > 1 ? atoi(argv[1]) : 2048;
  const int64_t stride = argc > 2 ? atoll(argv[2]) : 4096;  // bytes between operand rows (the GEMM's K)
  const bool brief = argc > 3;
  printf("row stride %lld\n", (long long)stride);
  int8_t* src;
  const size_t bytes = static_cast<size_t>(256) * 512 * stride;
  CHECK(hipMalloc(&src, bytes));
  int8_t* h = static_cast<int8_t*>(malloc(bytes));
  uint32_t s = 12345;
  for (size_t i = 0; i < bytes; ++i) {
    s = s * 1664525u + 1013904223u;
    h[i] = static_cast<int8_t>(s >> 24);
  }
  CHECK(hipMemcpy(src, h, bytes, hipMemcpyHostToDevice));
  int* out;
  uint64_t* cyc;
  CHECK(hipMalloc(&out, 1024));
  CHECK(hipMalloc(&cyc, 1024 * 8));
  if (!brief) {
  run<0, true>(src, stride, steps, out, cyc, "matrix operations only");
  run<1, true>(src, stride, steps, out, cyc, "+ 16 fragment reads");
  run<9, true>(src, stride, steps, out, cyc, "+ reads + barrier");
  run<5, true>(src, stride, steps, out, cyc, "+ reads + 8 LDS writes (no loads)");
  run<3, true>(src, stride, steps, out, cyc, "+ reads + 8 global loads");
  }
  run<7, true>(src, stride, steps, out, cyc, "+ reads + loads + writes");
  run<15, true>(src, stride, steps, out, cyc, "+ reads + loads + writes + barrier (register-staged loop)");
  if (!brief) run<17, true>(src, stride, steps, out, cyc, "+ reads + 8 LDS-DMA");
  run<25, true>(src, stride, steps, out, cyc, "+ reads + 8 LDS-DMA + barrier (the DMA loop)");
  if (!brief) run<3, false>(src, stride, steps, out, cyc, "+ reads + 8 global loads");
  run<15, false>(src, stride, steps, out, cyc, "+ reads + loads + writes + barrier");
  run<25, false>(src, stride, steps, out, cyc, "+ reads + 8 LDS-DMA + barrier");
  return 0;
}
