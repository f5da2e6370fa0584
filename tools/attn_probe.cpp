// Phase costs of the exact decode attention launch (attention_exact.hip) at the Transformer-base
// shape: 32 rows x 8 heads x d_head 64, T cached keys, fused Quantize. The kernel's
// CT2X_ATTN_STAMP hooks record s_memtime for workgroup 0 / wave 0.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -x hip tools/attn_probe.cpp \
//         -Ictranslate2_amd/csrc -Iinclude -Lctranslate2_amd -lct2_mi355 \
//         -Wl,-rpath,$PWD/ctranslate2_amd -o /tmp/ap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ long long g_stamp[8];
#define CT2X_ATTN_STAMP(i) if (blockIdx.x == 0 && threadIdx.x == 0) g_stamp[i] = clock64();
#include "attention_exact.hip"
using namespace ct2x;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s\n", hipGetErrorString(e)); return 1; } } while (0)

int main() {
  const int rows = 32, H = 8, dh = 64, TMAX = 64;
  float *q, *kc, *vc, *qs; int8_t* qo; int32_t* step;
  CK(hipMalloc(&q, rows * 3 * H * dh * 4)); CK(hipMalloc(&kc, (size_t)rows * H * TMAX * dh * 4));
  CK(hipMalloc(&vc, (size_t)rows * H * TMAX * dh * 4)); CK(hipMalloc(&qs, rows * 4));
  CK(hipMalloc(&qo, rows * H * dh)); CK(hipMalloc(&step, 4));
  std::vector<float> h((size_t)rows * H * TMAX * dh);
  for (size_t i = 0; i < h.size(); ++i) h[i] = ((i * 2654435761u) % 2001) / 1000.f - 1.f;
  CK(hipMemcpy(kc, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(vc, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(q, h.data(), (size_t)rows * 3 * H * dh * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int T : {1, 16, 32, 48}) {
    CK(hipMemcpy(step, &T, 4, hipMemcpyHostToDevice));
    AttnDecodeParams p;
    p.q = q; p.q_row_stride = 3 * H * dh;
    p.k_new = q + H * dh; p.v_new = q + 2 * H * dh; p.new_row_stride = 3 * H * dh;
    p.k_cache = kc; p.v_cache = vc;
    p.kv_row_stride = (int64_t)H * TMAX * dh; p.kv_head_stride = (int64_t)TMAX * dh; p.kv_time_stride = dh;
    p.step_ptr = step; p.rows = rows; p.heads = H; p.heads_kv = H; p.dhead = dh; p.kv_group = 1;
    p.t_max = TMAX; p.scale = 0.125f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, 0));
      launch_attention_exact_decode(p, 3, qo, qs, 0);
      CK(hipEventRecord(e1, 0));
      CK(hipDeviceSynchronize());
      long long hs[8]; CK(hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_stamp), sizeof hs));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 2)
        printf("T=%2d: bracket %.2f us | cycles: loads issued+q parked %lld  QK^T (K tile to LDS, dots) %lld  softmax %lld  P.V %lld  quantize (block amax, store) %lld | total %lld\n",
               T, ms * 1e3, hs[1] - hs[0], hs[2] - hs[1], hs[3] - hs[2], hs[4] - hs[3], hs[5] - hs[4], hs[5] - hs[0]);
    }
  }
  return 0;
}
