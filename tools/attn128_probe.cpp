// Phase costs of the exact d_head-128 decode attention launch (attention_exact128.hip) at the
// Llama-3-8B decode shape: 16 rows x 32 / 8 heads, T cached keys, RoPE, cache append. The kernel's
// CT2X_X128_STAMP hooks record s_memtime for workgroup 0 / wave 0; the launch is also timed with HIP
// events back to back (hot caches) and behind a 512 MB memset (cold L2 / MALL, cold I-cache).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -x hip tools/attn128_probe.cpp \
//         -Ictranslate2_amd/csrc -Iinclude -Lctranslate2_amd -lct2_mi355 \
//         -Wl,-rpath,$PWD/ctranslate2_amd -o build/tools/attn128_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ long long g_stamp[16];
#ifdef PROBE_NO_STAMPS
#define CT2X_X128_STAMP(i)
#define CT2X_X128_TICK(i)
__device__ long long g_acq[2];
#else
#define CT2X_X128_STAMP(i) if (blockIdx.x == 0 && threadIdx.x == 0) g_stamp[i] = clock64();
// TICK(0) / TICK(1) bracket the tile acquisition (DMA wait + barrier) of the QK^T loop: summed
__device__ long long g_acq[2];
#define CT2X_X128_TICK(i) if (blockIdx.x == 0 && threadIdx.x == 0) { if (i == 0) g_acq[0] = clock64(); else g_acq[1] += clock64() - g_acq[0]; }
#endif
#include "attention_exact128.hip"
using namespace ct2x;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s\n", hipGetErrorString(e)); return 1; } } while (0)

int main() {
  const int rows = 16, H = 32, HK = 8, dh = 128, TMAX = 320, LAYERS = 8;
  const int64_t qkv = (H + 2 * HK) * dh;
  float *q, *kc, *vc, *out, *rs, *rc, *amax, *scratch; int32_t* step;
  const size_t cache = (size_t)rows * HK * TMAX * dh;
  CK(hipMalloc(&q, rows * qkv * 4)); CK(hipMalloc(&kc, cache * 4 * LAYERS)); CK(hipMalloc(&vc, cache * 4 * LAYERS));
  CK(hipMalloc(&out, rows * H * dh * 4)); CK(hipMalloc(&rs, (TMAX + 1) * dh * 4)); CK(hipMalloc(&rc, (TMAX + 1) * dh * 4));
  CK(hipMalloc(&amax, rows * 8 * 4)); CK(hipMalloc(&step, 4)); CK(hipMalloc(&scratch, 512u << 20));
  std::vector<float> h(cache);
  for (size_t i = 0; i < h.size(); ++i) h[i] = ((i * 2654435761u) % 2001) / 1000.f - 1.f;
  for (int l = 0; l < LAYERS; ++l) {
    CK(hipMemcpy(kc + l * cache, h.data(), cache * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(vc + l * cache, h.data(), cache * 4, hipMemcpyHostToDevice));
  }
  CK(hipMemcpy(q, h.data(), rows * qkv * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(rs, h.data(), (TMAX + 1) * dh * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(rc, h.data(), (TMAX + 1) * dh * 4, hipMemcpyHostToDevice));
  CK(hipMemset(amax, 0, rows * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int T : {1, 64, 128, 192, 255}) {
    CK(hipMemcpy(step, &T, 4, hipMemcpyHostToDevice));
    AttnDecodeParams p;
    p.q = q; p.q_row_stride = qkv;
    p.k_new = q + H * dh; p.v_new = q + (H + HK) * dh; p.new_row_stride = qkv;
    p.kv_row_stride = (int64_t)HK * TMAX * dh; p.kv_head_stride = (int64_t)TMAX * dh; p.kv_time_stride = dh;
    p.step_ptr = step; p.rows = rows; p.heads = H; p.heads_kv = HK; p.dhead = dh; p.kv_group = 1;
    p.t_max = TMAX; p.scale = 0.08838835f; p.rope_sin = rs; p.rope_cos = rc; p.rot_dim = dh;
    p.out = out; p.row_amax_out = amax; p.row_amax_parts = HK;
    if (!attention_exact128_decode_applicable(p)) { printf("not applicable\n"); return 1; }
    float hot = 0, cold = 0;
    long long hs[16];
    for (int rep = 0; rep < 4; ++rep) {       // hot: the same layer's cache again and again
      p.k_cache = kc; p.v_cache = vc;
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < 8; ++i) launch_attention_exact128_decode(p, 0);
      CK(hipEventRecord(e1, 0));
      CK(hipDeviceSynchronize());
      CK(hipEventElapsedTime(&hot, e0, e1));
    }
    for (int rep = 0; rep < 3; ++rep) {       // cold: a different layer's cache after a 512 MB memset
      p.k_cache = kc + (rep + 1) * cache; p.v_cache = vc + (rep + 1) * cache;
      CK(hipMemsetAsync(scratch, rep, 512u << 20, 0));
      { long long z[2] = {0, 0}; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_acq), z, sizeof z)); }
      CK(hipEventRecord(e0, 0));
      launch_attention_exact128_decode(p, 0);
      CK(hipEventRecord(e1, 0));
      CK(hipDeviceSynchronize());
      CK(hipEventElapsedTime(&cold, e0, e1));
      CK(hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_stamp), sizeof hs));
      CK(hipMemcpyFromSymbol(&hs[12], HIP_SYMBOL(g_acq), 2 * sizeof(long long)));
    }
    printf("T=%3d: hot %.2f us/launch, cold bracket %.2f us | cold cycles: early requests %lld  args+prologue %lld  barrier %lld  QK^T %lld  barrier %lld  softmax %lld  barrier %lld  P.V %lld | total %lld | of QK^T: tile waits %lld\n",
           T, hot * 1e3 / 8, cold * 1e3, hs[1] - hs[0], hs[2] - hs[1], hs[3] - hs[2], hs[4] - hs[3], hs[5] - hs[4],
           hs[6] - hs[5], hs[7] - hs[6], hs[8] - hs[7], hs[8] - hs[0], hs[13]);
  }
  return 0;
}
