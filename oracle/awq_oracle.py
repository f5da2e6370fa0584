"""TEST INFRASTRUCTURE ONLY - numpy restatement of the reference's AWQ int4 ops.

PARITY UNPINNED: the reference has no CPU implementation of these ops (src/ops/awq/*_cpu.cc throw)
and no test vectors; its only implementation is CUDA fp16 (src/ops/awq/*_gpu.cu, derived from
mit-han-lab/llm-awq). This file restates the DATA FORMATS and the dequantisation arithmetic those
kernels implement, each function citing the lines it follows; products are formed in float64 from
the float16-rounded dequantised weights, so a device result is compared with a tolerance, never
bit for bit. Only tests/ may import this module.
"""
import numpy as np

# Nibble i of a packed word of the GEMM format holds output channel ORDER[i] of the group of eight
# (AutoAWQ packs with order_map = [0, 2, 4, 6, 1, 3, 5, 7]; dequantize_s4_to_fp16x2,
# src/ops/awq/dequantize.cuh:16-76, emits nibbles 0,4,1,5,2,6,3,7 as channels 0..7).
GEMM_ORDER = np.array([0, 2, 4, 6, 1, 3, 5, 7])


def bf16_bits(a):
    """float32 array -> bfloat16 bit patterns (uint16), round to nearest even"""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7fff + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def bf16_to_f32(bits):
    """bfloat16 bit patterns (uint16) -> float32"""
    return (np.ascontiguousarray(bits, np.uint16).astype(np.uint32) << 16).view(np.float32)


def _round_16(product_f32, bf16):
    """one rounding of the exact fp32 product to the arm's 16-bit type: float16 values, or bfloat16
    bit patterns (uint16) for the bfloat16 arm (awq.hip, round 6 - defined there, the reference has
    float16 only)"""
    return bf16_bits(product_f32) if bf16 else product_f32.astype(np.float16)


def _scales_f32(scales, bf16):
    return bf16_to_f32(scales) if bf16 else scales.astype(np.float32)


def _weights_f64(w16, bf16):
    return (bf16_to_f32(w16) if bf16 else w16).astype(np.float64)


def _nibbles(words):
    """[..., W] int32 -> [..., W, 8] values 0..15, nibble 0 = least significant."""
    w = words.astype(np.uint32)[..., None]
    return ((w >> (4 * np.arange(8, dtype=np.uint32))) & np.uint32(15)).astype(np.int32)


def pack_gemm(q, zeros, group):
    """q [in_c, out_c] in 0..15, zeros [in_c / group, out_c] in 0..15 -> (qweight [in_c, out_c / 8],
    qzeros [in_c / group, out_c / 8]) int32 in the AWQ GEMM layout."""
    def pack(m):
        r, c = m.shape
        blocks = m.reshape(r, c // 8, 8).astype(np.uint32)
        out = np.zeros((r, c // 8), np.uint32)
        for i in range(8):
            out |= blocks[:, :, GEMM_ORDER[i]] << np.uint32(4 * i)
        return out.view(np.int32)
    return pack(q), pack(zeros)


def dequantize_gemm(qweight, scales, qzeros, bf16=False):
    """DequantizeAwq (src/ops/awq/dequantize_gpu.cu:8-64, shapes :71-76): qweight [in_c, out_c / 8]
    int32, scales [in_c / G, out_c] float16, qzeros [in_c / G, out_c / 8] int32 -> float16
    [in_c, out_c]: w = (q - z) * s, the subtraction exact and the product rounded once to float16
    (sub.f16x2 then fma.rn.f16x2 with a zero addend)."""
    in_c, qout = qweight.shape
    out_c = qout * 8
    group = in_c // scales.shape[0]
    inv = np.argsort(GEMM_ORDER)  # channel c lives in nibble inv[c]
    q = _nibbles(qweight)[:, :, inv].reshape(in_c, out_c)
    z = _nibbles(qzeros)[:, :, inv].reshape(-1, out_c)
    z = np.repeat(z, group, axis=0)
    s = np.repeat(_scales_f32(scales, bf16), group, axis=0)
    return _round_16((q - z).astype(np.float32) * s, bf16)


def gemm(x, qweight, scales, qzeros, bf16=False):
    """GemmAwq (src/ops/awq/gemm.cc:9-31 + gemm_gpu.cu:470-530): x [m, in_c] float16 -> float16
    [m, out_c] = x . dequantize(qweight). The CUDA kernel accumulates eight K-interleaved partial
    sums in fp32, rounds each to float16 and ops::Sum adds them; the restatement keeps the exact
    product of the rounded operands (float64) - the comparison tolerance covers that difference."""
    w = _weights_f64(dequantize_gemm(qweight, scales, qzeros, bf16), bf16)
    return _weights_f64(x, bf16) @ w


def gemv_widths(in_c, group):
    """zeros_w / sf_w of the GEMV layout (src/ops/awq/gemv_gpu.cu:296-302, 365-369)."""
    md = lambda c, d: (c + d - 1) // d
    if group == 64:
        zeros_w = md(md(in_c // group, 8), 2) * 2
    else:
        zeros_w = md(in_c // group, 8)
    return zeros_w, zeros_w * 8


def pack_gemv(q, zeros, scales, group):
    """q [out_c, in_c] in 0..15, zeros / scales [out_c, in_c / group] -> (qweight [out_c, in_c / 8],
    qzeros [out_c, zeros_w] int32, scales [out_c, sf_w] float16) in the AWQ GEMV layout: nibble i
    of word w is input channel 8 w + i, zeros nibble g % 8 of word g / 8 is group g, both widths
    padded as gemv_widths says."""
    out_c, in_c = q.shape
    zeros_w, sf_w = gemv_widths(in_c, group)
    qw = np.zeros((out_c, in_c // 8), np.uint32)
    qq = q.reshape(out_c, in_c // 8, 8).astype(np.uint32)
    for i in range(8):
        qw |= qq[:, :, i] << np.uint32(4 * i)
    groups = in_c // group
    zp = np.zeros((out_c, zeros_w * 8), np.uint32)
    zp[:, :groups] = zeros
    zz = zp.reshape(out_c, zeros_w, 8)
    qz = np.zeros((out_c, zeros_w), np.uint32)
    for i in range(8):
        qz |= zz[:, :, i] << np.uint32(4 * i)
    sc = np.zeros((out_c, sf_w), np.float16)
    sc[:, :groups] = scales
    return qw.view(np.int32), qz.view(np.int32), sc


def dequantize_gemv(qweight, scales, qzeros, group, bf16=False):
    """The weight the GEMV kernels multiply with (src/ops/awq/gemv_gpu.cu:140-150 for the
    tensor-core form: __float2half(scale * (q - z)); :318-331 for the vector form, which keeps the
    product in fp32): returned as float16 [out_c, in_c]."""
    out_c, words = qweight.shape
    in_c = words * 8
    groups = in_c // group
    q = _nibbles(qweight).reshape(out_c, in_c)
    z = _nibbles(qzeros).reshape(out_c, -1)[:, :groups]
    z = np.repeat(z, group, axis=1)
    s = np.repeat(_scales_f32(scales[:, :groups], bf16), group, axis=1)
    return _round_16((q - z).astype(np.float32) * s, bf16)


def gemv(x, qweight, scales, qzeros, group, bf16=False):
    """GemvAwq (src/ops/awq/gemv.cc:9-37): x [m, in_c] float16 -> [m, out_c] = x . W^T."""
    w = _weights_f64(dequantize_gemv(qweight, scales, qzeros, group, bf16), bf16)
    return _weights_f64(x, bf16) @ w.T
