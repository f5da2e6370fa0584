"""oracle/awq_oracle.py is the CPU restatement of the reference's AWQ int4 formats (parity
unpinned: the reference has no CPU implementation and no vectors). What CAN be pinned on the CPU:
pack -> dequantise round trips of both layouts, the nibble order against the unpack sequence of the
reference's dequantize_s4_to_fp16x2 (src/ops/awq/dequantize.cuh:16-76, executed here on integers),
and the padded widths of the GEMV layout against the formulas in gemv_gpu.cu."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import awq_oracle as A  # noqa: E402


def _reference_unpack(word):
    """dequantize_s4_to_fp16x2 on one word, with the fp16 magic-number arithmetic replaced by its
    exact integer meaning: h[0] = nibbles (0, 4), h[1] = (1, 5), h[2] = (2, 6), h[3] = (3, 7); the
    eight halves it returns, in memory order, are the eight output channels of the group."""
    n = [(int(word) >> (4 * i)) & 15 for i in range(8)]
    return [n[0], n[4], n[1], n[5], n[2], n[6], n[3], n[7]]


def test_gemm_layout_nibble_order_follows_the_cuda_unpack():
    rng = np.random.default_rng(1)
    q = rng.integers(0, 16, (4, 16))
    qw, _ = A.pack_gemm(q, np.zeros((1, 16), np.int64), 4)
    for r in range(4):
        for cw in range(2):
            assert _reference_unpack(qw.view(np.uint32)[r, cw]) == q[r, cw * 8:(cw + 1) * 8].tolist()


def test_round_trips_and_widths():
    rng = np.random.default_rng(2)
    in_c, out_c, g = 256, 64, 64
    q = rng.integers(0, 16, (in_c, out_c))
    z = rng.integers(0, 16, (in_c // g, out_c))
    s = (rng.random((in_c // g, out_c)) * 0.02 + 0.001).astype(np.float16)
    qw, qz = A.pack_gemm(q, z, g)
    want = ((q - np.repeat(z, g, 0)).astype(np.float32) * np.repeat(s.astype(np.float32), g, 0)).astype(np.float16)
    assert np.array_equal(A.dequantize_gemm(qw, s, qz), want)
    qv = rng.integers(0, 16, (out_c, in_c))
    zv = rng.integers(0, 16, (out_c, in_c // g))
    sv = (rng.random((out_c, in_c // g)) * 0.02 + 0.001).astype(np.float16)
    qw2, qz2, sc2 = A.pack_gemv(qv, zv, sv, g)
    want2 = ((qv - np.repeat(zv, g, 1)).astype(np.float32) * np.repeat(sv.astype(np.float32), g, 1)).astype(np.float16)
    assert np.array_equal(A.dequantize_gemv(qw2, sc2, qz2, g), want2)
    # gemv_gpu.cu:296-302 (g64) and :365-369 (g128): zeros_w / sf_w for IC = 4096
    assert A.gemv_widths(4096, 64) == (8, 64) and A.gemv_widths(4096, 128) == (4, 32)
    # the reference recovers the group size as IC / scale.dim(-1) (gemv_gpu.cu:447): consistent
    assert 4096 // A.gemv_widths(4096, 64)[1] == 64 and 4096 // A.gemv_widths(4096, 128)[1] == 128


def test_model_builder_packing_agrees_with_the_restatement():
    """ctranslate2_amd.model_builder writes the AWQ variables of the synthetic models; what it packs
    must dequantise (oracle) to what it says the twin float16 weight is."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from ctranslate2_amd import model_builder as mb
    rng = np.random.default_rng(4)
    w = rng.normal(0, 0.05, (48, 256)).astype(np.float32)
    for group in (64, 128):
        q, z, s = mb.awq_quantize(w, group)
        want = mb.awq_dequantize(q, z, s, group)                      # [out_c, in_c]
        qw, sc, qz = mb.awq_pack(q, z, s, group, "gemm")
        assert np.array_equal(A.dequantize_gemm(qw, sc, qz), want.T)
        qw, sc, qz = mb.awq_pack(q, z, s, group, "gemv")
        assert qz.shape[1] == A.gemv_widths(256, group)[0] and sc.shape[1] == A.gemv_widths(256, group)[1]
        assert np.array_equal(A.dequantize_gemv(qw, sc, qz, group), want)
        assert np.abs(want.astype(np.float32) - w).max() < 0.02  # 4-bit groups: coarse but sane


def test_autoawq_published_order_known_answer():
    """The GEMM layout's nibble order as AutoAWQ publishes it (awq/utils/packing_utils.py:
    AWQ_ORDER = [0, 2, 4, 6, 1, 3, 5, 7], AWQ_REVERSE_ORDER = [0, 4, 1, 5, 2, 6, 3, 7]; the packer
    ORs intweight[:, col * 8 + AWQ_ORDER[i]] << (4 * i) into the word) as known answers that do not
    pass through this repository's own unpacker: the eight channels 0 .. 7 of one input row pack to
    0x75316420, their complement 15 - c to 0x8ACE9BDF, and the unpack permutation is the published
    reverse order."""
    order, reverse = [0, 2, 4, 6, 1, 3, 5, 7], [0, 4, 1, 5, 2, 6, 3, 7]
    assert A.GEMM_ORDER.tolist() == order and np.argsort(A.GEMM_ORDER).tolist() == reverse
    qw, qz = A.pack_gemm(np.arange(8)[None, :], np.zeros((1, 8), np.int64), 1)
    assert int(qw.view(np.uint32)[0, 0]) == 0x75316420 and int(qz.view(np.uint32)[0, 0]) == 0
    qw, _ = A.pack_gemm(15 - np.arange(8)[None, :], np.zeros((1, 8), np.int64), 1)
    assert int(qw.view(np.uint32)[0, 0]) == 0x8ACE9BDF
    # ... and the dequantiser reads them back in channel order with scale 1, zero 0
    s = np.ones((1, 8), np.float16)
    assert A.dequantize_gemm(*A.pack_gemm(np.arange(8)[None, :], np.zeros((1, 8), np.int64), 1)[:1], s,
                             qz).astype(np.int64).tolist() == [list(range(8))]


def test_bfloat16_arm_of_the_restatement():
    """Round 6: the bfloat16 arm (defined in awq.hip, the reference has float16 only). The rounding
    helper against hand-computed patterns (ties to even, carry into the exponent), and the arm's
    dequantised weight - scale rounded to bfloat16 once, (q - z) * s rounded to bfloat16 once -
    against ctranslate2_amd.model_builder's own statement of it, both layouts."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from ctranslate2_amd import model_builder as mb
    x = np.array([1.0, 1.00390625, 1.01171875, 1.99609375, -3.140625, 0.0, 65280.0], np.float32)
    # 1 + 2^-8 is a tie -> even (1.0); 1 + 3 * 2^-8 is a tie -> even (1 + 2^-6); 2 - 2^-8 is a tie -> 2.0
    assert A.bf16_bits(x).tolist() == [0x3f80, 0x3f80, 0x3f82, 0x4000, 0xc049, 0x0000, 0x477f]
    assert np.array_equal(A.bf16_to_f32(A.bf16_bits(x)),
                          np.array([1.0, 1.0, 1.015625, 2.0, -3.140625, 0.0, 65280.0], np.float32))
    rng = np.random.default_rng(9)
    w = rng.normal(0, 0.05, (48, 256)).astype(np.float32)
    for group in (64, 128):
        q, z, s = mb.awq_quantize(w, group)
        want = mb.awq_dequantize_bf16(q, z, s, group)                 # float32 holding bfloat16 values
        assert np.array_equal(A.bf16_to_f32(A.bf16_bits(want)), want)
        qw, sc, qz = mb.awq_pack(q, z, s, group, "gemm")
        got = A.dequantize_gemm(qw, A.bf16_bits(sc.astype(np.float32)), qz, bf16=True)
        assert np.array_equal(A.bf16_to_f32(got), want.T)
        qw, sc, qz = mb.awq_pack(q, z, s, group, "gemv")
        got = A.dequantize_gemv(qw, A.bf16_bits(sc.astype(np.float32)), qz, group, bf16=True)
        assert np.array_equal(A.bf16_to_f32(got), want)
        assert np.abs(want - w).max() < 0.02
