"""CPU tests pinning the parity oracle (oracle/ct2_oracle.c):
 (1) against the reference's own known-answer vectors (tests/golden/ops_kat.json, extracted
     from /root/reference/tests/ops_test.cc and layers_test.cc by tests/golden/extract_kat.py);
 (2) against the REAL reference CPU library (oracle/_ref) on random inputs when that build is
     present (always in the authoring container; on the GPU box if the .so files travelled).
"""
import json
import os

import numpy as np
import pytest

import _oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "ops_kat.json")))
# tests/ops_test.cc:1434-1436: CPU float32 ops are compared at 1e-5.
TOL = 1e-5


def arr(values, shape, dtype=np.float32):
    return np.array(values, dtype).reshape(shape)


def test_kat_gemm_int8():
    k = KAT["GemmInt8"]
    a = arr(k["a"], k["a_shape"], np.int8)
    b = arr(k["b"], k["b_shape"], np.int8)  # [K, N] in the reference test (no transpose)
    c = O.gemm_s8(a, np.ascontiguousarray(b.T))
    assert np.array_equal(c.reshape(-1), np.array(k["c"], np.int32))


@pytest.mark.parametrize("name", ["QuantizeINT8", "QuantizeINT8ZeroRow"])
def test_kat_quantize(name):
    k = KAT[name]
    x = arr(k["x"], k["shape"])
    q, s = O.quantize_s8(x, True)
    np.testing.assert_allclose(s, np.array(k["scale"], np.float32), rtol=1e-6)
    assert np.array_equal(q.reshape(-1), np.array(k["q_round"], np.int8))
    q, s = O.quantize_s8(x, False)  # legacy truncation (binary_version < 5)
    assert np.array_equal(q.reshape(-1), np.array(k["q_trunc"], np.int8))
    if "q_round_shift_u8" in k:  # shift_to_uint8 arm: the KAT lists the uint8 bit patterns
        q, s = O.quantize_s8(x, 3)
        u8 = (q.astype(np.int32) + 128).astype(np.uint8).view(np.int8)
        assert np.array_equal(u8.reshape(-1), np.array(k["q_round_shift_u8"], np.int8))


def test_kat_topk():
    k = KAT["TopK"]
    v, i = O.topk(arr(k["x"], k["shape"]), k["k"])
    assert np.array_equal(i.reshape(-1), np.array(k["indices"], np.int32))
    np.testing.assert_allclose(v.reshape(-1), np.array(k["values"], np.float32), atol=TOL)


def test_kat_softmax_family():
    k = KAT["SoftMax"]
    np.testing.assert_allclose(O.softmax(arr(k["x"], k["shape"])).reshape(-1), k["y"], atol=TOL)
    k = KAT["LogSoftMax"]
    np.testing.assert_allclose(O.softmax(arr(k["x"], k["shape"]), log=True).reshape(-1), k["y"], atol=TOL)
    k = KAT["MaskedSoftMax"]
    y = O.softmax(arr(k["x"], k["shape"]), np.array(k["lengths"], np.int32))
    np.testing.assert_allclose(y.reshape(-1), k["y"], atol=TOL)


def test_kat_norms():
    k = KAT["LayerNorm"]
    y = O.layer_norm(arr(k["x"], k["shape"]), np.array(k["gamma"], np.float32),
                     np.array(k["beta"], np.float32), k["eps"])
    np.testing.assert_allclose(y.reshape(-1), k["y"], atol=TOL)
    k = KAT["RMSNorm"]
    y = O.rms_norm(arr(k["x"], k["shape"]), np.array(k["gamma"], np.float32), k["eps"])
    np.testing.assert_allclose(y.reshape(-1), k["y"], atol=TOL)


def rotary_tables(positions, dim, base, interleave):
    """RotaryEmbeddings::initialize (src/layers/attention_layer.cc:252-343), default scaling."""
    inv = (1.0 / np.power(np.float32(base), np.arange(0, dim, 2, dtype=np.float32) / np.float32(dim))).astype(np.float32)
    t = np.arange(positions, dtype=np.float32)[:, None] * inv[None, :]
    emb = np.repeat(t, 2, axis=1) if interleave else np.concatenate([t, t], axis=1)
    return np.sin(emb).astype(np.float32), np.cos(emb).astype(np.float32)


def test_kat_rotary():
    """tests/layers_test.cc:108-191 (interleaved, offset 2)."""
    k = KAT["RotaryEmbedding"]
    x = arr(k["x"], k["shape"])  # [2, 4, 2, 6] = [batch, heads, time, dim]
    b, h, t, d = x.shape
    sin, cos = rotary_tables(k["offset"] + t, d, k["base"], True)
    y = O.rotary(x.reshape(b * h, t, d), sin[k["offset"]:], cos[k["offset"]:], d, True)
    np.testing.assert_allclose(y.reshape(-1), k["y"], atol=TOL)


# ---- against the real reference library ------------------------------------------------------

def _ref():
    import _refshim as R
    if not R.available():
        pytest.skip("oracle/_ref (the compiled reference) is not present")
    return R


@pytest.mark.parametrize("rbc", [0, 1, 2, 3])  # bit 0: round, bit 1: shift_to_uint8 arm
def test_ref_quantize_exact(rbc):
    R = _ref()
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((333, 512)) * 3).astype(np.float32)
    x[5] = 0
    q1, s1 = O.quantize_s8(x, rbc)
    q2, s2 = R.quantize_s8(x, rbc)
    assert np.array_equal(s1, s2) and np.array_equal(q1, q2)


def test_ref_gemm_s8_exact_on_this_host():
    R = _ref()
    rng = np.random.default_rng(1)
    a = rng.integers(-127, 128, (32, 512), dtype=np.int8)
    b = rng.integers(-127, 128, (96, 512), dtype=np.int8)
    ref = R.gemm_s8(a, b)
    mine = O.gemm_s8(a, b)
    if not np.array_equal(ref, mine):
        # The reference's MKL u8s8 path saturates 16-bit intermediates on CPUs without VNNI
        # (SURVEY.md appendix A); the oracle is the exact integer result by construction.
        a2 = (a // 4).astype(np.int8)
        assert np.array_equal(R.gemm_s8(a2, b), O.gemm_s8(a2, b))
        pytest.xfail("reference int8 GEMM saturates on this host CPU (no VNNI path in MKL)")


@pytest.mark.parametrize("act", [-1, 0, 1, 2, 3, 4, 5, 6])
def test_ref_dequantize_exact(act):
    """Dequantize + bias + EVERY activation of ops::ActivationType against the reference library,
    bit for bit: the fused epilogue runs every column (masked remainder included: n = 130 has a
    remainder of 2) through the AVX2 functors - exp256_ps, vec_tanh (Eigen's rational form),
    vec_erf (Abramowitz-Stegun with exp256_ps), src/cpu/vec.h:155-237 - restated in
    oracle/ct2_oracle.c; wide value ranges so that the clamps and the tiny-argument arm are hit."""
    R = _ref()
    rng = np.random.default_rng(2 + act)
    for scale_exp in (-2, 0, 1):
        c = rng.integers(-300000, 300000, (17, 130), dtype=np.int32)
        c[0, :8] = [0, 1, -1, 2, -2, 3, 40, -40]  # tiny arguments
        sa = (rng.uniform(3, 90, 17) * 10.0 ** scale_exp).astype(np.float32)
        sb = rng.uniform(40, 5000, 130).astype(np.float32)
        bias = (rng.standard_normal(130) * 0.01).astype(np.float32)
        y1 = O.dequantize_gemm_output(c, sa, sb, bias, act)
        y2 = R.dequantize_gemm_output(c, sa, sb, bias, act)
        assert np.array_equal(y1.view(np.uint32), y2.view(np.uint32))


def test_ref_float_ops_close():
    R = _ref()
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((9, 512)) * 2).astype(np.float32)
    g = rng.standard_normal(512).astype(np.float32)
    b = rng.standard_normal(512).astype(np.float32)
    np.testing.assert_allclose(O.layer_norm(x, g, b), R.layer_norm(x, g, b), atol=TOL, rtol=TOL)
    np.testing.assert_allclose(O.rms_norm(x, g), R.rms_norm(x, g), atol=TOL, rtol=TOL)
    L = rng.integers(0, 513, 9).astype(np.int32)
    for log in (False, True):
        np.testing.assert_allclose(O.softmax(x, L, log), R.softmax(x, L, log), atol=TOL, rtol=TOL)
    big = (rng.standard_normal((3, 32000)) * 4).astype(np.float32)
    np.testing.assert_allclose(O.softmax(big, None, True), R.softmax(big, None, True), atol=TOL, rtol=TOL)
    v1, i1 = O.topk(big, 8)
    v2, i2 = R.topk(big, 8)
    assert np.array_equal(i1, i2) and np.array_equal(v1, v2)
    xr = rng.standard_normal((2, 3, 5, 64)).astype(np.float32)
    sin, cos = rotary_tables(5, 64, 10000.0, False)
    np.testing.assert_allclose(O.rotary(xr.reshape(6, 5, 64), sin, cos, 64, False).reshape(xr.shape),
                               R.rotary(xr, sin, cos, 64, False), atol=1e-6)


# ------------------------------------------------------------------------------------------
# Bit-exact pins: the oracle's fp32 row kernels and MatMul restate the arithmetic of oracle/_ref
# AS COMPILED (GCC FMA contraction, scalar fast-math LayerNorm, MKL's summation trees), so the
# comparison is array_equal, not a tolerance. These are what the HIP kernels are held to.
# ------------------------------------------------------------------------------------------
def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.int32)


@pytest.mark.parametrize("depth", [1, 3, 7, 8, 9, 11, 12, 13, 15, 16, 20, 64, 100, 512, 1000, 1001, 4096])
def test_exact_norms(depth):
    R = _ref()
    rng = np.random.default_rng(depth)
    x = (rng.standard_normal((11, depth)) * 3 + 0.3).astype(np.float32)
    g = rng.standard_normal(depth).astype(np.float32)
    b = rng.standard_normal(depth).astype(np.float32)
    assert np.array_equal(_bits(O.layer_norm(x, g, b)), _bits(R.layer_norm(x, g, b)))
    assert np.array_equal(_bits(O.layer_norm(x, g, b, 1e-6)), _bits(R.layer_norm(x, g, b, 1e-6)))
    for res in (False, True):
        assert np.array_equal(_bits(O.rms_norm(x, g, 1e-6, res)), _bits(R.rms_norm(x, g, 1e-6, res)))
    const = np.full((2, depth), 1.25, np.float32)  # variance clamps at 0
    assert np.array_equal(_bits(O.layer_norm(const, g, b)), _bits(R.layer_norm(const, g, b)))


@pytest.mark.parametrize("depth", [1, 2, 5, 8, 9, 15, 16, 17, 33, 64, 100, 1000, 32000])
@pytest.mark.parametrize("log", [False, True])
def test_exact_softmax(depth, log):
    R = _ref()
    rng = np.random.default_rng(depth + 7 * log)
    rows = 9 if depth < 5000 else 3
    x = (rng.standard_normal((rows, depth)) * 4).astype(np.float32)
    assert np.array_equal(_bits(O.softmax(x, None, log)), _bits(R.softmax(x, None, log)))
    L = rng.integers(0, depth + 1, rows).astype(np.int32)
    assert np.array_equal(_bits(O.softmax(x, L, log)), _bits(R.softmax(x, L, log)))
    wide = (x * 30).astype(np.float32)  # exercises the exp clamp at -88.376
    assert np.array_equal(_bits(O.softmax(wide, L, log)), _bits(R.softmax(wide, L, log)))


def test_exact_dequantize_rows():
    """Embedding rows: the CPU multiplies by 1/scale (src/ops/dequantize_cpu.cc:10-19)."""
    R = _ref()
    rng = np.random.default_rng(20)
    x = rng.integers(-127, 128, (50, 512)).astype(np.int8)
    s = (127.0 / (rng.random(50) * 0.1 + 0.01)).astype(np.float32)
    assert np.array_equal(_bits(O.dequantize_rows(x, s)), _bits(R.dequantize_rows(x, s)))


def test_exact_matmul_decode_qk():
    """m = 1, trans_b, k = 64: every key count 1..70 (main groups of 4 + the three tail forms)."""
    R = _ref()
    rng = np.random.default_rng(21)
    for n in range(1, 71):
        q = rng.standard_normal((24, 1, 64)).astype(np.float32)
        k = rng.standard_normal((24, n, 64)).astype(np.float32)
        assert np.array_equal(_bits(O.matmul(q, k, True, 0.125)), _bits(R.matmul(q, k, True, 0.125))), n


def test_exact_matmul_decode_qk_every_head_size():
    """... and every head size 4, 8, .. 128 plus 160 / 192 / 256 (the fixtures use d_head 4, 16, 32,
    64; Llama-family models 128): the 32-lane "type A" tail form covers every k % 4 == 0."""
    R = _ref()
    rng = np.random.default_rng(25)
    for kk in list(range(4, 129, 4)) + [160, 192, 256]:
        for n in (1, 2, 3, 4, 5, 6, 7, 8, 9, 33):
            q = rng.standard_normal((10, 1, kk)).astype(np.float32)
            k = rng.standard_normal((10, n, kk)).astype(np.float32)
            assert np.array_equal(_bits(O.matmul(q, k, True, 0.5)), _bits(R.matmul(q, k, True, 0.5))), (kk, n)


def test_exact_matmul_decode_pv():
    """m = 1, no transpose: every k 1..70 and a long one; every head size in use."""
    R = _ref()
    rng = np.random.default_rng(22)
    for k in list(range(1, 71)) + [100, 257]:
        for n in (4, 8, 16, 32, 64, 128):
            p = rng.random((6, 1, k)).astype(np.float32)
            v = rng.standard_normal((6, k, n)).astype(np.float32)
            assert np.array_equal(_bits(O.matmul(p, v, False, 1.0)), _bits(R.matmul(p, v, False, 1.0))), (k, n)


def test_exact_matmul_sequence():
    """m >= 3 (encoder / beam >= 3 on the memory): k-ordered FMA chains."""
    R = _ref()
    rng = np.random.default_rng(23)
    for m, n, k, tb in [(32, 32, 64, True), (4, 32, 64, True), (3, 5, 64, True), (17, 40, 64, True),
                        (32, 64, 32, False), (4, 64, 32, False), (2, 64, 9, False), (17, 64, 40, False),
                        (32, 32, 128, True), (20, 128, 37, False)]:
        a = rng.standard_normal((5, m, k)).astype(np.float32)
        b = rng.standard_normal((5, n, k) if tb else (5, k, n)).astype(np.float32)
        assert np.array_equal(_bits(O.matmul(a, b, tb, 0.125 if tb else 1.0)),
                              _bits(R.matmul(a, b, tb, 0.125 if tb else 1.0))), (m, n, k, tb)


def test_exact_matmul_sequence_alpha_forms():
    """Where alpha enters MKL's sequence-mode product (round 3): scaled sum for small problems,
    carried by the packed copy of B once n >= 192 or (n > m and n >= max(k, 12)). One MKL thread
    per matrix = the batched regime (batch >= threads), checked both ways here. Head sizes whose
    scale is not a power of two (32, 128, 96) are the ones that can tell."""
    R = _ref()
    rng = np.random.default_rng(31)
    shapes = [(6, 127, 128), (6, 128, 128), (100, 128, 128), (128, 128, 128), (128, 129, 128),
              (150, 151, 128), (191, 191, 128), (192, 192, 128), (300, 200, 128), (256, 256, 128),
              (20, 20, 32), (4, 31, 32), (4, 32, 32), (4, 40, 32), (40, 41, 32), (200, 192, 32),
              (16, 95, 96), (16, 96, 96), (33, 300, 64), (6, 12, 12), (6, 11, 12)]
    import os
    for threads, batch in ((os.cpu_count() or 4, 16), (1, 2)):
        R.set_num_threads(threads)
        for m, n, k in shapes:
            alpha = float(np.float32(1.0 / np.sqrt(k)))
            a = rng.standard_normal((batch, m, k)).astype(np.float32)
            b = rng.standard_normal((batch, n, k)).astype(np.float32)
            assert np.array_equal(_bits(O.matmul(a, b, True, alpha)), _bits(R.matmul(a, b, True, alpha))), (threads, m, n, k)
    R.set_num_threads(os.cpu_count() or 4)


def test_exact_matmul_k_blocking():
    """Sequence-mode P.V with long rows: MKL's k partition (384-blocks while > 768 remain, then
    ceil / floor halves above 384), every block a chain from zero, blocks added in order."""
    R = _ref()
    rng = np.random.default_rng(32)
    for m, n, k in [(16, 128, 384), (16, 128, 385), (16, 128, 389), (8, 64, 500), (128, 128, 768),
                    (16, 128, 769), (16, 64, 1024), (5, 128, 1153), (16, 128, 2049), (3, 32, 4097)]:
        p = rng.random((9, m, k)).astype(np.float32)
        v = rng.standard_normal((9, k, n)).astype(np.float32)
        assert np.array_equal(_bits(O.matmul(p, v, False, 1.0)), _bits(R.matmul(p, v, False, 1.0))), (m, n, k)


def test_vml_sin_cos_restated():
    """orc_vml_sin / orc_vml_cos against the MKL library itself (vsSin / vsCos of libmkl_rt, the
    entry points the reference's ops::Sin / ops::Cos call, src/cpu/primitives.cc:378-394): every
    float of a binade, 2 million random arguments in the restated range, sinusoidal-table and RoPE
    arguments - bit for bit. (On an AVX-512 host MKL runs its Z0 kernels, not the L9 kernels the
    restatement was read from: same bits.)"""
    import ctypes as C
    import glob
    import _refshim as RS
    _ref()
    RS.lib()  # loads the reference library and, in the global scope, the MKL runtime it is linked to
    # (the ILP64 interface: MKL_INT is 64 bits wide)
    mkl = C.CDLL(sorted(glob.glob(os.path.join(RS.MKL_DIR, "libmkl_intel_ilp64.so*")))[0],
                 mode=os.RTLD_GLOBAL | os.RTLD_LAZY)
    rng = np.random.default_rng(41)
    half = 256
    inc = np.float32(np.log(np.float32(10000.0)) / np.float32(half - 1))
    ts = np.exp(-inc * np.arange(half, dtype=np.float32)).astype(np.float32)
    inv = (1.0 / np.power(np.float32(500000.0), np.arange(0, 128, 2, dtype=np.float32) / np.float32(128))).astype(np.float32)
    xs = [np.arange(0x3F800000, 0x40000000, dtype=np.uint32).view(np.float32),
          rng.uniform(-10000, 10000, 2_000_000).astype(np.float32),
          (np.arange(1, 5001, dtype=np.float32)[:, None] * ts[None, :]).astype(np.float32).ravel(),
          (np.arange(0, 9000, dtype=np.float32)[:, None] * inv[None, :]).astype(np.float32).ravel(),
          np.array([0.0, -0.0, 1e-30, 1.5e-45, np.pi, -np.pi, np.pi / 2, 10000.0, -10000.0], np.float32)]
    for x in xs:
        for fn, mine in (("vsSin", O.vml_sin), ("vsCos", O.vml_cos)):
            want = np.empty_like(x)
            getattr(mkl, fn)(C.c_int64(len(x)), x.ctypes.data_as(C.c_void_p), want.ctypes.data_as(C.c_void_p))
            assert np.array_equal(_bits(mine(x)), _bits(want)), fn


def test_matmul_envelope_is_explicit():
    with pytest.raises(ValueError):   # one-row form: k % 4 != 0 was not recovered
        O.matmul(np.zeros((1, 1, 130), np.float32), np.zeros((1, 5, 130), np.float32), True)
    with pytest.raises(ValueError):   # two query rows
        O.matmul(np.zeros((1, 2, 64), np.float32), np.zeros((1, 8, 64), np.float32), True)
    with pytest.raises(ValueError):   # few query rows with long dot products (k > 64)
        O.matmul(np.zeros((1, 3, 128), np.float32), np.zeros((1, 8, 128), np.float32), True)


@pytest.mark.parametrize("kv_group", [1, 4])
def test_exact_attention_composition(kv_group):
    """orc_attention_decode == the reference's MatMul -> SoftMax(lengths) -> MatMul, bit for bit
    (self-attention: n = cache length; memory attention: n = padded length + lengths)."""
    R = _ref()
    rng = np.random.default_rng(24 + kv_group)
    B, H, dh = 5, 8, 64
    rows = B * kv_group
    for T, masked in [(1, False), (6, False), (31, False), (64, False), (33, True), (8, True)]:
        q = rng.standard_normal((rows, H, dh)).astype(np.float32)
        k = rng.standard_normal((B, H, T, dh)).astype(np.float32)
        v = rng.standard_normal((B, H, T, dh)).astype(np.float32)
        lengths = rng.integers(1, T + 1, B).astype(np.int32) if masked else None
        qm = q.reshape(B, kv_group, H, dh).transpose(0, 2, 1, 3).reshape(B * H, kv_group, dh)
        scores = R.matmul(qm, k.reshape(B * H, T, dh), True, 0.125)
        rep = np.repeat(lengths if masked else np.full(B, T, np.int32), H * kv_group)
        probs = R.softmax(scores.reshape(B * H * kv_group, T), rep, False)
        out = R.matmul(probs.reshape(B * H, kv_group, T), v.reshape(B * H, T, dh), False, 1.0)
        out = out.reshape(B, H, kv_group, dh).transpose(0, 2, 1, 3).reshape(rows, H * dh)
        mine = O.attention_decode(q, k, v, lengths, T, kv_group, 0.125)
        assert np.array_equal(_bits(mine), _bits(out)), (T, masked)


def test_ref_attention_composition():
    """The oracle's fused single-query attention == the reference's MatMul/SoftMax/MatMul chain."""
    R = _ref()
    rng = np.random.default_rng(4)
    rows, H, dh, T = 6, 4, 16, 11
    q = rng.standard_normal((rows, H, dh)).astype(np.float32)
    k = rng.standard_normal((rows, H, T, dh)).astype(np.float32)
    v = rng.standard_normal((rows, H, T, dh)).astype(np.float32)
    lengths = rng.integers(1, T + 1, rows).astype(np.int32)
    scale = 1 / np.sqrt(dh)
    scores = R.matmul(q.reshape(rows * H, 1, dh), k.reshape(rows * H, T, dh), True, scale)
    probs = R.softmax(scores.reshape(rows * H, T), np.repeat(lengths, H), False)
    out = R.matmul(probs.reshape(rows * H, 1, T), v.reshape(rows * H, T, dh), False, 1.0)
    mine = O.attention_decode(q, k, v, lengths, 0, 1, scale)
    np.testing.assert_allclose(mine, out.reshape(rows, H * dh), atol=2e-6, rtol=1e-5)


def _processor_cases():
    """(logits, history, kwargs) covering every branch of the three processors."""
    rng = np.random.default_rng(5)
    rows, vocab = 6, 50
    cases = []
    for length in (0, 1, 2, 5, 17):
        lg = rng.standard_normal((rows, vocab)).astype(np.float32) * 3
        # a small alphabet makes repeated tokens and repeated n-grams certain
        seq = rng.integers(0, 6, (rows, length)).astype(np.int32)
        tail2 = seq[0, -2:].tolist() if length >= 2 else [1, 2]
        suppress = [[7], tail2 + [9], [seq[1, -1] if length else 3, 11], [], [3, 4, 5, 6, 8]]
        for kw in ({"penalty": 1.5}, {"penalty": 0.7}, {"ngram": 1}, {"ngram": 2}, {"ngram": 3},
                   {"ngram": 6}, {"suppress": suppress},
                   {"penalty": 1.2, "ngram": 2, "suppress": suppress}):
            cases.append((lg, seq, kw))
    return cases


def test_ref_logits_processors_exact():
    """orc_process_logits == the reference's RepetitionPenalty / NoRepeatNgram / SuppressSequences
    classes (src/decoding_utils.cc) bit for bit, incl. empty histories, n-grams longer than the
    history, tokens repeated in the history (penalised once) and empty banned sequences."""
    R = _ref()
    for lg, seq, kw in _processor_cases():
        a = O.process_logits(lg, seq, **kw)
        b = R.process_logits(lg, seq, **kw)
        assert np.array_equal(a, b), (seq.shape, kw)
    # the penalty must not compound on a repeated token
    lg = np.full((1, 8), 2.0, np.float32)
    out = O.process_logits(lg, np.array([[3, 3, 3]], np.int32), penalty=2.0)
    assert out[0, 3] == 1.0 and out[0, 2] == 2.0


PENALIZE_KAT = dict(  # tests/primitives_test.cc:33-52 (PrimitiveTest.PenalizePreviousTokens)
    scores=np.array([[0.6, 0.2, -1.2, 0.1], [0.3, 0.5, -1.3, 0.2]], np.float32),
    previous_ids=np.array([[2, 2], [1, 2]], np.int32), penalty=1.2)


def penalize_kat_expected():
    e = PENALIZE_KAT["scores"].copy()
    p = np.float32(PENALIZE_KAT["penalty"])
    e[0, 2] *= p
    e[1, 1] /= p
    e[1, 2] *= p
    return e


def test_kat_penalize_previous_tokens_and_logsumexp():
    """The reference's own vectors: PrimitiveTest.PenalizePreviousTokens (a token repeated in the
    history is penalised once) and PrimitiveTest.LogSumExp (tests/primitives_test.cc:25-52)."""
    got = O.process_logits(PENALIZE_KAT["scores"], PENALIZE_KAT["previous_ids"],
                           penalty=PENALIZE_KAT["penalty"])
    assert np.array_equal(got, penalize_kat_expected())
    x = np.array([[0.6, 0.2, -1.2, 0.1, 0.3, 0.5, -1.3, 0.2]], np.float32)
    lse = x[0, 0] - O.softmax(x, None, True)[0, 0]
    assert abs(float(lse) - 2.1908040046691895) < 1e-6


def test_kat_float_gemm_family():
    """OpDeviceFPTest.Gemm / GemmBias / GemmResidual / GemmGELU (tests/ops_test.cc:516-680): the
    float Gemm with fused bias, residual and activation. beta = 1 cases (`+ y`) are expressed as a
    residual, which is how layers::Dense uses the op."""
    g = KAT["GemmFloat"]
    a = np.array(g["a"], np.float32).reshape(g["a_shape"])
    b = np.array(g["b"], np.float32).reshape(g["b_shape"])
    y = np.array(g["y"], np.float32).reshape(g["y_shape"])
    res = np.array(g["residual"], np.float32).reshape(g["y_shape"])

    def run(**kw):
        return O.gemm_f32(a, b, trans_b=False, **kw).ravel()

    tol = dict(atol=1e-5, rtol=0)
    np.testing.assert_allclose(run(residual=y), g["a_b_plus_y"], **tol)
    np.testing.assert_allclose(O.gemm_f32(a, np.ascontiguousarray(b.T), trans_b=True, residual=y).ravel(),
                               g["a_b_plus_y"], **tol)
    np.testing.assert_allclose(run(bias=np.array(g["bias"], np.float32), residual=y),
                               g["a_b_plus_bias_plus_y"], **tol)
    np.testing.assert_allclose(run(bias=np.array(g["residual_bias"], np.float32), residual=y + res),
                               g["a_b_plus_bias_plus_y_plus_residual"], **tol)
    np.testing.assert_allclose(run(residual=y + res), g["a_b_plus_y_plus_residual"], **tol)
    np.testing.assert_allclose(run(bias=np.array(g["gelu_bias"], np.float32), act=3),
                               g["gelu_a_b_plus_bias"], **tol)
    np.testing.assert_allclose(run(act=3), g["gelu_a_b"], **tol)


def test_kat_topk_variants_and_topp_mask():
    """OpDeviceTest.TopKVariableDepth / TopKChangeK and OpDeviceFPTest.TopPMask
    (tests/ops_test.cc:726-790)."""
    t = KAT["TopKVariableDepth"]
    v, i = O.topk(np.array(t["x"], np.float32).reshape(t["shape"]), t["k"])
    assert np.array_equal(v.ravel(), np.array(t["values"], np.float32))
    assert np.array_equal(i.ravel(), np.array(t["indices"], np.int32))
    t = KAT["TopKChangeK"]
    x = np.array(t["x"], np.float32).reshape(t["shape"])
    for k in (2, 3):
        v, i = O.topk(x, k)
        assert np.array_equal(v.ravel(), np.array(t["values_k%d" % k], np.float32))
        assert np.array_equal(i.ravel(), np.array(t["indices_k%d" % k], np.int32))
    t = KAT["TopPMask"]
    x = np.array(t["x"], np.float32).reshape(t["shape"])
    y = O.topp_mask(x, t["p"])
    kept = np.array(t["kept"], bool)
    assert np.array_equal(np.isfinite(y), kept)
    assert np.array_equal(y[kept], x[kept])


def test_kat_masked_softmax_triangular_and_activations():
    """OpDeviceFPTest.MaskedSoftMaxTriangular (tests/ops_test.cc:841-878: lengths from
    prepare_length_mask(lengths, heads = 2, queries = 3, triangular) = min(length, q + 1)) and the
    activation functors ReLU / GELU / GELUTanh / GELUSigmoid / Sigmoid / Swish / Tanh (:1015-1095)."""
    t = KAT["MaskedSoftMaxTriangular"]
    b, h, q, kk = t["shape"]
    x = np.array(t["x"], np.float32).reshape(b * h * q, kk)
    lengths = np.array([min(t["lengths"][bi], qi + 1) for bi in range(b) for _ in range(h)
                        for qi in range(q)], np.int32)
    np.testing.assert_allclose(O.softmax(x, lengths, False).ravel(), t["y"], atol=1e-5, rtol=0)
    for name, a in KAT["Activations"].items():
        np.testing.assert_allclose(O.activation(np.array(a["x"], np.float32), a["act"]),
                                   np.array(a["y"], np.float32), atol=1e-5, rtol=0, err_msg=name)


def test_exact_division_identity():
    """The five-operation division of the MFMA-bound GEMM epilogue (device_utils.h exact_div; the
    reference's Dequantize divides by the weight scale, src/cpu/kernels.cc:666-673) returns the bits
    of t / d on the guarded domain: 2e8 random and adversarial pairs here, 1e11 when it was
    introduced (round 2)."""
    import ctypes as C

    lib = O.lib()
    lib.orc_exact_div_check.restype = C.c_int64
    lib.orc_exact_div_check.argtypes = [C.c_int64, C.c_uint64]
    assert lib.orc_exact_div_check(200_000_000, 20260921) == 0
    lib.orc_exact_div.restype = C.c_float
    lib.orc_exact_div.argtypes = [C.c_float, C.c_float]
    for t, d in [(0.0, 3.0), (1.0, 3.0), (-7.5, 0.3), (2.0 ** 60, 2.0 ** -30), (2.0 ** -29, 2.0 ** 30)]:
        assert np.float32(lib.orc_exact_div(t, d)) == np.float32(t) / np.float32(d)
