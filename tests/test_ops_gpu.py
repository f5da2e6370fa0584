"""-m gpu parity tests: every hot-path op through the C ABI (include/ct2_mi355.h) on a real
MI355X against the plain-C oracle (oracle/ct2_oracle.c) on the same seeded inputs.

Bar: bit-exact for integer/byte/index outputs (quantised values, int32 GEMM, top-k indices,
gathers); float outputs within the tolerance written next to each assert (the op is computed in
fp32 with the oracle's formula order; differences come only from reduction order / libm)."""
import numpy as np
import pytest

import _oracle as O

pytestmark = pytest.mark.gpu

# Decoder shapes from SURVEY.md section 8(a): Transformer-base and Llama-3-8B rows/depths.
QUANT_SHAPES = [(32, 512), (128, 2048), (16, 4096), (3, 96), (1, 32), (5, 7), (2, 14336)]


# quantisation flag word: bit 0 round-to-nearest-even, bit 1 the CPU backends' u8 shift
QUANT_MODES = [0, 1, 2, 3]


@pytest.mark.parametrize("rows,depth", QUANT_SHAPES)
@pytest.mark.parametrize("rbc", QUANT_MODES)
def test_quantize_s8(gpu, rows, depth, rbc):
    import _gpuops as G
    rng = np.random.default_rng(rows * 1000 + depth)
    x = (rng.standard_normal((rows, depth)) * 3).astype(np.float32)
    x[0, :] = 0.0  # zero row -> scale 1 (tests/ops_test.cc:969-991)
    q, s = G.quantize_s8(x, rbc)
    qo, so = O.quantize_s8(x, rbc)
    assert np.array_equal(s, so)  # 127/amax is one IEEE division: must be bit-identical
    assert np.array_equal(q, qo)


GEMM_SHAPES = [
    (32, 1536, 512), (32, 512, 512), (32, 2048, 512), (32, 512, 2048), (32, 32000, 512),
    (128, 1536, 512), (1024, 2048, 512), (16, 6144, 4096), (1, 64, 64), (3, 43, 32),
    (33, 70, 48), (65, 130, 144), (200, 100, 16), (7, 9, 13),
    # >= 96 tiles of 256x256 with K % 128 == 0: the DMA-staged 256x256 kernel (gemm_s8_big.hip),
    # full tiles and ragged M/N edges
    (2560, 2560, 256), (2500, 2700, 384), (513, 12289, 640),
    # > 256 tiles: the persistent workgroups of gemm_s8_big3.hip walk more than one tile each
    # (full tiles; ragged edges with a row pitch that allows / forbids 16-byte stores)
    (4352, 4352, 256), (4200, 4404, 384), (4100, 4299, 256),
]


@pytest.mark.parametrize("m,n,k", GEMM_SHAPES)
def test_gemm_s8_exact(gpu, m, n, k):
    import _gpuops as G
    rng = np.random.default_rng(m + n + k)
    a = rng.integers(-127, 128, (m, k), dtype=np.int8)
    b = rng.integers(-127, 128, (n, k), dtype=np.int8)
    c = G.gemm_s8(a, b)
    assert np.array_equal(c, O.gemm_s8(a, b))


def test_gemm_s8_transpose_detecting(gpu):
    """A = I-like pattern with ASYMMETRIC B: catches row/col swaps in the MFMA C layout."""
    import _gpuops as G
    m, n, k = 32, 64, 64
    a = np.zeros((m, k), np.int8)
    for i in range(m):
        a[i, i] = 1
    b = (np.arange(n)[:, None] * 2 + np.arange(k)[None, :] * 1 - 60).clip(-127, 127).astype(np.int8)
    c = G.gemm_s8(a, b)
    assert np.array_equal(c, O.gemm_s8(a, b))
    assert np.array_equal(c, b[:, :m].T.astype(np.int32))


def test_gemm_s8_extremes(gpu):
    """|acc| up to 127*127*K: no saturation anywhere (the reference CPU u8s8 path can saturate
    without VNNI; the MFMA path must not)."""
    import _gpuops as G
    m, n, k = 16, 32, 4096
    a = np.full((m, k), 127, np.int8)
    b = np.full((n, k), -127, np.int8)
    b[::2] = 127
    c = G.gemm_s8(a, b)
    assert np.array_equal(c, O.gemm_s8(a, b))
    assert abs(int(c[0, 0])) == 127 * 127 * k


@pytest.mark.parametrize("act", [-1, 0, 1, 2, 3, 4, 5, 6])
def test_dequantize_gemm_output(gpu, act):
    import _gpuops as G
    rng = np.random.default_rng(7 + act)
    m, n = 33, 130
    c = rng.integers(-200000, 200000, (m, n), dtype=np.int32)
    sa = rng.uniform(5, 80, m).astype(np.float32)
    sb = rng.uniform(50, 4000, n).astype(np.float32)
    bias = rng.standard_normal(n).astype(np.float32)
    y = G.dequantize_gemm_output(c, sa, sb, bias, act)
    yo = O.dequantize_gemm_output(c, sa, sb, bias, act)
    # every activation is the reference's AVX2 functor restated from IEEE operations
    # (activation_exact.h = oracle = reference library, bit for bit on the CPU): identical here too
    assert np.array_equal(y.view(np.uint32), yo.view(np.uint32))


@pytest.mark.parametrize("m,n,k", [(32, 1536, 512), (32, 512, 2048), (128, 2048, 512),
                                   (16, 4096, 4096), (5, 43, 32), (70, 96, 32),
                                   (2500, 2700, 256), (4200, 4404, 256)])
@pytest.mark.parametrize("act,with_res", [(-1, True), (0, False), (2, False), (-1, False)])
def test_dense_fused_vs_unfused_oracle(gpu, m, n, k, act, with_res):
    """layers::Dense int8 (Quantize -> Gemm -> Dequantize(+bias,act) -> Add residual) as ONE fused
    launch after the quantize kernel, against the oracle's four separate steps."""
    import _gpuops as G
    rng = np.random.default_rng(m * 7 + n + k + act)
    x = rng.standard_normal((m, k)).astype(np.float32)
    w = rng.standard_normal((n, k)).astype(np.float32) * 0.02
    wq, ws = O.quantize_s8(w, True)  # weights quantised offline (model_spec.py:233-240)
    bias = (rng.standard_normal(n) * 0.01).astype(np.float32)
    res = rng.standard_normal((m, n)).astype(np.float32) if with_res else None
    q, s = G.quantize_s8(x, True)
    y = G.gemm_s8_dequant(q, s, wq, ws, bias, res, act)
    yo = O.dense_s8(x, wq, ws, bias, res, act, True)
    assert np.array_equal(y, yo)  # Swish included: the epilogue's activations are exact


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


STREAM_SHAPES = [(16, 6144, 4096), (16, 4096, 4096), (16, 4096, 14336), (16, 128256, 4096),
                 (5, 14336, 4096), (1, 4096, 8192), (16, 8208, 4096)]


@pytest.mark.parametrize("m,n,k", STREAM_SHAPES)
def test_dense_streaming_form(gpu, m, n, k, monkeypatch):
    """The persistent weight-streaming GEMM (gemm_s8_stream.hip; Llama-shape decode rows, weights
    >= 8 MB) against the one-shot kernel it replaces (CT2X_STREAM_GEMM is read once per process, so
    the comparison partner is the oracle's Dense chain) - bit-identical; int32 form too."""
    import _gpuops as G
    rng = np.random.default_rng(m + n + k)
    x = rng.standard_normal((m, k)).astype(np.float32)
    wq = rng.integers(-127, 128, (n, k), dtype=np.int8)
    ws = rng.uniform(50, 4000, n).astype(np.float32)
    bias = (rng.standard_normal(n) * 0.01).astype(np.float32)
    res = rng.standard_normal((m, n)).astype(np.float32)
    q, s = G.quantize_s8(x, 3)
    c = G.gemm_s8(q, wq)
    co = O.gemm_s8(q, wq)
    assert np.array_equal(c, co)
    y = G.gemm_s8_dequant(q, s, wq, ws, bias, res, 0)
    yo = _f32(res) + O.dequantize_gemm_output(co, s, ws, bias, 0)
    assert np.array_equal(y, yo)
    # the same launch reading the fragment-major copy of the weight
    if G.stream_packable(n, k):
        assert np.array_equal(G.gemm_s8_dequant_packed(q, s, wq, ws, bias, res, 0), yo)
    else:
        assert (m, n, k) == (16, 8208, 4096)
        with pytest.raises(RuntimeError):
            G.gemm_s8_dequant_packed(q, s, wq, ws, bias, res, 0)


@pytest.mark.parametrize("m,n,k", [(16, 14336, 4096), (7, 14336, 4096), (16, 4096, 8192)])
def test_dense_glu_streaming_form_with_row_amax(gpu, m, n, k):
    """Gated pair in the streaming form + the row absmax it leaves for the following Quantize."""
    import _gpuops as G
    rng = np.random.default_rng(m + n + k)
    x = rng.standard_normal((m, k)).astype(np.float32)
    w0q = rng.integers(-127, 128, (n, k), dtype=np.int8)
    w1q = rng.integers(-127, 128, (n, k), dtype=np.int8)
    w0s = rng.uniform(2000, 9000, n).astype(np.float32)
    w1s = rng.uniform(2000, 9000, n).astype(np.float32)
    q, s = G.quantize_s8(x, 3)
    y, amax = G.gemm_s8_glu_amax(q, s, w0q, w0s, None, w1q, w1s, None, 0)
    c0, c1 = O.gemm_s8(q, w0q), O.gemm_s8(q, w1q)
    yo = O.dequantize_gemm_output(c1, s, w1s, None, -1) * O.dequantize_gemm_output(c0, s, w0s, None, 0)
    assert np.array_equal(y, yo)
    assert np.array_equal(amax, np.abs(yo).max(axis=1))
    y2 = G.gemm_s8_glu(q, s, w0q, w0s, None, w1q, w1s, None, 0)
    assert np.array_equal(y2, yo)
    y3, amax3 = G.gemm_s8_glu_packed(q, s, w0q, w0s, None, w1q, w1s, None, 0)
    assert np.array_equal(y3, yo) and np.array_equal(amax3, amax)


@pytest.mark.parametrize("m,n", [(16, 6144), (16, 4096), (16, 128256), (5, 6144), (1, 4096)])
@pytest.mark.parametrize("use_residual,flags", [(False, 3), (True, 3), (False, 1), (False, 0)])
def test_dense_streaming_norm_prologue(no_leader_forms, m, n, use_residual, flags):
    """RMSNorm + Quantize as the prologue of the persistent streaming Dense (gemm_s8_stream.hip
    AMODE 2: the reference's eight-lane statistics chain run chunk-wise through LDS by four waves
    while the weight stream is already in flight) against the launches it replaces
    (ct2x_norm_quantize_s8 + ct2x_gemm_s8_dequant_packed) and against the oracle's RMSNorm ->
    Quantize -> Gemm -> Dequantize chain: bit for bit, both gamma forms, every Quantize mode."""
    import _gpuops as G
    k = 4096
    rng = np.random.default_rng(m + n + flags + int(use_residual))
    x = (rng.standard_normal((m, k)) * 2.5).astype(np.float32)
    x[:, 7] *= 30.0   # an outlier channel, as residual streams have
    if m > 2:
        x[2, :] = 0.0  # zero row -> absmax 0 -> scale 1
    gamma = (1 + 0.2 * rng.standard_normal(k)).astype(np.float32)
    if use_residual:
        gamma -= 1.0
    wq = rng.integers(-127, 128, (n, k), dtype=np.int8)
    ws = rng.uniform(50, 4000, n).astype(np.float32)
    bias = (rng.standard_normal(n) * 0.01).astype(np.float32)
    res = rng.standard_normal((m, n)).astype(np.float32)
    y, amax = G.norm_gemm_s8_dequant_packed(x, gamma, 1e-5, use_residual, flags, wq, ws, bias, res, 2)
    _, q, s = G.norm_quantize_s8(x, gamma, None, 1e-5, use_residual, flags, False)
    y2 = G.gemm_s8_dequant_packed(q, s, wq, ws, bias, res, 2)
    assert np.array_equal(y, y2)
    assert np.array_equal(amax, np.abs(y).max(axis=1))
    if True:
        xn = O.rms_norm(x, gamma, 1e-5, use_residual)
        qo, so = O.quantize_s8(xn, flags)
        assert np.array_equal(q, qo) and np.array_equal(s, so)
        yo = _f32(res) + O.dequantize_gemm_output(O.gemm_s8(qo, wq), so, ws, bias, 2)
        assert np.array_equal(y, yo)


@pytest.mark.parametrize("m", [16, 3])
def test_dense_glu_streaming_norm_prologue(no_leader_forms, m):
    """The same prologue in front of the gated pair (Llama-3-8B feed-forward input at decode rows)."""
    import _gpuops as G
    n, k = 14336, 4096
    rng = np.random.default_rng(m)
    x = (rng.standard_normal((m, k)) * 2.5).astype(np.float32)
    gamma = (1 + 0.2 * rng.standard_normal(k)).astype(np.float32)
    w0q = rng.integers(-127, 128, (n, k), dtype=np.int8)
    w1q = rng.integers(-127, 128, (n, k), dtype=np.int8)
    w0s = rng.uniform(2000, 9000, n).astype(np.float32)
    w1s = rng.uniform(2000, 9000, n).astype(np.float32)
    y, amax = G.norm_gemm_s8_glu_packed(x, gamma, 1e-5, False, 3, w0q, w0s, None, w1q, w1s, None, 2)
    _, q, s = G.norm_quantize_s8(x, gamma, None, 1e-5, False, 3, False)
    y2, amax2 = G.gemm_s8_glu_packed(q, s, w0q, w0s, None, w1q, w1s, None, 2)
    assert np.array_equal(y, y2) and np.array_equal(amax, amax2)
    xn = O.rms_norm(x, gamma, 1e-5, False)
    qo, so = O.quantize_s8(xn, 3)
    c0, c1 = O.gemm_s8(qo, w0q), O.gemm_s8(qo, w1q)
    yo = O.dequantize_gemm_output(c1, so, w1s, None, -1) * O.dequantize_gemm_output(c0, so, w0s, None, 2)
    assert np.array_equal(y, yo)


@pytest.mark.parametrize("m,n,parts", [(16, 4096, 8), (16, 4096, 1), (7, 6144, 4), (16, 128256, 1)])
@pytest.mark.parametrize("flags", [3, 1, 0])
def test_dense_streaming_quantize_on_load(no_leader_forms, m, n, parts, flags):
    """ops::Quantize folded into the streaming Dense's fragment fetch (AMODE 1), the row absmax given
    as per-workgroup partial maxima (what the d_head 128 attention leaves): against Quantize +
    ct2x_gemm_s8_dequant_packed and the oracle's Dense chain, bit for bit."""
    import _gpuops as G
    k = 4096
    rng = np.random.default_rng(m + n + parts + flags)
    x = (rng.standard_normal((m, k)) * 1.7).astype(np.float32)
    if m > 2:
        x[2, :] = 0.0
    wq = rng.integers(-127, 128, (n, k), dtype=np.int8)
    ws = rng.uniform(50, 4000, n).astype(np.float32)
    bias = (rng.standard_normal(n) * 0.01).astype(np.float32)
    res = rng.standard_normal((m, n)).astype(np.float32)
    amax_parts = np.abs(x).reshape(m, parts, k // parts).max(axis=2)
    y = G.gemm_s8_dequant_quantizing_packed(x, amax_parts, flags, wq, ws, bias, res, -1)
    q, s = G.quantize_s8(x, flags)
    y2 = G.gemm_s8_dequant_packed(q, s, wq, ws, bias, res, -1)
    assert np.array_equal(y, y2)
    assert np.array_equal(y, O.dense_s8(x, wq, ws, bias, res, -1, flags))


@pytest.fixture
def leader_forms(gpu):
    """The streaming Dense's prologues done once by the launch's first sixteen workgroups
    (CT2X_STREAM_PROLOGUE bits 2 and 3, gemm_s8_stream.hip AMODE 3 / 4)."""
    import _gpuops as G
    prev = G.set_stream_prologue(12)
    yield
    G.set_stream_prologue(prev)


@pytest.fixture
def no_leader_forms(gpu):
    import _gpuops as G
    prev = G.set_stream_prologue(3)
    yield
    G.set_stream_prologue(prev)


@pytest.mark.parametrize("m,n", [(16, 6144), (16, 4096), (16, 128256), (5, 6144), (1, 4096)])
@pytest.mark.parametrize("use_residual,flags", [(False, 3), (True, 3), (False, 1), (False, 0)])
def test_dense_streaming_leader_norm(leader_forms, m, n, use_residual, flags):
    """RMSNorm + Quantize by sixteen leader workgroups of the streaming Dense's own launch (AMODE 3:
    one row each, int8 row + scale stored write-through, a tag word per row; every workgroup's first
    weight ring is in flight meanwhile): the same bits as the launches it replaces and as the
    oracle's RMSNorm -> Quantize -> Gemm -> Dequantize chain; repeated launches (fresh tags each
    time the scratch is fresh) agree with themselves."""
    import _gpuops as G
    k = 4096
    rng = np.random.default_rng(1000 + m + n + flags + int(use_residual))
    x = (rng.standard_normal((m, k)) * 2.5).astype(np.float32)
    x[:, 7] *= 30.0
    if m > 2:
        x[2, :] = 0.0
    gamma = (1 + 0.2 * rng.standard_normal(k)).astype(np.float32)
    if use_residual:
        gamma -= 1.0
    wq = rng.integers(-127, 128, (n, k), dtype=np.int8)
    ws = rng.uniform(50, 4000, n).astype(np.float32)
    bias = (rng.standard_normal(n) * 0.01).astype(np.float32)
    res = rng.standard_normal((m, n)).astype(np.float32)
    y, amax = G.norm_gemm_s8_dequant_packed(x, gamma, 1e-5, use_residual, flags, wq, ws, bias, res, 2)
    _, q, s = G.norm_quantize_s8(x, gamma, None, 1e-5, use_residual, flags, False)
    y2 = G.gemm_s8_dequant_packed(q, s, wq, ws, bias, res, 2)
    assert np.array_equal(y, y2)
    assert np.array_equal(amax, np.abs(y).max(axis=1))
    xn = O.rms_norm(x, gamma, 1e-5, use_residual)
    qo, so = O.quantize_s8(xn, flags)
    yo = _f32(res) + O.dequantize_gemm_output(O.gemm_s8(qo, wq), so, ws, bias, 2)
    assert np.array_equal(y, yo)
    for _ in range(3):
        y3, _ = G.norm_gemm_s8_dequant_packed(x, gamma, 1e-5, use_residual, flags, wq, ws, bias, res, 2)
        assert np.array_equal(y3, y)


@pytest.mark.parametrize("m", [16, 3])
def test_dense_glu_streaming_leader_norm(leader_forms, m):
    """The leader norm in front of the gated pair (Llama-3-8B feed-forward input at decode rows)."""
    import _gpuops as G
    n, k = 14336, 4096
    rng = np.random.default_rng(100 + m)
    x = (rng.standard_normal((m, k)) * 2.5).astype(np.float32)
    gamma = (1 + 0.2 * rng.standard_normal(k)).astype(np.float32)
    w0q = rng.integers(-127, 128, (n, k), dtype=np.int8)
    w1q = rng.integers(-127, 128, (n, k), dtype=np.int8)
    w0s = rng.uniform(2000, 9000, n).astype(np.float32)
    w1s = rng.uniform(2000, 9000, n).astype(np.float32)
    y, amax = G.norm_gemm_s8_glu_packed(x, gamma, 1e-5, False, 3, w0q, w0s, None, w1q, w1s, None, 2)
    _, q, s = G.norm_quantize_s8(x, gamma, None, 1e-5, False, 3, False)
    y2, amax2 = G.gemm_s8_glu_packed(q, s, w0q, w0s, None, w1q, w1s, None, 2)
    assert np.array_equal(y, y2) and np.array_equal(amax, amax2)
    xn = O.rms_norm(x, gamma, 1e-5, False)
    qo, so = O.quantize_s8(xn, 3)
    c0, c1 = O.gemm_s8(qo, w0q), O.gemm_s8(qo, w1q)
    yo = O.dequantize_gemm_output(c1, so, w1s, None, -1) * O.dequantize_gemm_output(c0, so, w0s, None, 2)
    assert np.array_equal(y, yo)


@pytest.mark.parametrize("m,n,k,parts", [(16, 4096, 4096, 8), (16, 4096, 4096, 1), (7, 6144, 4096, 4),
                                         (16, 4096, 14336, 1), (9, 4096, 14336, 1), (16, 4096, 8192, 2),
                                         (16, 128256, 4096, 1)])
@pytest.mark.parametrize("flags", [3, 1, 0])
def test_dense_streaming_leader_quantize(leader_forms, m, n, k, parts, flags):
    """ops::Quantize by sixteen leader workgroups of the streaming Dense's own launch (AMODE 4), the row
    absmax given as the producer's partial maxima - any K with a streaming form (4096: the attention
    output projection, 14336: the Llama-3-8B feed-forward output): against Quantize +
    ct2x_gemm_s8_dequant_packed and the oracle's Dense chain, bit for bit."""
    import _gpuops as G
    rng = np.random.default_rng(m + n + k + parts + flags)
    x = (rng.standard_normal((m, k)) * 1.7).astype(np.float32)
    if m > 2:
        x[2, :] = 0.0
    wq = rng.integers(-127, 128, (n, k), dtype=np.int8)
    ws = rng.uniform(50, 4000, n).astype(np.float32)
    bias = (rng.standard_normal(n) * 0.01).astype(np.float32)
    res = rng.standard_normal((m, n)).astype(np.float32)
    amax_parts = np.abs(x).reshape(m, parts, k // parts).max(axis=2)
    y = G.gemm_s8_dequant_quantizing_packed(x, amax_parts, flags, wq, ws, bias, res, -1)
    q, s = G.quantize_s8(x, flags)
    y2 = G.gemm_s8_dequant_packed(q, s, wq, ws, bias, res, -1)
    assert np.array_equal(y, y2)
    assert np.array_equal(y, O.dense_s8(x, wq, ws, bias, res, -1, flags))


def test_streaming_prologue_envelope(no_leader_forms):
    """Shapes outside the prologue forms are refused, not computed some other way."""
    import _gpuops as G
    rng = np.random.default_rng(0)
    for m, n, k in [(17, 6144, 4096), (16, 4096, 8192), (16, 512, 512)]:
        x = rng.standard_normal((m, k)).astype(np.float32)
        wq = rng.integers(-127, 128, (n, k), dtype=np.int8)
        ws = np.ones(n, np.float32)
        if not G.stream_packable(n, k):
            continue
        with pytest.raises(RuntimeError):
            G.norm_gemm_s8_dequant_packed(x, np.ones(k, np.float32), 1e-6, False, 3, wq, ws)
        with pytest.raises(RuntimeError):
            G.gemm_s8_dequant_quantizing_packed(x, np.abs(x).max(axis=1, keepdims=True), 3, wq, ws)


@pytest.mark.parametrize("scale_exp", [0, 31, -31])
def test_dense_big_tile_division_guard(gpu, scale_exp):
    """The 256 x 256 kernel divides by the weight scale with exact_div (device_utils.h) when every
    scale of the tile is inside [2^-30, 2^30] and with the IEEE sequence otherwise: both arms return
    the oracle's bits (Dequantize formula order: src/cpu/kernels.cc:666-673)."""
    import _gpuops as G
    m, n, k = 2500, 2700, 256
    rng = np.random.default_rng(99 + scale_exp)
    x = rng.standard_normal((m, k)).astype(np.float32)
    w = rng.standard_normal((n, k)).astype(np.float32) * 0.02
    wq, ws = O.quantize_s8(w, True)
    ws = ws.copy()
    if scale_exp:
        ws[::97] = np.float32(2.0 ** scale_exp) * np.float32(1.37)  # outside the guarded range
    bias = (rng.standard_normal(n) * 0.01).astype(np.float32)
    q, s = G.quantize_s8(x, True)
    y = G.gemm_s8_dequant(q, s, wq, ws, bias, None, 0)
    c = O.gemm_s8(q, wq)
    yo = O.dequantize_gemm_output(c, s, ws, bias, 0)
    assert np.array_equal(y, yo)


@pytest.mark.parametrize("act", [1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("with_bias", [True, False])
def test_dense_big_tile_activation_epilogue(gpu, act, with_bias):
    """The 256 x 256 kernel's straight-line epilogue of interior tiles (round 6: constant activation,
    four interleaved elements, the exponential activations' division on the reduced IEEE sequence
    with a wave-uniform fall-back to the plain division) and the general epilogue of the edge tiles
    return the oracle's bits for every activation of the reference (src/cpu/kernels.cc:146-207).
    Rows of zeros and rows scaled by 2^12 put exact zeros, -0.0 and values far outside the reduced
    sequence's range next to ordinary ones."""
    import _gpuops as G
    m, n, k = 1100, 1300, 256  # 4 x 5 interior tiles + an edge row and column of tiles
    rng = np.random.default_rng(31 * act + with_bias)
    x = rng.standard_normal((m, k)).astype(np.float32)
    x[5] = 0
    x[300:310] *= np.float32(4096)
    x[700] = -np.abs(x[700]) * np.float32(1e-6)
    w = rng.standard_normal((n, k)).astype(np.float32) * 0.02
    w[17] = 0
    wq, ws = O.quantize_s8(w, True)
    bias = (rng.standard_normal(n) * 0.01).astype(np.float32) if with_bias else None
    q, s = G.quantize_s8(x, True)
    y = G.gemm_s8_dequant(q, s, wq, ws, bias, None, act)
    yo = O.dequantize_gemm_output(O.gemm_s8(q, wq), s, ws, bias, act)
    assert np.array_equal(y.view(np.uint32), yo.view(np.uint32))


def test_reduced_division_equals_plain_division(gpu):
    """div_noscale (gemm_s8_big3.hip) = the compiler's IEEE division minus the steps that are the
    identity inside the admitted range: against `/` on 2 x 10^9 results on the device."""
    import ctypes as C
    from ctranslate2_amd import _lib
    bad = C.c_uint64(1)
    for seed in (1, 77):
        _lib.check(_lib.lib().ct2x_debug_fast_div_check(C.c_uint64(500_000_000), C.c_uint32(seed), C.byref(bad)))
        assert bad.value == 0


@pytest.mark.parametrize("m,n,k", [(16, 14336, 4096), (32, 2048, 512), (5, 96, 64), (16, 1000, 2048)])
@pytest.mark.parametrize("act,with_bias", [(2, False), (0, True), (3, True)])
def test_dense_glu_fused(gpu, m, n, k, act, with_bias):
    """ffn_glu input pair (src/layers/transformer.cc:21-51): act(Dense0(x)) * Dense1(x) as one
    launch against the oracle's two Dense chains and the element-wise product."""
    import _gpuops as G
    rng = np.random.default_rng(m + n + k + act)
    x = rng.standard_normal((m, k)).astype(np.float32)
    w0q, w0s = O.quantize_s8((rng.standard_normal((n, k)) * 0.02).astype(np.float32), True)
    w1q, w1s = O.quantize_s8((rng.standard_normal((n, k)) * 0.02).astype(np.float32), True)
    b0 = (rng.standard_normal(n) * 0.01).astype(np.float32) if with_bias else None
    b1 = (rng.standard_normal(n) * 0.01).astype(np.float32) if with_bias else None
    q, s = G.quantize_s8(x, True)
    y = G.gemm_s8_glu(q, s, w0q, w0s, b0, w1q, w1s, b1, act)
    yo = O.dense_s8(x, w1q, w1s, b1, None, -1, True) * O.dense_s8(x, w0q, w0s, b0, None, act, True)
    if act == 0:
        assert np.array_equal(y, yo)
    else:
        np.testing.assert_allclose(y, yo, rtol=2e-5, atol=1e-6)  # libm expf/erff in the gate, times up


def _to_bf16_bits(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def _from_bf16_bits(b):
    return (b.astype(np.uint32) << 16).view(np.float32)


@pytest.mark.parametrize("m,n,k", [(32, 1536, 512), (16, 14336, 4096), (128, 2048, 512), (5, 43, 40),
                                   (1000, 777, 264), (2048, 512, 2048),
                                   # decode rows with K slices over workgroups (2 / 8 / 8 / 4 / 4 slices)
                                   (40, 1280, 1280), (40, 1280, 5120), (16, 4096, 4096),
                                   (64, 2048, 2048), (33, 1000, 2048),
                                   # 129 .. 256 rows (round 6, gemm_h_rows256_kernel): all rows x 128 columns
                                   # x a K slice per workgroup - 4 / 16 / 8 / 4 slices, ragged rows and
                                   # columns, and the single-slice epilogue
                                   (256, 8192, 2048), (256, 2048, 8192), (200, 2048, 2048), (129, 1000, 2048),
                                   (256, 6144, 2048), (250, 640, 256), (256, 33000, 512),
                                   # more than 256 rows on the 128-row stage machinery (sequence mode:
                                   # at least 256 tiles), ragged rows and columns
                                   (4000, 1280, 640), (3001, 1500, 512)])
@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("act,with_res", [(-1, True), (0, False)])
def test_gemm_half_weights(gpu, m, n, k, bf16, act, with_res):
    """float16 / bfloat16 arm of ops::Gemm: weights in the 16-bit type, activation rounded to bfloat16
    (bf16) or split into two float16 halves (fp16, gemm_h.hip split_f16), products summed in fp32
    (MFMA). Oracle: the same rounded operands multiplied in float64; the only difference left
    is the fp32 accumulation order (<= ~sqrt(k) * 2^-24 relative to the sum of |products|)."""
    import _gpuops as G
    rng = np.random.default_rng(m + n + k + int(bf16))
    a = rng.standard_normal((m, k)).astype(np.float32)
    w = (rng.standard_normal((n, k)) * 0.05).astype(np.float32)
    bias = (rng.standard_normal(n) * 0.1).astype(np.float32)
    res = rng.standard_normal((m, n)).astype(np.float32) if with_res else None
    if bf16:
        wb = _to_bf16_bits(w)
        wr, ar = _from_bf16_bits(wb), _from_bf16_bits(_to_bf16_bits(a))
    else:
        wh = w.astype(np.float16)
        # (float16: the activation enters as hi + lo float16 halves - 22 bits - so the oracle keeps it
        # in fp32; only the WEIGHT is a float16 value)
        wb, wr, ar = wh.view(np.uint16), wh.astype(np.float32), a
    y = G.gemm_f16w(a, wb, bf16, bias, res, act)
    ref = ar.astype(np.float64) @ wr.astype(np.float64).T + bias
    if act == 0:
        ref = np.maximum(ref, 0)
    if res is not None:
        ref = ref + res
    bound = (np.abs(ar).astype(np.float64) @ np.abs(wr).astype(np.float64).T + 4) * 3e-6
    assert np.all(np.abs(y - ref) <= bound), float(np.max(np.abs(y - ref) / bound))


def test_dequantize_rows(gpu):
    import _gpuops as G
    rng = np.random.default_rng(3)
    x = rng.integers(-127, 128, (9, 512), dtype=np.int8)
    s = rng.uniform(10, 500, 9).astype(np.float32)
    assert np.array_equal(G.dequantize_rows(x, s), O.dequantize_rows(x, s))


NORM_SHAPES = [(32, 512), (100, 512), (16, 4096), (3, 32), (2, 1280), (5, 96)]


@pytest.mark.parametrize("rows,depth", NORM_SHAPES)
def test_layer_norm(gpu, rows, depth):
    import _gpuops as G
    rng = np.random.default_rng(rows + depth)
    x = (rng.standard_normal((rows, depth)) * 2 + 0.3).astype(np.float32)
    g = rng.standard_normal(depth).astype(np.float32)
    b = rng.standard_normal(depth).astype(np.float32)
    y = G.layer_norm(x, g, b, 1e-5)
    # the kernel reproduces the reference's compiled arithmetic (ordered sums, FMA placement):
    # bit for bit, not the 1e-5 of the reference's own CPU-vs-GPU tests
    assert np.array_equal(y, O.layer_norm(x, g, b, 1e-5))


@pytest.mark.parametrize("rows,depth", NORM_SHAPES)
@pytest.mark.parametrize("use_residual", [False, True])
def test_rms_norm(gpu, rows, depth, use_residual):
    import _gpuops as G
    rng = np.random.default_rng(rows + depth + 1)
    x = (rng.standard_normal((rows, depth)) * 2).astype(np.float32)
    g = rng.standard_normal(depth).astype(np.float32)
    y = G.rms_norm(x, g, 1e-6, use_residual)
    assert np.array_equal(y, O.rms_norm(x, g, 1e-6, use_residual))


@pytest.mark.parametrize("rows,depth", [(32, 512), (16, 4096), (3, 32)])
@pytest.mark.parametrize("rms", [False, True])
@pytest.mark.parametrize("rbc", QUANT_MODES)
def test_norm_quantize_fused(gpu, rows, depth, rms, rbc):
    """Fused norm+quantize: the float output must match the standalone norm kernel bit for bit,
    and the int8 output must be the oracle quantisation OF THAT float output (exact)."""
    import _gpuops as G
    rng = np.random.default_rng(rows * 3 + depth)
    x = (rng.standard_normal((rows, depth)) * 2).astype(np.float32)
    g = rng.standard_normal(depth).astype(np.float32)
    b = None if rms else rng.standard_normal(depth).astype(np.float32)
    eps = 1e-6 if rms else 1e-5
    y, q, s = G.norm_quantize_s8(x, g, b, eps, False, rbc, True)
    y_alone = G.rms_norm(x, g, eps) if rms else G.layer_norm(x, g, b, eps)
    assert np.array_equal(y, y_alone)
    qo, so = O.quantize_s8(y, rbc)
    assert np.array_equal(s, so)
    assert np.array_equal(q, qo)
    _, q2, s2 = G.norm_quantize_s8(x, g, b, eps, False, rbc, False)
    assert np.array_equal(q2, q) and np.array_equal(s2, s)


@pytest.mark.parametrize("m,n,k", [(32, 1536, 512), (32, 512, 512), (32, 2048, 512), (16, 512, 256),
                                   (17, 100, 256), (5, 48, 64), (1, 512, 512), (9, 640, 128),
                                   (128, 512, 512), (100, 1536, 512)])
@pytest.mark.parametrize("rms", [False, True])
@pytest.mark.parametrize("act,with_res", [(-1, True), (0, False)])
def test_norm_gemm_fused(gpu, m, n, k, rms, act, with_res):
    """Pre-norm Dense as one launch (LayerNorm / RMSNorm + Quantize as the GEMM prologue, eight
    rows per wave) against the two-launch path (ct2x_norm_quantize_s8 + ct2x_gemm_s8_dequant): the
    prologue runs the reference's compiled summation order (row_norm.h), so both must agree BIT
    FOR BIT."""
    import _gpuops as G
    rng = np.random.default_rng(m + n + k + int(rms))
    x = (rng.standard_normal((m, k)) * 1.5).astype(np.float32)
    gamma = (1 + 0.1 * rng.standard_normal(k)).astype(np.float32)
    beta = None if rms else (0.1 * rng.standard_normal(k)).astype(np.float32)
    eps = 1e-6 if rms else 1e-5
    wq, ws = O.quantize_s8((rng.standard_normal((n, k)) * 0.02).astype(np.float32), True)
    bias = (rng.standard_normal(n) * 0.01).astype(np.float32)
    res = rng.standard_normal((m, n)).astype(np.float32) if with_res else None
    y = G.norm_gemm_s8_dequant(x, gamma, beta, eps, 3, wq, ws, bias, res, act)
    _, q, s = G.norm_quantize_s8(x, gamma, beta, eps, False, 3, False)
    y2 = G.gemm_s8_dequant(q, s, wq, ws, bias, res, act)
    assert np.array_equal(y, y2)


@pytest.mark.parametrize("m,n,k", [(32, 512, 2048), (128, 512, 2048), (16, 512, 512), (7, 100, 64),
                                   (33, 48, 192), (1, 4096, 1024), (100, 640, 2048)])
@pytest.mark.parametrize("flags", [3, 1, 0])
@pytest.mark.parametrize("with_res", [True, False])
def test_dense_quantize_on_load(gpu, m, n, k, flags, with_res):
    """ops::Quantize folded into the consuming GEMM's operand fetch (gemm_s8.hip QOL) against the
    two launches it replaces and against the oracle's Dense chain: bit-identical. flags: 3 = the
    x86 default (u8 shift + round), 1 = round only (the reference's GPU form), 0 = truncate."""
    import _gpuops as G
    rng = np.random.default_rng(m * 3 + n + k + flags)
    x = np.maximum(rng.standard_normal((m, k)), 0).astype(np.float32) * 2.5  # a ReLU output
    if m > 2:
        x[2, :] = 0.0  # zero row -> scale 1
    wq, ws = O.quantize_s8((rng.standard_normal((n, k)) * 0.02).astype(np.float32), True)
    bias = (rng.standard_normal(n) * 0.01).astype(np.float32)
    res = rng.standard_normal((m, n)).astype(np.float32) if with_res else None
    amax = np.abs(x).max(axis=1).astype(np.float32)
    y = G.gemm_s8_dequant_quantizing(x, amax, flags, wq, ws, bias, res, -1)
    q, s = G.quantize_s8(x, flags)
    y2 = G.gemm_s8_dequant(q, s, wq, ws, bias, res, -1)
    assert np.array_equal(y, y2)
    assert np.array_equal(y, O.dense_s8(x, wq, ws, bias, res, -1, flags))


def test_norm_gemm_row_amax_feeds_quantize_on_load(gpu):
    """The feed-forward block of a decode step as the engine runs it: pre-norm Dense + ReLU that also
    leaves the row absmax (ct2x_norm_gemm_s8_dequant_amax), then the Dense that quantises on load -
    against norm + Quantize + Dense + Quantize + Dense, bit for bit."""
    import _gpuops as G
    rng = np.random.default_rng(11)
    m, d, inner = 32, 512, 2048
    x = (rng.standard_normal((m, d)) * 1.5).astype(np.float32)
    gamma = (1 + 0.1 * rng.standard_normal(d)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(d)).astype(np.float32)
    w0q, w0s = O.quantize_s8((rng.standard_normal((inner, d)) * 0.02).astype(np.float32), True)
    w1q, w1s = O.quantize_s8((rng.standard_normal((d, inner)) * 0.02).astype(np.float32), True)
    b0 = (rng.standard_normal(inner) * 0.01).astype(np.float32)
    b1 = (rng.standard_normal(d) * 0.01).astype(np.float32)
    h, amax = G.norm_gemm_s8_dequant_amax(x, gamma, beta, 1e-5, 3, w0q, w0s, b0, None, 0)
    assert np.array_equal(amax, np.abs(h).max(axis=1))
    y = G.gemm_s8_dequant_quantizing(h, amax, 3, w1q, w1s, b1, x, -1)
    _, q, s = G.norm_quantize_s8(x, gamma, beta, 1e-5, False, 3, False)
    h2 = G.gemm_s8_dequant(q, s, w0q, w0s, b0, None, 0)
    q2, s2 = G.quantize_s8(h2, 3)
    y2 = G.gemm_s8_dequant(q2, s2, w1q, w1s, b1, x, -1)
    assert np.array_equal(h, h2) and np.array_equal(y, y2)


def test_dense_quantize_on_load_envelope(gpu):
    import _gpuops as G
    rng = np.random.default_rng(0)
    for m, n, k in [(129, 512, 512), (32, 8192, 512), (16, 512, 96), (128, 512, 4096)]:
        x = rng.standard_normal((m, k)).astype(np.float32)
        wq, ws = O.quantize_s8(rng.standard_normal((n, k)).astype(np.float32), True)
        with pytest.raises(RuntimeError):
            G.gemm_s8_dequant_quantizing(x, np.abs(x).max(axis=1), 3, wq, ws, None, None, -1)


def test_norm_gemm_fused_envelope(gpu):
    """Outside the envelope (K not 64 / 128 / 256 / 512, more than 128 rows, the vocabulary projection) the entry point
    refuses instead of computing something else."""
    import _gpuops as G
    rng = np.random.default_rng(0)
    for m, n, k in [(16, 512, 1024), (129, 512, 512), (32, 32000, 512), (8, 64, 96), (16, 512, 448)]:
        x = rng.standard_normal((m, k)).astype(np.float32)
        wq, ws = O.quantize_s8(rng.standard_normal((n, k)).astype(np.float32), True)
        with pytest.raises(RuntimeError):
            G.norm_gemm_s8_dequant(x, np.ones(k, np.float32), None, 1e-6, 3, wq, ws, None, None, -1)


@pytest.mark.parametrize("rows,depth", [(256, 36), (32, 32000), (8, 128256), (5, 7), (64, 64)])
@pytest.mark.parametrize("log", [False, True])
@pytest.mark.parametrize("masked", [False, True])
def test_softmax(gpu, rows, depth, log, masked):
    import _gpuops as G
    rng = np.random.default_rng(rows + depth)
    x = (rng.standard_normal((rows, depth)) * 4).astype(np.float32)
    lengths = None
    if masked:
        lengths = rng.integers(0, depth + 1, rows).astype(np.int32)
        lengths[0] = 0       # fully masked row -> all zeros (kernels.cc:417-428)
        lengths[-1] = depth
    y = G.softmax(x, lengths, log)
    yo = O.softmax(x, lengths, log)
    if masked:
        for r in range(rows):
            assert np.all(y[r, lengths[r]:] == 0.0)
    if not log:
        # exp256_ps polynomial + the AVX2 sum order of the reference: bit for bit
        assert np.array_equal(y, yo)
    else:
        # glibc's logf and - for the CPU's scalar tail (depth % 8 elements, a whole row of <= 8) -
        # glibc's expf are restated on the device (logf_glibc, expf_glibc: the latter compared with
        # libm over all 2^32 floats on the host build of activation_exact.h)
        np.testing.assert_allclose(y, yo, rtol=3e-7, atol=2e-6)
        if not masked:
            assert np.array_equal(y, yo)


@pytest.mark.parametrize("interleave", [True, False])
@pytest.mark.parametrize("depth,ndims", [(128, 128), (64, 64), (64, 32), (8, 4)])
def test_rotary(gpu, interleave, depth, ndims):
    import _gpuops as G
    rng = np.random.default_rng(depth + ndims)
    batch, time = 6, 5
    x = rng.standard_normal((batch, time, depth)).astype(np.float32)
    ang = rng.uniform(0, 6.28, (time, ndims)).astype(np.float32)
    s, c = np.sin(ang).astype(np.float32), np.cos(ang).astype(np.float32)
    y = G.rotary(x, s, c, ndims, interleave)
    assert np.array_equal(y, O.rotary(x, s, c, ndims, interleave))


@pytest.mark.parametrize("rows,depth,k", [(32, 32000, 1), (32, 128000, 8), (8, 128256, 1),
                                          (3, 43, 4), (4, 10, 10), (2, 5, 1)])
def test_topk(gpu, rows, depth, k):
    import _gpuops as G
    rng = np.random.default_rng(rows + depth + k)
    x = rng.standard_normal((rows, depth)).astype(np.float32)
    # duplicates force the tie-break: lowest index first (std::max_element, topk_cpu.cc:27)
    x[:, depth // 2] = x[:, 0] = x.max(axis=1) + 1.0
    v, i = G.topk(x, k)
    vo, io = O.topk(x, k)
    assert np.array_equal(i, io)
    assert np.array_equal(v, vo)
    assert np.all(i[:, 0] == 0)


def test_process_logits(gpu):
    """ct2x_process_logits (RepetitionPenalty / NoRepeatNgram / SuppressSequences + the
    indexed_fill of DisableTokens, src/decoding_utils.cc) == the oracle bit for bit."""
    import _gpuops as G
    from test_oracle import _processor_cases
    for lg, seq, kw in _processor_cases():
        got = G.process_logits(lg, seq, **kw)
        want = O.process_logits(lg, seq, **kw)
        assert np.array_equal(got, want), (seq.shape, kw)
    # vocabulary-sized rows, long history (several passes of the 256-thread loops)
    rng = np.random.default_rng(2)
    lg = rng.standard_normal((32, 32000)).astype(np.float32)
    seq = rng.integers(0, 40, (32, 700)).astype(np.int32)
    kw = {"penalty": 1.3, "ngram": 2, "suppress": [[5], [seq[0, -1], 17], [1, 2, 3]]}
    assert np.array_equal(G.process_logits(lg, seq, **kw), O.process_logits(lg, seq, **kw))


def test_kat_penalize_previous_tokens_and_logsumexp(gpu):
    """tests/primitives_test.cc:25-52 on the device entry points (the CUDA instantiation of the
    reference's PrimitiveTest)."""
    import _gpuops as G
    from test_oracle import PENALIZE_KAT, penalize_kat_expected
    got = G.process_logits(PENALIZE_KAT["scores"], PENALIZE_KAT["previous_ids"],
                           penalty=PENALIZE_KAT["penalty"])
    assert np.array_equal(got, penalize_kat_expected())
    x = np.array([[0.6, 0.2, -1.2, 0.1, 0.3, 0.5, -1.3, 0.2]], np.float32)
    lse = x[0, 0] - G.softmax(x, None, True)[0, 0]
    assert abs(float(lse) - 2.1908040046691895) < 1e-6


@pytest.mark.parametrize("beam", [1, 2, 4, 32])
def test_beam_select(gpu, beam):
    """ct2x_beam_select (the device side of a beam-search step after TopK: unflatten_ids, hard
    prefix, candidate selection; src/decoding.cc:21-66, 95-116, 618-655) == the oracle
    restatement, bit for bit, over steps inside / at the end of / after the prefix and candidate
    sets with several EOS among primary and secondary candidates."""
    import _gpuops as G
    rng = np.random.default_rng(beam)
    batch, vocab, eos = 9, 50, 2
    nc = 2 * beam
    prefix = [rng.integers(3, vocab, n).tolist() for n in (0, 1, 3, 2, 0, 3, 1, 2, 3)]
    for step in range(5):
        for use_prefix in (False, True):
            words = rng.integers(0, vocab, (batch, nc))
            words[rng.random((batch, nc)) < 0.35] = eos        # many ended candidates
            words[0, :] = eos                                   # nothing left to promote
            words[1, :beam] = eos                               # every primary ended
            origin = rng.integers(0, beam, (batch, nc))
            ids = (origin * vocab + words).astype(np.int32)
            scores = -np.sort(rng.random((batch, nc)).astype(np.float32) * 9, axis=1)
            got = G.beam_select(scores, ids, beam, vocab, eos, prefix if use_prefix else None, step,
                                first_step=3)
            want = O.beam_select(scores, ids, beam, vocab, eos, prefix if use_prefix else None, step)
            for a, b, name in zip(got, want, ("word", "score", "origin", "next_ids", "next_scores",
                                              "parents")):
                assert np.array_equal(a, b), (name, step, use_prefix)


def test_gather_rows(gpu):
    import _gpuops as G
    rng = np.random.default_rng(5)
    data = rng.standard_normal((50, 8, 16)).astype(np.float32)
    idx = rng.integers(0, 50, 77).astype(np.int32)
    assert np.array_equal(G.gather_rows(data, idx), O.gather_rows(data, idx))
    odd = rng.integers(-100, 100, (20, 7), dtype=np.int8)
    assert np.array_equal(G.gather_rows(odd, idx % 20), O.gather_rows(odd, idx % 20))


@pytest.mark.parametrize("m,n,k,trans_b", [(32, 96, 32, True), (70, 130, 50, True),
                                           (64, 64, 64, False), (5, 7, 3, False)])
def test_gemm_f32(gpu, m, n, k, trans_b):
    import _gpuops as G
    rng = np.random.default_rng(m + n)
    a = rng.standard_normal((m, k)).astype(np.float32)
    b = rng.standard_normal((n, k) if trans_b else (k, n)).astype(np.float32)
    bias = rng.standard_normal(n).astype(np.float32)
    res = rng.standard_normal((m, n)).astype(np.float32)
    y = G.gemm_f32(a, b, trans_b, bias, res, 0)
    yo = O.gemm_f32(a, b, trans_b, 1.0, bias, res, 0)
    # same k-ordered fmaf chain on both sides
    np.testing.assert_allclose(y, yo, rtol=1e-6, atol=1e-6)


ATTN_CASES = [
    # rows, heads, heads_kv, dhead, t_max, kv_group
    (32, 8, 8, 64, 64, 1),     # Transformer-base self-attention
    (128, 8, 8, 64, 40, 4),    # cross-attention over encoder memory, beam 4
    (16, 32, 8, 128, 256, 1),  # Llama-3 8B GQA without head replication
    (3, 8, 8, 4, 12, 1),       # aren-transliteration fixture (d_head 4)
    (4, 2, 1, 96, 33, 2),      # MQA, non power-of-two chunk count
    (2, 1, 1, 256, 9, 1),      # two float4 chunks per lane
    (16, 8, 2, 64, 100, 1),    # GQA group 4, keys split over 4 waves (attention_decode2 split mode)
    (8, 4, 2, 128, 70, 2),     # GQA group 2, d_head 128, shared cache rows
    (5, 8, 8, 64, 200, 1),     # 4 chunks of 64 keys
]


@pytest.mark.parametrize("rows,heads,heads_kv,dhead,t_max,kv_group", ATTN_CASES)
def test_attention_decode(gpu, rows, heads, heads_kv, dhead, t_max, kv_group):
    import _gpuops as G
    rng = np.random.default_rng(rows + heads + dhead + t_max)
    q = rng.standard_normal((rows, heads, dhead)).astype(np.float32)
    k = rng.standard_normal((rows // kv_group, heads_kv, t_max, dhead)).astype(np.float32)
    v = rng.standard_normal((rows // kv_group, heads_kv, t_max, dhead)).astype(np.float32)
    lengths = rng.integers(1, t_max + 1, rows // kv_group).astype(np.int32)
    lengths[0] = t_max
    if rows // kv_group > 1:
        lengths[1] = 0  # fully masked row -> zero context
    scale = 1.0 / np.sqrt(dhead)
    y = G.attention_decode(q, k, v, lengths, 0, kv_group, scale)
    yo = O.attention_decode(q, k, v, lengths, 0, kv_group, scale)
    y2 = G.attention_decode(q, k, v, None, t_max // 2 + 1, kv_group, scale)
    yo2 = O.attention_decode(q, k, v, None, t_max // 2 + 1, kv_group, scale)
    if dhead in (4, 8, 16, 32, 64):
        # attention_exact.hip: MKL's summation trees + the reference's softmax, bit for bit
        # (the oracle is pinned bit for bit against the reference library)
        assert np.array_equal(y, yo)
        assert np.array_equal(y2, yo2)
    else:
        np.testing.assert_allclose(y, yo, rtol=2e-5, atol=2e-6)
        if dhead == 128 and kv_group == 1:
            # attention_exact128.hip (self-attention form: one cache length, no length vector)
            assert np.array_equal(y2, yo2)
        else:
            np.testing.assert_allclose(y2, yo2, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("rows,kv_group,dhead", [(32, 1, 64), (24, 3, 64), (128, 4, 64), (8, 1, 4),
                                                  (8, 1, 8), (6, 1, 16), (8, 1, 32), (12, 4, 8)])
def test_attention_decode_exact_every_length(gpu, rows, kv_group, dhead):
    """Every key count 1..70: the three tail forms of MKL's one-row product, the block-of-eight
    P.V (plain chains for d_head 4 / 8), the sequence forms of beam search on the memory - bit
    for bit, at every head size of the fixtures."""
    import _gpuops as G
    rng = np.random.default_rng(rows + kv_group + dhead)
    scale = float(np.float32(1.0 / np.sqrt(dhead)))
    for t_max in list(range(1, 71)) + [130]:
        b = rows // kv_group
        q = rng.standard_normal((rows, 8, dhead)).astype(np.float32)
        k = rng.standard_normal((b, 8, t_max, dhead)).astype(np.float32)
        v = rng.standard_normal((b, 8, t_max, dhead)).astype(np.float32)
        if kv_group >= 3 and t_max < 5:
            continue  # outside the recovered MKL envelope (m >= 3 needs n >= 5)
        y = G.attention_decode(q, k, v, None, t_max, kv_group, scale)
        assert np.array_equal(y, O.attention_decode(q, k, v, None, t_max, kv_group, scale)), t_max
        lengths = rng.integers(1, t_max + 1, b).astype(np.int32)
        y = G.attention_decode(q, k, v, lengths, 0, kv_group, scale)
        assert np.array_equal(y, O.attention_decode(q, k, v, lengths, 0, kv_group, scale)), t_max


@pytest.mark.parametrize("group", [1, 2, 4, 8])
def test_attention_decode_exact128_every_length(gpu, group):
    """d_head 128 (Llama-3 geometry), every cache length 1..140 and a few long ones: MKL's main /
    tail-A / tail-B dot forms on 16-lane DPP rows, the block-of-eight P.V, every GQA group size -
    bit for bit against the oracle (pinned against the reference library for k = 128)."""
    import _gpuops as G
    rng = np.random.default_rng(100 + group)
    scale = float(np.float32(1.0 / np.sqrt(128)))
    heads_kv = 2
    heads = heads_kv * group
    for t_max in list(range(1, 141)) + [255, 256, 257, 600, 1031]:
        rows = 3 if t_max < 200 else 2
        q = rng.standard_normal((rows, heads, 128)).astype(np.float32)
        k = rng.standard_normal((rows, heads_kv, t_max, 128)).astype(np.float32)
        v = rng.standard_normal((rows, heads_kv, t_max, 128)).astype(np.float32)
        y = G.attention_decode(q, k, v, None, t_max, 1, scale)
        assert np.array_equal(y, O.attention_decode(q, k, v, None, t_max, 1, scale)), t_max


@pytest.mark.parametrize("b,tq,tk,h,hkv,dh,causal", [
    (4, 32, 32, 8, 8, 64, False),     # Transformer-base encoder tile
    (3, 128, 128, 8, 2, 128, True),   # Llama-style prefill: GQA, causal
    (2, 100, 100, 4, 4, 64, True),    # ragged query tile (100 = 64 + 36)
    (2, 70, 45, 4, 2, 128, False),    # Tq != Tk
    (2, 20, 20, 2, 2, 32, True),      # head size of the per-row kernel
    (1, 400, 400, 2, 1, 128, True),   # MKL packs K (n >= 192) and cuts the P.V chain (400 > 384)
    (1, 200, 200, 2, 2, 64, False),   # packed K at d_head 64 (power-of-two scale: same bits)
    (2, 150, 150, 2, 1, 32, True),    # d_head 32: scale not a power of two, m == n < 192
    (1, 100, 160, 2, 2, 32, False),   # n > m, n >= k: packed K below 192 keys
    (1, 40, 1200, 1, 1, 16, False),   # long memory: three blocks of the k partition
])
def test_attention_full(gpu, b, tq, tk, h, hkv, dh, causal):
    """Sequence-mode attention against a float64 statement of dot_product_attention
    (src/layers/attention.cc:178-287): scores = scale * q.k, length mask (and causal mask),
    softmax, P.V; masked rows give zeros."""
    import _gpuops as G
    rng = np.random.default_rng(b + tq + tk + h + dh)
    q = rng.standard_normal((b, tq, h, dh)).astype(np.float32)
    k = rng.standard_normal((b, tk, hkv, dh)).astype(np.float32)
    v = rng.standard_normal((b, tk, hkv, dh)).astype(np.float32)
    lengths = rng.integers(1, tk + 1, b).astype(np.int32)
    lengths[0] = tk
    if b > 2:
        lengths[1] = 0
    scale = 1.0 / np.sqrt(dh)
    y = G.attention_full(q, k, v, lengths, causal, 0, scale)
    group = h // hkv
    ref = np.zeros((b, tq, h, dh))
    for bi in range(b):
        for hi in range(h):
            kk, vv = k[bi, :, hi // group].astype(np.float64), v[bi, :, hi // group].astype(np.float64)
            s = scale * (q[bi, :, hi].astype(np.float64) @ kk.T)            # [tq, tk]
            mask = np.arange(tk)[None, :] < lengths[bi]
            if causal:
                mask = mask & (np.arange(tk)[None, :] <= np.arange(tq)[:, None])
            s = np.where(mask, s, -np.inf)
            with np.errstate(invalid="ignore"):
                e = np.exp(s - np.max(s, axis=1, keepdims=True))
            e = np.where(mask, e, 0.0)
            den = e.sum(axis=1, keepdims=True)
            pmat = np.divide(e, den, out=np.zeros_like(e), where=den > 0)
            ref[bi, :, hi] = pmat @ vv
    np.testing.assert_allclose(y, ref, rtol=2e-5, atol=3e-6)
    if dh in (4, 8, 16, 32, 64, 128):
        # bit for bit against the reference's MatMul -> SoftMax -> MatMul chain (oracle restatement;
        # attention_exact.hip in its envelope, attention_seq_exact.hip - fp32 MFMA chains - beyond)
        for bi in range(b):
            for hi in range(h):
                sc = O.matmul(q[bi, :, hi][None], k[bi, :, hi // group][None], True, scale)[0]
                size = np.minimum(lengths[bi], np.arange(tq) + 1) if causal else np.full(tq, lengths[bi])
                pm = O.softmax(sc, size.astype(np.int32), False)
                want = O.matmul(pm[None], v[bi, :, hi // group][None], False, 1.0)[0]
                assert np.array_equal(y[bi, :, hi], want), (bi, hi)


def _attention_f64(q, k, v, lengths, causal, q_offset, scale):
    b, tq, h, dh = q.shape
    tk, hkv = k.shape[1], k.shape[2]
    group = h // hkv
    ref = np.zeros((b, tq, h, dh))
    for bi in range(b):
        for hi in range(h):
            kk, vv = k[bi, :, hi // group].astype(np.float64), v[bi, :, hi // group].astype(np.float64)
            s = scale * (q[bi, :, hi].astype(np.float64) @ kk.T)
            mask = np.arange(tk)[None, :] < lengths[bi]
            if causal:
                mask = mask & (np.arange(tk)[None, :] <= np.arange(tq)[:, None] + q_offset)
            s = np.where(mask, s, -np.inf)
            with np.errstate(invalid="ignore"):
                e = np.exp(s - np.max(s, axis=1, keepdims=True))
            e = np.where(mask, e, 0.0)
            den = e.sum(axis=1, keepdims=True)
            ref[bi, :, hi] = np.divide(e, den, out=np.zeros_like(e), where=den > 0) @ vv
    return ref


@pytest.mark.parametrize("kind", [1, 2])
@pytest.mark.parametrize("b,tq,tk,h,hkv,dh,causal,q_offset", [
    (4, 32, 32, 8, 8, 64, False, 0),      # Transformer-base encoder tile; one empty batch entry
    (3, 128, 128, 8, 2, 128, True, 0),    # Llama-style prefill: GQA, causal
    (2, 100, 100, 4, 4, 64, True, 0),     # ragged query tile (100 = 64 + 36), ragged key tile (100 = 3 * 32 + 4)
    (2, 70, 45, 4, 2, 128, False, 0),     # Tq != Tk
    (9, 64, 333, 3, 3, 64, False, 0),     # 9 (batch, head) units over the 8 XCD slots: a partly empty round
    (1, 200, 230, 6, 2, 64, True, 30),    # causal with an offset (chunked prefill), odd key count
    (2, 17, 1500, 2, 2, 64, False, 0),    # Whisper memory length
])
def test_attention_full_half(gpu, b, tq, tk, h, hkv, dh, causal, q_offset, kind):
    """Sequence-mode attention as float16 / bfloat16 compute runs it (ct2x_attention_full_half: the
    16-bit matrix cores with hi + lo operand splitting) against the float64 statement of
    dot_product_attention (src/layers/attention.cc:178-287). The split keeps 22 (float16) / 16
    (bfloat16) significant bits per operand, so the bound is the fp32 kernel's for float16 and
    2^-16-sized for bfloat16 - two to three orders tighter than one rounding to the 16-bit type,
    which is what the reference's own 16-bit GPU path does to these tensors."""
    import _gpuops as G
    rng = np.random.default_rng(b + tq + tk + h + dh + kind)
    q = rng.standard_normal((b, tq, h, dh)).astype(np.float32)
    k = rng.standard_normal((b, tk, hkv, dh)).astype(np.float32)
    v = rng.standard_normal((b, tk, hkv, dh)).astype(np.float32)
    lengths = rng.integers(1, tk + 1, b).astype(np.int32)
    lengths[0] = tk
    if b > 2:
        lengths[1] = 0
    scale = 1.0 / np.sqrt(dh)
    y = G.attention_full(q, k, v, lengths, causal, q_offset, scale, compute_kind=kind)
    ref = _attention_f64(q, k, v, lengths, causal, q_offset, scale)
    if kind == 1:
        np.testing.assert_allclose(y, ref, rtol=2e-5, atol=3e-6)
    else:
        np.testing.assert_allclose(y, ref, rtol=1e-3, atol=2e-4)
    if b > 2:
        assert not y[1].any()   # no visible key -> zeros


@pytest.mark.parametrize("batch,m,n,k,trans_b,alpha", [
    (256, 1, 33, 64, True, 0.125),     # decode QK^T: one query per (row, head), tail of 1 key
    (256, 1, 64, 33, False, 1.0),      # decode P.V: blocks of eight + one key
    (64, 1, 7, 32, True, 0.17677669),  # d_head 32, non power-of-two scale
    (64, 1, 4, 9, False, 1.0),         # d_head 4: narrow output, plain chain
    (32, 4, 40, 64, True, 0.125),      # beam 4 on the encoder memory
    (32, 4, 64, 40, False, 1.0),
    (8, 32, 32, 64, True, 0.125),      # encoder self-attention tile
    (8, 32, 64, 32, False, 1.0),
])
def test_matmul_f32_exact(gpu, batch, m, n, k, trans_b, alpha):
    """ops::MatMul through the C ABI = the reference CPU path's MKL sgemm, bit for bit
    (orc_matmul is pinned against the reference library for exactly these shape classes)."""
    import _gpuops as G
    rng = np.random.default_rng(batch + m + n + k)
    a = rng.standard_normal((batch, m, k)).astype(np.float32)
    b = rng.standard_normal((batch, n, k) if trans_b else (batch, k, n)).astype(np.float32)
    y = G.matmul_f32(a, b, trans_b, alpha)
    assert np.array_equal(y, O.matmul(a, b, trans_b, alpha))


def test_matmul_f32_outside_the_envelope_is_close(gpu):
    """Shapes whose MKL order was not recovered (k > 64 with one query row) run the plain fp32
    GEMM: within fp32 accumulation error of a float64 product."""
    import _gpuops as G
    rng = np.random.default_rng(11)
    a = rng.standard_normal((6, 1, 128)).astype(np.float32)
    b = rng.standard_normal((6, 50, 128)).astype(np.float32)
    y = G.matmul_f32(a, b, True, 0.5)
    ref = 0.5 * np.einsum("bmk,bnk->bmn", a.astype(np.float64), b.astype(np.float64))
    np.testing.assert_allclose(y, ref, rtol=2e-5, atol=2e-5)


def test_runtime_abi_round_trip(gpu):
    """The runtime entry points of the boundary (allocator, copies, events, streams):
    include/ct2_mi355.h 'runtime'."""
    import ctypes as C
    from ctranslate2_amd import _lib
    L = _lib.lib()
    L.ct2x_thread_stream.restype = C.c_void_p
    L.ct2x_allocated_bytes.restype = C.c_int64
    stream = C.c_void_p(L.ct2x_thread_stream())
    _lib.check(L.ct2x_clear_cache())  # earlier tests leave cached blocks: start from the driver
    before = L.ct2x_allocated_bytes()
    src = np.arange(1 << 16, dtype=np.float32)
    d1, d2 = C.c_void_p(), C.c_void_p()
    _lib.check(L.ct2x_malloc(C.byref(d1), C.c_size_t(src.nbytes)))
    _lib.check(L.ct2x_malloc(C.byref(d2), C.c_size_t(src.nbytes)))
    assert d1.value and d2.value and d1.value != d2.value
    assert L.ct2x_allocated_bytes() >= before + 2 * src.nbytes
    e0, e1 = C.c_void_p(), C.c_void_p()
    _lib.check(L.ct2x_event_create(C.byref(e0)))
    _lib.check(L.ct2x_event_create(C.byref(e1)))
    _lib.check(L.ct2x_event_record(e0, stream))
    _lib.check(L.ct2x_copy_to_device(src.ctypes.data_as(C.c_void_p), d1, C.c_size_t(src.nbytes), stream))
    _lib.check(L.ct2x_copy_device(d1, d2, C.c_size_t(src.nbytes), stream))
    out = np.zeros_like(src)
    _lib.check(L.ct2x_copy_to_host(d2, out.ctypes.data_as(C.c_void_p), C.c_size_t(src.nbytes), stream))
    assert np.array_equal(out, src)  # the D2H copy has landed when the call returns
    _lib.check(L.ct2x_event_record(e1, stream))
    ms = C.c_float(-1)
    _lib.check(L.ct2x_event_elapsed_ms(e0, e1, C.byref(ms)))
    assert ms.value >= 0
    # stream-ordered reuse: a block freed with work queued behind it comes back to the same
    # stream at once (and only to it)
    _lib.check(L.ct2x_free(d1))
    d3 = C.c_void_p()
    _lib.check(L.ct2x_malloc(C.byref(d3), C.c_size_t(src.nbytes)))
    assert d3.value == d1.value
    _lib.check(L.ct2x_free(d2))
    _lib.check(L.ct2x_free(d3))
    _lib.check(L.ct2x_stream_synchronize(stream))
    _lib.check(L.ct2x_device_synchronize())
    _lib.check(L.ct2x_clear_cache())
    assert L.ct2x_allocated_bytes() <= before
    assert L.ct2x_free(d3) != 0  # double free is an error, not a crash
    _lib.check(L.ct2x_event_destroy(e0))
    _lib.check(L.ct2x_event_destroy(e1))


@pytest.mark.parametrize("topk,topp,temp", [(8, 0.7, 1.0), (40, 0.9, 0.8), (64, 1.0, 1.3), (2, 0.5, 1.0),
                                            (0, 0.8, 1.0), (0, 1.0, 0.7), (0, 0.3, 1.5),
                                            (65, 1.0, 1.0), (200, 0.9, 0.8), (1000, 0.6, 1.2), (1499, 1.0, 1.0),
                                            (5000, 0.9, 1.0)])
def test_random_sample_with_injected_uniforms(gpu, topk, topp, temp):
    """RandomSampler's chain TopK -> temperature -> TopPMask -> SoftMax -> Multinomial
    (src/sampling.cc:41-101) with the uniform draws INJECTED: the id each row returns must be the
    inverse CDF, at its draw, of the distribution the oracle's TopPMask (orc_topp_mask, pinned against
    the reference's TopPMask KAT) leaves - candidates in TopK order for sampling_topk > 0, in
    index order for the whole vocabulary and for candidate sets longer than 64 (found by a value
    threshold instead of a list, so any sampling_topk up to the vocabulary works) - and the score the log-probability under the original
    distribution. Rows whose draw or whose top-p mass sits within 1e-5 of a boundary are not
    decidable across float formats and are skipped (a handful at most)."""
    import _gpuops as G
    rng = np.random.default_rng(topk * 31 + int(topp * 100) + int(temp * 10))
    rows, vocab = 96, 1500
    logits = (rng.standard_normal((rows, vocab)) * 3).astype(np.float32)
    u = rng.random(rows).astype(np.float32)
    u[:4] = [0.0, 0.999999, 0.5, 1e-7]
    ids, scores = G.random_sample(logits, topk, topp, temp, u)
    checked = 0
    for r in range(rows):
        x = logits[r]
        if 0 < topk <= 64:
            order = np.argsort(-x, kind="stable")[:topk]
        elif 64 < topk < vocab:
            order = np.sort(np.argsort(-x, kind="stable")[:topk])
        else:
            order = np.arange(vocab)
        z = (x[order].astype(np.float64) - x.max()) / temp
        y = O.topp_mask((x[order] / np.float32(temp))[None, :].astype(np.float32), topp, -np.inf)[0]
        kept = np.isfinite(y)
        p_all = np.exp(z) / np.exp(z).sum()
        # decidability of the mask: mass ranked before each candidate vs topp
        srt = np.argsort(-z, kind="stable")
        before = np.concatenate([[0.0], np.cumsum(p_all[srt])[:-1]])
        if topp < 1.0 and np.min(np.abs(before - topp)) < 1e-5:
            continue
        w = np.where(kept, np.exp(z), 0.0)
        cdf = np.cumsum(w) / w.sum()
        target = float(u[r])
        if np.min(np.abs(cdf - target)) < 1e-5:
            continue
        pick = int(np.searchsorted(cdf, target, side="right"))
        pick = min(pick, len(cdf) - 1)
        assert kept[pick]
        assert ids[r] == order[pick], (r, ids[r], order[pick])
        lse = np.log(np.exp(x.astype(np.float64) - x.max()).sum()) + x.max()
        assert abs(scores[r] - (x[ids[r]] - lse)) < 1e-4
        checked += 1
    assert checked >= rows - 8


def test_random_sample_long_candidate_list_breaks_ties_by_index(gpu):
    """ops::TopK keeps the LOWEST indices among equal values (src/ops/topk_cpu.cc:13-55): with a
    plateau of equal logits straddling the k-th rank, only the plateau's first tokens are candidates."""
    import _gpuops as G
    rows, vocab, topk = 64, 3000, 300
    rng = np.random.default_rng(17)
    logits = np.full((rows, vocab), -5.0, np.float32)
    high = rng.permutation(vocab)[:250]
    logits[:, high] = 2.0                       # 250 clear winners, then a plateau of 2750 equal values
    u = rng.random(rows).astype(np.float32)
    ids, _ = G.random_sample(logits, topk, 1.0, 1.0, u)
    plateau = np.setdiff1d(np.arange(vocab), high)[:topk - 250]
    allowed = set(high.tolist()) | set(plateau.tolist())
    assert all(int(i) in allowed for i in ids)
    # the plateau members are reachable: force the draw into the tail of the CDF
    ids_tail, _ = G.random_sample(logits, topk, 1.0, 1.0, np.full(rows, 0.999999, np.float32))
    assert all(int(i) == max(allowed) for i in ids_tail)


def test_random_sample_hash_is_reproducible(gpu):
    """Without injected draws the sampler uses its counter-based hash of (seed, step, row): same
    seed, same ids; another seed, another sample."""
    import _gpuops as G
    rng = np.random.default_rng(5)
    logits = rng.standard_normal((64, 2000)).astype(np.float32)
    a, _ = G.random_sample(logits, 16, 1.0, 1.0, None, seed=7)
    b, _ = G.random_sample(logits, 16, 1.0, 1.0, None, seed=7)
    c, _ = G.random_sample(logits, 16, 1.0, 1.0, None, seed=8)
    assert np.array_equal(a, b) and not np.array_equal(a, c)


def test_gumbel_noise_is_exponential_and_seeded(gpu):
    """ops::GumbelMax::add_gumbel_noise (src/ops/gumbel_max_gpu.cu:12-38): y = x - log(u), u uniform
    in (0, 1]: the added noise is a unit exponential variate (mean 1, variance 1, median ln 2); the
    arg-max of log-probabilities + that noise follows... whatever the reference's TopK makes of it -
    what is pinned here is the op: the noise distribution, x == y in place, fresh draws per call."""
    import ctypes as C
    import torch
    from ctranslate2_amd import _lib
    n = 1 << 20
    x = torch.full((n,), 3.0, dtype=torch.float32, device="cuda")
    y1, y2 = torch.empty_like(x), torch.empty_like(x)
    p = lambda t: C.c_void_p(t.data_ptr())
    _lib.check(gpu.ct2x_gumbel_noise(p(x), p(y1), C.c_int64(n), C.c_uint32(7), None))
    _lib.check(gpu.ct2x_gumbel_noise(p(x), p(y2), C.c_int64(n), C.c_uint32(7), None))
    z1, z2 = (y1 - 3.0).cpu().numpy(), (y2 - 3.0).cpu().numpy()
    assert z1.min() >= 0 and np.isfinite(z1).all()
    assert abs(z1.mean() - 1.0) < 0.01 and abs(z1.var() - 1.0) < 0.02
    assert abs(np.median(z1) - np.log(2.0)) < 0.01
    assert not np.array_equal(z1, z2)  # successive calls draw fresh numbers
    _lib.check(gpu.ct2x_gumbel_noise(p(x), p(x), C.c_int64(n), C.c_uint32(7), None))  # in place
    assert abs((x - 3.0).mean().item() - 1.0) < 0.01


@pytest.mark.parametrize("batch,beam,heads,hkv,s_max,dh", [(3, 5, 4, 4, 1500, 64), (2, 1, 8, 8, 37, 64),
                                                          (4, 4, 8, 2, 260, 64), (1, 8, 2, 1, 8, 64),
                                                          (5, 3, 20, 20, 129, 64),
                                                          # d_head 128 (NLLB-3.3B geometry: 16 heads, 32 positions)
                                                          (6, 4, 16, 16, 32, 128), (2, 5, 4, 2, 700, 128),
                                                          (3, 1, 8, 8, 5, 128), (1, 8, 2, 1, 1500, 128)])
@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("split", [False, True])
def test_cross_attention_16bit_memory(gpu, batch, beam, heads, hkv, s_max, dh, bf16, split):
    """attention_cross_h.hip: the decode steps' encoder-decoder attention over float16 / bfloat16
    memory keys and values, one workgroup per (batch entry, head) for all hypothesis rows. Oracle:
    the same 16-bit values in float64 (softmax(q . K^T * scale, lengths) . V); what is left is the
    fp32 evaluation order."""
    import _gpuops as G
    rng = np.random.default_rng(batch * 1000 + beam * 100 + heads + s_max + int(bf16))
    q = rng.standard_normal((batch * beam, heads, dh)).astype(np.float32)
    kv = rng.standard_normal((batch, s_max, 2, hkv, dh)).astype(np.float32)
    lengths = rng.integers(max(1, s_max // 2), s_max + 1, batch).astype(np.int32)
    lengths[0] = s_max
    if bf16:
        bits = _to_bf16_bits(kv)
        kvr = _from_bf16_bits(bits)
    else:
        h16 = kv.astype(np.float16)
        bits, kvr = h16.view(np.uint16), h16.astype(np.float32)
    scale = 0.125
    if split and batch > 1:
        lengths[1] = max(1, min(s_max, 130))   # the later chunks of this entry see no valid key
    # split: the memory of an (entry, head) over several workgroups merged by the last arrival
    # (1500 / 700 / 260 positions cut into 8 / 5 / 2 chunks here); twice on one scratch = tickets back at zero
    y = G.cross_attention_h(q, bits, bf16, lengths, beam, hkv, scale, split=split, calls=2)
    group = heads // hkv
    ref = np.zeros_like(q, dtype=np.float64)
    for r in range(batch * beam):
        b = r // beam
        n = int(lengths[b])
        for h in range(heads):
            k = kvr[b, :n, 0, h // group].astype(np.float64)
            v = kvr[b, :n, 1, h // group].astype(np.float64)
            sc = (k @ q[r, h].astype(np.float64)) * scale
            w = np.exp(sc - sc.max())
            ref[r, h] = (w / w.sum()) @ v
    assert np.max(np.abs(y - ref)) <= 2e-5 * max(1.0, float(np.abs(ref).max()))
