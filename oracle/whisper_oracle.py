"""TEST INFRASTRUCTURE ONLY - never imported by the product.

numpy restatement of the CPU algorithms that are specific to the Whisper path, each
citing the reference lines it follows, pinned against the reference library on CPU
(tests/test_whisper_oracle.py, through oracle/_ref):

  conv1d_k3_s8      ops::Conv1D on Device::CPU with int8 weights (src/ops/conv1d_cpu.cc:139-207)
  layer_norm_axis   ops::LayerNorm(axis = -2) without gamma / beta (src/cpu/kernels.cc:499-539)
  timestamp_rules   ApplyTimestampRules + SuppressTokensBegin (src/models/whisper.cc:741-861,
                    src/decoding_utils.cc:172-194) - parity of this one is pinned only through the
                    generate goldens (the class is private to whisper.cc)

Element arithmetic that the reference compiles for AVX2 (Quantize, the GELU functor) comes from the
plain-C oracle (oracle/ct2_oracle.c through tests/_oracle.py).
"""
import numpy as np


def im2col_k3(x, stride, padding):
    """im2col_transposed (conv1d_cpu.cc:209-243), kernel 3, groups 1: x [batch, C, L] ->
    [batch, out_len, C * 3] with column c * 3 + k."""
    b, c, length = x.shape
    out_len = (length + 2 * padding - 3) // stride + 1
    xp = np.zeros((b, c, length + 2 * padding), np.float32)
    xp[:, :, padding:padding + length] = x
    cols = np.empty((b, out_len, c, 3), np.float32)
    for t in range(out_len):
        cols[:, t] = xp[:, :, t * stride:t * stride + 3]
    return cols.reshape(b, out_len, c * 3)


def conv1d_k3_s8(O, x, wq, w_scale, bias, stride, padding, act):
    """The int8 arm of Conv1D::compute_with_gemm (conv1d_cpu.cc:176-203): every im2col row is
    quantised on its own (ops::Quantize INT8 arm: per row, round before cast, no u8 shift), the
    product is weight x rows^T in exact int32, and Dequantize runs with the WEIGHT scale as the
    row scale (multiplied in as a reciprocal) and the input row's scale as the column scale
    (divided by) - cpu::dequantize_gemm_output, src/cpu/kernels.cc:653-685 - then bias and
    activation over the channel axis (apply_bias_and_activation, src/ops/gemm.cc:11-25).
    O: tests/_oracle.py (the C oracle). Returns [batch, out_c, out_len]."""
    b = x.shape[0]
    out_c = wq.shape[0]
    cols = im2col_k3(x, stride, padding)
    w2 = np.ascontiguousarray(wq.reshape(out_c, -1))
    y = np.empty((b, out_c, cols.shape[1]), np.float32)
    for i in range(b):
        q, xs = O.quantize_s8(cols[i], 1)          # rounding, no shift (flags: bit 0 only)
        acc = O.gemm_s8(w2, q)                      # [out_c, out_len] exact
        r = np.float32(1.0) / w_scale.astype(np.float32)
        v = (acc.astype(np.float32) * r[:, None]) / xs[None, :]
        if bias is not None:
            v = v + bias.astype(np.float32)[:, None]
        y[i] = O.activation(v, act) if act >= 0 else v
    return y


def layer_norm_axis(x, eps=0.0):
    """cpu::layer_norm_axis (src/cpu/kernels.cc:499-539) on [outer, axis, inner] without gamma /
    beta, AS COMPILED: the function sits inside CT2_FFAST_MATH_BEGIN / END (kernels.cc:461-574:
    `#pragma GCC optimize("-ffast-math")`), so GCC vectorises the reduction over the axis and
    rewrites the divisions. Read off the AVX-512 object of oracle/_ref (objdump; the AVX2 object
    has the same shape) and pinned bit for bit against it (tests/test_whisper_oracle.py):
      * n > 4: FOUR partial sums over k = l, l + 4, ... (sum += x; sum_squares = fma(x, x,
        sum_squares)), folded as (p0 + p2) + (p1 + p3), then the n % 4 last elements one by one;
        n <= 4: one chain from zero;
      * mean = sum * (1 / n), variance = max(fma(sum_squares, 1 / n, -(mean * mean)), 0),
        rstd = 1 / sqrtf(variance + eps), y = (x - mean) * rstd (no contraction in the last one).
    fma is emulated in float64 (exact product of two floats, one rounding)."""
    f32, f64 = np.float32, np.float64
    outer, n, inner = x.shape
    x = x.astype(f32)

    def fma(a, b, c):
        return (a.astype(f64) * b.astype(f64) + c.astype(f64)).astype(f32)

    shape = (outer, inner)
    if n > 4:
        s4 = [np.zeros(shape, f32) for _ in range(4)]
        q4 = [np.zeros(shape, f32) for _ in range(4)]
        n4 = n // 4
        for g in range(n4):
            for l in range(4):
                v = x[:, 4 * g + l, :]
                q4[l] = fma(v, v, q4[l])
                s4[l] = (s4[l] + v).astype(f32)
        s = ((s4[1] + s4[3]).astype(f32) + (s4[0] + s4[2]).astype(f32)).astype(f32)
        q = ((q4[1] + q4[3]).astype(f32) + (q4[0] + q4[2]).astype(f32)).astype(f32)
        first_tail = 4 * n4
    else:
        s, q, first_tail = np.zeros(shape, f32), np.zeros(shape, f32), 0
    for k in range(first_tail, n):
        v = x[:, k, :]
        q = fma(v, v, q)
        s = (s + v).astype(f32)
    inv_n = f32(1) / f32(n)
    mean = (s * inv_n).astype(f32)
    m2 = (mean * mean).astype(f32)
    var = (q.astype(f64) * f64(inv_n) - m2.astype(f64)).astype(f32)
    var = np.maximum(var, f32(0))
    with np.errstate(divide="ignore", invalid="ignore"):
        rstd = (f32(1) / np.sqrt((var + f32(eps)).astype(f32))).astype(f32)
        return ((x - mean[:, None, :]).astype(f32) * rstd[:, None, :]).astype(f32)


def timestamp_rules(logits, history, sample_begin, step, eot, no_timestamps, max_initial_index,
                    begin_ids=()):
    """One row of SuppressTokensBegin + ApplyTimestampRules in place (whisper.cc:741-861): `history`
    = the row's tokens of steps 0 .. step - 1 (prefix included). Returns the masked logits."""
    lg = logits.astype(np.float32).copy()
    low = np.finfo(np.float32).min
    ts_begin, ts_end = no_timestamps + 1, lg.size - 1
    if step == sample_begin:
        for i in begin_ids:
            lg[i] = low
    lg[no_timestamps] = low
    check = False
    if step == sample_begin and step == 0:
        lg[:ts_begin] = low
        lg[ts_begin + max_initial_index + 1:] = low
    elif step > sample_begin:
        last = history[step - 1]
        if last >= ts_begin:
            penultimate = history[step - 2] if step - 1 > sample_begin else last
            if penultimate >= ts_begin:
                lg[ts_begin:] = low
            else:
                lg[:eot] = low
                lg[ts_begin:last] = low
                check = True
        else:
            check = True
            for t in range(step - 1, sample_begin - 1, -1):
                if history[t] >= ts_begin:
                    lg[ts_begin:history[t] + 1] = low
                    break
    if check:
        x = lg.astype(np.float64)
        lp = x - (np.log(np.sum(np.exp(x - x.max()))) + x.max())
        ts = lp[ts_begin:]
        if np.log(np.sum(np.exp(ts - ts.max()))) + ts.max() > lp[:ts_begin].max():
            lg[:ts_begin] = low
    return lg
