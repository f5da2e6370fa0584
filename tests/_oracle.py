"""TEST INFRASTRUCTURE ONLY: ctypes/numpy binding of oracle/libct2_oracle.so, the plain-C
restatement of the reference's CPU algorithm (oracle/ct2_oracle.c). The checker for every
parity test; never imported by the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ORACLE_DIR, "libct2_oracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "oracle"])
        _lib = C.CDLL(path)
        _lib.orc_activation.restype = C.c_float
        _lib.orc_activation.argtypes = [C.c_float, C.c_int]
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _i64(v):
    return C.c_int64(int(v))


def quantize_s8(x, round_before_cast=True):
    x = _f32(x)
    rows, depth = x.shape
    q = np.empty((rows, depth), np.int8)
    s = np.empty((rows,), np.float32)
    lib().orc_quantize_s8(_p(x), _i64(rows), _i64(depth), C.c_int(int(round_before_cast)), _p(q), _p(s))
    return q, s


def gemm_s8(a, b):
    a = np.ascontiguousarray(a, np.int8)
    b = np.ascontiguousarray(b, np.int8)
    m, k = a.shape
    n = b.shape[0]
    c = np.empty((m, n), np.int32)
    lib().orc_gemm_s8(_p(a), _p(b), _i64(m), _i64(n), _i64(k), _p(c))
    return c


def dequantize_gemm_output(c, a_scale, b_scale, bias=None, act=-1):
    c = np.ascontiguousarray(c, np.int32)
    a_scale, b_scale, bias = _f32(a_scale), _f32(b_scale), _f32(bias)
    m, n = c.shape
    y = np.empty((m, n), np.float32)
    lib().orc_dequantize_gemm_output(_p(c), _p(a_scale), _p(b_scale), _p(bias), _i64(m), _i64(n),
                                     C.c_int(act), _p(y))
    return y


def dequantize_rows(x, scale):
    x = np.ascontiguousarray(x, np.int8)
    scale = _f32(scale)
    rows, depth = x.shape
    y = np.empty((rows, depth), np.float32)
    lib().orc_dequantize_rows(_p(x), _p(scale), _i64(rows), _i64(depth), _p(y))
    return y


def gemm_f32(a, b, trans_b=True, alpha=1.0, bias=None, residual=None, act=-1):
    a, b, bias, residual = _f32(a), _f32(b), _f32(bias), _f32(residual)
    m, k = a.shape
    n = b.shape[0] if trans_b else b.shape[1]
    c = np.empty((m, n), np.float32)
    lib().orc_gemm_f32(_p(a), _p(b), _i64(m), _i64(n), _i64(k), C.c_int(int(trans_b)),
                       C.c_float(alpha), _p(bias), _p(residual), C.c_int(act), _p(c))
    return c


def dense_s8(x, w, w_scale, bias=None, residual=None, act=-1, round_before_cast=True):
    """layers::Dense int8 path: Quantize -> Gemm -> Dequantize(+bias, act) -> Add(residual)
    (src/layers/common.cc:353-401)."""
    q, s = quantize_s8(x, round_before_cast)
    c = gemm_s8(q, w)
    y = dequantize_gemm_output(c, s, w_scale, bias, act)
    if residual is not None:
        y = _f32(residual) + y
    return y


def layer_norm(x, gamma, beta, eps=1e-5):
    x, gamma, beta = _f32(x), _f32(gamma), _f32(beta)
    rows, depth = x.shape
    y = np.empty_like(x)
    lib().orc_layer_norm(_p(x), _p(gamma), _p(beta), _i64(rows), _i64(depth), C.c_float(eps), _p(y))
    return y


def rms_norm(x, gamma, eps=1e-6, use_residual=False):
    x, gamma = _f32(x), _f32(gamma)
    rows, depth = x.shape
    y = np.empty_like(x)
    lib().orc_rms_norm(_p(x), _p(gamma), _i64(rows), _i64(depth), C.c_float(eps),
                       C.c_int(int(use_residual)), _p(y))
    return y


def softmax(x, lengths=None, log=False):
    x = _f32(x)
    rows, depth = x.shape
    if lengths is not None:
        lengths = np.ascontiguousarray(lengths, np.int32)
    y = np.empty_like(x)
    lib().orc_softmax(_p(x), _p(lengths), _i64(rows), _i64(depth), C.c_int(int(log)), _p(y))
    return y


def rotary(x, sin, cos, ndims, interleave):
    x, sin, cos = _f32(x), _f32(sin), _f32(cos)
    batch, time, depth = x.shape
    y = np.empty_like(x)
    lib().orc_rotary(_p(x), _p(sin), _p(cos), _i64(batch), _i64(time), _i64(depth), _i64(ndims),
                     C.c_int(int(interleave)), _p(y))
    return y


def _pack_suppress(suppress):
    seqs = [list(q) for q in suppress if len(q) > 0]
    ids = np.array([t for q in seqs for t in q] or [0], np.int32)
    off = np.concatenate([[0], np.cumsum([len(q) for q in seqs])]).astype(np.int32)
    return ids, off, len(seqs)


def process_logits(logits, seq, penalty=1.0, ngram=0, suppress=()):
    """RepetitionPenalty / NoRepeatNgram / SuppressSequences on a copy of logits [rows, vocab]."""
    lg = _f32(logits).copy()
    rows, vocab = lg.shape
    seq = np.ascontiguousarray(seq, np.int32).reshape(rows, -1)
    ids, off, n = _pack_suppress(suppress)
    lib().orc_process_logits(_p(lg), _i64(rows), _i64(vocab), _p(seq), C.c_int32(seq.shape[1]),
                             C.c_float(penalty), C.c_int32(ngram), _p(ids), _p(off), C.c_int32(n))
    return lg


def beam_select(topk_scores, topk_ids, beam, vocab, eos, prefix=None, step=0):
    """One beam step after TopK. prefix: list of per-batch id lists or None. Returns the
    candidates (word, score, origin) [batch, 2*beam] and next ids / scores / parents [batch*beam]."""
    sc = _f32(topk_scores)
    ids = np.ascontiguousarray(topk_ids, np.int32)
    batch, nc = ids.shape
    pid = plen = None
    stride = 0
    if prefix is not None:
        stride = max(1, max(len(q) for q in prefix))
        pid = np.zeros((batch, stride), np.int32)
        for b, q in enumerate(prefix):
            pid[b, :len(q)] = q
        plen = np.array([len(q) for q in prefix], np.int32)
    cw = np.empty((batch, nc), np.int32)
    cs = np.empty((batch, nc), np.float32)
    co = np.empty((batch, nc), np.int32)
    ni = np.empty((batch * beam,), np.int32)
    ns = np.empty((batch * beam,), np.float32)
    pa = np.empty((batch * beam,), np.int32)
    lib().orc_beam_select(_p(sc), _p(ids), _i64(batch), C.c_int32(beam), C.c_int32(vocab),
                          C.c_int32(eos), _p(pid), _p(plen), C.c_int32(stride), C.c_int32(step),
                          _p(cw), _p(cs), _p(co), _p(ni), _p(ns), _p(pa))
    return cw, cs, co, ni, ns, pa


def topp_mask(x, p, mask=-np.inf):
    x = _f32(x)
    rows, depth = x.shape
    y = np.empty_like(x)
    lib().orc_topp_mask(_p(x), _i64(rows), _i64(depth), C.c_float(p), C.c_float(mask), _p(y))
    return y


def activation(x, act):
    return np.array([lib().orc_activation(C.c_float(float(v)), C.c_int(act)) for v in np.ravel(x)],
                    np.float32).reshape(np.shape(x))


def topk(x, k):
    x = _f32(x)
    rows, depth = x.shape
    v = np.empty((rows, k), np.float32)
    i = np.empty((rows, k), np.int32)
    lib().orc_topk(_p(x), _i64(rows), _i64(depth), _i64(k), _p(v), _p(i))
    return v, i


def gather_rows(data, idx):
    data = np.ascontiguousarray(data)
    idx = np.ascontiguousarray(idx, np.int32)
    row_bytes = data.strides[0]
    out = np.empty((len(idx),) + data.shape[1:], data.dtype)
    lib().orc_gather_rows(_p(data), _p(idx), _i64(len(idx)), _i64(row_bytes), _p(out))
    return out


def attention_decode(q, k, v, lengths=None, length=0, kv_group=1, scale=1.0):
    q, k, v = _f32(q), _f32(k), _f32(v)
    rows, heads, dhead = q.shape
    _, heads_kv, t_max, _ = k.shape
    if lengths is not None:
        lengths = np.ascontiguousarray(lengths, np.int32)
    out = np.empty((rows, heads * dhead), np.float32)
    lib().orc_attention_decode(_p(q), _p(k), _p(v), _p(lengths), _i64(length), _i64(rows),
                               _i64(heads), _i64(heads_kv), _i64(dhead), _i64(t_max),
                               _i64(kv_group), C.c_float(scale), _p(out))
    return out


def matmul(a, b, trans_b=False, alpha=1.0):
    """orc_matmul: ops::MatMul with MKL's recovered summation order; raises outside the envelope."""
    a, b = _f32(a), _f32(b)
    batch, m, k = a.shape
    n = b.shape[1] if trans_b else b.shape[2]
    c = np.zeros((batch, m, n), np.float32)
    L = lib()
    L.orc_matmul.restype = C.c_int
    rc = L.orc_matmul(_p(a), _p(b), _i64(batch), _i64(m), _i64(n), _i64(k), C.c_int(int(trans_b)),
                      C.c_float(alpha), _p(c))
    if rc != 0:
        raise ValueError("shape outside the validated MKL envelope")
    return c


def exp_avx2(x):
    x = _f32(x)
    L = lib()
    L.orc_exp_avx2.restype = C.c_float
    return np.array([L.orc_exp_avx2(C.c_float(float(v))) for v in x.ravel()], np.float32).reshape(x.shape)


def _vml(fn, x):
    x = _f32(x).ravel()
    y = np.empty_like(x)
    f = getattr(lib(), fn)
    f.restype = C.c_int
    f(_p(x), _i64(len(x)), _p(y))
    return y


def vml_sin(x):
    """orc_vml_sin: MKL VML vsSin (HA) restated; arguments beyond 10000 rad give 0."""
    return _vml("orc_vml_sin", x)


def vml_cos(x):
    return _vml("orc_vml_cos", x)
