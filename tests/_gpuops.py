"""Helpers for the -m gpu parity tests: call the C ABI (include/ct2_mi355.h) on torch device
tensors. torch is only the device-memory plumbing here."""
import ctypes as C

import numpy as np
import torch

from ctranslate2_amd import _lib

F32 = 0


def dev(a):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    torch.cuda.synchronize()
    return t


def host(t):
    sync()
    return t.cpu().numpy()


def sync():
    _lib.check(_lib.lib().ct2x_stream_synchronize(None))
    torch.cuda.synchronize()


def i64(v):
    return C.c_int64(int(v))


def quantize_s8(x, round_before_cast=True):
    L = _lib.lib()
    xd = dev(x.astype(np.float32))
    rows, depth = x.shape
    q = torch.empty((rows, depth), dtype=torch.int8, device="cuda")
    s = torch.empty((rows,), dtype=torch.float32, device="cuda")
    _lib.check(L.ct2x_quantize_s8(_lib.ptr(xd), F32, i64(rows), i64(depth),
                                  C.c_int(int(round_before_cast)), _lib.ptr(q), _lib.ptr(s), None))
    return host(q), host(s)


def gemm_s8(a, b):
    L = _lib.lib()
    ad, bd = dev(a.astype(np.int8)), dev(b.astype(np.int8))
    m, k = a.shape
    n = b.shape[0]
    c = torch.empty((m, n), dtype=torch.int32, device="cuda")
    _lib.check(L.ct2x_gemm_s8s8s32(_lib.ptr(ad), _lib.ptr(bd), i64(m), i64(n), i64(k), _lib.ptr(c), None))
    return host(c)


def dequantize_gemm_output(c, sa, sb, bias=None, act=-1):
    L = _lib.lib()
    cd, sad, sbd = dev(c.astype(np.int32)), dev(sa.astype(np.float32)), dev(sb.astype(np.float32))
    bd = dev(bias.astype(np.float32)) if bias is not None else None
    m, n = c.shape
    y = torch.empty((m, n), dtype=torch.float32, device="cuda")
    _lib.check(L.ct2x_dequantize_gemm_output(_lib.ptr(cd), _lib.ptr(sad), _lib.ptr(sbd), _lib.ptr(bd),
                                             i64(m), i64(n), C.c_int(act), _lib.ptr(y), F32, None))
    return host(y)


def gemm_s8_dequant(q, sa, w, sw, bias=None, residual=None, act=-1):
    L = _lib.lib()
    qd, sad, wd, swd = dev(q.astype(np.int8)), dev(sa.astype(np.float32)), dev(w.astype(np.int8)), dev(sw.astype(np.float32))
    bd = dev(bias.astype(np.float32)) if bias is not None else None
    rd = dev(residual.astype(np.float32)) if residual is not None else None
    m, k = q.shape
    n = w.shape[0]
    y = torch.empty((m, n), dtype=torch.float32, device="cuda")
    _lib.check(L.ct2x_gemm_s8_dequant(_lib.ptr(qd), _lib.ptr(sad), _lib.ptr(wd), _lib.ptr(swd),
                                      _lib.ptr(bd), _lib.ptr(rd), i64(m), i64(n), i64(k),
                                      C.c_int(act), _lib.ptr(y), F32, None))
    return host(y)


def norm_gemm_s8_dequant(x, gamma, beta, eps, flags, w, sw, bias=None, residual=None, act=-1):
    L = _lib.lib()
    xd, gd = dev(x.astype(np.float32)), dev(gamma.astype(np.float32))
    bd = dev(beta.astype(np.float32)) if beta is not None else None
    wd, swd = dev(w.astype(np.int8)), dev(sw.astype(np.float32))
    biasd = dev(bias.astype(np.float32)) if bias is not None else None
    rd = dev(residual.astype(np.float32)) if residual is not None else None
    m, k = x.shape
    n = w.shape[0]
    y = torch.empty((m, n), dtype=torch.float32, device="cuda")
    _lib.check(L.ct2x_norm_gemm_s8_dequant(_lib.ptr(xd), _lib.ptr(gd), _lib.ptr(bd), C.c_float(eps),
                                           C.c_int(0), C.c_int(flags), _lib.ptr(wd), _lib.ptr(swd),
                                           _lib.ptr(biasd), _lib.ptr(rd), i64(m), i64(n), i64(k),
                                           C.c_int(act), _lib.ptr(y), None))
    return host(y)


def norm_gemm_s8_dequant_amax(x, gamma, beta, eps, flags, w, sw, bias=None, residual=None, act=-1):
    """-> (y, row absmax of y as the launch left it in a zeroed buffer)"""
    L = _lib.lib()
    xd, gd = dev(x.astype(np.float32)), dev(gamma.astype(np.float32))
    bd = dev(beta.astype(np.float32)) if beta is not None else None
    wd, swd = dev(w.astype(np.int8)), dev(sw.astype(np.float32))
    biasd = dev(bias.astype(np.float32)) if bias is not None else None
    rd = dev(residual.astype(np.float32)) if residual is not None else None
    m, k = x.shape
    n = w.shape[0]
    y = torch.empty((m, n), dtype=torch.float32, device="cuda")
    amax = torch.zeros((m,), dtype=torch.float32, device="cuda")
    _lib.check(L.ct2x_norm_gemm_s8_dequant_amax(_lib.ptr(xd), _lib.ptr(gd), _lib.ptr(bd), C.c_float(eps),
                                                C.c_int(0), C.c_int(flags), _lib.ptr(wd), _lib.ptr(swd),
                                                _lib.ptr(biasd), _lib.ptr(rd), i64(m), i64(n), i64(k),
                                                C.c_int(act), _lib.ptr(y), _lib.ptr(amax), None))
    return host(y), host(amax)


def gemm_s8_dequant_quantizing(x, amax, flags, w, sw, bias=None, residual=None, act=-1):
    L = _lib.lib()
    xd, ad = dev(x.astype(np.float32)), dev(amax.astype(np.float32))
    wd, swd = dev(w.astype(np.int8)), dev(sw.astype(np.float32))
    biasd = dev(bias.astype(np.float32)) if bias is not None else None
    rd = dev(residual.astype(np.float32)) if residual is not None else None
    m, k = x.shape
    n = w.shape[0]
    y = torch.empty((m, n), dtype=torch.float32, device="cuda")
    _lib.check(L.ct2x_gemm_s8_dequant_quantizing(_lib.ptr(xd), _lib.ptr(ad), C.c_int(flags), _lib.ptr(wd),
                                                 _lib.ptr(swd), _lib.ptr(biasd), _lib.ptr(rd), i64(m),
                                                 i64(n), i64(k), C.c_int(act), _lib.ptr(y), None))
    return host(y)


def gemm_s8_glu(q, sa, w0, sw0, b0, w1, sw1, b1, act):
    L = _lib.lib()
    qd, sad = dev(q.astype(np.int8)), dev(sa.astype(np.float32))
    w0d, s0d, w1d, s1d = dev(w0.astype(np.int8)), dev(sw0.astype(np.float32)), dev(w1.astype(np.int8)), dev(sw1.astype(np.float32))
    b0d = dev(b0.astype(np.float32)) if b0 is not None else None
    b1d = dev(b1.astype(np.float32)) if b1 is not None else None
    m, k = q.shape
    n = w0.shape[0]
    y = torch.empty((m, n), dtype=torch.float32, device="cuda")
    _lib.check(L.ct2x_gemm_s8_glu(_lib.ptr(qd), _lib.ptr(sad), _lib.ptr(w0d), _lib.ptr(s0d), _lib.ptr(b0d),
                                  _lib.ptr(w1d), _lib.ptr(s1d), _lib.ptr(b1d), i64(m), i64(n), i64(k),
                                  C.c_int(act), _lib.ptr(y), F32, None))
    return host(y)


def gemm_s8_glu_amax(q, sa, w0, sw0, b0, w1, sw1, b1, act):
    L = _lib.lib()
    qd, sad = dev(q.astype(np.int8)), dev(sa.astype(np.float32))
    w0d, s0d, w1d, s1d = dev(w0.astype(np.int8)), dev(sw0.astype(np.float32)), dev(w1.astype(np.int8)), dev(sw1.astype(np.float32))
    b0d = dev(b0.astype(np.float32)) if b0 is not None else None
    b1d = dev(b1.astype(np.float32)) if b1 is not None else None
    m, k = q.shape
    n = w0.shape[0]
    y = torch.empty((m, n), dtype=torch.float32, device="cuda")
    amax = torch.zeros((m,), dtype=torch.float32, device="cuda")
    _lib.check(L.ct2x_gemm_s8_glu_amax(_lib.ptr(qd), _lib.ptr(sad), _lib.ptr(w0d), _lib.ptr(s0d), _lib.ptr(b0d),
                                       _lib.ptr(w1d), _lib.ptr(s1d), _lib.ptr(b1d), i64(m), i64(n), i64(k),
                                       C.c_int(act), _lib.ptr(y), _lib.ptr(amax), None))
    return host(y), host(amax)


def stream_packable(n, k, dual=False):
    return bool(_lib.lib().ct2x_gemm_s8_stream_packable(i64(n), i64(k), C.c_int(int(dual))))


def _stream_pack(wd, n, k, dual):
    packed = torch.empty((n * k,), dtype=torch.int8, device="cuda")
    _lib.check(_lib.lib().ct2x_gemm_s8_stream_pack(_lib.ptr(wd), i64(n), i64(k), C.c_int(int(dual)),
                                                   _lib.ptr(packed), None))
    return packed


def gemm_s8_dequant_packed(q, sa, w, sw, bias=None, residual=None, act=-1):
    L = _lib.lib()
    qd, sad, wd, swd = dev(q.astype(np.int8)), dev(sa.astype(np.float32)), dev(w.astype(np.int8)), dev(sw.astype(np.float32))
    bd = dev(bias.astype(np.float32)) if bias is not None else None
    rd = dev(residual.astype(np.float32)) if residual is not None else None
    m, k = q.shape
    n = w.shape[0]
    packed = _stream_pack(wd, n, k, False)
    y = torch.empty((m, n), dtype=torch.float32, device="cuda")
    _lib.check(L.ct2x_gemm_s8_dequant_packed(_lib.ptr(qd), _lib.ptr(sad), _lib.ptr(packed), _lib.ptr(swd),
                                             _lib.ptr(bd), _lib.ptr(rd), i64(m), i64(n), i64(k),
                                             C.c_int(act), _lib.ptr(y), None))
    return host(y)


def gemm_s8_glu_packed(q, sa, w0, sw0, b0, w1, sw1, b1, act):
    L = _lib.lib()
    qd, sad = dev(q.astype(np.int8)), dev(sa.astype(np.float32))
    w0d, s0d, w1d, s1d = dev(w0.astype(np.int8)), dev(sw0.astype(np.float32)), dev(w1.astype(np.int8)), dev(sw1.astype(np.float32))
    b0d = dev(b0.astype(np.float32)) if b0 is not None else None
    b1d = dev(b1.astype(np.float32)) if b1 is not None else None
    m, k = q.shape
    n = w0.shape[0]
    p0, p1 = _stream_pack(w0d, n, k, True), _stream_pack(w1d, n, k, True)
    y = torch.empty((m, n), dtype=torch.float32, device="cuda")
    amax = torch.zeros((m,), dtype=torch.float32, device="cuda")
    _lib.check(L.ct2x_gemm_s8_glu_packed(_lib.ptr(qd), _lib.ptr(sad), _lib.ptr(p0), _lib.ptr(s0d), _lib.ptr(b0d),
                                         _lib.ptr(p1), _lib.ptr(s1d), _lib.ptr(b1d), i64(m), i64(n), i64(k),
                                         C.c_int(act), _lib.ptr(y), _lib.ptr(amax), None))
    return host(y), host(amax)


def set_stream_prologue(mask):
    """CT2X_STREAM_PROLOGUE for this process (ct2x_debug_set_stream_prologue) -> the previous mask"""
    return int(_lib.lib().ct2x_debug_set_stream_prologue(C.c_int(mask)))


def norm_gemm_s8_dequant_packed(x, gamma, eps, use_residual, flags, w, sw, bias=None, residual=None,
                                act=-1):
    """RMSNorm + Quantize as the prologue of the streaming Dense -> (y, row absmax of y)"""
    L = _lib.lib()
    xd, gd = dev(x.astype(np.float32)), dev(gamma.astype(np.float32))
    wd, swd = dev(w.astype(np.int8)), dev(sw.astype(np.float32))
    biasd = dev(bias.astype(np.float32)) if bias is not None else None
    rd = dev(residual.astype(np.float32)) if residual is not None else None
    m, k = x.shape
    n = w.shape[0]
    packed = _stream_pack(wd, n, k, False)
    y = torch.empty((m, n), dtype=torch.float32, device="cuda")
    amax = torch.zeros((m,), dtype=torch.float32, device="cuda")
    _lib.check(L.ct2x_norm_gemm_s8_dequant_packed(_lib.ptr(xd), _lib.ptr(gd), C.c_float(eps),
                                                  C.c_int(int(use_residual)), C.c_int(flags),
                                                  _lib.ptr(packed), _lib.ptr(swd), _lib.ptr(biasd),
                                                  _lib.ptr(rd), i64(m), i64(n), i64(k), C.c_int(act),
                                                  _lib.ptr(y), _lib.ptr(amax), None))
    return host(y), host(amax)


def norm_gemm_s8_glu_packed(x, gamma, eps, use_residual, flags, w0, sw0, b0, w1, sw1, b1, act):
    L = _lib.lib()
    xd, gd = dev(x.astype(np.float32)), dev(gamma.astype(np.float32))
    w0d, s0d, w1d, s1d = dev(w0.astype(np.int8)), dev(sw0.astype(np.float32)), dev(w1.astype(np.int8)), dev(sw1.astype(np.float32))
    b0d = dev(b0.astype(np.float32)) if b0 is not None else None
    b1d = dev(b1.astype(np.float32)) if b1 is not None else None
    m, k = x.shape
    n = w0.shape[0]
    p0, p1 = _stream_pack(w0d, n, k, True), _stream_pack(w1d, n, k, True)
    y = torch.empty((m, n), dtype=torch.float32, device="cuda")
    amax = torch.zeros((m,), dtype=torch.float32, device="cuda")
    _lib.check(L.ct2x_norm_gemm_s8_glu_packed(_lib.ptr(xd), _lib.ptr(gd), C.c_float(eps),
                                              C.c_int(int(use_residual)), C.c_int(flags), _lib.ptr(p0),
                                              _lib.ptr(s0d), _lib.ptr(b0d), _lib.ptr(p1), _lib.ptr(s1d),
                                              _lib.ptr(b1d), i64(m), i64(n), i64(k), C.c_int(act),
                                              _lib.ptr(y), _lib.ptr(amax), None))
    return host(y), host(amax)


def gemm_s8_dequant_quantizing_packed(x, amax_parts, flags, w, sw, bias=None, residual=None, act=-1):
    """amax_parts [m, parts]: partial row maxima as a producer's workgroups leave them"""
    L = _lib.lib()
    amax_parts = np.ascontiguousarray(amax_parts, np.float32)
    xd, ad = dev(x.astype(np.float32)), dev(amax_parts)
    wd, swd = dev(w.astype(np.int8)), dev(sw.astype(np.float32))
    biasd = dev(bias.astype(np.float32)) if bias is not None else None
    rd = dev(residual.astype(np.float32)) if residual is not None else None
    m, k = x.shape
    n = w.shape[0]
    packed = _stream_pack(wd, n, k, False)
    y = torch.empty((m, n), dtype=torch.float32, device="cuda")
    _lib.check(L.ct2x_gemm_s8_dequant_quantizing_packed(_lib.ptr(xd), _lib.ptr(ad),
                                                        C.c_int(amax_parts.shape[1]), C.c_int(flags),
                                                        _lib.ptr(packed), _lib.ptr(swd), _lib.ptr(biasd),
                                                        _lib.ptr(rd), i64(m), i64(n), i64(k),
                                                        C.c_int(act), _lib.ptr(y), None))
    return host(y)


def dequantize_rows(x, scale):
    L = _lib.lib()
    xd, sd = dev(x.astype(np.int8)), dev(scale.astype(np.float32))
    rows, depth = x.shape
    y = torch.empty((rows, depth), dtype=torch.float32, device="cuda")
    _lib.check(L.ct2x_dequantize_rows(_lib.ptr(xd), _lib.ptr(sd), i64(rows), i64(depth), _lib.ptr(y), F32, None))
    return host(y)


def gemm_f32(a, b, trans_b=True, bias=None, residual=None, act=-1):
    L = _lib.lib()
    ad, bd = dev(a.astype(np.float32)), dev(b.astype(np.float32))
    biasd = dev(bias.astype(np.float32)) if bias is not None else None
    rd = dev(residual.astype(np.float32)) if residual is not None else None
    m, k = a.shape
    n = b.shape[0] if trans_b else b.shape[1]
    c = torch.empty((m, n), dtype=torch.float32, device="cuda")
    _lib.check(L.ct2x_gemm_f32(_lib.ptr(ad), _lib.ptr(bd), i64(m), i64(n), i64(k), C.c_int(int(trans_b)),
                               _lib.ptr(biasd), _lib.ptr(rd), C.c_int(act), _lib.ptr(c), None))
    return host(c)


def gemm_f16w(a, w_bits, bf16, bias=None, residual=None, act=-1):
    """a fp32 [m, k]; w_bits uint16 [n, k] holding float16 or bfloat16 bit patterns."""
    L = _lib.lib()
    ad = dev(a.astype(np.float32))
    wd = torch.from_numpy(np.ascontiguousarray(w_bits).view(np.int16)).cuda()
    biasd = dev(bias.astype(np.float32)) if bias is not None else None
    rd = dev(residual.astype(np.float32)) if residual is not None else None
    m, k = a.shape
    n = w_bits.shape[0]
    c = torch.empty((m, n), dtype=torch.float32, device="cuda")
    _lib.check(L.ct2x_gemm_f16w(_lib.ptr(ad), _lib.ptr(wd), C.c_int(5 if bf16 else 4), i64(m), i64(n),
                                i64(k), _lib.ptr(biasd), _lib.ptr(rd), C.c_int(act), _lib.ptr(c), None))
    return host(c)


def matmul_f32(a, b, trans_b=False, alpha=1.0):
    L = _lib.lib()
    ad, bd = dev(a.astype(np.float32)), dev(b.astype(np.float32))
    batch, m, k = a.shape
    n = b.shape[1] if trans_b else b.shape[2]
    c = torch.empty((batch, m, n), dtype=torch.float32, device="cuda")
    _lib.check(L.ct2x_matmul_f32(_lib.ptr(ad), _lib.ptr(bd), i64(batch), i64(m), i64(n), i64(k),
                                 C.c_int(int(trans_b)), C.c_float(alpha), _lib.ptr(c), None))
    return host(c)


def layer_norm(x, gamma, beta, eps=1e-5):
    L = _lib.lib()
    xd, gd, bd = dev(x.astype(np.float32)), dev(gamma.astype(np.float32)), dev(beta.astype(np.float32))
    rows, depth = x.shape
    y = torch.empty_like(xd)
    _lib.check(L.ct2x_layer_norm(_lib.ptr(xd), _lib.ptr(gd), _lib.ptr(bd), i64(rows), i64(depth),
                                 C.c_float(eps), _lib.ptr(y), F32, None))
    return host(y)


def rms_norm(x, gamma, eps=1e-6, use_residual=False):
    L = _lib.lib()
    xd, gd = dev(x.astype(np.float32)), dev(gamma.astype(np.float32))
    rows, depth = x.shape
    y = torch.empty_like(xd)
    _lib.check(L.ct2x_rms_norm(_lib.ptr(xd), _lib.ptr(gd), i64(rows), i64(depth), C.c_float(eps),
                               C.c_int(int(use_residual)), _lib.ptr(y), F32, None))
    return host(y)


def norm_quantize_s8(x, gamma, beta, eps, use_residual=False, round_before_cast=True, want_y=True):
    L = _lib.lib()
    xd, gd = dev(x.astype(np.float32)), dev(gamma.astype(np.float32))
    bd = dev(beta.astype(np.float32)) if beta is not None else None
    rows, depth = x.shape
    y = torch.empty_like(xd) if want_y else None
    q = torch.empty((rows, depth), dtype=torch.int8, device="cuda")
    s = torch.empty((rows,), dtype=torch.float32, device="cuda")
    _lib.check(L.ct2x_norm_quantize_s8(_lib.ptr(xd), _lib.ptr(gd), _lib.ptr(bd), i64(rows), i64(depth),
                                       C.c_float(eps), C.c_int(int(use_residual)),
                                       C.c_int(int(round_before_cast)), _lib.ptr(y), _lib.ptr(q),
                                       _lib.ptr(s), None))
    return (host(y) if want_y else None), host(q), host(s)


def softmax(x, lengths=None, log=False):
    L = _lib.lib()
    xd = dev(x.astype(np.float32))
    ld = dev(np.asarray(lengths, np.int32)) if lengths is not None else None
    rows, depth = x.shape
    y = torch.full_like(xd, 7.0)
    _lib.check(L.ct2x_softmax(_lib.ptr(xd), _lib.ptr(ld), i64(rows), i64(depth), C.c_int(int(log)),
                              _lib.ptr(y), F32, None))
    return host(y)


def rotary(x, sin, cos, ndims, interleave):
    L = _lib.lib()
    xd, sd, cd = dev(x.astype(np.float32)), dev(sin.astype(np.float32)), dev(cos.astype(np.float32))
    batch, time, depth = x.shape
    y = torch.empty_like(xd)
    _lib.check(L.ct2x_rotary(_lib.ptr(xd), _lib.ptr(sd), _lib.ptr(cd), i64(batch), i64(time), i64(depth),
                             i64(ndims), C.c_int(int(interleave)), _lib.ptr(y), None))
    return host(y)


def topk(x, k):
    L = _lib.lib()
    xd = dev(x.astype(np.float32))
    rows, depth = x.shape
    v = torch.empty((rows, k), dtype=torch.float32, device="cuda")
    i = torch.empty((rows, k), dtype=torch.int32, device="cuda")
    _lib.check(L.ct2x_topk(_lib.ptr(xd), F32, i64(rows), i64(depth), i64(k), _lib.ptr(v), _lib.ptr(i), None))
    return host(v), host(i)


def random_sample(logits, topk, topp, temperature, uniforms=None, seed=1234):
    """ct2x_random_sample on [rows, vocab] logits; uniforms [rows] in [0, 1) injects the draws."""
    L = _lib.lib()
    lg = dev(np.ascontiguousarray(logits, np.float32))
    rows, vocab = logits.shape
    ud = dev(np.ascontiguousarray(uniforms, np.float32)) if uniforms is not None else None
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    ids = torch.empty(rows, dtype=torch.int32, device="cuda")
    scores = torch.empty(rows, dtype=torch.float32, device="cuda")
    _lib.check(L.ct2x_random_sample(_lib.ptr(lg), i64(rows), i64(vocab), C.c_int32(topk),
                                    C.c_float(topp), C.c_float(temperature), C.c_uint32(seed),
                                    _lib.ptr(ud), _lib.ptr(step), _lib.ptr(ids), _lib.ptr(scores), None))
    return host(ids), host(scores)


def process_logits(logits, seq, penalty=1.0, ngram=0, suppress=()):
    L = _lib.lib()
    lg = dev(np.ascontiguousarray(logits, np.float32))
    rows, vocab = logits.shape
    seq = np.ascontiguousarray(seq, np.int32).reshape(rows, -1)
    length = seq.shape[1]
    sd = dev(seq if length else np.zeros((rows, 1), np.int32))
    seqs = [list(q) for q in suppress if len(q) > 0]
    ids = dev(np.array([t for q in seqs for t in q] or [0], np.int32))
    off = dev(np.concatenate([[0], np.cumsum([len(q) for q in seqs])]).astype(np.int32))
    _lib.check(L.ct2x_process_logits(_lib.ptr(lg), i64(rows), i64(vocab), _lib.ptr(sd),
                                     C.c_int32(length), C.c_float(penalty), C.c_int32(ngram),
                                     _lib.ptr(ids), _lib.ptr(off), C.c_int32(len(seqs)), None))
    return host(lg)


def beam_select(topk_scores, topk_ids, beam, vocab, eos, prefix=None, step=0, first_step=0):
    L = _lib.lib()
    sc = dev(np.ascontiguousarray(topk_scores, np.float32))
    ids = dev(np.ascontiguousarray(topk_ids, np.int32))
    batch, nc = topk_ids.shape
    pid = plen = None
    stride = 0
    if prefix is not None:
        stride = max(1, max(len(q) for q in prefix))
        host_pid = np.zeros((batch, stride), np.int32)
        for b, q in enumerate(prefix):
            host_pid[b, :len(q)] = q
        pid = dev(host_pid)
        plen = dev(np.array([len(q) for q in prefix], np.int32))
    d_step = dev(np.array([first_step + step], np.int32))
    # the kernel records the candidates at history index (step - first_step)
    hist = (step + 1, batch, nc)
    cw = torch.zeros(hist, dtype=torch.int32, device="cuda")
    cs = torch.zeros(hist, dtype=torch.float32, device="cuda")
    co = torch.zeros(hist, dtype=torch.int32, device="cuda")
    ni = torch.empty((batch * beam,), dtype=torch.int32, device="cuda")
    ns = torch.empty((batch * beam,), dtype=torch.float32, device="cuda")
    pa = torch.empty((batch * beam,), dtype=torch.int32, device="cuda")
    _lib.check(L.ct2x_beam_select(_lib.ptr(sc), _lib.ptr(ids), i64(batch), C.c_int32(beam), i64(vocab),
                                  C.c_int32(eos), _lib.ptr(pid), _lib.ptr(plen), C.c_int32(stride),
                                  _lib.ptr(d_step), C.c_int32(first_step), _lib.ptr(cw), _lib.ptr(cs),
                                  _lib.ptr(co), _lib.ptr(ni), _lib.ptr(ns), _lib.ptr(pa), None))
    return host(cw)[step], host(cs)[step], host(co)[step], host(ni), host(ns), host(pa)


def gather_rows(data, idx):
    L = _lib.lib()
    dd, idd = dev(data), dev(np.asarray(idx, np.int32))
    row_bytes = data.strides[0]
    out = torch.empty((len(idx),) + tuple(data.shape[1:]), dtype=dd.dtype, device="cuda")
    _lib.check(L.ct2x_gather_rows(_lib.ptr(dd), _lib.ptr(idd), i64(len(idx)), i64(row_bytes), _lib.ptr(out), None))
    return host(out)


def attention_decode(q, k, v, lengths=None, length=0, kv_group=1, scale=1.0):
    L = _lib.lib()
    qd, kd, vd = dev(q.astype(np.float32)), dev(k.astype(np.float32)), dev(v.astype(np.float32))
    ld = dev(np.asarray(lengths, np.int32)) if lengths is not None else None
    rows, heads, dhead = q.shape
    _, heads_kv, t_max, _ = k.shape
    out = torch.empty((rows, heads * dhead), dtype=torch.float32, device="cuda")
    _lib.check(L.ct2x_attention_decode(_lib.ptr(qd), _lib.ptr(kd), _lib.ptr(vd), _lib.ptr(ld), i64(length),
                                       i64(rows), i64(heads), i64(heads_kv), i64(dhead), i64(t_max),
                                       i64(kv_group), C.c_float(scale), _lib.ptr(out), None))
    return host(out)


def attention_full(q, k, v, lengths, causal, q_offset, scale, compute_kind=0):
    """q [B, Tq, H, dh], k / v [B, Tk, Hkv, dh] -> [B, Tq, H, dh]. compute_kind 1 / 2: the form the
    engine runs under float16 / bfloat16 compute (ct2x_attention_full_half)."""
    L = _lib.lib()
    if compute_kind:
        b, tq, h, dh = q.shape
        tk, hkv = k.shape[1], k.shape[2]
        qd, kd, vd = dev(q.astype(np.float32)), dev(k.astype(np.float32)), dev(v.astype(np.float32))
        ld = torch.from_numpy(np.ascontiguousarray(lengths, np.int32)).cuda() if lengths is not None else None
        out = torch.empty((b, tq, h, dh), dtype=torch.float32, device="cuda")
        _lib.check(L.ct2x_attention_full_half(_lib.ptr(qd), _lib.ptr(kd), _lib.ptr(vd), _lib.ptr(ld), i64(b),
                                              i64(tq), i64(tk), i64(h), i64(hkv), i64(dh),
                                              C.c_int(int(causal)), i64(q_offset), C.c_float(scale),
                                              C.c_int(compute_kind), _lib.ptr(out), None))
        return host(out)
    b, tq, h, dh = q.shape
    tk, hkv = k.shape[1], k.shape[2]
    qd, kd, vd = dev(q.astype(np.float32)), dev(k.astype(np.float32)), dev(v.astype(np.float32))
    ld = torch.from_numpy(np.ascontiguousarray(lengths, np.int32)).cuda() if lengths is not None else None
    out = torch.empty((b, tq, h, dh), dtype=torch.float32, device="cuda")
    _lib.check(L.ct2x_attention_full(_lib.ptr(qd), _lib.ptr(kd), _lib.ptr(vd), _lib.ptr(ld), i64(b),
                                     i64(tq), i64(tk), i64(h), i64(hkv), i64(dh), C.c_int(int(causal)),
                                     i64(q_offset), C.c_float(scale), _lib.ptr(out), None))
    return host(out)


def cross_attention_h(q, kv_bits, bf16, lengths, beam, heads_kv, scale, split=False, calls=1):
    """q [batch * beam, heads, 64] fp32, kv_bits [batch, s_max, 2, heads_kv, 64] uint16 bit patterns of
    the 16-bit type -> [batch * beam, heads, 64]. split: ct2x_cross_attention_hd_split (the memory
    of an (entry, head) over several workgroups), `calls` launches on the same scratch."""
    L = _lib.lib()
    rows, heads, dh = q.shape
    batch, s_max = kv_bits.shape[0], kv_bits.shape[1]
    qd = dev(q.astype(np.float32))
    kvd = torch.from_numpy(np.ascontiguousarray(kv_bits).view(np.int16)).cuda()
    ld = torch.from_numpy(np.ascontiguousarray(lengths, np.int32)).cuda() if lengths is not None else None
    out = torch.empty((rows, heads, dh), dtype=torch.float32, device="cuda")
    if split:
        L.ct2x_cross_attention_split_scratch_bytes.restype = C.c_int64
        nbytes = int(L.ct2x_cross_attention_split_scratch_bytes(i64(batch), i64(beam), i64(heads), i64(dh), i64(s_max)))
        scratch = torch.full((max(nbytes, 4),), 0x7f, dtype=torch.uint8, device="cuda")  # content irrelevant
        for _ in range(calls):
            out.fill_(float("nan"))
            _lib.check(L.ct2x_cross_attention_hd_split(_lib.ptr(qd), _lib.ptr(kvd), C.c_int(5 if bf16 else 4),
                                                       _lib.ptr(ld), i64(batch), i64(beam), i64(heads),
                                                       i64(heads_kv), i64(dh), i64(s_max), C.c_float(scale),
                                                       _lib.ptr(out), _lib.ptr(scratch), i64(nbytes), None))
        return host(out)
    _lib.check(L.ct2x_cross_attention_hd(_lib.ptr(qd), _lib.ptr(kvd), C.c_int(5 if bf16 else 4),
                                         _lib.ptr(ld), i64(batch), i64(beam), i64(heads), i64(heads_kv),
                                         i64(dh), i64(s_max), C.c_float(scale), _lib.ptr(out), None))
    return host(out)


def gemm_s8_argmax(q, sa, w, sw, bias=None):
    """ct2x_gemm_s8_argmax: the vocabulary projection with the greedy arg-max riding in the launch"""
    L = _lib.lib()
    qd, sad, wd, swd = dev(q.astype(np.int8)), dev(sa.astype(np.float32)), dev(w.astype(np.int8)), dev(sw.astype(np.float32))
    bd = dev(bias.astype(np.float32)) if bias is not None else None
    m, k = q.shape
    n = w.shape[0]
    L.ct2x_gemm_s8_argmax_scratch_bytes.restype = C.c_int64
    scratch = torch.zeros((int(L.ct2x_gemm_s8_argmax_scratch_bytes(i64(m))),), dtype=torch.uint8, device="cuda")
    ids = torch.full((m,), -1, dtype=torch.int32, device="cuda")
    _lib.check(L.ct2x_gemm_s8_argmax(_lib.ptr(qd), _lib.ptr(sad), _lib.ptr(wd), _lib.ptr(swd), _lib.ptr(bd),
                                     i64(m), i64(n), i64(k), _lib.ptr(ids), _lib.ptr(scratch), None))
    assert int(scratch.sum().item()) == 0, "the scratch must be zero again after the launch"
    return host(ids)
