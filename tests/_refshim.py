"""TEST INFRASTRUCTURE ONLY: ctypes loader for oracle/_ref/libct2ref_capi.so -- the REAL
reference CPU library (compiled from /root/reference by oracle/Makefile) behind a C ABI.
Used to pin the C restatement oracle, to generate tests/golden fixtures and to time the
`cpu_baseline`. Never imported by the product package."""
import ctypes as C
import glob
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(os.path.dirname(_HERE), "oracle", "_ref")
MKL_DIR = os.environ.get("CT2REF_MKL_DIR", "/opt/conda/lib")

_lib = None


VARIANT = os.environ.get("CT2REF_VARIANT", "cpu")


def available():
    return os.path.exists(os.path.join(REF_DIR, "libct2ref_capi.so")) and bool(
        glob.glob(os.path.join(MKL_DIR, "libmkl_core.so*")))


def lib():
    global _lib
    if _lib is not None:
        return _lib
    # The reference selects MKL only on GenuineIntel CPUs unless told otherwise
    # (src/cpu/backend.cc:10-31); the GPU box has AMD EPYC host cores.
    os.environ.setdefault("CT2_USE_MKL", "1")
    mode = os.RTLD_GLOBAL | os.RTLD_LAZY  # MKL's core library needs lazy binding
    for name in ("libmkl_core.so", "libmkl_gnu_thread.so", "libmkl_intel_ilp64.so"):
        cands = sorted(glob.glob(os.path.join(MKL_DIR, name + "*")))
        if not cands:
            raise OSError("MKL runtime %s not found under %s" % (name, MKL_DIR))
    try:
        C.CDLL("libgomp.so.1", mode=mode)
    except OSError:
        pass
    for name in ("libmkl_core.so", "libmkl_gnu_thread.so", "libmkl_intel_ilp64.so"):
        C.CDLL(sorted(glob.glob(os.path.join(MKL_DIR, name + "*")))[0], mode=mode)
    # Op-level taps by symbol interposition (oracle/ref_tap.cc): must enter the global scope
    # BEFORE the reference library so that the reference's own PLT calls resolve to the shims.
    global _tap
    if VARIANT == "mi355":
        # Level-1 integration build (oracle/Makefile target mi355): the reference's own front-end
        # linked against the MI355X backend. Same exported names as the CPU build, so the two
        # cannot share a process: CT2REF_VARIANT=mi355 selects this one (tests run it in a
        # subprocess, tests/integration/run_level1.py).
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        C.CDLL(os.path.join(root, "ctranslate2_amd", "libct2_mi355.so"), mode=mode)
        C.CDLL(os.path.join(root, "integration", "libctranslate2_mi355.so"), mode=mode)
        L = C.CDLL(os.path.join(REF_DIR, "libct2ref_capi_mi355.so"), mode=mode)
    else:
        tap_path = os.path.join(REF_DIR, "libct2ref_tap.so")
        if os.path.exists(tap_path):
            _tap = C.CDLL(tap_path, mode=mode)
        C.CDLL(os.path.join(REF_DIR, "libctranslate2_ref.so"), mode=mode)
        L = C.CDLL(os.path.join(REF_DIR, "libct2ref_capi.so"), mode=mode)
    L.ct2ref_model_load_device.restype = C.c_void_p
    L.ct2ref_model_load_device.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int]
    L.ct2ref_last_error.restype = C.c_char_p
    L.ct2ref_model_load.restype = C.c_void_p
    L.ct2ref_model_load.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int]
    L.ct2ref_model_free.argtypes = [C.c_void_p]
    _lib = L
    return L


_tap = None


class taps:
    """with taps() as t: ...reference calls...; t.records -> [(kind, array)] of every
    ops::Quantize (in / out / scale), LayerNorm, MatMul, SoftMax and Dense (Dequantize) call the
    reference made, in call order."""

    def __enter__(self):
        lib()
        if _tap is None:
            raise RuntimeError("oracle/_ref/libct2ref_tap.so is not built")
        _tap.ct2tap_begin()
        self.records = []
        return self

    def __exit__(self, *exc):
        _tap.ct2tap_end()
        n = _tap.ct2tap_count()
        name = C.create_string_buffer(64)
        dtype, rank = C.c_int32(), C.c_int32()
        shape = (C.c_int64 * 8)()
        for i in range(n):
            _tap.ct2tap_info(i, name, 64, C.byref(dtype), C.byref(rank), shape)
            shp = tuple(shape[d] for d in range(rank.value))
            a = np.empty(shp, np.float32 if dtype.value == 0 else np.int8)
            if a.size:
                _tap.ct2tap_data(i, a.ctypes.data_as(C.c_void_p))
            self.records.append((name.value.decode(), a))
        return False


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _check(rc):
    if rc != 0:
        raise RuntimeError(lib().ct2ref_last_error().decode())


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def set_num_threads(n):
    _check(lib().ct2ref_set_num_threads(C.c_int(n)))


def quantize_s8(x, round_before_cast=True):
    x = _f32(x)
    rows, depth = x.shape
    q = np.empty((rows, depth), np.int8)
    s = np.empty((rows,), np.float32)
    _check(lib().ct2ref_quantize_s8(_p(x), C.c_int64(rows), C.c_int64(depth),
                                    C.c_int(int(round_before_cast)), _p(q), _p(s)))
    return q, s


def gemm_s8(a, b):
    a = np.ascontiguousarray(a, np.int8)
    b = np.ascontiguousarray(b, np.int8)
    m, k = a.shape
    n = b.shape[0]
    c = np.empty((m, n), np.int32)
    _check(lib().ct2ref_gemm_s8(_p(a), _p(b), C.c_int64(m), C.c_int64(n), C.c_int64(k), _p(c)))
    return c


def gemm_f32(a, b, trans_b=True, bias=None, residual=None, act=-1):
    a, b, bias, residual = _f32(a), _f32(b), _f32(bias), _f32(residual)
    m, k = a.shape
    n = b.shape[0] if trans_b else b.shape[1]
    c = np.empty((m, n), np.float32)
    _check(lib().ct2ref_gemm_f32(_p(a), _p(b), C.c_int64(m), C.c_int64(n), C.c_int64(k),
                                 C.c_int(int(trans_b)), _p(bias), _p(residual), C.c_int(act), _p(c)))
    return c


def dequantize_gemm_output(c, a_scale, b_scale, bias=None, act=-1):
    c = np.ascontiguousarray(c, np.int32)
    a_scale, b_scale, bias = _f32(a_scale), _f32(b_scale), _f32(bias)
    m, n = c.shape
    y = np.empty((m, n), np.float32)
    _check(lib().ct2ref_dequantize_gemm_output(_p(c), _p(a_scale), _p(b_scale), _p(bias),
                                               C.c_int64(m), C.c_int64(n), C.c_int(act), _p(y)))
    return y


def dequantize_rows(x, scale):
    x = np.ascontiguousarray(x, np.int8)
    scale = _f32(scale)
    rows, depth = x.shape
    y = np.empty((rows, depth), np.float32)
    _check(lib().ct2ref_dequantize_rows(_p(x), _p(scale), C.c_int64(rows), C.c_int64(depth), _p(y)))
    return y


def layer_norm(x, gamma, beta, eps=1e-5):
    x, gamma, beta = _f32(x), _f32(gamma), _f32(beta)
    rows, depth = x.shape
    y = np.empty_like(x)
    _check(lib().ct2ref_layer_norm(_p(x), _p(gamma), _p(beta), C.c_int64(rows), C.c_int64(depth),
                                   C.c_float(eps), _p(y)))
    return y


def rms_norm(x, gamma, eps=1e-6, use_residual=False):
    x, gamma = _f32(x), _f32(gamma)
    rows, depth = x.shape
    y = np.empty_like(x)
    _check(lib().ct2ref_rms_norm(_p(x), _p(gamma), C.c_int64(rows), C.c_int64(depth),
                                 C.c_float(eps), C.c_int(int(use_residual)), _p(y)))
    return y


def softmax(x, lengths=None, log=False):
    x = _f32(x)
    rows, depth = x.shape
    if lengths is not None:
        lengths = np.ascontiguousarray(lengths, np.int32)
    y = np.empty_like(x)
    _check(lib().ct2ref_softmax(_p(x), _p(lengths), C.c_int64(rows), C.c_int64(depth),
                                C.c_int(int(log)), _p(y)))
    return y


def rotary(x, sin, cos, ndims, interleave):
    x, sin, cos = _f32(x), _f32(sin), _f32(cos)
    b, h, t, d = x.shape
    y = np.empty_like(x)
    _check(lib().ct2ref_rotary(_p(x), _p(sin), _p(cos), C.c_int64(b), C.c_int64(h), C.c_int64(t),
                               C.c_int64(d), C.c_int64(ndims), C.c_int(int(interleave)), _p(y)))
    return y


def topk(x, k):
    x = _f32(x)
    rows, depth = x.shape
    v = np.empty((rows, k), np.float32)
    i = np.empty((rows, k), np.int32)
    _check(lib().ct2ref_topk(_p(x), C.c_int64(rows), C.c_int64(depth), C.c_int64(k), _p(v), _p(i)))
    return v, i


def matmul(a, b, trans_b=False, alpha=1.0):
    a, b = _f32(a), _f32(b)
    batch, m, k = a.shape
    n = b.shape[1] if trans_b else b.shape[2]
    c = np.empty((batch, m, n), np.float32)
    _check(lib().ct2ref_matmul(_p(a), _p(b), C.c_int64(batch), C.c_int64(m), C.c_int64(n),
                               C.c_int64(k), C.c_int(int(trans_b)), C.c_float(alpha), _p(c)))
    return c


def sinusoidal_table(depth, positions):
    out = np.empty((positions, depth), np.float32)
    _check(lib().ct2ref_sinusoidal_table(C.c_int64(depth), C.c_int64(positions), _p(out)))
    return out


def rotary_tables(dim, positions, interleave=False, scaling_type=-1, scaling_factor=1.0,
                  base=10000.0, low_freq_factor=1.0, high_freq_factor=4.0,
                  original_max_position_embeddings=0, num_initial_positions=2048):
    """The reference's RotaryEmbeddings sin / cos tables [positions, dim] (read back through
    unit patterns, oracle/ref_capi.cc ct2ref_rotary_tables)."""
    sin = np.empty((positions, dim), np.float32)
    cos = np.empty((positions, dim), np.float32)
    _check(lib().ct2ref_rotary_tables(C.c_int64(dim), C.c_int(int(interleave)), C.c_int(scaling_type),
                                      C.c_float(scaling_factor), C.c_float(base),
                                      C.c_int64(num_initial_positions), C.c_float(low_freq_factor),
                                      C.c_float(high_freq_factor),
                                      C.c_int64(original_max_position_embeddings),
                                      C.c_int64(positions), _p(sin), _p(cos)))
    return sin, cos


def process_logits(logits, seq, penalty=1.0, ngram=0, suppress=()):
    """The reference's LogitsProcessor classes on a copy of logits [rows, vocab]."""
    lg = np.ascontiguousarray(logits, np.float32).copy()
    rows, vocab = lg.shape
    seq = np.ascontiguousarray(seq, np.int32).reshape(rows, -1)
    seqs = [list(q) for q in suppress if len(q) > 0]
    ids = np.array([t for q in seqs for t in q] or [0], np.int32)
    off = np.concatenate([[0], np.cumsum([len(q) for q in seqs])]).astype(np.int32)
    _check(lib().ct2ref_process_logits(_p(lg), C.c_int64(rows), C.c_int64(vocab), _p(seq),
                                       C.c_int32(seq.shape[1]), C.c_float(penalty),
                                       C.c_int32(ngram), _p(ids), _p(off), C.c_int32(len(seqs))))
    return lg


class RefModel:
    """The reference's Translator / Generator on CPU."""

    def __init__(self, path, compute_type="int8", inter_threads=1, intra_threads=0, device="cpu"):
        self._h = lib().ct2ref_model_load_device(path.encode(), device.encode(),
                                                 compute_type.encode(), C.c_int(inter_threads),
                                                 C.c_int(intra_threads))
        if not self._h:
            raise RuntimeError(lib().ct2ref_last_error().decode())

    def close(self):
        if self._h:
            lib().ct2ref_model_free(C.c_void_p(self._h))
            self._h = None

    def compute_type(self):
        buf = C.create_string_buffer(64)
        _check(lib().ct2ref_model_compute_type(C.c_void_p(self._h), buf, C.c_int(64)))
        return buf.value.decode()

    @staticmethod
    def _pack(seqs):
        batch = len(seqs)
        lengths = np.array([len(s) for s in seqs], np.int32)
        stride = max(1, int(lengths.max()) if batch else 1)
        ids = np.zeros((batch, stride), np.int32)
        for i, s in enumerate(seqs):
            ids[i, :len(s)] = s
        return ids, lengths, stride

    def translate_ids(self, src, beam_size=1, max_len=256, min_len=1, length_penalty=1.0,
                      num_hypotheses=1, return_scores=True, max_batch_size=0):
        ids, lengths, stride = self._pack(src)
        batch = len(src)
        out = np.zeros((batch, num_hypotheses, max_len), np.int32)
        out_len = np.zeros((batch, num_hypotheses), np.int32)
        scores = np.zeros((batch, num_hypotheses), np.float32)
        _check(lib().ct2ref_translate(C.c_void_p(self._h), _p(ids), _p(lengths), C.c_int(batch),
                                      C.c_int(stride), C.c_int(beam_size), C.c_int(max_len),
                                      C.c_int(min_len), C.c_float(length_penalty),
                                      C.c_int(num_hypotheses), C.c_int(int(return_scores)),
                                      C.c_int(max_batch_size), _p(out), _p(out_len), _p(scores)))
        hyps = [[out[b, k, :out_len[b, k]].tolist() for k in range(num_hypotheses)]
                for b in range(batch)]
        return hyps, scores

    def translate_prefix_ids(self, sources, prefixes, beam_size=1, max_len=64, min_len=1,
                             length_penalty=1.0, return_scores=True):
        ids, lengths, stride = self._pack(sources)
        pids, plens, pstride = self._pack(prefixes)
        batch = len(sources)
        out = np.zeros((batch, max_len), np.int32)
        out_len = np.zeros((batch,), np.int32)
        scores = np.zeros((batch,), np.float32)
        _check(lib().ct2ref_translate_prefix(C.c_void_p(self._h), _p(ids), _p(lengths), C.c_int(batch),
                                             C.c_int(stride), _p(pids), _p(plens), C.c_int(pstride),
                                             C.c_int(beam_size), C.c_int(max_len), C.c_int(min_len),
                                             C.c_float(length_penalty), C.c_int(int(return_scores)),
                                             _p(out), _p(out_len), _p(scores)))
        return [out[b, :out_len[b]].tolist() for b in range(batch)], scores

    def translate_processors_ids(self, sources, beam_size=1, max_len=64, min_len=1,
                                 length_penalty=1.0, return_scores=True, repetition_penalty=1.0,
                                 no_repeat_ngram_size=0, suppress_sequences_ids=()):
        ids, lengths, stride = self._pack(sources)
        batch = len(sources)
        out = np.zeros((batch, max_len), np.int32)
        out_len = np.zeros((batch,), np.int32)
        scores = np.zeros((batch,), np.float32)
        seqs = [list(q) for q in suppress_sequences_ids]
        flat = np.array([t for q in seqs for t in q] or [0], np.int32)
        lens = np.array([len(q) for q in seqs] or [0], np.int32)
        _check(lib().ct2ref_translate_processors(
            C.c_void_p(self._h), _p(ids), _p(lengths), C.c_int(batch), C.c_int(stride),
            C.c_int(beam_size), C.c_int(max_len), C.c_int(min_len), C.c_float(length_penalty),
            C.c_int(int(return_scores)), C.c_float(repetition_penalty),
            C.c_int(no_repeat_ngram_size), _p(flat), _p(lens), C.c_int(len(seqs)), _p(out),
            _p(out_len), _p(scores)))
        return [out[b, :out_len[b]].tolist() for b in range(batch)], scores

    def generate_ids(self, prompts, beam_size=1, max_len=64, min_len=0, return_scores=True):
        ids, lengths, stride = self._pack(prompts)
        batch = len(prompts)
        out = np.zeros((batch, max_len), np.int32)
        out_len = np.zeros((batch,), np.int32)
        scores = np.zeros((batch,), np.float32)
        _check(lib().ct2ref_generate(C.c_void_p(self._h), _p(ids), _p(lengths), C.c_int(batch),
                                     C.c_int(stride), C.c_int(beam_size), C.c_int(max_len),
                                     C.c_int(min_len), C.c_int(int(return_scores)), _p(out),
                                     _p(out_len), _p(scores)))
        return [out[b, :out_len[b]].tolist() for b in range(batch)], scores

    def generate_ids_with_prompt(self, prompts, beam_size=1, max_len=64, min_len=0, return_scores=True):
        """Generator::generate_batch with include_prompt_in_result = True (ct2ref_generate_with_prompt)."""
        ids, lengths, stride = self._pack(prompts)
        batch = len(prompts)
        width = max_len + 1
        out = np.zeros((batch, width), np.int32)
        out_len = np.zeros((batch,), np.int32)
        scores = np.zeros((batch,), np.float32)
        _check(lib().ct2ref_generate_with_prompt(
            C.c_void_p(self._h), _p(ids), _p(lengths), C.c_int(batch), C.c_int(stride),
            C.c_int(beam_size), C.c_int(max_len), C.c_int(min_len), C.c_int(int(return_scores)),
            C.c_int(width), _p(out), _p(out_len), _p(scores)))
        return [out[b, :out_len[b]].tolist() for b in range(batch)], scores

    def generate_ids_static(self, prompts, static_len, include_prompt, beam_size=1, max_len=64, min_len=0):
        """Generator::generate_batch with static_prompt = the first static_len tokens of every prompt
        (ct2ref_generate_static); returns (sequences, scores)."""
        ids, lengths, stride = self._pack(prompts)
        batch = len(prompts)
        width = max_len + 1
        out = np.zeros((batch, width), np.int32)
        out_len = np.zeros((batch,), np.int32)
        scores = np.zeros((batch,), np.float32)
        _check(lib().ct2ref_generate_static(
            C.c_void_p(self._h), _p(ids), _p(lengths), C.c_int(batch), C.c_int(stride), C.c_int(static_len),
            C.c_int(int(include_prompt)), C.c_int(beam_size), C.c_int(max_len), C.c_int(min_len), C.c_int(1),
            C.c_int(width), _p(out), _p(out_len), _p(scores)))
        return [out[b, :out_len[b]].tolist() for b in range(batch)], scores

    def generate_ex_ids(self, prompts, beam_size=1, max_len=64, min_len=0, num_hypotheses=1,
                        return_alternatives=False, min_alternative_expansion_prob=0.0):
        """Generator::generate_batch with num_hypotheses / return_alternatives (ct2ref_generate_ex).
        Returns (sequences [batch][<= num_hypotheses], scores likewise)."""
        ids, lengths, stride = self._pack(prompts)
        batch = len(prompts)
        out = np.zeros((batch, num_hypotheses, max_len), np.int32)
        out_len = np.zeros((batch, num_hypotheses), np.int32)
        scores = np.zeros((batch, num_hypotheses), np.float32)
        _check(lib().ct2ref_generate_ex(
            C.c_void_p(self._h), _p(ids), _p(lengths), C.c_int(batch), C.c_int(stride),
            C.c_int(beam_size), C.c_int(max_len), C.c_int(min_len), C.c_int(num_hypotheses),
            C.c_int(int(return_alternatives)), C.c_float(min_alternative_expansion_prob), _p(out),
            _p(out_len), _p(scores)))
        seqs = [[out[b, k, :min(out_len[b, k], max_len)].tolist() for k in range(num_hypotheses) if out_len[b, k] >= 0]
                for b in range(batch)]
        scs = [[float(scores[b, k]) for k in range(num_hypotheses) if out_len[b, k] >= 0] for b in range(batch)]
        return seqs, scs

    def generate_processors_ids(self, prompts, beam_size=1, max_len=64, min_len=0,
                                return_scores=True, repetition_penalty=1.0, no_repeat_ngram_size=0,
                                suppress_sequences_ids=()):
        ids, lengths, stride = self._pack(prompts)
        batch = len(prompts)
        out = np.zeros((batch, max_len), np.int32)
        out_len = np.zeros((batch,), np.int32)
        scores = np.zeros((batch,), np.float32)
        seqs = [list(q) for q in suppress_sequences_ids]
        flat = np.array([t for q in seqs for t in q] or [0], np.int32)
        lens = np.array([len(q) for q in seqs] or [0], np.int32)
        _check(lib().ct2ref_generate_processors(
            C.c_void_p(self._h), _p(ids), _p(lengths), C.c_int(batch), C.c_int(stride),
            C.c_int(beam_size), C.c_int(max_len), C.c_int(min_len), C.c_int(int(return_scores)),
            C.c_float(repetition_penalty), C.c_int(no_repeat_ngram_size), _p(flat), _p(lens),
            C.c_int(len(seqs)), _p(out), _p(out_len), _p(scores)))
        return [out[b, :out_len[b]].tolist() for b in range(batch)], scores

    def translate_ex_ids(self, sources, prefixes=None, beam_size=1, max_len=64, min_len=1,
                         length_penalty=1.0, coverage_penalty=0.0, num_hypotheses=1,
                         return_alternatives=False, min_alternative_expansion_prob=0.0,
                         prefix_bias_beta=0.0, return_attention=False):
        """Translator::translate_batch with the remaining decoding options (ct2ref_translate_ex).
        Returns (hypotheses [batch][<= num_hypotheses], scores likewise, attention or None)."""
        ids, lengths, stride = self._pack(sources)
        batch = len(sources)
        if prefixes is not None:
            pids, plens, pstride = self._pack(prefixes)
        out = np.zeros((batch, num_hypotheses, max_len), np.int32)
        out_len = np.zeros((batch, num_hypotheses), np.int32)
        scores = np.zeros((batch, num_hypotheses), np.float32)
        att_stride = stride + 2
        att = np.zeros((batch, num_hypotheses, max_len, att_stride), np.float32) if return_attention else None
        _check(lib().ct2ref_translate_ex(
            C.c_void_p(self._h), _p(ids), _p(lengths), C.c_int(batch), C.c_int(stride),
            _p(pids) if prefixes is not None else None, _p(plens) if prefixes is not None else None,
            C.c_int(pstride if prefixes is not None else 0), C.c_int(beam_size), C.c_int(max_len),
            C.c_int(min_len), C.c_float(length_penalty), C.c_float(coverage_penalty),
            C.c_int(num_hypotheses), C.c_int(int(return_alternatives)),
            C.c_float(min_alternative_expansion_prob), C.c_float(prefix_bias_beta), _p(out),
            _p(out_len), _p(scores), _p(att) if return_attention else None, C.c_int(att_stride)))
        hyps = [[out[b, k, :out_len[b, k]].tolist() for k in range(num_hypotheses) if out_len[b, k] >= 0]
                for b in range(batch)]
        scs = [[float(scores[b, k]) for k in range(num_hypotheses) if out_len[b, k] >= 0]
               for b in range(batch)]
        atts = None
        if return_attention:
            atts = [[att[b, k, :out_len[b, k]].copy() for k in range(num_hypotheses) if out_len[b, k] >= 0]
                    for b in range(batch)]
        return hyps, scs, atts

    def generator_score_ids(self, sequences, max_input_length=1024):
        ids, lengths, stride = self._pack(sequences)
        batch = len(sequences)
        scores = np.zeros((batch, stride), np.float32)
        out_len = np.zeros((batch,), np.int32)
        _check(lib().ct2ref_generator_score(C.c_void_p(self._h), _p(ids), _p(lengths), C.c_int(batch),
                                            C.c_int(stride), C.c_int(max_input_length), _p(scores),
                                            _p(out_len)))
        return [scores[b, :out_len[b]].copy() for b in range(batch)]

    def generator_forward_ids(self, sequences, vocab_size, return_log_probs=False):
        ids, lengths, stride = self._pack(sequences)
        batch = len(sequences)
        out = np.empty((batch, int(lengths.max()), vocab_size), np.float32)
        v = C.c_int64(0)
        _check(lib().ct2ref_generator_forward(C.c_void_p(self._h), _p(ids), _p(lengths), C.c_int(batch),
                                              C.c_int(stride), C.c_int(int(return_log_probs)), _p(out),
                                              C.byref(v)))
        assert v.value == vocab_size, (v.value, vocab_size)
        return out

    def step_logits(self, src, tgt_in, vocab_size):
        """Teacher-forced per-step logits [T, batch, vocab]. `src` includes special tokens."""
        batch = len(tgt_in)
        if src:
            ids, lengths, stride = self._pack(src)
        else:
            ids, lengths, stride = np.zeros((1, 1), np.int32), np.zeros((1,), np.int32), 1
        tgt = np.ascontiguousarray(tgt_in, np.int32)
        T = tgt.shape[1]
        logits = np.empty((T, batch, vocab_size), np.float32)
        v = C.c_int64(0)
        _check(lib().ct2ref_step_logits(C.c_void_p(self._h), _p(ids), _p(lengths), C.c_int(batch),
                                        C.c_int(stride), _p(tgt), C.c_int(T), _p(logits),
                                        C.byref(v)))
        assert v.value == vocab_size, (v.value, vocab_size)
        return logits


class WhisperOptions(C.Structure):
    """models::WhisperOptions (include/ctranslate2/models/whisper.h:11-58); the product's
    ct2x_whisper_options (include/ct2_mi355.h) has the same layout."""
    _fields_ = [("beam_size", C.c_int32), ("patience", C.c_float), ("length_penalty", C.c_float),
                ("repetition_penalty", C.c_float), ("no_repeat_ngram_size", C.c_int32),
                ("max_length", C.c_int32), ("sampling_topk", C.c_int32),
                ("sampling_temperature", C.c_float), ("num_hypotheses", C.c_int32),
                ("return_scores", C.c_int32), ("return_no_speech_prob", C.c_int32),
                ("max_initial_timestamp_index", C.c_int32), ("suppress_blank", C.c_int32),
                ("num_suppress_tokens", C.c_int32), ("suppress_tokens", C.POINTER(C.c_int32))]


def whisper_options(beam_size=5, patience=1.0, length_penalty=1.0, repetition_penalty=1.0,
                    no_repeat_ngram_size=0, max_length=448, sampling_topk=1,
                    sampling_temperature=1.0, num_hypotheses=1, return_scores=False,
                    return_no_speech_prob=False, max_initial_timestamp_index=50,
                    suppress_blank=True, suppress_tokens=(-1,)):
    sup = np.array(list(suppress_tokens), np.int32)
    o = WhisperOptions(beam_size, patience, length_penalty, repetition_penalty,
                       no_repeat_ngram_size, max_length, sampling_topk, sampling_temperature,
                       num_hypotheses, int(return_scores), int(return_no_speech_prob),
                       max_initial_timestamp_index, int(suppress_blank), len(sup),
                       sup.ctypes.data_as(C.POINTER(C.c_int32)))
    o._keep = sup
    return o


class WhisperCalls:
    """models::Whisper behind a C ABI: `prefix` + {encode, generate, detect_language, align}.
    Shared by the reference shim (ct2ref_whisper_*) and the product (ct2x_whisper_*), whose entry
    points take the same arguments."""

    def __init__(self, L, prefix, handle, check):
        self._L, self._prefix, self._h, self._check = L, prefix, handle, check

    def _fn(self, name):
        return getattr(self._L, self._prefix + name)

    def info(self):
        v = [C.c_int32() for _ in range(3)]
        self._check(self._fn("info")(C.c_void_p(self._h), *[C.byref(x) for x in v]))
        return dict(is_multilingual=bool(v[0].value), n_mels=v[1].value, num_languages=v[2].value)

    def encode(self, features, d_model):
        f = _f32(features)
        b, m, t = f.shape
        out = np.zeros((b, (t + 1) // 2 + 1, d_model), np.float32)
        dims = np.zeros(3, np.int64)
        self._check(self._fn("encode")(C.c_void_p(self._h), _p(f), C.c_int(b), C.c_int(m), C.c_int(t),
                                       _p(out), C.c_int64(out.size), _p(dims)))
        n = int(dims[0] * dims[1] * dims[2])
        return out.reshape(-1)[:n].reshape(tuple(int(x) for x in dims)).copy()

    def generate(self, features, prompts, max_len=64, **kw):
        f = _f32(features)
        b, m, t = f.shape
        o = whisper_options(**kw)
        ids, lengths, stride = RefModel._pack(prompts)
        nh = o.num_hypotheses
        out = np.zeros((b, nh, max_len), np.int32)
        out_len = np.zeros((b, nh), np.int32)
        scores = np.zeros((b, nh), np.float32)
        nsp = np.zeros(b, np.float32)
        self._check(self._fn("generate")(C.c_void_p(self._h), _p(f), C.c_int(b), C.c_int(m), C.c_int(t),
                                         _p(ids), _p(lengths), C.c_int(stride), C.byref(o),
                                         C.c_int(max_len), _p(out), _p(out_len), _p(scores), _p(nsp)))
        hyps = [[out[i, k, :out_len[i, k]].tolist() for k in range(nh) if out_len[i, k] >= 0]
                for i in range(b)]
        return hyps, scores, nsp

    def detect_language(self, features, num_languages):
        f = _f32(features)
        b, m, t = f.shape
        ids = np.zeros((b, num_languages), np.int32)
        probs = np.zeros((b, num_languages), np.float32)
        self._check(self._fn("detect_language")(C.c_void_p(self._h), _p(f), C.c_int(b), C.c_int(m),
                                                C.c_int(t), C.c_int(num_languages), _p(ids), _p(probs)))
        return ids, probs

    def align(self, features, start_sequence, text_tokens, num_frames, median_filter_width=7):
        f = _f32(features)
        b, m, t = f.shape
        start = np.array(start_sequence, np.int32)
        ids, lengths, stride = RefModel._pack(text_tokens)
        nf = np.array(num_frames, np.int32)
        cap = int(stride + 2 + t)
        pairs = np.zeros((b, cap, 2), np.int32)
        n_pairs = np.zeros(b, np.int32)
        probs = np.zeros((b, stride), np.float32)
        self._check(self._fn("align")(C.c_void_p(self._h), _p(f), C.c_int(b), C.c_int(m), C.c_int(t),
                                      _p(start), C.c_int(len(start)), _p(ids), _p(lengths),
                                      C.c_int(stride), _p(nf), C.c_int(median_filter_width),
                                      C.c_int(cap), _p(pairs), _p(n_pairs), _p(probs)))
        return ([pairs[i, :n_pairs[i]].tolist() for i in range(b)],
                [probs[i, :lengths[i]].copy() for i in range(b)])


class RefWhisper(WhisperCalls):
    """The reference's models::Whisper (CPU, or Device::CUDA = this backend in the Level-1 build)."""

    def __init__(self, path, compute_type="float32", intra_threads=0, device="cpu"):
        L = lib()
        L.ct2ref_whisper_load.restype = C.c_void_p
        L.ct2ref_whisper_load.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
        L.ct2ref_whisper_free.argtypes = [C.c_void_p]
        h = L.ct2ref_whisper_load(path.encode(), device.encode(), compute_type.encode(),
                                  C.c_int(intra_threads))
        if not h:
            raise RuntimeError(L.ct2ref_last_error().decode())
        super().__init__(L, "ct2ref_whisper_", h, _check)

    def close(self):
        if self._h:
            self._L.ct2ref_whisper_free(C.c_void_p(self._h))
            self._h = None


def conv1d(x, w, bias=None, w_scale=None, stride=1, padding=0, act=-1):
    """ops::Conv1D on CPU: x [batch, in_c, length], w [out_c, in_c, kernel] float32 or int8 (+ w_scale)."""
    x = _f32(x)
    b, c, length = x.shape
    out_c, _, kernel = w.shape
    w = np.ascontiguousarray(w, np.int8 if w_scale is not None else np.float32)
    out_cap = (length + 2 * padding - kernel) // stride + 1
    y = np.empty((b, out_c, out_cap), np.float32)
    out_len = C.c_int64()
    _check(lib().ct2ref_conv1d(_p(x), _p(w), _p(_f32(w_scale)), _p(_f32(bias)), C.c_int64(b), C.c_int64(c),
                               C.c_int64(length), C.c_int64(out_c), C.c_int64(kernel), C.c_int64(stride),
                               C.c_int64(padding), C.c_int(act), _p(y), C.byref(out_len)))
    assert out_len.value == out_cap
    return y


def layer_norm_axis(x, eps=0.0):
    x = _f32(x)
    outer, n, inner = x.shape
    y = np.empty_like(x)
    _check(lib().ct2ref_layer_norm_axis(_p(x), C.c_int64(outer), C.c_int64(n), C.c_int64(inner),
                                        C.c_float(eps), _p(y)))
    return y
