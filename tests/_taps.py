"""Alignment of the engine's debugging taps (ct2x_debug_step_taps) with the op-level taps on the
real reference library (oracle/ref_tap.cc, _refshim.taps) for a teacher-forced run of a pre-norm
encoder-decoder Transformer: both are reduced to the same ordered list of EVENTS, one per int8
Dense input of the run (the activations + row scales entering every layers::Dense, plus the fp32
tensor the Quantize consumed), under canonical names:

  enc.l<l>.self.qin   LayerNorm(x) -> Quantize                    -> fused QKV projection
  enc.l<l>.self.ctxq  attention context -> Quantize               -> output projection
  enc.l<l>.ffn.qin    LayerNorm(x) -> Quantize                    -> FFN linear_0
  enc.l<l>.ffn.hq     ReLU(linear_0) -> Quantize                  -> FFN linear_1
  mem.l<l>.qin        encoder output -> Quantize                  -> cross-attention K/V projection
  s<t>.l<l>.self.qin / .self.ctxq / .cross.qin / .cross.ctxq / .ffn.qin / .ffn.hq
  s<t>.out.qin        final LayerNorm -> Quantize                 -> vocabulary projection

Encoder-side events hold the VALID rows only, in (batch, position) order (the reference removes
padding); decoder-side events hold one row per batch entry. The producing op of an event's fp32
input follows from its name (PRODUCER)."""
import zlib

import numpy as np

PRODUCER = {
    "self.qin": "LayerNorm (of the residual stream)",
    "cross.qin": "LayerNorm (of the residual stream)",
    "ffn.qin": "LayerNorm (of the residual stream)",
    "out.qin": "LayerNorm (of the residual stream)",
    "self.ctxq": "self-attention (MatMul / SoftMax / MatMul)",
    "cross.ctxq": "cross-attention (MatMul / SoftMax / MatMul)",
    "ffn.hq": "Dense linear_0 + ReLU (int8 GEMM + Dequantize)",
    "qin": "encoder output LayerNorm",
}


def producer_of(name):
    for k, v in PRODUCER.items():
        if name.endswith(k):
            return v
    return "?"


def expected_names(n_enc, n_dec, steps):
    names = []
    for l in range(n_enc):
        names += ["enc.l%d.self.qin" % l, "enc.l%d.self.ctxq" % l, "enc.l%d.ffn.qin" % l,
                  "enc.l%d.ffn.hq" % l]
    for s in range(steps):
        for l in range(n_dec):
            names += ["s%d.l%d.self.qin" % (s, l), "s%d.l%d.self.ctxq" % (s, l),
                      "s%d.l%d.cross.qin" % (s, l)]
            if s == 0:
                names.append("mem.l%d.qin" % l)
            names += ["s%d.l%d.cross.ctxq" % (s, l), "s%d.l%d.ffn.qin" % (s, l),
                      "s%d.l%d.ffn.hq" % (s, l)]
        names.append("s%d.out.qin" % s)
    return names


def ref_events(records, n_enc, n_dec, steps):
    """records: _refshim.taps().records of ONE RefModel.step_logits call."""
    triples = []
    i = 0
    while i < len(records):
        if records[i][0] == "quantize.in":
            f, q, s = records[i][1], records[i + 1][1], records[i + 2][1]
            depth = f.shape[-1]
            # layers::Dense quantises with shift_to_uint8 on x86 (src/layers/common.cc:296-298):
            # the stored bytes are q + 128 as uint8
            qs = (q.reshape(-1, depth).view(np.uint8).astype(np.int16) - 128).astype(np.int8)
            triples.append((f.reshape(-1, depth), qs, s.reshape(-1)))
            i += 3
        else:
            i += 1
    names = expected_names(n_enc, n_dec, steps)
    assert len(triples) == len(names), (len(triples), len(names))
    return [(n, t[0], t[1], t[2]) for n, t in zip(names, triples)]


def engine_events(taps, lengths, s_max, n_enc, n_dec, steps):
    """taps: [(name, array)] of ct2x_debug_step_taps (Translator.step_taps)."""
    lengths = np.asarray(lengths)
    valid = (np.arange(s_max)[None, :] < lengths[:, None]).reshape(-1)
    by_scope = {}
    for name, a in taps:
        parts = name.split(".")
        scope, kind = ".".join(parts[:-2]), parts[-1]
        by_scope.setdefault(scope, []).append((kind, a))

    def quant_events(scope):
        items = by_scope.get(scope, [])
        out = []
        for j, (kind, a) in enumerate(items):
            if kind == "q":
                f = items[j - 1][1] if j > 0 and items[j - 1][0] == "f" else None
                out.append((f, a, items[j + 1][1].reshape(-1)))
        return out

    ev = {}
    for l in range(n_enc):
        q = quant_events("enc.l%d.self" % l)
        ev["enc.l%d.self.qin" % l], ev["enc.l%d.self.ctxq" % l] = q[0], q[1]
        q = quant_events("enc.l%d.ffn" % l)
        ev["enc.l%d.ffn.qin" % l], ev["enc.l%d.ffn.hq" % l] = q[0], q[1]
    mem = quant_events("enc.kv")
    for l in range(n_dec):
        ev["mem.l%d.qin" % l] = mem[0]
    for s in range(steps):
        for l in range(n_dec):
            for sub, tags in (("self", ("qin", "ctxq")), ("cross", ("qin", "ctxq")),
                              ("ffn", ("qin", "hq"))):
                q = quant_events("s%d.dec.l%d.%s" % (s, l, sub))
                ev["s%d.l%d.%s.%s" % (s, l, sub, tags[0])] = q[0]
                ev["s%d.l%d.%s.%s" % (s, l, sub, tags[1])] = q[1]
        ev["s%d.out.qin" % s] = quant_events("s%d.dec.out" % s)[0]
    out = []
    for n in expected_names(n_enc, n_dec, steps):
        f, q, s = ev[n]
        if n.startswith("enc.") or n.startswith("mem."):
            f = None if f is None else f[valid]
            q, s = q[valid], s[valid]
        out.append((n, f, q, s))
    return out


def row_hashes(q, s):
    """One CRC per row over the int8 values and the fp32 scale."""
    q = np.ascontiguousarray(q, np.int8)
    s = np.ascontiguousarray(s, np.float32)
    return np.array([zlib.crc32(q[r].tobytes() + s[r].tobytes()) for r in range(q.shape[0])],
                    np.uint32)


def ulp_distance(a, b):
    def key(x):
        i = np.ascontiguousarray(x, np.float32).view(np.int32).astype(np.int64)
        return np.where(i < 0, -(i & 0x7FFFFFFF), i)
    return np.abs(key(a) - key(b))


def first_divergence(engine_ev, ref_hashes=None, ref_ev=None):
    """Walks the events in execution order. Returns None when every int8 Dense input (and scale)
    is identical, else a dict describing the first event / row / element that differs."""
    for idx, (name, f, q, s) in enumerate(engine_ev):
        h = row_hashes(q, s)
        if ref_ev is not None:
            rn, rf, rq, rs = ref_ev[idx]
            assert rn == name
            want = row_hashes(rq, rs)
        else:
            want = ref_hashes[name]
        if h.shape != want.shape:
            return {"event": name, "index": idx, "error": "row count %s vs %s" % (h.shape, want.shape)}
        bad = np.nonzero(h != want)[0]
        if bad.size == 0:
            continue
        d = {"event": name, "index": idx, "producer": producer_of(name), "row": int(bad[0]),
             "rows_differing": int(bad.size), "rows": int(h.size)}
        if ref_ev is not None:
            r = int(bad[0])
            cols = np.nonzero(q[r] != rq[r])[0]
            d["scale_engine"], d["scale_ref"] = float(s[r]), float(rs[r])
            d["int8_differing_in_row"] = int(cols.size)
            if cols.size:
                c = int(cols[0])
                d.update(col=c, int8_engine=int(q[r, c]), int8_ref=int(rq[r, c]))
                if f is not None:
                    d.update(fp32_engine=float(f[r, c]), fp32_ref=float(rf[r, c]),
                             ulp_distance=int(ulp_distance(f[r, c], rf[r, c])))
            if f is not None:
                u = ulp_distance(f[r], rf[r])
                d["row_fp32_max_ulp"] = int(u.max())
                d["row_fp32_elements_off"] = int((u > 0).sum())
        return d
    return None
