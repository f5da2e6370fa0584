"""Host-side check of the operand-splitting arithmetic the 16-bit kernels rely on
(attention_prefill.hip `split16`, attention_cross_h.hip Q^T terms, gemm_h.hip `split_f16`): an fp32
value enters the 16-bit matrix cores as a sum of 16-bit terms, x = hi + lo (+ lo2) with
hi = RN16(x), lo = RN16(x - hi), ... - restated here in numpy, with the bounds DESIGN.md and the
kernel headers quote: 22 significant bits for two float16 terms (an absolute 2^-25 below 1/8), 16 / 24 for two / three bfloat16
terms, and a.b = hi.hi + hi.lo + lo.hi exact up to the dropped lo.lo term."""
import numpy as np


def rn_bf16(x):
    """float32 -> bfloat16 (round to nearest even) -> float32"""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32) << 16
    return r.view(np.float32)


def rn_f16(x):
    return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)


def split(x, rn, terms):
    parts, rest = [], np.asarray(x, np.float32)
    for _ in range(terms):
        p = rn(rest)
        parts.append(p)
        rest = (rest - p).astype(np.float32)   # exact in fp32: p agrees with rest in its leading bits
    return parts


def test_split_carries_the_quoted_bits():
    rng = np.random.default_rng(0)
    # magnitudes the activations of the tested models take; float16 terms must stay in its normal range
    x = (rng.standard_normal(200000) * np.exp(rng.uniform(-6, 6, 200000))).astype(np.float32)
    x = x[np.abs(x) > 1e-3]
    for rn, terms, bits in ((rn_f16, 2, 22), (rn_bf16, 2, 16), (rn_bf16, 3, 24)):
        parts = split(x, rn, terms)
        back = np.zeros_like(x, dtype=np.float64)
        for p in parts:
            back += p.astype(np.float64)
        err = np.abs(back - x.astype(np.float64))
        # float16: below |x| = 1/8 the low term is a float16 SUBNORMAL (steps of 2^-24): the error is
        # then an absolute 2^-25 instead of 2^-22 |x| - still far below anything the tests resolve
        floor = 2.0 ** -25 if rn is rn_f16 else 0.0
        assert np.all(err <= np.maximum(2.0 ** -bits * np.abs(x), floor)), (rn.__name__, terms)
    # three bfloat16 terms are the fp32 value itself wherever the last term is not subnormal-small
    parts = split(x, rn_bf16, 3)
    assert np.array_equal((parts[0].astype(np.float64) + parts[1] + parts[2]).astype(np.float32), x)


def test_three_product_terms_drop_only_lo_lo():
    """a.b as hi.hi + hi.lo + lo.hi (three MFMAs, fp32 accumulate): what is left out is lo_a.lo_b,
    2^-22 (float16) / 2^-16 (bfloat16) of the product - the bound of test_attention_full_half."""
    rng = np.random.default_rng(1)
    a = rng.standard_normal(100000).astype(np.float32)
    b = rng.standard_normal(100000).astype(np.float32)
    keep = (np.abs(a) > 0.125) & (np.abs(b) > 0.125)   # (low float16 terms in the normal range, see above)
    a, b = a[keep], b[keep]
    for rn, bits in ((rn_f16, 22), (rn_bf16, 16)):
        ah, al = split(a, rn, 2)
        bh, bl = split(b, rn, 2)
        three = ah.astype(np.float64) * bh + ah.astype(np.float64) * bl + al.astype(np.float64) * bh
        exact = a.astype(np.float64) * b.astype(np.float64)
        rel = np.abs(three - exact) / np.abs(exact)
        # residual of each operand <= 2^-(bits of two terms); dropped lo.lo <= 2^-(2 * 11) resp. 2^-(2 * 8)
        assert rel.max() <= 2.0 ** -(bits - 1) + 2.0 ** -bits, (rn.__name__, float(rel.max()))


def test_single_rounding_is_what_the_split_avoids():
    """One rounding to the 16-bit type costs 2^-11 (float16) / 2^-8 (bfloat16) per operand - the
    1.2e-3 measured on the NLLB-geometry logits with single-rounded float16 activations (gemm_h.hip)
    is that error; the split forms above stay five to three orders of magnitude below it."""
    rng = np.random.default_rng(2)
    x = rng.standard_normal(100000).astype(np.float32)
    x = x[np.abs(x) > 1e-3]
    assert (np.abs(rn_f16(x) - x) / np.abs(x)).max() > 2.0 ** -12.5
    assert (np.abs(rn_bf16(x) - x) / np.abs(x)).max() > 2.0 ** -9.5
