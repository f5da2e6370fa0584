"""AWQ int4 ops (ops::DequantizeAwq / GemmAwq / GemvAwq, src/ops/awq/*) against the numpy
restatement in oracle/awq_oracle.py. PARITY UNPINNED: the reference implements these for CUDA only
(no CPU path, no vectors), so the checks are (a) the dequantised weight bit for bit against the
restated arithmetic - integer subtraction, one float16 rounding of the product - and (b) the fused
products against the float64 product of the float16-rounded operands, within the error of an fp32
accumulation rounded to float16 once."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import awq_oracle as A  # noqa: E402

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _i64(v):
    return C.c_int64(int(v))


def _case_gemm(rng, in_c, out_c, group):
    q = rng.integers(0, 16, (in_c, out_c))
    z = rng.integers(0, 16, (in_c // group, out_c))
    s = (rng.random((in_c // group, out_c)) * 0.02 + 0.001).astype(np.float16)
    qw, qz = A.pack_gemm(q, z, group)
    return qw, s, qz


@pytest.mark.parametrize("in_c,out_c,group", [(256, 128, 128), (4096, 4096, 128), (512, 1032, 64),
                                              (128, 8, 32)])
def test_awq_dequantize_gemm_layout(gpu, in_c, out_c, group):
    import torch
    from ctranslate2_amd import _lib
    rng = np.random.default_rng(in_c + out_c)
    qw, s, qz = _case_gemm(rng, in_c, out_c, group)
    out = torch.empty((in_c, out_c), dtype=torch.float16, device="cuda")
    dq, ds, dz = _dev(qw), _dev(s), _dev(qz)
    _lib.check(gpu.ct2x_awq_dequantize(_ptr(dq), _ptr(ds), _ptr(dz), _i64(in_c), _i64(out_c),
                                       _i64(group), _ptr(out), None))
    assert np.array_equal(out.cpu().numpy(), A.dequantize_gemm(qw, s, qz))


@pytest.mark.parametrize("m,in_c,out_c,group", [(16, 4096, 4096, 128), (1, 4096, 11008, 128),
                                                (5, 512, 1032, 64), (40, 1024, 256, 128),
                                                (16, 128, 8, 32)])
def test_awq_gemm_fused(gpu, m, in_c, out_c, group):
    import torch
    from ctranslate2_amd import _lib
    rng = np.random.default_rng(m + in_c + out_c)
    qw, s, qz = _case_gemm(rng, in_c, out_c, group)
    x = (rng.standard_normal((m, in_c)) * 0.5).astype(np.float16)
    y16 = torch.empty((m, out_c), dtype=torch.float16, device="cuda")
    y32 = torch.empty((m, out_c), dtype=torch.float32, device="cuda")
    dx, dq, ds, dz = _dev(x), _dev(qw), _dev(s), _dev(qz)
    _lib.check(gpu.ct2x_awq_gemm(_ptr(dx), _ptr(dq), _ptr(ds), _ptr(dz), _i64(m), _i64(in_c),
                                 _i64(out_c), _i64(group), _ptr(y16), _ptr(y32), None))
    ref = A.gemm(x, qw, s, qz)
    scale = np.abs(ref).max()
    # fp32 accumulation over in_c products: relative error ~ in_c * 2^-24 of the row's magnitude
    assert np.abs(y32.cpu().numpy() - ref).max() <= 2e-5 * scale * max(1, in_c // 1024)
    assert np.array_equal(y16.cpu().numpy(), y32.cpu().numpy().astype(np.float16))


@pytest.mark.parametrize("m,in_c,out_c,group", [(1, 4096, 4096, 128), (8, 4096, 4096, 64),
                                                (16, 1024, 11008, 128), (33, 256, 48, 128),
                                                (3, 128, 16, 64)])
def test_awq_gemv_layout(gpu, m, in_c, out_c, group):
    import torch
    from ctranslate2_amd import _lib
    rng = np.random.default_rng(m + in_c + out_c + group)
    q = rng.integers(0, 16, (out_c, in_c))
    z = rng.integers(0, 16, (out_c, in_c // group))
    s = (rng.random((out_c, in_c // group)) * 0.02 + 0.001).astype(np.float16)
    qw, qz, sc = A.pack_gemv(q, z, s, group)
    zeros_w, sf_w = A.gemv_widths(in_c, group)
    assert qz.shape[1] == zeros_w and sc.shape[1] == sf_w
    dq, ds, dz = _dev(qw), _dev(sc), _dev(qz)
    wd = torch.empty((out_c, in_c), dtype=torch.float16, device="cuda")
    _lib.check(gpu.ct2x_awq_dequantize_gemv(_ptr(dq), _ptr(ds), _ptr(dz), _i64(out_c), _i64(in_c),
                                            _i64(group), _i64(zeros_w), _i64(sf_w), _ptr(wd), None))
    assert np.array_equal(wd.cpu().numpy(), A.dequantize_gemv(qw, sc, qz, group))
    x = (rng.standard_normal((m, in_c)) * 0.5).astype(np.float16)
    y32 = torch.empty((m, out_c), dtype=torch.float32, device="cuda")
    y16 = torch.empty((m, out_c), dtype=torch.float16, device="cuda")
    dx = _dev(x)
    _lib.check(gpu.ct2x_awq_gemv(_ptr(dx), _ptr(dq), _ptr(ds), _ptr(dz), _i64(m), _i64(in_c),
                                 _i64(out_c), _i64(group), _i64(zeros_w), _i64(sf_w), _ptr(y16),
                                 _ptr(y32), None))
    ref = A.gemv(x, qw, sc, qz, group)
    scale = np.abs(ref).max()
    assert np.abs(y32.cpu().numpy() - ref).max() <= 2e-5 * scale * max(1, in_c // 1024)
    assert np.array_equal(y16.cpu().numpy(), y32.cpu().numpy().astype(np.float16))


def test_awq_envelope_is_explicit(gpu):
    """Shapes outside the fused kernels' envelope are refused, not mis-computed."""
    import torch
    from ctranslate2_amd import _lib
    buf = torch.zeros(1 << 16, dtype=torch.int32, device="cuda")
    with pytest.raises(RuntimeError):
        _lib.check(gpu.ct2x_awq_gemm(_ptr(buf), _ptr(buf), _ptr(buf), _ptr(buf), _i64(4), _i64(48),
                                     _i64(64), _i64(16), _ptr(buf), None, None))
    with pytest.raises(RuntimeError):
        _lib.check(gpu.ct2x_awq_gemv(_ptr(buf), _ptr(buf), _ptr(buf), _ptr(buf), _i64(4), _i64(192),
                                     _i64(64), _i64(64), _i64(1), _i64(8), _ptr(buf), None, None))


@pytest.mark.parametrize("layout,group", [("gemm", 128), ("gemv", 128), ("gemv", 64)])
def test_awq_model_runs_like_its_float16_twin(gpu, layout, group, tmp_path):
    """A Llama-style decoder whose layer Dense weights are AWQ int4 (model.bin variables weight /
    weight_scale / weight_zero + config.json quantization_type, as the reference's converter writes
    them) against the float16 model holding exactly the weights those int4 values dequantise to:
    same teacher-forced logits up to the summation order of two different GEMM kernels, same greedy
    tokens. (The reference itself refuses AWQ on ROCm and has no CPU path: src/layers/common.cc:402.)"""
    import ctranslate2_amd as ct2
    from ctranslate2_amd import model_builder as mb
    kw = dict(num_layers=2, d_model=256, ffn_dim=512, num_heads=4, num_heads_kv=2, vocab_size=320,
              seed=3, quantization="float16", w_std=0.05)
    d_awq = mb.build_decoder_lm(str(tmp_path / "awq"), awq=(layout, group, False), **kw)
    d_twin = mb.build_decoder_lm(str(tmp_path / "twin"), awq=(layout, group, True), **kw)
    a = ct2.Translator(d_awq, device="cuda", compute_type="float16")
    b = ct2.Translator(d_twin, device="cuda", compute_type="float16")
    rng = np.random.default_rng(0)
    tgt = rng.integers(4, 320, (3, 12)).astype(np.int32)
    la, lb = a.step_logits([], tgt), b.step_logits([], tgt)
    assert np.isfinite(la).all() and np.abs(la).max() > 0.1
    assert np.abs(la - lb).max() <= 2e-3 * np.abs(lb).max()
    prompts = [rng.integers(4, 320, n).tolist() for n in (5, 9, 3)]
    out_a, _ = a.generate_ids(prompts, max_length=10, min_length=10)
    out_b, _ = b.generate_ids(prompts, max_length=10, min_length=10)
    assert out_a == out_b


# ---- round 6: the bfloat16 arm ("a bf16 variant the reference lacks") ---------------
def _bf16_case_gemm(rng, in_c, out_c, group):
    qw, s, qz = _case_gemm(rng, in_c, out_c, group)
    return qw, A.bf16_bits(s.astype(np.float32)), qz


@pytest.mark.parametrize("in_c,out_c,group", [(256, 128, 128), (4096, 4096, 128), (512, 1032, 64),
                                              (128, 8, 32)])
def test_awq_bf16_dequantize_gemm_layout(gpu, in_c, out_c, group):
    """(q - z) * s with a bfloat16 scale, rounded to bfloat16 once: bit for bit the restated arithmetic."""
    import torch
    from ctranslate2_amd import _lib
    rng = np.random.default_rng(7 + in_c + out_c)
    qw, s, qz = _bf16_case_gemm(rng, in_c, out_c, group)
    out = torch.empty((in_c, out_c), dtype=torch.int16, device="cuda")
    dq, ds, dz = _dev(qw), _dev(s.view(np.int16)), _dev(qz)
    _lib.check(gpu.ct2x_awq_dequantize_bf16(_ptr(dq), _ptr(ds), _ptr(dz), _i64(in_c), _i64(out_c),
                                            _i64(group), _ptr(out), None))
    assert np.array_equal(out.cpu().numpy().view(np.uint16), A.dequantize_gemm(qw, s, qz, bf16=True))


@pytest.mark.parametrize("m,in_c,out_c,group", [(16, 4096, 4096, 128), (1, 4096, 11008, 128),
                                                (5, 512, 1032, 64), (40, 1024, 256, 128),
                                                (16, 128, 8, 32)])
def test_awq_bf16_gemm_fused(gpu, m, in_c, out_c, group):
    import torch
    from ctranslate2_amd import _lib
    rng = np.random.default_rng(11 + m + in_c + out_c)
    qw, s, qz = _bf16_case_gemm(rng, in_c, out_c, group)
    x = A.bf16_bits((rng.standard_normal((m, in_c)) * 0.5).astype(np.float32))
    y16 = torch.empty((m, out_c), dtype=torch.int16, device="cuda")
    y32 = torch.empty((m, out_c), dtype=torch.float32, device="cuda")
    dx, dq, ds, dz = _dev(x.view(np.int16)), _dev(qw), _dev(s.view(np.int16)), _dev(qz)
    _lib.check(gpu.ct2x_awq_gemm_bf16(_ptr(dx), _ptr(dq), _ptr(ds), _ptr(dz), _i64(m), _i64(in_c),
                                      _i64(out_c), _i64(group), _ptr(y16), _ptr(y32), None))
    ref = A.gemm(x, qw, s, qz, bf16=True)
    scale = np.abs(ref).max()
    # exact products of the rounded operands, fp32 accumulation: ~ in_c * 2^-24 of the row's magnitude
    assert np.abs(y32.cpu().numpy() - ref).max() <= 2e-5 * scale * max(1, in_c // 1024)
    assert np.array_equal(y16.cpu().numpy().view(np.uint16), A.bf16_bits(y32.cpu().numpy()))


@pytest.mark.parametrize("m,in_c,out_c,group", [(1, 4096, 4096, 128), (8, 4096, 4096, 64),
                                                (16, 1024, 11008, 128), (33, 256, 48, 128),
                                                (3, 128, 16, 64)])
def test_awq_bf16_gemv_layout(gpu, m, in_c, out_c, group):
    import torch
    from ctranslate2_amd import _lib
    rng = np.random.default_rng(13 + m + in_c + out_c + group)
    q = rng.integers(0, 16, (out_c, in_c))
    z = rng.integers(0, 16, (out_c, in_c // group))
    s = (rng.random((out_c, in_c // group)) * 0.02 + 0.001).astype(np.float16)
    qw, qz, sc16 = A.pack_gemv(q, z, s, group)
    sc = A.bf16_bits(sc16.astype(np.float32))
    zeros_w, sf_w = A.gemv_widths(in_c, group)
    dq, ds, dz = _dev(qw), _dev(sc.view(np.int16)), _dev(qz)
    wd = torch.empty((out_c, in_c), dtype=torch.int16, device="cuda")
    _lib.check(gpu.ct2x_awq_dequantize_gemv_bf16(_ptr(dq), _ptr(ds), _ptr(dz), _i64(out_c), _i64(in_c),
                                                 _i64(group), _i64(zeros_w), _i64(sf_w), _ptr(wd), None))
    assert np.array_equal(wd.cpu().numpy().view(np.uint16), A.dequantize_gemv(qw, sc, qz, group, bf16=True))
    x = A.bf16_bits((rng.standard_normal((m, in_c)) * 0.5).astype(np.float32))
    y32 = torch.empty((m, out_c), dtype=torch.float32, device="cuda")
    y16 = torch.empty((m, out_c), dtype=torch.int16, device="cuda")
    dx = _dev(x.view(np.int16))
    _lib.check(gpu.ct2x_awq_gemv_bf16(_ptr(dx), _ptr(dq), _ptr(ds), _ptr(dz), _i64(m), _i64(in_c),
                                      _i64(out_c), _i64(group), _i64(zeros_w), _i64(sf_w), _ptr(y16),
                                      _ptr(y32), None))
    ref = A.gemv(x, qw, sc, qz, group, bf16=True)
    scale = np.abs(ref).max()
    assert np.abs(y32.cpu().numpy() - ref).max() <= 2e-5 * scale * max(1, in_c // 1024)
    assert np.array_equal(y16.cpu().numpy().view(np.uint16), A.bf16_bits(y32.cpu().numpy()))


@pytest.mark.parametrize("layout,group", [("gemm", 128), ("gemv", 128), ("gemv", 64)])
def test_awq_bf16_model_runs_like_its_bfloat16_twin(gpu, layout, group, tmp_path):
    """BASELINE configs[4]'s arithmetic (bfloat16 + AWQ), which the reference cannot run (float16
    only, src/ops/awq/gemm_gpu.cu:541; FLOAT16 forced, src/models/model.cc:754-757): an AWQ model
    loaded with compute type bfloat16 against the model holding exactly the bfloat16 weights the
    int4 values dequantise to in that arm - same teacher-forced logits up to the summation order of
    two GEMM kernels, same greedy tokens."""
    import ctranslate2_amd as ct2
    from ctranslate2_amd import model_builder as mb
    kw = dict(num_layers=2, d_model=256, ffn_dim=512, num_heads=4, num_heads_kv=2, vocab_size=320,
              seed=3, quantization="float32", w_std=0.05)
    d_awq = mb.build_decoder_lm(str(tmp_path / "awq"), awq=(layout, group, False), **kw)
    d_twin = mb.build_decoder_lm(str(tmp_path / "twin"), awq=(layout, group, "bf16"), **kw)
    a = ct2.Translator(d_awq, device="cuda", compute_type="bfloat16")
    b = ct2.Translator(d_twin, device="cuda", compute_type="bfloat16")
    rng = np.random.default_rng(0)
    tgt = rng.integers(4, 320, (3, 12)).astype(np.int32)
    la, lb = a.step_logits([], tgt), b.step_logits([], tgt)
    assert np.isfinite(la).all() and np.abs(la).max() > 0.1
    # two summation orders ahead of a rounding of the activations to bfloat16 (2^-9 relative, eight
    # times coarser than the float16 twin test's 2e-3): measured 3e-3 of the largest logit
    assert np.abs(la - lb).max() <= 1.5e-2 * np.abs(lb).max()
    prompts = [rng.integers(4, 320, n).tolist() for n in (5, 9, 3)]
    out_a, _ = a.generate_ids(prompts, max_length=10, min_length=10)
    out_b, _ = b.generate_ids(prompts, max_length=10, min_length=10)
    same = [x == y for x, y in zip(out_a, out_b)]
    assert sum(same) >= len(same) - 1  # (a near-tie may flip)


@pytest.mark.parametrize("layout,group,compute_type", [("gemm", 128, "float16"), ("gemv", 128, "float16"),
                                                       ("gemv", 64, "bfloat16"), ("gemm", 128, "bfloat16")])
def test_awq_under_tensor_parallelism(gpu, layout, group, compute_type, tmp_path):
    """AWQ weights cut for two tensor-parallel ranks (packed words, scales and zero points along the
    same axis a float weight is cut: columns on words, rows on quantisation groups; the GEMV
    layout's nibble-packed zero points re-packed for the local width) against the unsharded model:
    the ranks' fp32 partial products are summed in another order, so logits agree to a tolerance
    and greedy tokens with at most a near-tie flip. (BASELINE configs[4] is AWQ + TP; the reference
    has no AWQ in its ROCm build, src/layers/common.cc:402-404.)"""
    import threading
    import ctranslate2_amd as ct2
    from ctranslate2_amd import model_builder as mb
    kw = dict(num_layers=2, d_model=512, ffn_dim=1024, num_heads=8, num_heads_kv=4, vocab_size=320,
              seed=5, quantization="float32", w_std=0.05)
    d_awq = mb.build_decoder_lm(str(tmp_path / "awq"), awq=(layout, group, False), **kw)
    single = ct2.Translator(d_awq, device="cuda", compute_type=compute_type)
    rng = np.random.default_rng(1)
    prompts = [rng.integers(4, 320, n).tolist() for n in (5, 9, 3, 7)]
    ref, _ = single.generate_ids(prompts, max_length=10, min_length=10)
    tgt = rng.integers(4, 320, (2, 8)).astype(np.int32)
    ref_logits = single.step_logits([], tgt)
    ranks = ct2.Translator.tensor_parallel_local(d_awq, 2, compute_type=compute_type)
    outs = [None] * len(ranks)
    errs = []

    def work(i):
        try:
            # (no graph capture: the in-process communicator's collectives synchronise on the host)
            outs[i] = (ranks[i].generate_ids(prompts, max_length=10, min_length=10, use_graph=False)[0],
                       ranks[i].step_logits([], tgt))
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    for phase in range(1):
        ts = [threading.Thread(target=work, args=(i,)) for i in range(len(ranks))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    assert not errs, errs
    for toks, logits in outs:
        bound = 3e-3 if compute_type == "float16" else 1.5e-2  # (bfloat16 activations: 2^-9 steps)
        assert np.abs(logits - ref_logits).max() <= bound * np.abs(ref_logits).max()
        same = [a == b for a, b in zip(toks, ref)]
        assert sum(same) >= len(same) - 1
    for t in ranks:
        t.close()
