"""The element-wise / layout members of primitives<Device::CUDA> (ctranslate2_amd/csrc/
primitives.hip; the reference's CPU bodies src/cpu/primitives.cc are the specification): integer
and float arithmetic is exact, the float functions are held to the oracle's activation functors
(which are pinned against the reference's vectors, tests/test_oracle.py) and to numpy."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu

DTYPES = {0: np.float32, 1: np.int8, 2: np.int16, 3: np.int32}


def _env():
    import torch
    import _gpuops as G
    from ctranslate2_amd import _lib
    return torch, G, _lib, _lib.lib()


def _rand(rng, shape, dtype):
    if dtype == np.float32:
        return rng.standard_normal(shape).astype(np.float32)
    return rng.integers(-50, 50, shape).astype(dtype)


@pytest.mark.parametrize("code", [0, 1, 2, 3])
def test_binary_vector_scalar_and_broadcast(gpu, code):
    torch, G, _lib, L = _env()
    dt = DTYPES[code]
    rng = np.random.default_rng(code)
    n = 70001
    a, b = _rand(rng, n, dt), _rand(rng, n, dt)
    ad, bd = G.dev(a), G.dev(b)      # named: a temporary would be freed (and reused) mid-call
    ops = {0: np.add, 1: np.subtract, 2: np.multiply, 3: np.minimum, 4: np.maximum}
    for op, fn in ops.items():
        c = torch.empty(n, dtype=getattr(torch, np.dtype(dt).name), device="cuda")
        _lib.check(L.ct2x_binary(op, _lib.ptr(ad), _lib.ptr(bd), _lib.ptr(c), G.i64(n), code, None))
        assert np.array_equal(G.host(c), fn(a, b).astype(dt)), op
        if op != 1:
            s = dt(3)
            _lib.check(L.ct2x_binary_scalar(op, C.c_double(float(s)), _lib.ptr(ad), _lib.ptr(c),
                                            G.i64(n), code, None))
            assert np.array_equal(G.host(c), fn(a, s).astype(dt)), op
    # in place, as the reference calls it: primitives::add(x, y, size) = add(x, y, y, size)
    acc = G.dev(b.copy())
    _lib.check(L.ct2x_binary(0, _lib.ptr(ad), _lib.ptr(acc), _lib.ptr(acc), G.i64(n), code, None))
    assert np.array_equal(G.host(acc), np.add(a, b).astype(dt))
    # broadcasts: batch (add, mul), depth, block
    a_size, reps = 37, 211
    av, bv = _rand(rng, a_size, dt), _rand(rng, a_size * reps, dt)
    avd, bvd = G.dev(av), G.dev(bv)
    out = torch.empty(a_size * reps, dtype=getattr(torch, np.dtype(dt).name), device="cuda")
    for op, fn in ((0, np.add), (2, np.multiply)):
        _lib.check(L.ct2x_broadcast(op, 0, _lib.ptr(avd), _lib.ptr(bvd), _lib.ptr(out),
                                    G.i64(a_size), G.i64(a_size * reps), G.i64(0), code, None))
        assert np.array_equal(G.host(out), fn(np.tile(av, reps), bv).astype(dt))
    _lib.check(L.ct2x_broadcast(0, 1, _lib.ptr(avd), _lib.ptr(bvd), _lib.ptr(out),
                                G.i64(a_size), G.i64(a_size * reps), G.i64(0), code, None))
    assert np.array_equal(G.host(out), (np.repeat(av, reps) + bv).astype(dt))
    block = 7                                   # add_block_broadcast: a[(i / block) % a_size]
    total = block * a_size * 5
    bv2 = _rand(rng, total, dt)
    bv2d = G.dev(bv2)
    out2 = torch.empty(total, dtype=getattr(torch, np.dtype(dt).name), device="cuda")
    _lib.check(L.ct2x_broadcast(0, 2, _lib.ptr(avd), _lib.ptr(bv2d), _lib.ptr(out2),
                                G.i64(a_size), G.i64(total), G.i64(block), code, None))
    idx = (np.arange(total) // block) % a_size
    assert np.array_equal(G.host(out2), (av[idx] + bv2).astype(dt))


@pytest.mark.parametrize("code", [0, 1, 2, 3])
def test_fills(gpu, code):
    torch, G, _lib, L = _env()
    dt = DTYPES[code]
    x = torch.zeros(1000, dtype=getattr(torch, np.dtype(dt).name), device="cuda")
    _lib.check(L.ct2x_fill(_lib.ptr(x), C.c_double(5), G.i64(1000), G.i64(1), code, None))
    assert np.all(G.host(x) == 5)
    _lib.check(L.ct2x_fill(_lib.ptr(x), C.c_double(-2), G.i64(334), G.i64(3), code, None))   # strided_fill
    want = np.full(1000, 5, dt)
    want[::3] = -2
    assert np.array_equal(G.host(x), want)
    idx = np.array([0, 2, 5, 999], np.int32)    # PrimitiveTest.IndexedFill pattern, primitives_test.cc:16
    idxd = G.dev(idx)
    _lib.check(L.ct2x_indexed_fill(_lib.ptr(x), C.c_double(1), _lib.ptr(idxd), G.i64(4), code, None))
    want[idx] = 1
    assert np.array_equal(G.host(x), want)


def test_unary_functions(gpu):
    torch, G, _lib, L = _env()
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(50000) * 3).astype(np.float32)
    y = torch.empty(x.size, dtype=torch.float32, device="cuda")
    xd = G.dev(x)
    for act in range(7):                        # ops::ActivationType order, against the oracle functors
        _lib.check(L.ct2x_unary(act, _lib.ptr(xd), _lib.ptr(y), G.i64(x.size), None))
        np.testing.assert_allclose(G.host(y), O.activation(x, act), rtol=2e-5, atol=2e-6, err_msg=str(act))
    for fn, ref, arg in ((7, np.exp, x), (8, np.log, np.abs(x) + 1e-3), (9, np.sin, x), (10, np.cos, x)):
        arg = arg.astype(np.float32)
        argd = G.dev(arg)
        _lib.check(L.ct2x_unary(fn, _lib.ptr(argd), _lib.ptr(y), G.i64(arg.size), None))
        np.testing.assert_allclose(G.host(y), ref(arg.astype(np.float64)), rtol=2e-6, atol=2e-6, err_msg=str(fn))
    # the reference's activation vectors (tests/ops_test.cc:1015-1095)
    import json
    kat = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ops_kat.json")))
    for name, a in kat["Activations"].items():
        xin = np.array(a["x"], np.float32)
        xind = G.dev(xin)
        yk = torch.empty(xin.size, dtype=torch.float32, device="cuda")
        _lib.check(L.ct2x_unary(a["act"], _lib.ptr(xind), _lib.ptr(yk), G.i64(xin.size), None))
        np.testing.assert_allclose(G.host(yk), np.array(a["y"], np.float32), atol=1e-5, rtol=0, err_msg=name)


@pytest.mark.parametrize("code", [0, 1, 2, 3])
def test_transposes(gpu, code):
    torch, G, _lib, L = _env()
    dt = DTYPES[code]
    rng = np.random.default_rng(code + 10)
    cases = [((5, 7), (1, 0)), ((3, 4, 5), (2, 0, 1)), ((3, 4, 5), (0, 2, 1)), ((3, 2, 3), (2, 1, 0)),
             ((2, 3, 4, 5), (0, 2, 1, 3)), ((2, 3, 4, 5), (3, 1, 0, 2)), ((32, 8, 6, 64), (0, 2, 1, 3))]
    for dims, perm in cases:
        a = _rand(rng, dims, dt)
        ad = G.dev(a)
        b = torch.empty(int(np.prod(dims)), dtype=getattr(torch, np.dtype(dt).name), device="cuda")
        d = (C.c_int64 * len(dims))(*dims)
        p = (C.c_int64 * len(perm))(*perm)
        _lib.check(L.ct2x_transpose(_lib.ptr(ad), d, p, len(dims), _lib.ptr(b), code, None))
        assert np.array_equal(G.host(b), np.transpose(a, perm).ravel()), (dims, perm)


def test_prepare_length_mask(gpu):
    torch, G, _lib, L = _env()
    lengths = np.array([3, 2, 7, 0], np.int32)
    heads, queries = 2, 5
    for mask_future in (0, 1):
        for multi_query in (0, 1):
            m = torch.empty(len(lengths) * heads * queries, dtype=torch.int32, device="cuda")
            ld = G.dev(lengths)
            _lib.check(L.ct2x_prepare_length_mask(_lib.ptr(ld), G.i64(len(lengths)), G.i64(heads),
                                                  G.i64(queries), mask_future, multi_query, _lib.ptr(m), None))
            want = np.empty((len(lengths), heads * queries), np.int32)
            for b, length in enumerate(lengths):   # src/cpu/primitives.cc:431-448
                for i in range(heads * queries):
                    q = (i // heads if multi_query else i % queries) + 1
                    want[b, i] = min(length, q) if mask_future else length
            assert np.array_equal(G.host(m), want.ravel()), (mask_future, multi_query)
