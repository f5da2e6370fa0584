"""CPU checks of the tensor-parallel design (no GPU): the partition the library uses
(ct2x_tp_shard_rows, host-only) and, with the oracle, the claim the exchange protocol rests on -
max-reducing the per-rank row absmax and sum-reducing the exact int32 partial products reproduces
the unsharded int8 Dense bit for bit (reference: all-gather + re-quantise + per-rank float
outputs + ReduceAll, src/layers/common.cc:348-401)."""
import ctypes as C

import numpy as np
import pytest

import _oracle as O
from ctranslate2_amd import _lib


def shard_rows(n, blocks, rank, world):
    L = _lib.lib()
    nb = len(blocks)
    arr = (C.c_int64 * max(nb, 1))(*blocks) if nb else None
    first = (C.c_int64 * max(nb, 1))()
    count = (C.c_int64 * max(nb, 1))()
    _lib.check(L.ct2x_tp_shard_rows(C.c_int64(n), arr, C.c_int32(nb), C.c_int32(rank),
                                    C.c_int32(world), first, count))
    return [(first[i], count[i]) for i in range(max(nb, 1))]


def test_partition_covers_every_row_once():
    # Llama-3-8B fused QKV: 32 query heads, 8 KV heads, d_head 128
    q, kv = 32 * 128, 8 * 128
    n = q + 2 * kv
    for world in (1, 2, 4, 8):
        seen = np.zeros(n, np.int32)
        for rank in range(world):
            segs = shard_rows(n, [q, kv, kv], rank, world)
            assert [c for _, c in segs] == [q // world, kv // world, kv // world]
            for first, count in segs:
                seen[first:first + count] += 1
        assert np.all(seen == 1)
    assert shard_rows(2048, [], 1, 4) == [(512, 512)]
    with pytest.raises(_lib.Ct2xError, match="not divisible"):
        shard_rows(n, [q, kv, kv], 0, 3)


@pytest.mark.parametrize("world", [2, 4])
def test_row_parallel_int8_dense_is_exact(world):
    rng = np.random.default_rng(world)
    m, k, n = 7, 256, 96
    x = rng.standard_normal((m, k)).astype(np.float32)
    wq, ws = O.quantize_s8((rng.standard_normal((n, k)) * 0.02).astype(np.float32), True)
    bias = (rng.standard_normal(n) * 0.01).astype(np.float32)
    res = rng.standard_normal((m, n)).astype(np.float32)
    ref = O.dense_s8(x, wq, ws, bias, res, -1, True)

    kl = k // world
    slices = [x[:, r * kl:(r + 1) * kl] for r in range(world)]
    amax = np.max(np.stack([np.abs(s).max(axis=1) for s in slices]), axis=0)  # all-reduce MAX
    scale = np.where(amax != 0, np.float32(127.0) / amax, np.float32(1.0)).astype(np.float32)
    acc = np.zeros((m, n), np.int64)
    for r in range(world):
        q, s = O.quantize_s8(np.concatenate([slices[r], amax[:, None]], axis=1), True)
        assert np.array_equal(s, scale)          # global scale on every rank
        acc += O.gemm_s8(q[:, :kl], np.ascontiguousarray(wq[:, r * kl:(r + 1) * kl]))  # all-reduce SUM
    y = O.dequantize_gemm_output(acc.astype(np.int32), scale, ws, bias, -1) + 0
    y = res + y
    assert np.array_equal(y, ref)
