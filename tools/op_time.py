#!/usr/bin/env python
"""GPU-side time of the fused norm+GEMM launch vs the two-launch path (HIP events, 300 launches)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ctranslate2_amd import _lib  # noqa: E402

L = _lib.lib()
m, n, k = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else (32, 512, 512)
dev = torch.device("cuda")
x = torch.randn(m, k, device=dev)
g = torch.ones(k, device=dev)
b = torch.zeros(k, device=dev)
w = torch.randint(-127, 128, (n, k), dtype=torch.int8, device=dev)
ws = torch.full((n,), 300.0, device=dev)
bias = torch.zeros(n, device=dev)
y = torch.empty(m, n, device=dev)
q = torch.empty(m, k, dtype=torch.int8, device=dev)
s = torch.empty(m, device=dev)
i64 = C.c_int64
P = lambda t: C.c_void_p(t.data_ptr())


def fused():
    L.ct2x_norm_gemm_s8_dequant(P(x), P(g), P(b), C.c_float(1e-5), 0, 3, P(w), P(ws), P(bias), None,
                                i64(m), i64(n), i64(k), -1, P(y), None)


def split():
    L.ct2x_norm_quantize_s8(P(x), P(g), P(b), i64(m), i64(k), C.c_float(1e-5), 0, 3, None, P(q), P(s), None)
    L.ct2x_gemm_s8_dequant(P(q), P(s), P(w), P(ws), P(bias), None, i64(m), i64(n), i64(k), -1, P(y), 0, None)


stream = torch.cuda.current_stream()
for name, fn in (("fused", fused), ("split", split)):
    for _ in range(20):
        fn()
    L.ct2x_stream_synchronize(None)
    e0 = C.c_void_p(); e1 = C.c_void_p()
    L.ct2x_event_create(C.byref(e0)); L.ct2x_event_create(C.byref(e1))
    L.ct2x_event_record(e0, None)
    for _ in range(300):
        fn()
    L.ct2x_event_record(e1, None)
    ms = C.c_float()
    L.ct2x_event_elapsed_ms(e0, e1, C.byref(ms))
    print("%s [%d,%d,%d]: %.2f us per Dense" % (name, m, n, k, ms.value * 1000 / 300))
