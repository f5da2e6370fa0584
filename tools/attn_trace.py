#!/usr/bin/env python
"""Per-phase s_memtime stamps of attention_decode2 (workgroup 0) for one launch shape.
  python tools/attn_trace.py rows heads heads_kv dhead t_max length"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ctranslate2_amd import _lib  # noqa: E402

L = _lib.lib()
rows, heads, hkv, dh, tmax, length = [int(x) for x in sys.argv[1:7]]
dev = torch.device("cuda")
q = torch.randn(rows, heads * dh, device=dev)
k = torch.randn(rows, hkv, tmax, dh, device=dev)
v = torch.randn(rows, hkv, tmax, dh, device=dev)
out = torch.empty(rows, heads * dh, device=dev)
trace = torch.zeros(16 * 16, dtype=torch.int64, device=dev)
L.ct2x_debug_set_attn_trace.argtypes = [C.c_void_p]
L.ct2x_debug_set_attn_trace.restype = None
names = ["start", "len known+prefetch issued", "prologue done", "barrier", "QK done", "softmax done",
         "PV done", "reduced"]
for rep in range(3):
    L.ct2x_debug_set_attn_trace(C.c_void_p(trace.data_ptr()))
    rc = L.ct2x_attention_decode(C.c_void_p(q.data_ptr()), C.c_void_p(k.data_ptr()), C.c_void_p(v.data_ptr()),
                                 None, C.c_int64(length), C.c_int64(rows), C.c_int64(heads), C.c_int64(hkv),
                                 C.c_int64(dh), C.c_int64(tmax), C.c_int64(1), C.c_float(dh ** -0.5),
                                 C.c_void_p(out.data_ptr()), None)
    assert rc == 0
    torch.cuda.synchronize()
    L.ct2x_debug_set_attn_trace(None)
t = trace.cpu().numpy().reshape(16, 16)
for w in range(4):
    if t[w, 0] == 0:
        continue
    base = t[0, 0]
    print("wave %d:" % w, "  ".join("%s=%d" % (names[i], t[w, i] - base) for i in range(8) if t[w, i]))
