#!/usr/bin/env python
"""Decode-row AWQ int4 GEMM / GEMV on one MI355X: microseconds per launch and the weight stream
rate against the HBM roofline (algorithmic bytes = in_c * out_c / 2 of packed weights + scales +
zeros + activations + result; weights rotated through > 512 MB so the 256 MB MALL cannot hold them).
  python tools/awq_bench.py [m] [bf16]      (bf16: the bfloat16 arm, round 6)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from ctranslate2_amd import _lib  # noqa: E402

L = _lib.lib()
m = int(sys.argv[1]) if len(sys.argv) > 1 else 16
BF = len(sys.argv) > 2 and sys.argv[2] == "bf16"
DT = torch.bfloat16 if BF else torch.float16
GEMM = L.ct2x_awq_gemm_bf16 if BF else L.ct2x_awq_gemm
GEMV = L.ct2x_awq_gemv_bf16 if BF else L.ct2x_awq_gemv
p = lambda t: C.c_void_p(t.data_ptr())
i64 = lambda v: C.c_int64(int(v))
for in_c, out_c in [(4096, 4096), (4096, 14336), (14336, 4096), (4096, 128256)]:
    g = 128
    wbytes = in_c * out_c // 2
    copies = (600 << 20) // wbytes + 1
    x = (torch.randn(m, in_c, device="cuda") * 0.5).to(DT)
    y = torch.empty(m, out_c, dtype=DT, device="cuda")
    for layout in ("gemm", "gemv"):
        if layout == "gemm":
            qw = torch.randint(-2**31, 2**31 - 1, (copies, in_c, out_c // 8), dtype=torch.int32, device="cuda")
            sc = (torch.rand(in_c // g, out_c, device="cuda") * 0.02 + 0.001).to(DT)
            qz = torch.randint(-2**31, 2**31 - 1, (in_c // g, out_c // 8), dtype=torch.int32, device="cuda")
            call = lambda i: GEMM(p(x), p(qw[i % copies]), p(sc), p(qz), i64(m), i64(in_c), i64(out_c),
                                             i64(g), p(y), None, None)
        else:
            zeros_w, sf_w = (in_c // g + 7) // 8, (in_c // g + 7) // 8 * 8
            qw = torch.randint(-2**31, 2**31 - 1, (copies, out_c, in_c // 8), dtype=torch.int32, device="cuda")
            sc = (torch.rand(out_c, sf_w, device="cuda") * 0.02 + 0.001).to(DT)
            qz = torch.randint(-2**31, 2**31 - 1, (out_c, zeros_w), dtype=torch.int32, device="cuda")
            call = lambda i: GEMV(p(x), p(qw[i % copies]), p(sc), p(qz), i64(m), i64(in_c), i64(out_c),
                                             i64(g), i64(zeros_w), i64(sf_w), p(y), None, None)
        stream = torch.cuda.current_stream()
        for i in range(5):
            assert call(i) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 200
        # the library launches on its own thread stream: bracket with device-wide syncs
        torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()
        for i in range(n):
            call(i)
        _lib.check(L.ct2x_device_synchronize())
        dt = (time.perf_counter() - t0) / n
        bytes_ = wbytes + sc.numel() * 2 + qz.numel() * 4 + x.numel() * 2 + y.numel() * 2
        print("awq %s%s m=%d in_c=%d out_c=%d: %.2f us/launch  %.0f GB/s (%.1f%% of 8 TB/s)" %
              (layout, " bf16" if BF else "", m, in_c, out_c, dt * 1e6, bytes_ / dt / 1e9, bytes_ / dt / 8e12 * 100))
        del qw
