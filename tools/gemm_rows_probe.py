#!/usr/bin/env python
"""Probe of the decode-row 16-bit GEMM (gemm_h.hip, gemm_h_rows_kernel with the K slices summed in
the launch) at the Whisper-large-v3 decode shapes (40 rows; bfloat16 weights): wall time per call
through ct2x_gemm_f16w (which also allocates and clears its workspace: an upper bound) and the phase
timeline of workgroup (0, 0) / of column tile 0's last arrival in shader clocks - a `make TUNING=1`
library (CT2X_LIBRARY)."""
# The TUNING library is built next to the shipped one (it never replaces it):
#   make -C ctranslate2_amd/csrc all TUNING=1 OBJDIR=build_tuning OUT=../libct2_mi355_tuning.so
#   CT2X_LIBRARY=$PWD/ctranslate2_amd/libct2_mi355_tuning.so python tools/<this file>
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    from ctranslate2_amd import _lib
    L = _lib.lib()
    have = hasattr(L, "ct2x_tuning_gemm_rows_stamps")
    i64 = C.c_int64
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    for (m, n, k) in ((40, 1280, 1280), (40, 3840, 1280), (40, 5120, 1280), (40, 1280, 5120)):
        a = torch.randn((m, k), device="cuda")
        w = (torch.randn((n, k), device="cuda") * 0.05).to(torch.bfloat16).contiguous()
        bias = torch.randn((n,), device="cuda")
        res = torch.randn((m, n), device="cuda")
        y = torch.empty((m, n), device="cuda")

        def launch():
            rc = L.ct2x_gemm_f16w(ptr(a), ptr(w), C.c_int(5), i64(m), i64(n), i64(k), ptr(bias), ptr(res),
                                  C.c_int(-1), ptr(y), None)
            assert rc == 0

        for _ in range(5):
            launch()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            launch()
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 200 * 1e6
        line = "[%d x %d] x [%d x %d]^T: %.1f us per call (with the workspace memset)" % (m, k, n, k, us)
        if have:
            st = (C.c_ulonglong * 16)()
            L.ct2x_tuning_gemm_rows_stamps(st)
            v = [int(x) for x in st]
            line += ("\n    workgroup (0, 0), shader clocks: loads + staging %d, barrier %d, MFMA %d, publish %d, "
                     "drain + ticket %d; last arrival of column tile 0: partial loads + sum + epilogue %d"
                     % (v[1] - v[0], v[2] - v[1], v[3] - v[2], v[4] - v[3], v[5] - v[4], v[7] - v[6]))
        print(line, flush=True)


if __name__ == "__main__":
    main()
