#!/usr/bin/env python
"""Time of ct2x_gemm_f16w (16-bit weights on the matrix cores, fp32 activations) at decode rows:
HIP events around 200 launches that walk a ring of distinct weight copies (> 512 MB in total, so
every launch streams its weight from HBM like a decode step does), per shape:

  python tools/gemm_h_bench.py [bf16|f16] m:n:k [m:n:k ...]

Prints us per launch and the weight stream rate. CT2X_HSKINNY_VARIANT=0 selects the round-2 form.
"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ctranslate2_amd import _lib  # noqa: E402

L = _lib.lib()
i64 = C.c_int64
P = lambda t: C.c_void_p(t.data_ptr())


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    shapes = [tuple(int(v) for v in s.split(":")) for s in sys.argv[2:]] or \
        [(40, 1280, 1280), (40, 3840, 1280), (40, 5120, 1280), (40, 1280, 5120), (40, 51866, 1280),
         (16, 4096, 4096), (8, 1280, 1280)]
    dt = torch.bfloat16 if kind == "bf16" else torch.float16
    code = 5 if kind == "bf16" else 4  # CT2X_BF16 / CT2X_F16 (DataType ids)
    dev = torch.device("cuda")
    for m, n, k in shapes:
        copies = max(2, int(600e6 // (n * k * 2)) + 1)
        copies = min(copies, 64)
        ws = [(torch.randn(n, k, device=dev) * 0.02).to(dt) for _ in range(copies)]
        x = torch.randn(m, k, device=dev)
        y = torch.empty(m, n, device=dev)
        bias = torch.zeros(n, device=dev)
        iters = 200

        def run(i):
            _lib.check(L.ct2x_gemm_f16w(P(x), P(ws[i % copies]), code, i64(m), i64(n), i64(k), P(bias),
                                        None, -1, P(y), None))

        torch.cuda.synchronize()  # the operands were made on torch's stream, the GEMMs run on ours
        for i in range(20):
            run(i)
        L.ct2x_stream_synchronize(None)
        ref = (x @ ws[19 % copies].float().t())
        err = float((y - ref).abs().max() / ref.abs().max())
        e0 = C.c_void_p()
        e1 = C.c_void_p()
        L.ct2x_event_create(C.byref(e0))
        L.ct2x_event_create(C.byref(e1))
        L.ct2x_event_record(e0, None)
        for i in range(iters):
            run(i)
        L.ct2x_event_record(e1, None)
        ms = C.c_float()
        L.ct2x_event_elapsed_ms(e0, e1, C.byref(ms))
        us = ms.value * 1000 / iters
        print("%s [%d x %d x %d]: %.2f us per launch, %.2f TB/s of weights, rel err %.2e"
              % (kind, m, n, k, us, n * k * 2 / us / 1e6, err))


if __name__ == "__main__":
    main()
