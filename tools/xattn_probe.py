#!/usr/bin/env python
"""Probe of the 16-bit encoder-decoder attention launch (attention_cross_h.hip) at the
Whisper-large-v3 decode shape - 8 windows x beam 5, 20 heads of 64, 1500 memory positions,
bfloat16 memory - with the memory of a (window, head) cut into 1 .. 8 chunks (CT2X_XATTN_SPLITS, read
by a `make TUNING=1` library only: CT2X_LIBRARY points at it), and the phase timeline of workgroup 0
(shader-clock stamps, TUNING builds). One line per variant: us per launch over 200 launches."""
# The TUNING library is built next to the shipped one (it never replaces it):
#   make -C ctranslate2_amd/csrc all TUNING=1 OBJDIR=build_tuning OUT=../libct2_mi355_tuning.so
#   CT2X_LIBRARY=$PWD/ctranslate2_amd/libct2_mi355_tuning.so python tools/<this file>
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    from ctranslate2_amd import _lib
    L = _lib.lib()
    batch, beam, heads, s_max, dh = 8, 5, 20, 1500, 64
    rng = np.random.default_rng(0)
    q = torch.from_numpy(rng.standard_normal((batch * beam, heads, dh)).astype(np.float32)).cuda()
    kv = torch.from_numpy(rng.standard_normal((batch, s_max, 2, heads, dh)).astype(np.float32)).cuda()
    kv16 = kv.to(torch.bfloat16).contiguous()
    out = torch.empty_like(q)
    L.ct2x_cross_attention_split_scratch_bytes.restype = C.c_int64
    i64 = C.c_int64
    nbytes = int(L.ct2x_cross_attention_split_scratch_bytes(i64(batch), i64(beam), i64(heads), i64(dh), i64(s_max)))
    scratch = torch.zeros((nbytes,), dtype=torch.uint8, device="cuda")
    ptr = lambda t: C.c_void_p(t.data_ptr())
    have_stamps = hasattr(L, "ct2x_tuning_xattn_stamps")

    def launch(split):
        if split:
            rc = L.ct2x_cross_attention_hd_split(ptr(q), ptr(kv16), C.c_int(5), None, i64(batch), i64(beam),
                                                 i64(heads), i64(heads), i64(dh), i64(s_max),
                                                 C.c_float(0.125), ptr(out), ptr(scratch), i64(nbytes), None)
        else:
            rc = L.ct2x_cross_attention_hd(ptr(q), ptr(kv16), C.c_int(5), None, i64(batch), i64(beam),
                                           i64(heads), i64(heads), i64(dh), i64(s_max), C.c_float(0.125),
                                           ptr(out), None)
        assert rc == 0, L.ct2x_last_error()

    ref = None
    for splits in (0, 1, 2, 3):
        if splits:
            os.environ["CT2X_XATTN_SPLITS"] = str(splits)
        for _ in range(5):
            launch(splits > 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            launch(splits > 0)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 200 * 1e6
        y = out.cpu().numpy().copy()
        if ref is None:
            ref = y
        line = "splits %d%s: %.1f us per launch, max |diff| to the one-workgroup form %.2e" % (
            splits, " (no scratch)" if splits == 0 else "", us, float(np.abs(y - ref).max()))
        if have_stamps:
            st = (C.c_ulonglong * 16)()
            L.ct2x_tuning_xattn_stamps(st)
            v = [int(x) for x in st]
            names = ["q", "scores", "max", "exp+sum", "context", "fold", "out/publish", "ticket", "merge"]
            seg = ["%s %d" % (names[i], v[i + 1] - v[i]) for i in range(0, 7)]
            if splits > 1:
                seg += ["ticket..merge(last wg) %d" % (v[9] - v[8])]
            line += "\n    workgroup 0, shader clocks: " + ", ".join(seg)
        print(line, flush=True)


if __name__ == "__main__":
    main()
