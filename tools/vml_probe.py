"""Does MKL VML on THIS host reproduce the authoring host's sinusoidal position table
(tests/golden/tap_golden.npz postable/<depth>)? Prints the fraction of identical entries for
MKL vsSin / vsCos and for libm sinf / cosf."""
import ctypes as C
import glob
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
z = np.load(os.path.join(HERE, "..", "tests", "golden", "tap_golden.npz"))
libm = C.CDLL("libm.so.6")
for f in ("logf", "expf", "sinf", "cosf"):
    getattr(libm, f).restype = C.c_float
mode = os.RTLD_GLOBAL | os.RTLD_LAZY
mk = None
for d in ("/opt/conda/lib", os.path.join(HERE, "..", "oracle", "_ref", "mkl")):
    try:
        C.CDLL("libgomp.so.1", mode=mode)
        for name in ("libmkl_core.so", "libmkl_gnu_thread.so", "libmkl_intel_ilp64.so"):
            mk = C.CDLL(sorted(glob.glob(os.path.join(d, name + "*")))[0], mode=mode)
        break
    except Exception as e:  # noqa
        mk = None
print("mkl loaded:", mk is not None, open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0])
for depth in (64, 256, 512):
    want = z["postable/%d" % depth]
    P, half = want.shape[0], depth // 2
    inc = np.float32(libm.logf(C.c_float(10000.0))) / np.float32(half - 1)
    ts = np.array([libm.expf(C.c_float(float(np.float32(-inc) * np.float32(i)))) for i in range(half)], np.float32)
    st = (np.arange(1, P + 1, dtype=np.float32)[:, None] * ts[None, :]).astype(np.float32)
    s_f = np.array([libm.sinf(C.c_float(float(v))) for v in st.ravel()], np.float32).reshape(P, half)
    c_f = np.array([libm.cosf(C.c_float(float(v))) for v in st.ravel()], np.float32).reshape(P, half)
    print(depth, "libm identical:", float((np.concatenate([s_f, c_f], 1) == want).mean()))
    if mk is not None:
        x = np.ascontiguousarray(st.ravel())
        ys, yc = np.empty_like(x), np.empty_like(x)
        mk.vsSin(C.c_longlong(x.size), x.ctypes.data_as(C.c_void_p), ys.ctypes.data_as(C.c_void_p))
        mk.vsCos(C.c_longlong(x.size), x.ctypes.data_as(C.c_void_p), yc.ctypes.data_as(C.c_void_p))
        got = np.concatenate([ys.reshape(P, half), yc.reshape(P, half)], 1)
        print(depth, "MKL VML identical:", float((got == want).mean()))
