"""The reference's own pybind11 module (python/cpp/*.cc -> `_ext`) built against the Level-1
library (oracle/Makefile `pyext`): SURVEY 8(f2). Run in a subprocess (tests/integration/
run_pyext.py). The CPU test proves the module and the library load and translate; the GPU test runs
`ctranslate2._ext.Translator(model, device="cuda")` = the reference's Python API on this backend."""
import glob
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
RUNNER = os.path.join(HERE, "integration", "run_pyext.py")
AREN = os.path.join(HERE, "golden", "models", "aren-transliteration-i8")
# tests/python/test_translator.py of the reference translates this fixture: "آتزمون" -> "atzmon"
AREN_SRC = [["آ", "ت", "ز", "م", "و", "ن"], ["آ", "ت", "ش", "ي", "س", "و", "ن"]]


def _built():
    return bool(glob.glob(os.path.join(ROOT, "oracle", "_ref", "pyext", "_ext*.so")))


def _run(spec):
    p = subprocess.run([sys.executable, RUNNER, json.dumps(spec)], capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads(p.stdout.strip().splitlines()[-1])


@pytest.mark.skipif(not _built(), reason="oracle/_ref/pyext is built where /root/reference exists")
def test_reference_python_extension_loads_and_translates_on_cpu():
    out = _run({"model": AREN, "device": "cpu", "compute_type": "int8", "tokens": AREN_SRC,
                "beams": [1, 4]})
    for run in out["runs"]:
        assert run["hypotheses"][0] == ["a", "t", "z", "m", "o", "n"]


@pytest.mark.gpu
@pytest.mark.skipif(not _built(), reason="oracle/_ref/pyext is built where /root/reference exists")
@pytest.mark.parametrize("compute_type", ["int8", "float32", "float16", "bfloat16"])
def test_reference_python_extension_on_this_backend(gpu, compute_type):
    """float16 / bfloat16: the reference's front-end keeps 16-bit activations end to end and every op
    goes through the 16-bit arms of the stubs (tests/integration/half_mi355_stub.cc)."""
    model = AREN if compute_type == "int8" else AREN.replace("-i8", "")
    if compute_type in ("float16", "bfloat16"):
        spec = {"model": model, "device": "cuda", "compute_type": compute_type, "tokens": AREN_SRC,
                "beams": [1, 4]}
        dev = _run(spec)
        assert compute_type in dev["compute_types"]
        for run in dev["runs"]:
            assert run["hypotheses"][0] == ["a", "t", "z", "m", "o", "n"]
        return
    spec = {"model": model, "device": "cuda", "compute_type": compute_type, "tokens": AREN_SRC,
            "beams": [1, 4]}
    dev = _run(spec)
    cpu = _run(dict(spec, device="cpu"))
    assert dev["device_count"] >= 1
    assert compute_type in dev["compute_types"]
    for d, c in zip(dev["runs"], cpu["runs"]):
        assert d["hypotheses"] == c["hypotheses"]
        assert d["hypotheses"][0] == ["a", "t", "z", "m", "o", "n"]
        # (the reference's Device::CUDA front-end is not its Device::CPU front-end op for op - no
        # u8 compensation term, unfused attention ops - so the int8 scores agree to quantisation
        # noise, the float32 ones to rounding)
        tol = 2e-2 if compute_type == "int8" else 1e-4
        for a, b in zip(d["scores"], c["scores"]):
            assert abs(a - b) < tol
