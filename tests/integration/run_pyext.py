"""Runner for the reference's OWN Python extension on this backend (own process: the module links
integration/libctranslate2_mi355.so, which exports the same names as the CPU reference build).

`oracle/_ref/pyext/_ext*.so` is python/cpp/*.cc of the reference (pybind11: Translator, Generator,
StorageView ...) compiled where it lies (`make -C oracle pyext`) and linked with the Level-1
library = the reference's host sources + the stub translation units + libct2_mi355.so. It is the
module the reference's `ctranslate2` package imports as `_ext`; here it is loaded directly.

  python tests/integration/run_pyext.py '<json spec>'
spec: {"model": dir, "device": "cuda"|"cpu", "compute_type": ..., "tokens": [[str]], "beams": [1, 4],
       "max_decoding_length": n}
prints one JSON line: {"device_count": n, "runs": [{"beam": b, "hypotheses": [[str]], "scores": [...]}]}
"""
import ctypes
import glob
import importlib.util
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def load_ext():
    # the reference only uses MKL's int8 GEMM on CPUs it identifies as Intel unless told otherwise
    # (src/cpu/backend.cc); the GPU box's host is an AMD EPYC
    os.environ.setdefault("CT2_USE_MKL", "1")
    found = glob.glob(os.path.join(ROOT, "oracle", "_ref", "pyext", "_ext*.so"))
    if not found:
        raise SystemExit("oracle/_ref/pyext/_ext*.so is not built (make -C oracle pyext)")
    # lazy binding: the 16-bit arms of the Level-1 library are unresolved functions that an
    # int8 / float32 model never calls
    sys.setdlopenflags(os.RTLD_LAZY | os.RTLD_GLOBAL)
    ctypes.CDLL(os.path.join(ROOT, "ctranslate2_amd", "libct2_mi355.so"), mode=os.RTLD_LAZY | os.RTLD_GLOBAL)
    spec = importlib.util.spec_from_file_location("_ext", found[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    spec = json.loads(sys.argv[1])
    ext = load_ext()
    out = {"device_count": ext.get_cuda_device_count(),
           "compute_types": sorted(ext.get_supported_compute_types(spec["device"])), "runs": []}
    if "gen_cases" in spec:
        # ctranslate2.Generator.generate_batch (include_prompt_in_result=False unless the case sets
        # it; Level 1.5: Mi355DecoderReplica::run_generation)
        g = ext.Generator(spec["model"], device=spec["device"], compute_type=spec["compute_type"])
        for case in spec["gen_cases"]:
            kw = dict(case)
            name = kw.pop("name")
            steps = []
            if kw.pop("callback", False):
                kw["callback"] = lambda r: steps.append([r.batch_id, r.step, r.token, r.is_last]) and False
            kw.setdefault("include_prompt_in_result", False)
            tokens = kw.pop("tokens", spec["tokens"])
            res = g.generate_batch(tokens, **kw)
            run = {"name": name, "sequences": [r.sequences for r in res],
                   "scores": [[float(x) for x in r.scores] for r in res], "steps": steps}
            if kw.get("return_logits_vocab"):
                import numpy as np
                run["logits"] = [[[np.array(l).astype(float).tolist() for l in hyp] for hyp in r.logits] for r in res]
            out["runs"].append(run)
        lib = ctypes.CDLL(os.path.join(ROOT, "integration", "libctranslate2_mi355.so"), mode=os.RTLD_LAZY | os.RTLD_GLOBAL)
        lib.ct2x_level15_served_calls.restype = ctypes.c_int64
        out["served_calls"] = int(lib.ct2x_level15_served_calls())
        print(json.dumps(out))
        return
    t = ext.Translator(spec["model"], device=spec["device"], compute_type=spec["compute_type"])
    # "cases": keyword sets of Translator.translate_batch (Level 1.5: which of them the fused engine
    # serves). A case may carry "target_prefix" and "callback": true (a Python callable collecting the
    # step results).
    for case in spec.get("cases", []):
        kw = dict(case)
        name = kw.pop("name")
        prefix = kw.pop("target_prefix", None)
        steps = []
        if kw.pop("callback", False):
            kw["callback"] = lambda r: steps.append([r.batch_id, r.step, r.token, r.is_last]) and False
        res = t.translate_batch(spec["tokens"], prefix, **kw)
        run = {"name": name, "hypotheses": [r.hypotheses for r in res],
               "scores": [[float(x) for x in r.scores] for r in res], "steps": steps}
        if kw.get("return_attention"):
            run["attention"] = [[[[float(x) for x in vec] for vec in hyp] for hyp in r.attention] for r in res]
        if kw.get("return_logits_vocab"):
            import numpy as np
            run["logits_shapes"] = [[[list(np.array(l).shape) for l in hyp] for hyp in r.logits] for r in res]
        out["runs"].append(run)
    lib = ctypes.CDLL(os.path.join(ROOT, "integration", "libctranslate2_mi355.so"), mode=os.RTLD_LAZY | os.RTLD_GLOBAL)
    lib.ct2x_level15_served_calls.restype = ctypes.c_int64
    out["served_calls"] = int(lib.ct2x_level15_served_calls())
    for beam in spec.get("beams", []):
        res = t.translate_batch(spec["tokens"], beam_size=beam, return_scores=True,
                                max_decoding_length=spec.get("max_decoding_length", 256))
        out["runs"].append({"beam": beam, "hypotheses": [r.hypotheses[0] for r in res],
                            "scores": [float(r.scores[0]) for r in res]})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
