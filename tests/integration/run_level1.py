"""Level-1 integration runner (own process: the library exports the same names as the CPU
reference build). Loads integration/libctranslate2_mi355.so - the reference's OWN sources
(Translator, ReplicaPool, layers::*, decoding.cc, ops front-ends) compiled with -DCT2_WITH_CUDA
and linked against the stub translation units + libct2_mi355.so - and runs
ctranslate2::Translator(model, Device::CUDA, ComputeType) on it.

  python tests/integration/run_level1.py <model_dir> <compute_type> <json: sources, beam, max_len, min_len>
prints one JSON line: {"compute_type": ..., "hypotheses": [[ids]...], "scores": [...]}
"""
import json
import os
import sys

os.environ["CT2REF_VARIANT"] = "mi355"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import _refshim as R  # noqa: E402


def main():
    model_dir, compute_type, spec = sys.argv[1], sys.argv[2], json.loads(sys.argv[3])
    m = R.RefModel(model_dir, compute_type, 1, 0, device="cuda")
    out = {"compute_type": m.compute_type(), "runs": []}
    for run in spec["runs"]:
        if "prompts" in spec:  # decoder-only: Generator::generate_batch, include_prompt_in_result = false
            ids, scores = m.generate_ids(spec["prompts"], beam_size=run["beam"], max_len=run["max_len"],
                                         min_len=run["min_len"], return_scores=True)
            out["runs"].append({"beam": run["beam"], "hypotheses": ids, "scores": [float(s) for s in scores]})
            continue
        hyps, scores = m.translate_ids(spec["sources"], beam_size=run["beam"],
                                       max_len=run["max_len"], min_len=run["min_len"],
                                       length_penalty=1.0, num_hypotheses=1, return_scores=True)
        out["runs"].append({"beam": run["beam"], "hypotheses": [h[0] for h in hyps],
                            "scores": [float(s[0]) for s in scores]})
    m.close()
    # Level 1.5: run_translation calls the fused engine served (level15_mi355_stub.cc)
    import ctypes
    try:
        fn = ctypes.CDLL(None).ct2x_level15_served_calls
        fn.restype = ctypes.c_int64
        out["fused_calls"] = int(fn())
    except AttributeError:
        out["fused_calls"] = None
    print(json.dumps(out))


if __name__ == "__main__":
    main()
