"""Dev-time script (needs /root/reference and oracle/_ref built): generates the MODEL-LEVEL golden
fixtures committed under tests/golden/ by running the REAL reference CPU library
(oracle/_ref, Translator / layers::TransformerDecoder on CPU, compute type int8 -> int8_float32)
on (a) the reference's own trained fixture tests/data/models/v2/aren-transliteration{,-i8}
(copied byte-for-byte into tests/golden/models/ -- binary test data, not source) and (b) the
synthetic models written by ctranslate2_amd/model_builder.py from fixed seeds.

The authoring host has AVX512-VNNI, so the reference's MKL u8s8 GEMM is exact here; the GPU box's
host CPU (AMD EPYC, MKL AVX2 path) may saturate, which is why the fixtures are generated here and
committed rather than recomputed there.
"""
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import _refshim as R  # noqa: E402
from ctranslate2_amd import model_builder as mb  # noqa: E402
import golden_cases as GC  # noqa: E402

REF = "/root/reference"


def copy_fixture(name):
    src = os.path.join(REF, "tests/data/models/v2", name)
    dst = os.path.join(HERE, "models", name)
    os.makedirs(dst, exist_ok=True)
    for f in os.listdir(src):
        shutil.copyfile(os.path.join(src, f), os.path.join(dst, f))
    return dst


def read_vocab(path):
    with open(path, encoding="utf-8") as f:
        tokens = [line.rstrip("\n") for line in f]
    if "<unk>" not in tokens:  # Vocabulary ctor appends the unknown token (src/vocabulary.cc:45-47)
        tokens.append("<unk>")
    return tokens


def run_case(model_dir, compute_type, sources, vocab_size, add_eos, beams, max_len, min_len,
             logits_steps, tmp_note):
    m = R.RefModel(model_dir, compute_type, 1, 0)
    out = {"compute_type": m.compute_type(), "sources": sources, "max_len": max_len,
           "min_len": min_len, "note": tmp_note, "runs": {}}
    for beam in beams:
        hyps, scores = m.translate_ids(sources, beam_size=beam, max_len=max_len, min_len=min_len,
                                       length_penalty=1.0, num_hypotheses=min(beam, 2),
                                       return_scores=True)
        out["runs"]["beam%d" % beam] = {"hypotheses": hyps, "scores": scores.tolist()}
    # Teacher-forced logits on the greedy output: summarised (argmax, top-2 values, sampled cols).
    greedy = [h[0] for h in out["runs"]["beam1"]["hypotheses"]]
    T = min(logits_steps, min(len(g) for g in greedy) + 1)
    tgt_in = np.array([[1] + g[:T - 1] for g in greedy], np.int32)  # <s> + previous tokens
    src_full = [s + ([2] if add_eos else []) for s in sources]
    logits = m.step_logits(src_full, tgt_in, vocab_size)  # [T, B, V]
    rng = np.random.default_rng(1234)
    cols = np.sort(rng.choice(vocab_size, size=min(16, vocab_size), replace=False))
    top2 = np.sort(logits, axis=-1)[..., -2:][..., ::-1]
    out["teacher_forced"] = {
        "tgt_in": tgt_in.tolist(),
        "argmax": logits.argmax(-1).tolist(),
        "top2": top2.tolist(),
        "cols": cols.tolist(),
        "values": logits[..., cols].tolist(),
    }
    m.close()
    return out


def main():
    golden = {}
    # (a) trained fixture: tests/translator_test.cc:53-97 expects "a t z m o n".
    for name in ("aren-transliteration-i8", "aren-transliteration"):
        d = copy_fixture(name)
        sv = read_vocab(os.path.join(d, "source_vocabulary.txt"))
        tv = read_vocab(os.path.join(d, "target_vocabulary.txt"))
        sources = [[sv.index(t) for t in words] for words in GC.AREN_SOURCES]
        ct = "int8" if name.endswith("i8") else "float32"
        golden[name] = run_case(d, ct, sources, len(tv), False, [1, 2, 4], 30, 1, 8,
                                "reference fixture; expected first output: a t z m o n")
        golden[name]["first_output_tokens"] = [tv[i] for i in golden[name]["runs"]["beam1"]["hypotheses"][0][0]]
    # (b) synthetic models.
    for key, cfg in GC.SYNTHETIC.items():
        d = "/tmp/ct2x_golden_%s" % key
        mb.build_transformer(d, **cfg["model"])
        rng = np.random.default_rng(cfg["data_seed"])
        V = cfg["model"]["vocab_size"]
        sources = [rng.integers(4, V, n).tolist() for n in cfg["src_lengths"]]
        golden[key] = run_case(d, "int8", sources, V, False, cfg["beams"], cfg["max_len"],
                               cfg["min_len"], cfg["logits_steps"], "synthetic " + json.dumps(cfg["model"]))
        # Same model directory run with compute_type=float32 (weights dequantised at load,
        # Model::ensure_dtype): the float pipeline without activation re-quantisation.
        golden[key + "_f32"] = run_case(d, "float32", sources, V, False, [1], cfg["max_len"],
                                        cfg["min_len"], min(cfg["logits_steps"], 16),
                                        "float32 compute on the int8 model " + key)
    # (c) synthetic decoder-only models through Generator::generate_batch
    # (include_prompt_in_result=False, greedy) + teacher-forced logits over prompt + continuation.
    for key, cfg in GC.SYNTHETIC_LM.items():
        d = "/tmp/ct2x_golden_%s" % key
        mb.build_decoder_lm(d, **cfg["model"])
        rng = np.random.default_rng(cfg["data_seed"])
        V = cfg["model"]["vocab_size"]
        prompts = [rng.integers(4, V, cfg["prompt_len"]).tolist() for _ in range(cfg["batch"])]
        for ct, suffix in (("int8", ""), ("float32", "_f32")):
            m = R.RefModel(d, ct, 1, 0)
            gen, scores = m.generate_ids(prompts, max_len=cfg["max_len"], min_len=cfg["max_len"],
                                         return_scores=True)
            full = [p + g for p, g in zip(prompts, gen)]
            tgt_in = np.array([f[:-1] for f in full], np.int32)
            logits = m.step_logits([], tgt_in, V)
            rng2 = np.random.default_rng(1234)
            cols = np.sort(rng2.choice(V, size=min(16, V), replace=False))
            top2 = np.sort(logits, axis=-1)[..., -2:][..., ::-1]
            golden[key + suffix] = {
                "compute_type": m.compute_type(), "prompts": prompts, "max_len": cfg["max_len"],
                "generated": gen, "scores": scores.tolist(),
                "teacher_forced": {"tgt_in": tgt_in.tolist(), "argmax": logits.argmax(-1).tolist(),
                                   "top2": top2.tolist(), "cols": cols.tolist(),
                                   "values": logits[..., cols].tolist()},
            }
            m.close()
    with open(os.path.join(HERE, "model_golden.json"), "w") as f:
        json.dump(golden, f)
    print({k: list(v.get("runs", {"generate": 0}).keys()) for k, v in golden.items()})
    print(golden["aren-transliteration-i8"]["first_output_tokens"],
          golden["aren-transliteration-i8"]["runs"]["beam1"]["scores"][0])


if __name__ == "__main__":
    main()
