"""Dev-time script (needs /root/reference and oracle/_ref built): golden outputs of the REAL
reference for the remaining decoding features:
  * Generator::forward_batch / score_batch on the synthetic Llama-shaped models,
  * Translator::translate_batch with return_alternatives (+ min_alternative_expansion_prob),
    biased decoding (prefix_bias_beta) and return_attention on the synthetic "small" / "tiny"
    encoder-decoder models
-> decoding_features_golden.json (tokens, scores) + decoding_features_golden.npz (arrays).

  python tests/golden/make_golden_decoding_features.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import _refshim as R  # noqa: E402
import golden_cases as GC  # noqa: E402
from ctranslate2_amd import model_builder as mb  # noqa: E402


def lm_cases(out, arrays):
    for key in ("lm_tiny", "lm_small"):
        cfg = GC.SYNTHETIC_LM[key]
        path = "/tmp/ct2x_golden_%s" % key
        if not os.path.exists(os.path.join(path, "model.bin")):
            mb.build_decoder_lm(path, **cfg["model"])
        vocab = cfg["model"]["vocab_size"]
        rng = np.random.default_rng(cfg["data_seed"] + 500)
        seqs = [rng.integers(4, vocab, n).tolist() for n in (9, 14, 1, 6, 14, 2)]
        for ct, suffix in (("float32", "_f32"), ("int8", "")):
            m = R.RefModel(path, ct, 1, 4)
            scores = m.generator_score_ids(seqs)
            scores_trunc = m.generator_score_ids(seqs, max_input_length=7)
            entry = {"sequences": seqs, "scores": [s.tolist() for s in scores],
                     "scores_max_input_length_7": [s.tolist() for s in scores_trunc]}
            if key == "lm_tiny":
                fwd = [s for s in seqs if len(s) > 0]
                arrays["%s%s_forward_logits" % (key, suffix)] = m.generator_forward_ids(fwd, vocab)
                arrays["%s%s_forward_log_probs" % (key, suffix)] = m.generator_forward_ids(fwd, vocab, True)
            out[key + suffix] = entry
            m.close()
            print(key, ct, [len(s) for s in scores], [len(s) for s in scores_trunc])


def s2s_cases(out, arrays):
    for key in ("tiny", "small"):
        cfg = GC.SYNTHETIC[key]
        path = "/tmp/ct2x_golden_%s" % key
        if not os.path.exists(os.path.join(path, "model.bin")):
            mb.build_transformer(path, **cfg["model"])
        vocab = cfg["model"]["vocab_size"]
        rng = np.random.default_rng(cfg["data_seed"] + 700)
        sources = [rng.integers(4, vocab, n).tolist() for n in (9, 12, 5, 12, 7, 3)]
        max_len = 14
        for ct, suffix in (("float32", "_f32"), ("int8", "")):
            m = R.RefModel(path, ct, 1, 4)
            base, _, _ = m.translate_ex_ids(sources, beam_size=2, max_len=max_len)
            # prefixes: what beam search says, cut at ragged lengths (one empty)
            prefixes = [base[b][0][:n] for b, n in enumerate((3, 0, 2, 5, 1, 2))]
            # a prefix the model would NOT choose (biased decoding then has something to do)
            odd = [[int((t * 7 + 13) % (vocab - 4)) + 4 for t in p] + [int(5 + b)] for b, p in enumerate(prefixes)]
            runs = {}

            def run(name, one_by_one=False, **kw):
                print("  run", name, flush=True)
                if one_by_one:
                    # The CPU reference's beam search returns the attention of ANOTHER sentence for
                    # the first step of sentence i >= 1 of a batch: its first step is not expanded
                    # yet (expand_after_first_step), the step's attention is repeated per beam
                    # (src/decoding.cc:589-591) and then gathered with the UNEXPANDED batch indices
                    # (:593) - sentence i receives row i of the repeated tensor, i.e. the vector of
                    # sentence i / beam_size. The coverage penalty inherits it. Sentences are
                    # translated one at a time here, where the indices coincide.
                    hyps, scores, att = [], [], []
                    for b in range(len(sources)):
                        kw_b = dict(kw)
                        if "prefixes" in kw_b:
                            kw_b["prefixes"] = [kw_b["prefixes"][b]]
                        h, sc, a = m.translate_ex_ids([sources[b]], max_len=max_len, **kw_b)
                        hyps += h
                        scores += sc
                        if a is not None:
                            att += [[np.pad(x, ((0, 0), (0, max(len(q) for q in sources) + 2 - x.shape[1])))
                                     for x in a[0]]]
                    att = att or None
                    kw = dict(kw, one_by_one=True)
                else:
                    hyps, scores, att = m.translate_ex_ids(sources, max_len=max_len, **kw)
                kw = dict(kw)
                kw.pop("return_attention", None)
                runs[name] = {"options": kw, "hypotheses": hyps, "scores": scores}
                if att is not None:
                    for b in range(len(sources)):
                        for k in range(len(att[b])):
                            arrays["%s%s_%s_att_%d_%d" % (key, suffix, name, b, k)] = att[b][k]
                return hyps

            # expansion threshold: every example keeps its two best alternatives, most lose the others
            # (the reference divides by zero when NO alternative passes)
            second = []
            for b in range(len(sources)):
                _, sc, _ = m.translate_ex_ids([sources[b]], prefixes=[prefixes[b]], beam_size=1,
                                              max_len=len(prefixes[b]) + 1, num_hypotheses=4,
                                              return_alternatives=True)
                second.append(np.exp(sc[0][1]))
            min_prob = float(min(second)) * 0.9999
            for beam in (1, 2):
                run("alt3_beam%d" % beam, beam_size=beam, num_hypotheses=3, return_alternatives=True)
                run("alt3_prefix_beam%d" % beam, prefixes=prefixes, beam_size=beam, num_hypotheses=3,
                    return_alternatives=True)
                run("alt4_minprob_beam%d" % beam, prefixes=prefixes, beam_size=beam, num_hypotheses=4,
                    return_alternatives=True, min_alternative_expansion_prob=min_prob)
                run("alt2_lp_beam%d" % beam, prefixes=prefixes, beam_size=beam, num_hypotheses=2,
                    return_alternatives=True, length_penalty=0.6, min_len=4)
                run("bias03_beam%d" % beam, prefixes=odd, beam_size=beam, prefix_bias_beta=0.3)
                run("bias08_beam%d" % beam, prefixes=odd, beam_size=beam, prefix_bias_beta=0.8)
                run("bias05_own_beam%d" % beam, prefixes=prefixes, beam_size=beam, prefix_bias_beta=0.5,
                    num_hypotheses=beam)
                run("attention_beam%d" % beam, one_by_one=beam > 1, beam_size=beam, num_hypotheses=beam,
                    return_attention=True)
                run("attention_prefix_beam%d" % beam, one_by_one=beam > 1, prefixes=prefixes,
                    beam_size=beam, return_attention=True)
            run("coverage_beam2", one_by_one=True, beam_size=2, num_hypotheses=2, coverage_penalty=0.3,
                length_penalty=0.4)
            # (coverage_penalty with beam_size 1 crashes the reference: GreedySearch gathers the
            # attention into results whose attention vector was never sized, src/decoding.cc:819-832)
            out[key + suffix] = {"sources": sources, "prefixes": prefixes, "odd_prefixes": odd,
                                 "max_len": max_len, "runs": runs}
            m.close()
            print(key, ct, {k: [len(h) for h in v["hypotheses"]] for k, v in runs.items() if "alt4" in k},
                  sum(a != b for a, b in zip(runs["bias08_beam2"]["hypotheses"], runs["bias03_beam2"]["hypotheses"])))


def main():
    assert R.available(), "build oracle/_ref first (make -C oracle ref)"
    out, arrays = {}, {}
    lm_cases(out, arrays)
    s2s_cases(out, arrays)
    with open(os.path.join(HERE, "decoding_features_golden.json"), "w") as f:
        json.dump(out, f)
    np.savez_compressed(os.path.join(HERE, "decoding_features_golden.npz"), **arrays)


if __name__ == "__main__":
    main()
