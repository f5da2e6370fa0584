"""Whisper parity cases shared by the golden generator (tests/golden/make_golden_whisper.py, runs
the reference's models::Whisper on CPU), the Level-1 runner (the reference's own Whisper on
Device::CUDA = this backend) and the Level-2 tests (ct2x_whisper_*, the fused engine). Every
case goes through the same `WhisperCalls` surface (tests/_refshim.py)."""
import numpy as np

MODEL = dict(num_encoder_layers=2, num_decoder_layers=2, d_model=64, ffn_dim=256, num_heads=4,
             n_mels=8, max_source_positions=12, max_target_positions=32, num_text_tokens=40,
             languages=("en", "fr", "de", "es"), num_timestamps=21, seed=3, w_std=0.15)
D_MODEL = MODEL["d_model"]
FRAMES = 24
# vocabulary layout of model_builder.whisper_vocab_tokens
EOT, SOT = 40, 41
EN, FR, DE, ES = 42, 43, 44, 45
TRANSLATE, TRANSCRIBE, STARTOFLM, STARTOFPREV, NOSPEECH, NOTIMESTAMPS = 46, 47, 48, 49, 50, 51
TS0 = 52
NUM_LANGUAGES = 4


def features(batch=3, frames=FRAMES, seed=11):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((batch, MODEL["n_mels"], frames)).astype(np.float32)


def build(path, quantization="float32"):
    import os
    from ctranslate2_amd import model_builder as mb
    if not os.path.exists(os.path.join(path, "model.bin")):
        mb.build_whisper(path, quantization=quantization, **MODEL)
    return path


_B = dict(max_length=24, return_scores=True, return_no_speech_prob=True)
GENERATE = {
    "greedy_notimestamps": ([[SOT, EN, TRANSCRIBE, NOTIMESTAMPS]] * 3, dict(_B, beam_size=1)),
    "greedy_timestamps": ([[SOT, EN, TRANSCRIBE], [SOT, FR, TRANSLATE], [SOT, DE, TRANSCRIBE]],
                          dict(_B, beam_size=1)),
    "beam5_timestamps": ([[SOT, EN, TRANSCRIBE]] * 3, dict(_B, beam_size=5, num_hypotheses=2)),
    "beam2_notimestamps": ([[SOT, ES, TRANSCRIBE, NOTIMESTAMPS]] * 3,
                           dict(_B, beam_size=2, patience=2.0, length_penalty=0.8)),
    "prev_text_greedy": ([[STARTOFPREV, 5, 6, 7, SOT, FR, TRANSCRIBE],
                          [STARTOFPREV, 11, 3, 19, SOT, EN, TRANSCRIBE],
                          [STARTOFPREV, 8, 8, 30, SOT, DE, TRANSLATE]], dict(_B, beam_size=1)),
    "prev_text_beam3": ([[STARTOFPREV, 5, 6, 7, SOT, FR, TRANSCRIBE, NOTIMESTAMPS]] * 2,
                        dict(_B, beam_size=3)),
    "sot_only": ([[SOT]] * 3, dict(_B, beam_size=1)),
    "text_after_task": ([[SOT, EN, TRANSCRIBE, NOTIMESTAMPS, 5, 9],
                         [SOT, EN, TRANSCRIBE, NOTIMESTAMPS, 13],
                         [SOT, EN, TRANSCRIBE, NOTIMESTAMPS, 21, 22]], dict(_B, beam_size=1)),
    "timestamp_in_prefix": ([[SOT, EN, TRANSCRIBE, TS0 + 4, 5], [SOT, EN, TRANSCRIBE, TS0 + 2, 17],
                             [SOT, EN, TRANSCRIBE, TS0 + 1, 9]], dict(_B, beam_size=1)),
    "timestamp_in_prefix_beam2": ([[SOT, EN, TRANSCRIBE, TS0 + 4, 5]] * 2, dict(_B, beam_size=2)),
    "options": ([[SOT, EN, TRANSCRIBE]] * 3,
                dict(_B, beam_size=1, suppress_blank=False, suppress_tokens=(4, 9, 33),
                     repetition_penalty=1.3, no_repeat_ngram_size=2, max_initial_timestamp_index=3)),
    "options_beam": ([[SOT, EN, TRANSCRIBE]] * 3,
                     dict(_B, beam_size=4, suppress_tokens=(), repetition_penalty=1.2,
                          max_initial_timestamp_index=1)),
    "no_scores": ([[SOT, EN, TRANSCRIBE, NOTIMESTAMPS]] * 3,
                  dict(max_length=30, beam_size=1, return_scores=False, return_no_speech_prob=False)),
}

ALIGN = {
    "uniform": (dict(start_sequence=[SOT, EN, TRANSCRIBE],
                     text_tokens=[[5, 6, 8, 9], [10, 11, 12], [3, 4, 7, 20, 21]],
                     num_frames=[24, 24, 24], median_filter_width=3)),
    "variable": (dict(start_sequence=[SOT, EN, TRANSCRIBE],
                      text_tokens=[[5, 6, 8, 9], [10, 11, 12], [3, 4, 7, 20, 21]],
                      num_frames=[24, 16, 20], median_filter_width=7)),
    "short": (dict(start_sequence=[SOT, FR],
                   text_tokens=[[15], [16, 2], [30, 31, 32]],
                   num_frames=[18, 18, 18], median_filter_width=5)),
}


def run_cases(w, only=None):
    """Runs every case through a WhisperCalls object; returns plain JSON-able results plus numpy
    arrays under keys ending in '@' (stored in the .npz)."""
    out = {}
    f = features()
    out["encode@"] = w.encode(f, D_MODEL)
    for name, (prompts, kw) in GENERATE.items():
        if only and name not in only:
            continue
        hyps, scores, nsp = w.generate(f[:len(prompts)], prompts, **kw)
        out["generate." + name] = {"hypotheses": hyps, "scores": np.asarray(scores).tolist(),
                                   "no_speech_prob": np.asarray(nsp).tolist()}
    ids, probs = w.detect_language(f, NUM_LANGUAGES)
    out["detect_language"] = {"ids": ids.tolist(), "probs": probs.tolist()}
    for name, kw in ALIGN.items():
        pairs, tprobs = w.align(f, **kw)
        out["align." + name] = {"pairs": pairs, "token_probs": [p.tolist() for p in tprobs]}
    return out
