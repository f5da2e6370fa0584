#!/usr/bin/env python
"""BASELINE.json configs[4] family on ONE GPU: Whisper-large-v3-shaped transcription (encoder 32
layers over 1500 positions from 30 s of audio = 3000 mel frames x 128 bins, decoder 32 layers,
d_model 1280, 20 heads of 64, FFN 5120, vocabulary 51866), 8 windows, beam 5 - models::Whisper
semantics through ct2x_whisper_generate with 16-bit weights on the matrix cores (bfloat16 by
default). Synthetic weights (model_builder.build_whisper, direct draw). The reference runs this
configuration with TP = 8; one MI355X holds it whole (3.1 GB of weights), so this is the one-GPU
line. <|endoftext|> is suppressed so that every window decodes the same number of steps.

Prints one JSON line: generated tokens/s (hypothesis 0), encoder ms per call, decode ms per step,
seconds of audio per second, and the encoder against the bf16 matrix roofline.

  python tools/bench_whisper.py [--batch 8] [--beam 5] [--steps 2] [--compute-type bfloat16]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MFMA_16BIT_PEAK_TFLOPS = 2500.0  # dense bf16 / fp16 (MI355X_MICROARCH.md)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--enc-layers", type=int, default=32)
    p.add_argument("--dec-layers", type=int, default=32)
    p.add_argument("--d-model", type=int, default=1280)
    p.add_argument("--ffn", type=int, default=5120)
    p.add_argument("--heads", type=int, default=20)
    p.add_argument("--n-mels", type=int, default=128)
    p.add_argument("--frames", type=int, default=3000)
    p.add_argument("--batch", type=int, default=8)
    p.add_argument("--beam", type=int, default=5)
    p.add_argument("--new-tokens", type=int, default=64)
    p.add_argument("--steps", type=int, default=2)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--compute-type", default="bfloat16")
    args = p.parse_args()

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench_whisper.py needs an MI355X (no CPU fallback)")
    import ctranslate2_amd as ct2
    from ctranslate2_amd import model_builder as mb

    quant = args.compute_type if args.compute_type in ("float16", "bfloat16") else "float32"
    model_dir = "/tmp/ct2x_bench_whisper_E%d_D%d_d%d_%s" % (args.enc_layers, args.dec_layers,
                                                           args.d_model, quant)
    t0 = time.perf_counter()
    if not os.path.exists(os.path.join(model_dir, "model.bin")):
        mb.build_whisper(model_dir, num_encoder_layers=args.enc_layers,
                         num_decoder_layers=args.dec_layers, d_model=args.d_model, ffn_dim=args.ffn,
                         num_heads=args.heads, n_mels=args.n_mels,
                         max_source_positions=(args.frames + 1) // 2, max_target_positions=448,
                         num_text_tokens=50257, languages=tuple("l%d" % i for i in range(100)),
                         num_timestamps=1501, quantization=quant, direct_f16=True, w_std=0.02,
                         alignment_heads=((args.dec_layers - 1, 0), (args.dec_layers - 1, 1)))
    t_build = time.perf_counter() - t0
    t0 = time.perf_counter()
    w = ct2.Whisper(model_dir, device="cuda", compute_type=args.compute_type)
    t_load = time.perf_counter() - t0
    rng = np.random.default_rng(1)
    feats = rng.standard_normal((args.batch, args.n_mels, args.frames)).astype(np.float32)
    sot = w._id("<|startoftranscript|>")
    eot = w._id("<|endoftext|>")
    prompt = [sot, w._id("<|l0|>"), w._id("<|transcribe|>"), w._id("<|notimestamps|>")]
    kw = dict(beam_size=args.beam, max_length=2 * args.new_tokens, suppress_tokens=(-1, eot))

    def encode_only():
        torch.cuda.synchronize()
        t = time.perf_counter()
        w.encode(feats)
        torch.cuda.synchronize()
        return (time.perf_counter() - t) * 1e3

    def one():
        res = w.generate(feats, [prompt] * args.batch, **kw)
        return sum(len(r.sequences_ids[0]) for r in res)

    for _ in range(args.warmup):
        one()
        encode_only()
    enc_ms = min(encode_only() for _ in range(2))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tokens = 0
    for _ in range(args.steps):
        tokens += one()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    call_ms = dt / args.steps * 1e3
    d, f, S, B = args.d_model, args.ffn, (args.frames + 1) // 2, args.batch
    # encoder FLOPs: convolutions + per layer (QKV, out, FFN) + attention; cross K/V projections
    conv = 2.0 * B * (args.frames * d * args.n_mels * 3 + S * d * d * 3)
    layer = 2.0 * B * S * (4 * d * d + 2 * d * f) + 4.0 * B * S * S * d
    cross_kv = 2.0 * B * S * 2 * d * d * args.dec_layers
    enc_flops = conv + args.enc_layers * layer + cross_kv
    # (encode() above stops before nothing: start_features also projects the cross-attention K/V)
    dec_step_ms = (call_ms - enc_ms) / max(args.new_tokens, 1)
    print(json.dumps({
        "metric": "generated tokens/sec (Whisper-large-v3 shape, %s, %d x 30 s windows, beam %d), 1 GPU"
                  % (args.compute_type, args.batch, args.beam),
        "value": tokens / dt, "unit": "tokens/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": call_ms, "dtype": args.compute_type, "data": "synthetic",
        "config": {
            "workload": "Whisper encoder %d + decoder %d layers d=%d ffn=%d heads=%d n_mels=%d vocab=51866, "
                        "%s weights (random), %d windows x %d frames, beam %d, prompt of 4 tokens, %d "
                        "new tokens (<|endoftext|> suppressed)"
                        % (args.enc_layers, args.dec_layers, d, f, args.heads, args.n_mels, quant, B,
                           args.frames, args.beam, args.new_tokens),
            "encode_ms": enc_ms, "decode_ms_per_step": dec_step_ms,
            "audio_seconds_per_second": B * (args.frames / 100.0) / (call_ms * 1e-3),
            "model_build_s": t_build, "model_load_s": t_load,
        },
        "encoder_roofline": {"bound": "mfma", "unit": "TFLOP/s", "peak": MFMA_16BIT_PEAK_TFLOPS,
                             "achieved": enc_flops / (enc_ms * 1e-3) / 1e12,
                             "frac": enc_flops / (enc_ms * 1e-3) / 1e12 / MFMA_16BIT_PEAK_TFLOPS,
                             "flops": enc_flops,
                             "note": "convolutions as GEMMs + 32 encoder layers over 1500 positions + the "
                                     "decoder's cross-attention K/V projections, per encode call"},
    }))


if __name__ == "__main__":
    main()
