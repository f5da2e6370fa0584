#!/usr/bin/env python
"""BASELINE.json configs[3] on ONE GPU: NLLB-200-3.3B-shaped float16 translation (encoder-decoder,
24 + 24 layers, d_model 2048, 16 heads of 128, FFN 8192, shared vocabulary 256206), batch 64, beam 4,
sources of 32 tokens, 64 forced decoding steps - Translator.translate_batch semantics through
ct2x_translate_batch with compute type float16 (16-bit weights on the fp16 matrix cores, fp32
accumulation). Synthetic weights (model_builder.py direct_f16: uniform float16 values of the
named variance, shared embeddings stored once: 3.3 B parameters = 6.6 GB). The reference runs this
configuration with TP = 4; one MI355X holds it whole (288 GB), so this is the one-GPU line.

Prints one JSON line: hypothesis-0 tokens/s (the reference's accounting), decode ms per step and
the decode step against the HBM roofline (decoder + projection weights read once per step).

  python tools/bench_nllb.py [--layers 24] [--batch 64] [--beam 4] [--steps 2]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--layers", type=int, default=24)
    p.add_argument("--d-model", type=int, default=2048)
    p.add_argument("--ffn", type=int, default=8192)
    p.add_argument("--heads", type=int, default=16)
    p.add_argument("--vocab", type=int, default=256206)
    p.add_argument("--batch", type=int, default=64)
    p.add_argument("--beam", type=int, default=4)
    p.add_argument("--src-len", type=int, default=32)
    p.add_argument("--max-len", type=int, default=64)
    p.add_argument("--steps", type=int, default=2)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--compute-type", default="float16")
    args = p.parse_args()

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench_nllb.py needs an MI355X (no CPU fallback)")
    import ctranslate2_amd as ct2
    from ctranslate2_amd import model_builder as mb

    model_dir = "/tmp/ct2x_bench_nllb_L%d_d%d" % (args.layers, args.d_model)
    t0 = time.perf_counter()
    if not os.path.exists(os.path.join(model_dir, "model.bin")):
        mb.build_transformer(model_dir, num_encoder_layers=args.layers, num_decoder_layers=args.layers,
                             d_model=args.d_model, ffn_dim=args.ffn, num_heads=args.heads,
                             vocab_size=args.vocab, seed=0, quantization="float16", direct_f16=True,
                             share_embeddings=True)
    t_build = time.perf_counter() - t0
    t0 = time.perf_counter()
    tr = ct2.Translator(model_dir, device="cuda", compute_type=args.compute_type)
    t_load = time.perf_counter() - t0
    rng = np.random.default_rng(1)
    sources = [rng.integers(4, args.vocab, args.src_len).tolist() for _ in range(args.batch)]
    kw = dict(beam_size=args.beam, max_decoding_length=args.max_len, min_decoding_length=args.max_len)

    def one():
        hyps, _ = tr.translate_ids(sources, **kw)
        return sum(len(h[0]) for h in hyps)

    for _ in range(args.warmup):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tokens = 0
    for _ in range(args.steps):
        tokens += one()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stats = tr.last_stats()
    d, f, v, L = args.d_model, args.ffn, args.vocab, args.layers
    dec_layer = 4 * d * d + 4 * d * d + 2 * d * f  # self-attention, cross-attention, FFN
    weight_bytes = 2 * (L * dec_layer + v * d)
    step_ms = stats["decode_ms"] / max(args.max_len - 1, 1)
    print(json.dumps({
        "metric": "generated tokens/sec (NLLB-3.3B shape, %s, batch=%d beam=%d), 1 GPU"
                  % (args.compute_type, args.batch, args.beam),
        "value": tokens / dt, "unit": "tokens/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "dtype": args.compute_type, "data": "synthetic",
        "config": {
            "workload": "encoder-decoder %d+%d layers d=%d ffn=%d heads=%d vocab=%d, float16 weights (random, "
                        "shared embeddings), %d sources x %d tokens, beam %d, %d forced decoding steps"
                        % (L, L, d, f, args.heads, v, args.batch, args.src_len, args.beam, args.max_len),
            "encode_ms_last_call": stats["encode_ms"], "decode_ms_last_call": stats["decode_ms"],
            "decode_ms_per_step": step_ms, "kernels_per_decode_step": stats["kernels_per_step"],
            "weight_bytes_per_step": weight_bytes, "model_build_s": t_build, "model_load_s": t_load,
        },
        "decode_roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                            "achieved": weight_bytes / (step_ms * 1e-3) / 1e9,
                            "frac": weight_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "note": "float16 decoder + projection weights read once per decode step / "
                                    "measured step time (256 rows: AI = 256 FLOP/B, still bandwidth bound)"},
    }))


if __name__ == "__main__":
    main()
