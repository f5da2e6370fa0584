#!/usr/bin/env python
"""BASELINE.json configs[2]: Llama-3-8B-shaped int8 decoder-only generation (RMSNorm + RoPE + GQA +
SwiGLU path), batch 16 greedy, on one MI355X: Generator.generate_batch semantics through
ct2x_generate_batch. Synthetic model (random int8 weights drawn directly, d=4096, 32 layers,
32/8 heads, ffn 14336, vocab 128256 => 7.5 GB of int8 weights streamed per decode step),
16 prompts x 128 tokens, 128 forced new tokens.

Prints one JSON line: generated tokens/s (prefill + decode, the reference's accounting) plus the
decode-only rate against the HBM roofline (weights are read once per step: bytes/step / t_step)
and the per-kernel profile of one call (HIP events on the engine's stream).

  python tools/bench_l8.py [--layers 32] [--batch 16] [--prompt 128] [--new 128] [--steps 3]
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
    p.add_argument("--layers", type=int, default=32)
    p.add_argument("--d-model", type=int, default=4096)
    p.add_argument("--ffn", type=int, default=14336)
    p.add_argument("--heads", type=int, default=32)
    p.add_argument("--heads-kv", type=int, default=8)
    p.add_argument("--vocab", type=int, default=128256)
    p.add_argument("--batch", type=int, default=16)
    p.add_argument("--prompt", type=int, default=128)
    p.add_argument("--new", type=int, default=128)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--no-profile", action="store_true")
    args = p.parse_args()

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench_l8.py needs an MI355X (no CPU fallback)")
    import ctranslate2_amd as ct2
    from ctranslate2_amd import model_builder as mb

    model_dir = "/tmp/ct2x_bench_l8_L%d_d%d" % (args.layers, args.d_model)
    t0 = time.perf_counter()
    if not os.path.exists(os.path.join(model_dir, "model.bin")):
        mb.build_decoder_lm(model_dir, num_layers=args.layers, d_model=args.d_model,
                            ffn_dim=args.ffn, num_heads=args.heads, num_heads_kv=args.heads_kv,
                            vocab_size=args.vocab, direct_int8=True)
    t_build = time.perf_counter() - t0
    t0 = time.perf_counter()
    gen = ct2.Translator(model_dir, device="cuda", compute_type="int8")
    t_load = time.perf_counter() - t0

    rng = np.random.default_rng(1)
    prompts = [rng.integers(4, args.vocab, args.prompt).tolist() for _ in range(args.batch)]
    kw = dict(max_length=args.new, min_length=args.new)

    def one():
        out, _ = gen.generate_ids(prompts, **kw)
        return sum(len(o) for o in out)

    for _ in range(args.warmup):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tokens = 0
    for _ in range(args.steps):
        tokens += one()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stats = gen.last_stats()

    d, f, v = args.d_model, args.ffn, args.vocab
    dh = d // args.heads
    per_layer = (args.heads + 2 * args.heads_kv) * dh * d + d * d + 3 * f * d
    weight_bytes = args.layers * per_layer + v * d
    decode_ms = stats["decode_ms"]
    decode_steps = args.new - 1 if args.new > 1 else 1
    step_ms = decode_ms / decode_steps
    result = {
        "metric": "generated tokens/sec (Llama-3-8B shape, int8, batch=%d greedy), 1 GPU" % args.batch,
        "value": tokens / dt,
        "unit": "tokens/s",
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "dtype": "int8",
        "data": "synthetic",
        "config": {
            "workload": "decoder-only %d layers d=%d ffn=%d heads=%d/%d vocab=%d int8 weights (random, "
                        "drawn directly), %d prompts x %d tokens, %d forced new tokens" %
                        (args.layers, d, f, args.heads, args.heads_kv, v, args.batch, args.prompt, args.new),
            "prefill_ms_last_call": stats["encode_ms"],
            "decode_ms_last_call": decode_ms,
            "decode_ms_per_step": step_ms,
            "kernels_per_decode_step": stats["kernels_per_step"],
            "weight_bytes_per_step": weight_bytes,
            "model_build_s": t_build,
            "model_load_s": t_load,
        },
        "decode_roofline": {
            "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
            "achieved": weight_bytes / (step_ms * 1e-3) / 1e9,
            "frac": weight_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "note": "int8 weights read once per decode step / measured step time (KV cache and "
                    "activations not counted)",
        },
    }
    if not args.no_profile:
        gen.profile_begin()
        one()
        prof = gen.profile_end()
        noop = prof.pop("noop[calibration]", None)  # the profile's own empty-kernel calibration
        total = sum(x["total_ms"] for x in prof.values())
        top = sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])[:12]
        result["profile"] = {
            "device_ms_eager": total,
            "top": [{"kernel": k, "launches": x["launches"], "total_ms": round(x["total_ms"], 3),
                     "share": round(x["total_ms"] / total, 4),
                     "avg_us": round(x["total_ms"] / max(x["launches"], 1) * 1e3, 2),
                     "GBps": round(x["bytes"] / (x["total_ms"] * 1e-3) / 1e9, 1) if x["bytes"] and x["total_ms"] else None,
                     "TOPS": round(x["ops"] / (x["total_ms"] * 1e-3) / 1e12, 1) if x["ops"] and x["total_ms"] else None}
                    for k, x in top],
        }
        if noop and noop["launches"]:
            result["profile"]["noop_bracket_us"] = noop["total_ms"] / noop["launches"] * 1e3
    print(json.dumps(result))


if __name__ == "__main__":
    main()
