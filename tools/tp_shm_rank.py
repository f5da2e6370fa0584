"""One rank of bench.py's `config.tp_shm` leg: the Transformer-base bench model sharded over
`world` PROCESSES that share the one visible GPU, tensor-parallel exchanges through the
shared-memory / hipIpc communicator (ShmComm, csrc/tp_comm.cc) - the multi-rank code path
(separate address spaces, IPC-mapped mailboxes) without a second device. Usage:
  python tools/tp_shm_rank.py <model_dir> <rank> <world> <segment> <out.json> <batch> <src_len> <max_len>"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    model_dir, rank, world, segment, out_path, batch, src_len, max_len = sys.argv[1:9]
    rank, world, batch, src_len, max_len = int(rank), int(world), int(batch), int(src_len), int(max_len)
    import ctranslate2_amd as ct2
    rng = np.random.default_rng(1)  # = bench.py make_sources(0, ...)
    src = [rng.integers(4, 32000, src_len).tolist() for _ in range(batch)]
    t = ct2.Translator.tensor_parallel_shm(model_dir, rank, world, segment, compute_type="int8")
    kw = dict(beam_size=1, max_decoding_length=max_len, min_decoding_length=max_len)
    t.translate_ids(src, **kw)  # warm-up (graph capture)
    t0 = time.perf_counter()
    for _ in range(3):
        hyps, _ = t.translate_ids(src, **kw)
    dt = (time.perf_counter() - t0) / 3
    json.dump({"hyps": hyps, "seconds_per_call": dt, "tp": t.tensor_parallel_info()}, open(out_path, "w"))
    t.close()


if __name__ == "__main__":
    main()
