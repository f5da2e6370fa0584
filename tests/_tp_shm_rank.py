"""One rank of the two-process tensor-parallel test (tests/test_tensor_parallel_gpu.py::
test_two_processes_one_gpu_*): builds the sharded engine on the shared-memory / hipIpc
communicator, translates, writes its result as JSON. Usage:
  python _tp_shm_rank.py <model_dir> <rank> <world> <segment> <compute_type> <out.json> <beam> <use_graph>"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    model_dir, rank, world, segment, compute_type, out_path, beam, use_graph = sys.argv[1:9]
    rank, world, beam, use_graph = int(rank), int(world), int(beam), bool(int(use_graph))
    import ctranslate2_amd as ct2
    import golden_cases as GC
    key = os.path.basename(model_dir).replace("ct2x_test_", "")
    cfg = GC.SYNTHETIC[key]
    rng = np.random.default_rng(cfg["data_seed"])
    vocab = cfg["model"]["vocab_size"]
    src = [rng.integers(4, vocab, n).tolist() for n in cfg["src_lengths"]]
    t = ct2.Translator.tensor_parallel_shm(model_dir, rank, world, segment, compute_type=compute_type)
    kw = dict(beam_size=beam, max_decoding_length=cfg["max_len"], min_decoding_length=cfg["min_len"],
              return_scores=True, use_graph=use_graph)
    hyps, scores = t.translate_ids(src, **kw)
    hyps2, scores2 = t.translate_ids(src, **kw)  # a second call reuses mailboxes, tags and graphs
    json.dump({"hyps": hyps, "scores": np.asarray(scores).tolist(), "hyps2": hyps2,
               "scores2": np.asarray(scores2).tolist(), "tp": t.tensor_parallel_info()},
              open(out_path, "w"))
    t.close()


if __name__ == "__main__":
    main()
