"""ctranslate2_amd -- MI355X (gfx950) native backend for CTranslate2's quantized decoder hot path.

Python surface: a thin ctypes mirror of the reference's Python API for this path
(`ctranslate2.Translator.translate_batch`, python/cpp/translator.cc:286-420) over the C ABI in
include/ct2_mi355.h. All compute happens in libct2_mi355.so (hand-written HIP kernels); there is
no CPU or PyTorch fallback.
"""
import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import Ct2xError, ExecStats, TranslationOptions

__version__ = "0.1.0"


class TranslationResult:
    """ctranslate2.TranslationResult: hypotheses (token lists), scores and, with return_attention,
    one source attention vector per token of every hypothesis."""

    def __init__(self, hypotheses, scores, hypotheses_ids, attention=None, logits=None):
        self.hypotheses = hypotheses
        self.scores = scores
        self.hypotheses_ids = hypotheses_ids
        self.attention = attention if attention is not None else []
        # return_logits_vocab: per hypothesis an array [steps, vocabulary] (the reference: a list
        # of StorageView per hypothesis, python/cpp/translation_result.cc:19-20)
        self.logits = logits if logits is not None else []

    def __repr__(self):
        return "TranslationResult(hypotheses=%r, scores=%r)" % (self.hypotheses, self.scores)


class GenerationStepResult:
    """ctranslate2.GenerationStepResult (python/cpp/generation_result.cc:11-42): what the
    `callback` of translate_batch / generate_ids receives for every generated token."""

    def __init__(self, step, batch_id, token_id, token, log_prob, is_last, hypothesis_id=0):
        self.step = step
        self.batch_id = batch_id
        self.token_id = token_id
        self.hypothesis_id = hypothesis_id
        self.token = token
        self.log_prob = log_prob
        self.logits = None
        self.is_last = is_last

    def __repr__(self):
        return ("GenerationStepResult(step=%r, batch_id=%r, token_id=%r, hypothesis_id=%r, "
                "token=%r, log_prob=%r, logits=None, is_last=%r)"
                % (self.step, self.batch_id, self.token_id, self.hypothesis_id, self.token,
                   self.log_prob, self.is_last))


class ScoringResult:
    """ctranslate2.ScoringResult: the scored tokens (target + </s>) and their log-probabilities."""

    def __init__(self, tokens, log_probs, tokens_ids):
        self.tokens = tokens
        self.log_probs = log_probs
        self.tokens_score = log_probs   # the C++ member name (include/ctranslate2/scoring.h)
        self.tokens_ids = tokens_ids

    def __repr__(self):
        return "ScoringResult(tokens=%r, log_probs=%r)" % (self.tokens, self.log_probs)


class Translator:  # also serves decoder-only models (ctranslate2.Generator) via generate_ids
    """ctranslate2.Translator for converted models, on one MI355X.

    Arguments follow the reference (python/cpp/translator.cc:286-297). `device` must be "cuda"
    (the reference's name for the GPU device, Device::CUDA) -- this backend has no CPU path.
    """

    def __init__(self, model_path: str, device: str = "cuda", device_index: int = 0,
                 compute_type: str = "default", tensor_parallel: bool = False, _handle=None):
        """tensor_parallel=True (reference: ctranslate2.Translator(..., tensor_parallel=True) under
        mpirun): one process per GPU launched by torch.distributed.run; torch.distributed must be
        initialised. Heads and the FFN inner dimension are split over the WORLD_SIZE ranks and the
        row-parallel layers all-reduce over RCCL/xGMI; every rank must make the same calls."""
        if device not in ("cuda", "auto"):
            raise ValueError("ctranslate2_amd only implements the GPU device (device='cuda'); "
                             "got device=%r" % device)
        self._L = _lib.lib()
        if _handle is not None:
            self._h = _handle
        elif tensor_parallel:
            self._h = self._create_tensor_parallel(model_path, device_index, compute_type)
        else:
            self._h = self._L.ct2x_translator_create(model_path.encode(), C.c_int(device_index),
                                                     compute_type.encode())
        if not self._h:
            raise Ct2xError(self._L.ct2x_last_error().decode("utf-8", "replace"))
        self.model_path = model_path
        self.device_index = device_index
        info = self._model_info()
        self.compute_type = info["compute_type"]
        self.vocab_size = info["vocab_size"]
        self.is_seq2seq = info["is_seq2seq"]
        self.eos_id = info["eos_id"]
        self.bos_id = info["bos_id"]

    def _create_tensor_parallel(self, model_path, device_index, compute_type):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("tensor_parallel=True needs an initialised torch.distributed "
                               "process group (one process per GPU)")
        rank, world = dist.get_rank(), dist.get_world_size()
        uid = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            buf = C.create_string_buffer(128)
            _lib.check(self._L.ct2x_tp_unique_id(buf))
            uid = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
        if world > 1:
            if dist.get_backend() == "nccl":  # RCCL group: the broadcast needs a device tensor
                dev = uid.to(torch.device("cuda", device_index))
                dist.broadcast(dev, src=0)
                uid = dev.cpu()
            else:
                dist.broadcast(uid, src=0)
        raw = bytes(uid.numpy().tobytes())
        return self._L.ct2x_translator_create_tp(model_path.encode(), C.c_int(device_index),
                                                 compute_type.encode(), C.c_int(rank), C.c_int(world),
                                                 C.c_char_p(raw))

    @classmethod
    def tensor_parallel_shm(cls, model_path: str, rank: int, world: int, segment: str,
                            device_index: int = 0, compute_type: str = "default"):
        """Rank `rank` of `world` tensor-parallel PROCESSES of one node connected without RCCL:
        shared-memory bootstrap under the name `segment`, hipIpc mailboxes for the decode-step
        exchanges (ct2x_translator_create_tp_shm). The ranks may share a device."""
        L = _lib.lib()
        h = L.ct2x_translator_create_tp_shm(model_path.encode(), C.c_int(device_index),
                                            compute_type.encode(), C.c_int(rank), C.c_int(world),
                                            segment.encode())
        if not h:
            raise Ct2xError(L.ct2x_last_error().decode("utf-8", "replace"))
        return cls(model_path, device_index=device_index, compute_type=compute_type, _handle=h)

    @classmethod
    def tensor_parallel_local(cls, model_path: str, world: int, device_index: int = 0,
                              compute_type: str = "default"):
        """`world` tensor-parallel ranks inside THIS process on one device, connected by the
        in-process communicator (tests: drive each returned Translator from its own thread)."""
        L = _lib.lib()
        handles = (C.c_void_p * world)()
        _lib.check(L.ct2x_translator_create_tp_local(model_path.encode(), C.c_int(device_index),
                                                     compute_type.encode(), C.c_int(world), handles))
        return [cls(model_path, device_index=device_index, compute_type=compute_type,
                    _handle=handles[r]) for r in range(world)]

    def close(self):
        if getattr(self, "_h", None):
            self._L.ct2x_translator_destroy(C.c_void_p(self._h))
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- model facts ---------------------------------------------------------------------
    def _model_info(self):
        vals = [C.c_int32() for _ in range(7)]
        buf = C.create_string_buffer(64)
        _lib.check(self._L.ct2x_model_info(C.c_void_p(self._h), *[C.byref(v) for v in vals], buf,
                                           C.c_int(64)))
        keys = ["is_seq2seq", "vocab_size", "d_model", "num_layers", "num_heads", "bos_id", "eos_id"]
        out = {k: v.value for k, v in zip(keys, vals)}
        out["is_seq2seq"] = bool(out["is_seq2seq"])
        out["compute_type"] = buf.value.decode()
        return out

    def tensor_parallel_info(self):
        """rank, world, backend name and whether the one-shot mailboxes are mapped (ct2x_tp_info)."""
        rank, world, boxes = C.c_int32(), C.c_int32(), C.c_int32()
        buf = C.create_string_buffer(32)
        _lib.check(self._L.ct2x_tp_info(C.c_void_p(self._h), C.byref(rank), C.byref(world),
                                        C.byref(boxes), buf, C.c_int(32)))
        return {"rank": rank.value, "world": world.value, "mailboxes": bool(boxes.value),
                "backend": buf.value.decode()}

    def token_to_id(self, token: str, target: bool = False) -> int:
        return int(self._L.ct2x_vocab_id(C.c_void_p(self._h), C.c_int(int(target)), token.encode("utf-8")))

    def id_to_token(self, idx: int, target: bool = True) -> str:
        s = self._L.ct2x_vocab_token(C.c_void_p(self._h), C.c_int(int(target)), C.c_int32(idx))
        if s is None:
            raise Ct2xError(self._L.ct2x_last_error().decode("utf-8", "replace"))
        return s.decode("utf-8")

    # -- translation -----------------------------------------------------------------------
    def _options(self, beam_size, max_decoding_length, min_decoding_length, num_hypotheses,
                 return_scores, return_end_token, max_batch_size, max_input_length,
                 length_penalty, patience, use_graph, num_streams=0, sampling_topk=1,
                 sampling_topp=1.0, sampling_temperature=1.0, seed=0):
        opt = TranslationOptions()
        self._L.ct2x_translation_options_default(C.byref(opt))
        opt.beam_size = beam_size
        opt.max_decoding_length = max_decoding_length
        opt.min_decoding_length = min_decoding_length
        opt.num_hypotheses = num_hypotheses
        opt.return_scores = int(return_scores)
        opt.return_end_token = int(return_end_token)
        opt.max_batch_size = max_batch_size
        opt.max_input_length = max_input_length
        opt.length_penalty = length_penalty
        opt.patience = patience
        opt.use_graph = int(use_graph)
        opt.num_streams = int(num_streams)
        # reference semantics: 0 = the whole vocabulary (-1 in the C struct, where 0 = arg-max)
        opt.sampling_topk = -1 if int(sampling_topk) <= 0 else int(sampling_topk)
        opt.sampling_topp = float(sampling_topp)
        opt.sampling_temperature = float(sampling_temperature)
        opt.seed = int(seed) & 0xFFFFFFFF
        return opt

    def _known_id(self, token: str, what: str) -> int:
        """Vocabulary::to_id with allow_unk = false: an out-of-vocabulary token is an error
        (std::invalid_argument in the reference, sequence_to_sequence.cc:330-361)."""
        idx = self.token_to_id(token, target=True)
        if self.id_to_token(idx) != token:
            raise ValueError("%s token '%s' is not in the vocabulary" % (what, token))
        return idx

    def _set_logits(self, opt, batch, hyps, steps):
        """return_logits_vocab: the host buffers the C entry fills ([batch, hyps, steps, vocab] and
        the per-hypothesis counts)."""
        # (np.empty: the pages are only touched where the C entry writes a step - a dense zero-fill
        # of batch x steps x vocabulary floats is gigabytes at a 256k vocabulary; `counts` says how
        # many step vectors of a hypothesis are valid)
        buf = np.empty((batch, hyps, steps, self.vocab_size), np.float32)
        counts = np.zeros((batch, hyps), np.int32)
        opt.logits_out = buf.ctypes.data_as(C.POINTER(C.c_float))
        opt.logits_counts_out = counts.ctypes.data_as(C.POINTER(C.c_int32))
        opt.logits_steps = steps
        return buf, counts

    def _set_callback(self, opt, callback):
        """TranslationOptions / GenerationOptions.callback (python/cpp/translator.cc:152-181): the
        Python callable is called with a GenerationStepResult per generated token; returning True
        stops that batch entry. Returns a holder whose `error` is an exception the callable raised
        (the decoding of every entry is then stopped and the exception re-raised by the caller)."""
        holder = type("CallbackHolder", (), {"error": None, "fn": None})()
        if callback is None:
            return holder

        def trampoline(_user, step, batch_id, hypothesis_id, token_id, has_score, score, is_last):
            if holder.error is not None:
                return 1
            try:
                r = GenerationStepResult(int(step), int(batch_id), int(token_id),
                                         self.id_to_token(int(token_id)),
                                         float(score) if has_score else None, bool(is_last),
                                         int(hypothesis_id))
                return 1 if callback(r) else 0
            except BaseException as e:  # noqa: BLE001 - must not unwind through the C frames
                holder.error = e
                return 1

        holder.fn = _lib.STEP_CALLBACK(trampoline)
        opt.step_callback = holder.fn
        return holder

    @staticmethod
    def _set_processors(opt, repetition_penalty, no_repeat_ngram_size, suppress_sequences_ids):
        """Fills the logits-processor fields of the options; returns the arrays the struct points
        into (keep them alive for the duration of the call)."""
        opt.repetition_penalty = float(repetition_penalty)
        opt.no_repeat_ngram_size = int(no_repeat_ngram_size)
        seqs = [list(q) for q in (suppress_sequences_ids or []) if len(q) > 0]
        flat = np.array([t for q in seqs for t in q] or [0], np.int32)
        lens = np.array([len(q) for q in seqs] or [0], np.int32)
        opt.num_suppress_sequences = len(seqs)
        opt.suppress_ids = flat.ctypes.data_as(C.POINTER(C.c_int32))
        opt.suppress_lengths = lens.ctypes.data_as(C.POINTER(C.c_int32))
        return flat, lens

    def translate_ids(self, source_ids: Sequence[Sequence[int]], beam_size: int = 2,
                      max_decoding_length: int = 256, min_decoding_length: int = 1,
                      num_hypotheses: int = 1, return_scores: bool = False,
                      return_end_token: bool = False, max_batch_size: int = 0,
                      max_input_length: int = 1024, length_penalty: float = 1.0,
                      patience: float = 1.0, use_graph: bool = True, num_streams: int = 0,
                      sampling_topk: int = 1, sampling_topp: float = 1.0,
                      sampling_temperature: float = 1.0, seed: int = 0,
                      target_prefix_ids: Optional[Sequence[Sequence[int]]] = None,
                      repetition_penalty: float = 1.0, no_repeat_ngram_size: int = 0,
                      suppress_sequences_ids: Optional[Sequence[Sequence[int]]] = None,
                      end_token_id: Optional[int] = None, return_alternatives: bool = False,
                      min_alternative_expansion_prob: float = 0.0, prefix_bias_beta: float = 0.0,
                      coverage_penalty: float = 0.0, return_attention: bool = False,
                      callback=None, return_logits_vocab: bool = False):
        """translate_batch on token ids. Returns (hypotheses ids [batch][hyp][tokens], scores), plus
        the attention ([batch][hyp] arrays [tokens, source length]) when return_attention is set.
        return_alternatives / min_alternative_expansion_prob / prefix_bias_beta / coverage_penalty /
        return_attention: as in the reference's TranslationOptions.
        end_token_id: another target id that ends decoding (TranslationOptions::end_token).
        sampling_* (reference names, python/cpp/translator.cc): random sampling among the best
        sampling_topk candidates (beam_size must be 1), reproducible for a `seed`.
        repetition_penalty / no_repeat_ngram_size as in the reference; suppress_sequences_ids:
        id sequences that must not be generated (a single id is disabled at every step).
        callback: called with a GenerationStepResult for every generated token (beam_size 1);
        returning True stops that batch entry.
        return_logits_vocab (beam_size 1): one more result, [batch][hyp] arrays [steps, vocabulary]
        of the logits every step sampled from (after the logits processors and the minimum-length
        mask; one row per step the entry ran, the end token's step included)."""
        batch = len(source_ids)
        lengths = np.array([len(s) for s in source_ids], np.int32)
        stride = max(1, int(lengths.max()) if batch else 1)
        ids = np.zeros((batch, stride), np.int32)
        for i, s in enumerate(source_ids):
            ids[i, :len(s)] = s
        out = np.zeros((batch, num_hypotheses, max_decoding_length), np.int32)
        out_len = np.zeros((batch, num_hypotheses), np.int32)
        scores = np.zeros((batch, num_hypotheses), np.float32)
        opt = self._options(beam_size, max_decoding_length, min_decoding_length, num_hypotheses,
                            return_scores, return_end_token, max_batch_size, max_input_length,
                            length_penalty, patience, use_graph, num_streams, sampling_topk,
                            sampling_topp, sampling_temperature, seed)
        _keep = self._set_processors(opt, repetition_penalty, no_repeat_ngram_size,
                                     suppress_sequences_ids)
        opt.end_token_plus_one = 0 if end_token_id is None else int(end_token_id) + 1
        opt.return_alternatives = int(return_alternatives)
        opt.min_alternative_expansion_prob = float(min_alternative_expansion_prob)
        opt.prefix_bias_beta = float(prefix_bias_beta)
        opt.coverage_penalty = float(coverage_penalty)
        cb = self._set_callback(opt, callback)
        if return_logits_vocab:
            lbuf, lcnt = self._set_logits(opt, batch, num_hypotheses, max_decoding_length)
        att = None
        if return_attention:
            att = np.zeros((batch, num_hypotheses, max_decoding_length, stride), np.float32)
            opt.attention_stride = stride
            opt.attention_out = att.ctypes.data_as(C.POINTER(C.c_float))
        if target_prefix_ids is not None:
            plen = np.array([len(p) for p in target_prefix_ids], np.int32)
            pstride = max(1, int(plen.max()) if batch else 1)
            pids = np.zeros((batch, pstride), np.int32)
            for i, pfx in enumerate(target_prefix_ids):
                pids[i, :len(pfx)] = pfx
            _lib.check(self._L.ct2x_translate_batch_prefix(
                C.c_void_p(self._h), _lib.ptr(ids), _lib.ptr(lengths), C.c_int32(batch),
                C.c_int32(stride), _lib.ptr(pids), _lib.ptr(plen), C.c_int32(pstride), C.byref(opt),
                _lib.ptr(out), _lib.ptr(out_len), _lib.ptr(scores)))
        else:
            _lib.check(self._L.ct2x_translate_batch(
                C.c_void_p(self._h), _lib.ptr(ids), _lib.ptr(lengths), C.c_int32(batch),
                C.c_int32(stride), C.byref(opt), _lib.ptr(out), _lib.ptr(out_len), _lib.ptr(scores)))
        if cb.error is not None:
            raise cb.error
        keep = [[k for k in range(num_hypotheses)
                 if (out_len[b, k] >= 0 if return_alternatives else (out_len[b, k] > 0 or k == 0))]
                for b in range(batch)]
        hyps = [[out[b, k, :max(out_len[b, k], 0)].tolist() for k in keep[b]] for b in range(batch)]
        extra = ()
        if return_logits_vocab:
            extra = ([[lbuf[b, k, :lcnt[b, k]].copy() for k in keep[b]] for b in range(batch)],)
        if return_attention:
            atts = [[att[b, k, :max(out_len[b, k], 0), :len(source_ids[b])].copy() for k in keep[b]]
                    for b in range(batch)]
            return (hyps, scores, atts) + extra
        return (hyps, scores) + extra

    def translate_batch(self, source: List[List[str]], target_prefix: Optional[list] = None,
                        **kwargs) -> List[TranslationResult]:
        """ctranslate2.Translator.translate_batch on tokens (same keyword names)."""
        return_scores = kwargs.get("return_scores", False)
        src_ids = [[self.token_to_id(t) for t in sent] for sent in source]
        # suppress_sequences / disable_unk (python/cpp/translator.cc) -> id sequences
        suppress = [[self._known_id(t, "suppress_sequences") for t in q]
                    for q in (kwargs.pop("suppress_sequences", None) or [])]
        end_token = kwargs.pop("end_token", None)
        if end_token is not None:
            kwargs["end_token_id"] = self._known_id(end_token, "end_token")
        if kwargs.pop("disable_unk", False):
            suppress.append([self.token_to_id("<unk>", target=True)])
        if suppress:
            kwargs["suppress_sequences_ids"] = suppress
        if target_prefix is not None:
            kwargs["target_prefix_ids"] = [[self.token_to_id(t, target=True) for t in (pfx or [])]
                                           for pfx in target_prefix]
        # replace_unknowns (TranslationOptions, include/ctranslate2/translation.h:81): an unknown
        # target token becomes the source token with the highest attention - the decoding runs
        # with return_attention (src/models/sequence_to_sequence.cc:352) and the vectors are dropped
        # afterwards unless the caller asked for them (:414-422)
        replace_unknowns = bool(kwargs.pop("replace_unknowns", False))
        return_attention = bool(kwargs.get("return_attention", False))
        if replace_unknowns:
            kwargs["return_attention"] = True
        atts = logits = None
        out = list(self.translate_ids(src_ids, **kwargs))
        if kwargs.get("return_logits_vocab", False):
            logits = out.pop()
        if kwargs.get("return_attention", False):
            atts = out.pop()
        hyps, scores = out
        results = []
        for b, sent_hyps in enumerate(hyps):
            tokens = [[self.id_to_token(i) for i in h] for h in sent_hyps]
            if replace_unknowns:
                # replace_unknown_tokens (sequence_to_sequence.cc:288-302): first maximum. (the id an
                # absent token maps to IS the unknown token's, whatever the model calls it)
                unk_id = self.token_to_id("<unk>", target=True)
                for k, hyp in enumerate(tokens):
                    for t in range(len(hyp)):
                        if sent_hyps[k][t] == unk_id and t < len(atts[b][k]) and len(source[b]) > 0:
                            hyp[t] = source[b][int(np.argmax(atts[b][k][t]))]
            sc = [float(scores[b, k]) for k in range(len(sent_hyps))] if return_scores else []
            att = [a.tolist() for a in atts[b]] if (atts is not None and return_attention) else None
            results.append(TranslationResult(tokens, sc, sent_hyps, att,
                                             logits[b] if logits is not None else None))
        return results

    def score_ids(self, source_ids: Sequence[Sequence[int]], target_ids: Sequence[Sequence[int]],
                  max_input_length: int = 1024):
        """Translator.score_batch on ids (no special tokens): per example the scored ids
        (target + </s>, truncated like the reference) and their log-probabilities."""
        batch = len(source_ids)
        if len(target_ids) != batch:
            raise ValueError("Batch size mismatch: got %d for source and %d for target"
                             % (batch, len(target_ids)))
        if batch == 0:
            return [], []

        def pack(seqs):
            lengths = np.array([len(q) for q in seqs], np.int32)
            stride = max(1, int(lengths.max()))
            ids = np.zeros((len(seqs), stride), np.int32)
            for i, q in enumerate(seqs):
                ids[i, :len(q)] = q
            return ids, lengths, stride

        src, src_len, src_stride = pack(source_ids)
        tgt, tgt_len, tgt_stride = pack(target_ids)
        out_stride = int(tgt_len.max()) + 1
        out_ids = np.zeros((batch, out_stride), np.int32)
        out_scores = np.zeros((batch, out_stride), np.float32)
        out_len = np.zeros((batch,), np.int32)
        _lib.check(self._L.ct2x_score_batch(
            C.c_void_p(self._h), _lib.ptr(src), _lib.ptr(src_len), C.c_int32(batch),
            C.c_int32(src_stride), _lib.ptr(tgt), _lib.ptr(tgt_len), C.c_int32(tgt_stride),
            C.c_int32(max_input_length), _lib.ptr(out_ids), _lib.ptr(out_scores),
            C.c_int32(out_stride), _lib.ptr(out_len)))
        return ([out_ids[b, :out_len[b]].tolist() for b in range(batch)],
                [out_scores[b, :out_len[b]].tolist() for b in range(batch)])

    def score_batch(self, source: List[List[str]], target: List[List[str]],
                    max_input_length: int = 1024) -> List[ScoringResult]:
        """ctranslate2.Translator.score_batch on tokens."""
        ids, scores = self.score_ids([[self.token_to_id(t) for t in q] for q in source],
                                     [[self.token_to_id(t, target=True) for t in q] for q in target],
                                     max_input_length=max_input_length)
        return [ScoringResult([self.id_to_token(i) for i in row], sc, row)
                for row, sc in zip(ids, scores)]

    def generate_ids(self, prompts: Sequence[Sequence[int]], max_length: int = 512,
                     min_length: int = 0, return_scores: bool = False,
                     return_end_token: bool = False, use_graph: bool = True,
                     sampling_topk: int = 1, sampling_topp: float = 1.0,
                     sampling_temperature: float = 1.0, seed: int = 0, beam_size: int = 1,
                     length_penalty: float = 1.0, repetition_penalty: float = 1.0,
                     no_repeat_ngram_size: int = 0,
                     suppress_sequences_ids: Optional[Sequence[Sequence[int]]] = None,
                     end_token_id: Optional[int] = None, callback=None, num_hypotheses: int = 1,
                     patience: float = 1.0, return_logits_vocab: bool = False,
                     return_alternatives: bool = False, min_alternative_expansion_prob: float = 0.0,
                     include_prompt_in_result: bool = False, static_prompt_length: int = 0):
        """ctranslate2.Generator.generate_batch on token ids (include_prompt_in_result=False unless
        asked for: the reference's default True decodes the prompt step by step as a RETURNED hard
        prefix - it counts towards max / min_length and the sequences begin with the prompt, token 0
        included unless it is the BOS token, src/models/language_model.cc:150-282; static_prompt_length:
        that many leading tokens of every prompt are GenerationOptions::static_prompt - forwarded in one
        pass, never returned or counted).
        return_alternatives: the num_hypotheses best FIRST tokens after the prompt, each continued
        independently with the requested search (decode_alternatives, src/decoding.cc:1122-1290);
        as in the reference, max_length / min_length then count from the first PROMPT position, and
        the sequences of a longer prompt of a ragged batch begin with the rest of that prompt.
        num_hypotheses > 1 (the best of a beam search, or that many random samples per prompt,
        ordered by score): the results become [batch][hypothesis][tokens] and scores [batch, hyp].
        return_logits_vocab (beam_size 1): a third result, per prompt (and hypothesis) an array
        [steps, vocabulary] of the logits every step sampled from - the forced steps of a longer
        prompt included.
        sampling_topk=1 is greedy; k >= 2 samples among the best k, 0 from the whole vocabulary (sampling_topp / temperature as
        in the reference). Returns (generated ids per prompt, scores)."""
        batch = len(prompts)
        lengths = np.array([len(p) for p in prompts], np.int32)
        stride = int(lengths.max())
        ids = np.zeros((batch, stride), np.int32)
        for i, p in enumerate(prompts):
            ids[i, :len(p)] = p
        nh = int(num_hypotheses)
        out = np.zeros((batch, nh, max_length), np.int32)
        out_len = np.zeros((batch, nh), np.int32)
        scores = np.zeros((batch, nh), np.float32)
        opt = self._options(beam_size, max_length, min_length, nh, return_scores, return_end_token,
                            0, 0, length_penalty, patience, use_graph, 1, sampling_topk, sampling_topp,
                            sampling_temperature, seed)
        _keep = self._set_processors(opt, repetition_penalty, no_repeat_ngram_size,
                                     suppress_sequences_ids)
        opt.end_token_plus_one = 0 if end_token_id is None else int(end_token_id) + 1
        opt.return_alternatives = int(return_alternatives)
        opt.min_alternative_expansion_prob = float(min_alternative_expansion_prob)
        opt.include_prompt_in_result = int(include_prompt_in_result)
        opt.static_prompt_length = int(static_prompt_length)
        cb = self._set_callback(opt, callback)
        if return_logits_vocab:
            lbuf, lcnt = self._set_logits(opt, batch, nh,
                                          max_length + int(lengths.max() - lengths.min()))
        _lib.check(self._L.ct2x_generate_batch(
            C.c_void_p(self._h), _lib.ptr(ids), _lib.ptr(lengths), C.c_int32(batch),
            C.c_int32(stride), C.byref(opt), _lib.ptr(out), _lib.ptr(out_len), _lib.ptr(scores)))
        if cb.error is not None:
            raise cb.error
        extra = ()
        # "Forward the start token to the output if it is not the special BOS token"
        # (language_model.cc:262-267)
        sp = int(static_prompt_length)  # (GenerationOptions::static_prompt: the head of every prompt row)
        head = [[int(p[sp])] if include_prompt_in_result and int(p[sp]) != self.bos_id else []
                for p in prompts]
        if nh == 1 and not return_alternatives:
            if return_logits_vocab:
                extra = ([lbuf[b, 0, :lcnt[b, 0]].copy() for b in range(batch)],)
            return ([head[b] + out[b, 0, :out_len[b, 0]].tolist() for b in range(batch)], scores[:, 0]) + extra
        if return_alternatives:  # (-1: fewer alternatives than num_hypotheses were expanded)
            kept = [[k for k in range(nh) if out_len[b, k] >= 0] for b in range(batch)]
            return ([[out[b, k, :out_len[b, k]].tolist() for k in kept[b]] for b in range(batch)],
                    [[float(scores[b, k]) for k in kept[b]] for b in range(batch)])
        kept = [[k for k in range(nh) if out_len[b, k] > 0 or k == 0] for b in range(batch)]
        if return_logits_vocab:
            extra = ([[lbuf[b, k, :lcnt[b, k]].copy() for k in kept[b]] for b in range(batch)],)
        return ([[head[b] + out[b, k, :out_len[b, k]].tolist() for k in kept[b]] for b in range(batch)],
                scores) + extra

    @staticmethod
    def _pack_ids(seqs):
        lengths = np.array([len(s) for s in seqs], np.int32)
        stride = max(1, int(lengths.max())) if len(seqs) else 1
        ids = np.zeros((len(seqs), stride), np.int32)
        for i, s in enumerate(seqs):
            ids[i, :len(s)] = s
        return ids, lengths, stride

    def forward_ids(self, sequences: Sequence[Sequence[int]], return_log_probs: bool = False):
        """ctranslate2.Generator.forward_batch on token ids: logits (or log-probabilities) of every
        position, [batch, longest sequence, vocabulary]."""
        ids, lengths, stride = self._pack_ids(sequences)
        out = np.empty((len(sequences), int(lengths.max()), self.vocab_size), np.float32)
        _lib.check(self._L.ct2x_forward_batch(
            C.c_void_p(self._h), _lib.ptr(ids), _lib.ptr(lengths), C.c_int32(len(sequences)),
            C.c_int32(stride), C.c_int32(1 if return_log_probs else 0), _lib.ptr(out)))
        return out

    def score_tokens_ids(self, sequences: Sequence[Sequence[int]], max_input_length: int = 1024):
        """ctranslate2.Generator.score_batch on token ids: for every sequence the log-probabilities
        of tokens[1:] given what precedes them (an empty list for sequences of fewer than 2
        tokens)."""
        ids, lengths, stride = self._pack_ids(sequences)
        scores = np.zeros((len(sequences), stride), np.float32)
        out_len = np.zeros((len(sequences),), np.int32)
        _lib.check(self._L.ct2x_generator_score_batch(
            C.c_void_p(self._h), _lib.ptr(ids), _lib.ptr(lengths), C.c_int32(len(sequences)),
            C.c_int32(stride), C.c_int32(max_input_length), _lib.ptr(scores), _lib.ptr(out_len)))
        return [scores[b, :out_len[b]].copy() for b in range(len(sequences))]

    def step_logits(self, source_ids, target_in):
        """Teacher-forced per-step logits [T, batch, vocab] (source ids INCLUDE special tokens)."""
        batch = len(target_in)
        tgt = np.ascontiguousarray(target_in, np.int32)
        T = tgt.shape[1]
        if source_ids:
            lengths = np.array([len(s) for s in source_ids], np.int32)
            stride = int(lengths.max())
            ids = np.zeros((batch, stride), np.int32)
            for i, s in enumerate(source_ids):
                ids[i, :len(s)] = s
        else:
            lengths = np.zeros((batch,), np.int32)
            stride = 1
            ids = np.zeros((batch, 1), np.int32)
        logits = np.empty((T, batch, self.vocab_size), np.float32)
        _lib.check(self._L.ct2x_step_logits(C.c_void_p(self._h), _lib.ptr(ids), _lib.ptr(lengths),
                                            C.c_int32(batch), C.c_int32(stride), _lib.ptr(tgt),
                                            C.c_int32(T), _lib.ptr(logits)))
        return logits

    def step_taps(self, source_ids, target_in):
        """Debugging (tests): step_logits with every producer on the path tapped. Returns
        (logits [T, batch, vocab], [(name, array)]) - see ct2x_debug_step_taps."""
        batch = len(target_in)
        tgt = np.ascontiguousarray(target_in, np.int32)
        T = tgt.shape[1]
        lengths = np.array([len(s) for s in source_ids], np.int32)
        stride = int(lengths.max())
        ids = np.zeros((batch, stride), np.int32)
        for i, s in enumerate(source_ids):
            ids[i, :len(s)] = s
        logits = np.empty((T, batch, self.vocab_size), np.float32)
        _lib.check(self._L.ct2x_debug_step_taps(C.c_void_p(self._h), _lib.ptr(ids),
                                                _lib.ptr(lengths), C.c_int32(batch),
                                                C.c_int32(stride), _lib.ptr(tgt), C.c_int32(T),
                                                _lib.ptr(logits)))
        taps = []
        name = C.create_string_buffer(96)
        dtype, rows, cols = C.c_int32(), C.c_int64(), C.c_int64()
        for i in range(self._L.ct2x_debug_tap_count()):
            self._L.ct2x_debug_tap_info(C.c_int(i), name, C.c_int(96), C.byref(dtype),
                                        C.byref(rows), C.byref(cols))
            a = np.empty((rows.value, cols.value), np.float32 if dtype.value == 0 else np.int8)
            if a.size:
                self._L.ct2x_debug_tap_data(C.c_int(i), _lib.ptr(a))
            taps.append((name.value.decode(), a))
        return logits, taps

    def set_position_table(self, table):
        """Debugging (tests): replaces the sinusoidal position table [positions, depth]."""
        table = np.ascontiguousarray(table, np.float32)
        _lib.check(self._L.ct2x_debug_set_position_table(C.c_void_p(self._h), _lib.ptr(table),
                                                         C.c_int64(table.shape[0]),
                                                         C.c_int64(table.shape[1])))

    def release_session(self):
        """Frees the decoding session kept for reuse by the next translate call of the same shape
        (ct2x_release_session; the reference's counterpart is Translator.unload_model's allocator
        cache release)."""
        _lib.check(self._L.ct2x_release_session(C.c_void_p(self._h)))

    def cluster_fallbacks(self) -> int:
        """Groups decoded again on the launch-per-op step after a cluster of the decoder-stack
        launch waited in vain (ct2x_cluster_fallbacks)."""
        self._L.ct2x_cluster_fallbacks.restype = C.c_int64
        return int(self._L.ct2x_cluster_fallbacks(C.c_void_p(self._h)))

    def debug_inject_cluster_timeout(self):
        _lib.check(self._L.ct2x_debug_inject_cluster_timeout(C.c_void_p(self._h)))

    def profile_begin(self):
        _lib.check(self._L.ct2x_profile_begin(C.c_void_p(self._h)))

    def profile_end(self) -> dict:
        import json
        buf = C.create_string_buffer(1 << 16)
        _lib.check(self._L.ct2x_profile_end(C.c_void_p(self._h), buf, C.c_int(1 << 16)))
        return json.loads(buf.value.decode())

    def last_stats(self) -> dict:
        st = ExecStats()
        _lib.check(self._L.ct2x_last_stats(C.c_void_p(self._h), C.byref(st)))
        out = {name: getattr(st, name) for name, _ in ExecStats._fields_}
        self._L.ct2x_last_compactions.restype = C.c_int64
        out["compactions"] = int(self._L.ct2x_last_compactions(C.c_void_p(self._h)))
        self._L.ct2x_session_hits.restype = C.c_int64
        out["session_hits"] = int(self._L.ct2x_session_hits(C.c_void_p(self._h)))
        return out


def tp_selftest_multi(device_index: int, rank: int, world: int, unique_id: Optional[bytes] = None,
                      shm_segment: Optional[str] = None) -> dict:
    """ct2x_tp_selftest_multi: the tensor-parallel communicator across real ranks (one process per
    GPU) - every collective the engine calls and, where the communicator has peer-mapped mailboxes,
    the one-shot exchanges of a row-parallel Dense, checked bit for bit. Every rank calls it with
    the RCCL unique id of rank 0 (ct2x_tp_unique_id, carried by the launcher) or a shared-memory
    segment name. Raises Ct2xError with the failing piece."""
    L = _lib.lib()
    boxes, us = C.c_int32(0), C.c_float(0.0)
    _lib.check(L.ct2x_tp_selftest_multi(C.c_int(device_index), C.c_int(rank), C.c_int(world),
                                        C.c_char_p(unique_id) if unique_id is not None else None,
                                        shm_segment.encode() if shm_segment else None,
                                        C.byref(boxes), C.byref(us)))
    return {"ok": True, "rank": rank, "world": world, "mailboxes": bool(boxes.value),
            "backend": "rccl" if unique_id is not None else "shared-memory",
            "oneshot_us_per_dense": float(us.value) if boxes.value else None}


def tp_unique_id() -> bytes:
    """128 bytes of rank 0 that every rank of a tensor-parallel group needs (ct2x_tp_unique_id)."""
    buf = C.create_string_buffer(128)
    _lib.check(_lib.lib().ct2x_tp_unique_id(buf))
    return bytes(buf.raw)


class WhisperOptions(C.Structure):
    """struct ct2x_whisper_options = models::WhisperOptions
    (include/ctranslate2/models/whisper.h:11-58)."""
    _fields_ = [("beam_size", C.c_int32), ("patience", C.c_float), ("length_penalty", C.c_float),
                ("repetition_penalty", C.c_float), ("no_repeat_ngram_size", C.c_int32),
                ("max_length", C.c_int32), ("sampling_topk", C.c_int32),
                ("sampling_temperature", C.c_float), ("num_hypotheses", C.c_int32),
                ("return_scores", C.c_int32), ("return_no_speech_prob", C.c_int32),
                ("max_initial_timestamp_index", C.c_int32), ("suppress_blank", C.c_int32),
                ("num_suppress_tokens", C.c_int32), ("suppress_tokens", C.POINTER(C.c_int32))]


class WhisperGenerationResult:
    """ctranslate2.models.WhisperGenerationResult (python/cpp/whisper.cc:229-249)."""

    def __init__(self, sequences, sequences_ids, scores, no_speech_prob):
        self.sequences = sequences
        self.sequences_ids = sequences_ids
        self.scores = scores
        self.no_speech_prob = no_speech_prob

    def __repr__(self):
        return ("WhisperGenerationResult(sequences=%r, sequences_ids=%r, scores=%r, "
                "no_speech_prob=%r)" % (self.sequences, self.sequences_ids, self.scores,
                                         self.no_speech_prob))


class WhisperAlignmentResult:
    """ctranslate2.models.WhisperAlignmentResult (python/cpp/whisper.cc:251-263)."""

    def __init__(self, alignments, text_token_probs):
        self.alignments = alignments
        self.text_token_probs = text_token_probs

    def __repr__(self):
        return "WhisperAlignmentResult(alignments=%r, text_token_probs=%r)" % (
            self.alignments, self.text_token_probs)


class Whisper:
    """ctranslate2.models.Whisper (python/cpp/whisper.cc:265-420) on one MI355X: encode, generate,
    detect_language, align over float32 features [batch, n_mels, frames] (numpy)."""

    def __init__(self, model_path: str, device: str = "cuda", device_index: int = 0,
                 compute_type: str = "default"):
        if device not in ("cuda", "auto"):
            raise ValueError("ctranslate2_amd only implements the GPU device (device='cuda'); "
                             "got device=%r" % device)
        self._L = L = _lib.lib()
        L.ct2x_whisper_create.restype = C.c_void_p
        L.ct2x_whisper_create.argtypes = [C.c_char_p, C.c_int, C.c_char_p]
        L.ct2x_whisper_destroy.argtypes = [C.c_void_p]
        L.ct2x_whisper_vocab_token.restype = C.c_char_p
        L.ct2x_whisper_vocab_token.argtypes = [C.c_void_p, C.c_int32]
        L.ct2x_whisper_vocab_id.restype = C.c_int32
        L.ct2x_whisper_vocab_id.argtypes = [C.c_void_p, C.c_char_p]
        L.ct2x_whisper_vocab_size.argtypes = [C.c_void_p]
        self._h = L.ct2x_whisper_create(model_path.encode(), C.c_int(device_index),
                                        compute_type.encode())
        if not self._h:
            raise Ct2xError(L.ct2x_last_error().decode("utf-8", "replace"))
        v = [C.c_int32() for _ in range(3)]
        _lib.check(L.ct2x_whisper_info(C.c_void_p(self._h), *[C.byref(x) for x in v]))
        self.is_multilingual = bool(v[0].value)
        self.n_mels = v[1].value
        self.num_languages = v[2].value

    def __del__(self):
        self.close()

    def close(self):
        if getattr(self, "_h", None):
            self._L.ct2x_whisper_destroy(C.c_void_p(self._h))
            self._h = None

    def _token(self, i):
        t = self._L.ct2x_whisper_vocab_token(C.c_void_p(self._h), C.c_int32(int(i)))
        return t.decode("utf-8", "replace") if t is not None else None

    def _id(self, token):
        return int(self._L.ct2x_whisper_vocab_id(C.c_void_p(self._h), token.encode("utf-8")))

    def _ids(self, seq):
        return [self._id(t) if isinstance(t, str) else int(t) for t in seq]

    @staticmethod
    def _features(features):
        f = np.ascontiguousarray(features, dtype=np.float32)
        if f.ndim != 3:
            raise ValueError("Expected input features to have 3 dimensions, but got %d "
                             "dimension(s) instead" % f.ndim)
        return f

    def encode(self, features) -> np.ndarray:
        f = self._features(features)
        b, m, t = f.shape
        self._L.ct2x_whisper_d_model.restype = C.c_int64
        d_model = int(self._L.ct2x_whisper_d_model(C.c_void_p(self._h)))
        out = np.zeros((b, (t - 1) // 2 + 1, d_model), np.float32)
        dims = np.zeros(3, np.int64)
        _lib.check(self._L.ct2x_whisper_encode(C.c_void_p(self._h), _lib.ptr(f), C.c_int32(b),
                                               C.c_int32(m), C.c_int32(t), _lib.ptr(out),
                                               C.c_int64(out.size), _lib.ptr(dims)))
        n = int(dims[0] * dims[1] * dims[2])
        return out.reshape(-1)[:n].reshape(tuple(int(x) for x in dims)).copy()

    def generate(self, features, prompts, *, beam_size=5, patience=1.0, num_hypotheses=1,
                 length_penalty=1.0, repetition_penalty=1.0, no_repeat_ngram_size=0,
                 max_length=448, return_scores=False, return_no_speech_prob=False,
                 max_initial_timestamp_index=50, suppress_blank=True, suppress_tokens=(-1,),
                 sampling_topk=1, sampling_temperature=1.0) -> List[WhisperGenerationResult]:
        f = self._features(features)
        b, m, t = f.shape
        prompts = [self._ids(p) for p in prompts]
        if len(prompts) != b:
            raise ValueError("one prompt per batch entry is required")
        lengths = np.array([len(p) for p in prompts], np.int32)
        stride = max(1, int(lengths.max()) if b else 1)
        ids = np.zeros((b, stride), np.int32)
        for i, p in enumerate(prompts):
            ids[i, :len(p)] = p
        sup = np.array(list(suppress_tokens or ()), np.int32)
        o = WhisperOptions(beam_size, patience, length_penalty, repetition_penalty,
                           no_repeat_ngram_size, max_length, sampling_topk, sampling_temperature,
                           num_hypotheses, int(return_scores), int(return_no_speech_prob),
                           max_initial_timestamp_index, int(suppress_blank), len(sup),
                           sup.ctypes.data_as(C.POINTER(C.c_int32)))
        max_len = max(1, max_length)
        out = np.zeros((b, num_hypotheses, max_len), np.int32)
        out_len = np.zeros((b, num_hypotheses), np.int32)
        scores = np.zeros((b, num_hypotheses), np.float32)
        nsp = np.zeros(b, np.float32)
        _lib.check(self._L.ct2x_whisper_generate(
            C.c_void_p(self._h), _lib.ptr(f), C.c_int32(b), C.c_int32(m), C.c_int32(t),
            _lib.ptr(ids), _lib.ptr(lengths), C.c_int32(stride), C.byref(o), C.c_int32(max_len),
            _lib.ptr(out), _lib.ptr(out_len), _lib.ptr(scores), _lib.ptr(nsp)))
        results = []
        for i in range(b):
            seq_ids = [out[i, k, :out_len[i, k]].tolist() for k in range(num_hypotheses)
                       if out_len[i, k] >= 0]
            results.append(WhisperGenerationResult(
                [[self._token(x) for x in s] for s in seq_ids], seq_ids,
                [float(scores[i, k]) for k in range(len(seq_ids))] if return_scores else [],
                float(nsp[i]) if return_no_speech_prob else 0.0))
        return results

    def detect_language(self, features):
        f = self._features(features)
        b, m, t = f.shape
        n = self.num_languages
        ids = np.zeros((b, n), np.int32)
        probs = np.zeros((b, n), np.float32)
        _lib.check(self._L.ct2x_whisper_detect_language(
            C.c_void_p(self._h), _lib.ptr(f), C.c_int32(b), C.c_int32(m), C.c_int32(t),
            C.c_int32(n), _lib.ptr(ids), _lib.ptr(probs)))
        return [[(self._token(ids[i, j]), float(probs[i, j])) for j in range(n)] for i in range(b)]

    def align(self, features, start_sequence, text_tokens, num_frames, *, median_filter_width=7):
        f = self._features(features)
        b, m, t = f.shape
        start = np.array(self._ids(start_sequence), np.int32)
        text_tokens = [self._ids(x) for x in text_tokens]
        lengths = np.array([len(x) for x in text_tokens], np.int32)
        stride = max(1, int(lengths.max()) if b else 1)
        ids = np.zeros((b, stride), np.int32)
        for i, x in enumerate(text_tokens):
            ids[i, :len(x)] = x
        nf = np.array(list(num_frames), np.int32)
        if len(nf) != b:
            raise ValueError("Invalid batch size for argument num_frames")
        cap = int(stride + 2 + t)
        pairs = np.zeros((b, cap, 2), np.int32)
        n_pairs = np.zeros(b, np.int32)
        probs = np.zeros((b, stride), np.float32)
        _lib.check(self._L.ct2x_whisper_align(
            C.c_void_p(self._h), _lib.ptr(f), C.c_int32(b), C.c_int32(m), C.c_int32(t),
            _lib.ptr(start), C.c_int32(len(start)), _lib.ptr(ids), _lib.ptr(lengths),
            C.c_int32(stride), _lib.ptr(nf), C.c_int32(median_filter_width), C.c_int32(cap),
            _lib.ptr(pairs), _lib.ptr(n_pairs), _lib.ptr(probs)))
        return [WhisperAlignmentResult([tuple(p) for p in pairs[i, :n_pairs[i]].tolist()],
                                       probs[i, :lengths[i]].tolist()) for i in range(b)]
