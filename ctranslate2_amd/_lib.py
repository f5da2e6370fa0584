"""ctypes binding of libct2_mi355.so (the C ABI in include/ct2_mi355.h).

The library is the product: there is no Python/CPU fallback. Importing this module without
the built extension raises; calling a compute entry point without a GPU returns the HIP
error from the runtime.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CT2X_LIBRARY: another build of the same library (A/B timing of two builds inside one gpurun call)
LIB_PATH = os.environ.get("CT2X_LIBRARY") or os.path.join(_HERE, "libct2_mi355.so")

_lib = None


class Ct2xError(RuntimeError):
    pass


# int (*step_callback)(void* user, int64 step, int64 batch_id, int64 hypothesis_id, int32 token_id,
#                       int32 has_score, float score, int32 is_last)
STEP_CALLBACK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32,
                            C.c_int32, C.c_float, C.c_int32)


class TranslationOptions(C.Structure):
    """struct ct2x_translation_options (include/ct2_mi355.h)."""
    _fields_ = [
        ("beam_size", C.c_int32),
        ("max_decoding_length", C.c_int32),
        ("min_decoding_length", C.c_int32),
        ("num_hypotheses", C.c_int32),
        ("return_scores", C.c_int32),
        ("return_end_token", C.c_int32),
        ("max_batch_size", C.c_int32),
        ("max_input_length", C.c_int32),
        ("length_penalty", C.c_float),
        ("patience", C.c_float),
        ("use_graph", C.c_int32),
        ("num_streams", C.c_int32),
        ("sampling_topk", C.c_int32),
        ("sampling_topp", C.c_float),
        ("sampling_temperature", C.c_float),
        ("seed", C.c_uint32),
        ("repetition_penalty", C.c_float),
        ("no_repeat_ngram_size", C.c_int32),
        ("num_suppress_sequences", C.c_int32),
        ("end_token_plus_one", C.c_int32),
        ("suppress_ids", C.POINTER(C.c_int32)),
        ("suppress_lengths", C.POINTER(C.c_int32)),
        ("return_alternatives", C.c_int32),
        ("min_alternative_expansion_prob", C.c_float),
        ("prefix_bias_beta", C.c_float),
        ("coverage_penalty", C.c_float),
        ("attention_stride", C.c_int32),
        ("static_prompt_length", C.c_int32),
        ("attention_out", C.POINTER(C.c_float)),
        ("step_callback", STEP_CALLBACK),
        ("step_callback_data", C.c_void_p),
        ("logits_out", C.POINTER(C.c_float)),
        ("logits_counts_out", C.POINTER(C.c_int32)),
        ("logits_steps", C.c_int32),
        ("include_prompt_in_result", C.c_int32),
    ]


class ExecStats(C.Structure):
    """struct ct2x_exec_stats (include/ct2_mi355.h)."""
    _fields_ = [
        ("num_tokens", C.c_int64),
        ("num_examples", C.c_int64),
        ("decode_steps", C.c_int64),
        ("kernels_per_step", C.c_int64),
        ("total_ms", C.c_double),
        ("encode_ms", C.c_double),
        ("decode_ms", C.c_double),
    ]


def lib():
    """Loads the HIP extension (once). Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    try:
        # Share PyTorch's HIP runtime when torch is in the process (same libamdhip64 soname),
        # so device pointers from torch tensors are valid in this library.
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the C ABI itself
        pass
    L = C.CDLL(LIB_PATH, mode=os.RTLD_GLOBAL | os.RTLD_NOW)
    L.ct2x_last_error.restype = C.c_char_p
    L.ct2x_thread_stream.restype = C.c_void_p
    L.ct2x_allocated_bytes.restype = C.c_int64
    if hasattr(L, "ct2x_translator_create"):
        L.ct2x_translator_create.restype = C.c_void_p
        L.ct2x_translator_create.argtypes = [C.c_char_p, C.c_int, C.c_char_p]
        L.ct2x_translator_destroy.argtypes = [C.c_void_p]
        L.ct2x_translator_create_tp.restype = C.c_void_p
        L.ct2x_translator_create_tp.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int,
                                                C.c_char_p]
        L.ct2x_translator_create_tp_shm.restype = C.c_void_p
        L.ct2x_translator_create_tp_shm.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int,
                                                    C.c_char_p]
        L.ct2x_tp_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                   C.POINTER(C.c_int32), C.c_char_p, C.c_int]
        L.ct2x_translator_create_tp_local.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int,
                                                      C.POINTER(C.c_void_p)]
        L.ct2x_tp_unique_id.argtypes = [C.c_char_p]
        L.ct2x_vocab_token.restype = C.c_char_p
        L.ct2x_vocab_token.argtypes = [C.c_void_p, C.c_int, C.c_int32]
        L.ct2x_vocab_id.restype = C.c_int32
        L.ct2x_vocab_id.argtypes = [C.c_void_p, C.c_int, C.c_char_p]
        L.ct2x_vocab_size.argtypes = [C.c_void_p, C.c_int]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise Ct2xError(lib().ct2x_last_error().decode("utf-8", "replace"))


def ptr(x):
    """Device/host pointer of a torch tensor / numpy array / int / None as c_void_p."""
    if x is None:
        return C.c_void_p(0)
    if isinstance(x, int):
        return C.c_void_p(x)
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    if hasattr(x, "ctypes"):
        return x.ctypes.data_as(C.c_void_p)
    raise TypeError("cannot take a pointer of %r" % type(x))
