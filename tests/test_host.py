"""CPU tests of the host side: the C-ABI library loads and exports every symbol the header
declares; model.bin / vocabulary parsing; the synthetic model writer round-trips; error behaviour
without a GPU is loud."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

from ctranslate2_amd import _lib, model_builder as mb

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = os.path.join(HERE, "golden")


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "ct2_mi355.h")).read()
    declared = set(re.findall(r"\b(ct2x_[a-z0-9_]+)\s*\(", header))
    declared -= {"ct2x_translation_options", "ct2x_exec_stats"}
    assert len(declared) > 40
    L = _lib.lib()
    missing = [name for name in sorted(declared) if not hasattr(L, name)]
    assert not missing, missing
    assert L.ct2x_version() == 1


def test_ctypes_structs_have_the_layout_of_the_header(tmp_path):
    """The ctypes mirrors of the C ABI's structs (ctranslate2_amd/_lib.py, the Whisper options in
    __init__.py) carry every field of the header at the header's offset: a C program prints
    offsetof of every member, the ctypes classes must agree."""
    import subprocess
    import ctranslate2_amd as ct2
    header = open(os.path.join(ROOT, "include", "ct2_mi355.h")).read()
    structs = {"ct2x_translation_options": _lib.TranslationOptions, "ct2x_exec_stats": _lib.ExecStats,
               "ct2x_whisper_options": ct2.WhisperOptions}
    lines = []
    for name, cls in structs.items():
        body = re.search(r"struct %s\s*\{(.*?)\n\};" % name, header, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            m = re.search(r"\(\*\s*(\w+)\)\s*\(", decl) or re.search(r"(\w+)\s*(\[\d+\])?$", decl)
            fields.append(m.group(1))
        assert fields == [f[0] for f in cls._fields_], (name, fields)
        for f in fields:
            lines.append('printf("%s.%s %%zu\\n", offsetof(struct %s, %s));' % (name, f, name, f))
        lines.append('printf("%s %%zu\\n", sizeof(struct %s));' % (name, name))
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "ct2_mi355.h"\n'
                   'int main(void) {\n%s\nreturn 0; }\n' % "\n".join(lines))
    exe = str(tmp_path / "layout")
    subprocess.check_call(["gcc", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe])
    got = dict(ln.split() for ln in subprocess.check_output([exe]).decode().splitlines())
    for name, cls in structs.items():
        assert int(got[name]) == C.sizeof(cls), name
        for f, _t in cls._fields_:
            assert int(got["%s.%s" % (name, f)]) == getattr(cls, f).offset, (name, f)


def test_no_gpu_is_a_loud_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import ctranslate2_amd as ct2
    with pytest.raises(ct2.Ct2xError):
        ct2.Translator(os.path.join(GOLDEN, "models", "aren-transliteration-i8"))
    L = _lib.lib()
    assert L.ct2x_device_count() == 0
    p = C.c_void_p()
    assert L.ct2x_malloc(C.byref(p), C.c_size_t(1024)) != 0
    assert b"HIP" in L.ct2x_last_error()


def inspect(path):
    L = _lib.lib()
    spec = C.create_string_buffer(128)
    bv, rev, na = C.c_int32(), C.c_int32(), C.c_int32()
    n = L.ct2x_inspect_model(path.encode(), spec, 128, C.byref(bv), C.byref(rev), C.byref(na))
    assert n >= 0, L.ct2x_last_error()
    out = {}
    for i in range(n):
        name = C.create_string_buffer(512)
        dt, rank, nbytes = C.c_int32(), C.c_int32(), C.c_int64()
        dims = (C.c_int64 * 8)()
        _lib.check(L.ct2x_inspect_variable(path.encode(), i, name, 512, C.byref(dt), C.byref(rank),
                                           dims, C.byref(nbytes)))
        out[name.value.decode()] = (dt.value, tuple(dims[:rank.value]), nbytes.value)
    return spec.value.decode(), bv.value, rev.value, na.value, out


def test_parse_reference_fixture_model():
    """Legacy binary version 2 fixture (item-size dtype encoding, no alias section)."""
    spec, bv, rev, na, variables = inspect(os.path.join(GOLDEN, "models", "aren-transliteration-i8"))
    assert (spec, bv, rev, na) == ("TransformerBase", 2, 2, 0)
    assert len(variables) == 269
    assert variables["decoder/embeddings/weight"] == (1, (43, 32), 1376)
    assert variables["decoder/layer_0/self_attention/linear_0/weight_scale"] == (0, (96,), 384)


def test_model_builder_round_trip(tmp_path):
    d = str(tmp_path / "m")
    mb.build_transformer(d, 2, 2, 64, 128, 4, 100, seed=0)
    spec, bv, rev, na, variables = inspect(d)
    assert (spec, bv, rev) == ("TransformerSpec", 6, 7)
    assert variables["decoder/num_heads"] == (2, (), 2)          # 0-d int16 attribute
    assert variables["decoder/projection/weight"] == (1, (100, 64), 6400)
    assert variables["encoder/layer_1/ffn/linear_0/weight_scale"] == (0, (128,), 512)
    cfg = json.load(open(os.path.join(d, "config.json")))
    assert cfg["decoder_start_token"] == "<s>" and cfg["add_source_eos"] is False
    # determinism: same seed -> identical bytes
    d2 = str(tmp_path / "m2")
    mb.build_transformer(d2, 2, 2, 64, 128, 4, 100, seed=0)
    assert open(os.path.join(d, "model.bin"), "rb").read() == open(os.path.join(d2, "model.bin"), "rb").read()


def test_model_builder_quantisation_matches_spec_formula():
    """python/ctranslate2/specs/model_spec.py:233-240 (KAT: python/tests/test_spec.py:101-164)."""
    w = np.array([[-10, -3, 5, 2], [0, 0, 0, 0], [0.1, -0.25, 0.05, 0.25]], np.float32)
    q, s = mb.quantize_int8(w)
    assert np.array_equal(q, np.array([[-127, -38, 64, 25], [0, 0, 0, 0], [51, -127, 25, 127]], np.int8))
    np.testing.assert_allclose(s, [12.7, 1.0, 508.0], rtol=1e-6)


def test_corrupt_model_is_rejected(tmp_path):
    d = tmp_path / "bad"
    d.mkdir()
    (d / "model.bin").write_bytes(b"\x07\x00\x00\x00garbage")
    L = _lib.lib()
    n = L.ct2x_inspect_model(str(d).encode(), None, 0, None, None, None)
    assert n < 0 and b"binary version" in L.ct2x_last_error()
    src = open(os.path.join(GOLDEN, "models", "aren-transliteration-i8", "model.bin"), "rb").read()
    (d / "model.bin").write_bytes(src[:5000])
    assert L.ct2x_inspect_model(str(d).encode(), None, 0, None, None, None) < 0
    assert L.ct2x_inspect_model(str(tmp_path / "missing").encode(), None, 0, None, None, None) < 0


def test_cpp_api_header_builds_and_fails_loudly_without_gpu():
    """include/ct2_mi355.hpp (the ctranslate2::Translator / Generator surface over the C ABI)
    compiles warning-free with plain g++ and links against the in-tree library; with no GPU the
    constructor throws the HIP error - there is no CPU arm to fall back to."""
    import torch
    import _cppapi
    _cppapi.build()
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    rc, out, err = _cppapi.run("translate", os.path.join(GOLDEN, "models", "aren-transliteration-i8"),
                               2, "a")
    assert rc == 2 and "HIP" in err and out == ""


def test_host_beam_bookkeeping_matches_the_oracle():
    """The rule the host replays on the recorded candidates (ct2x_host_beam_step_select =
    beam_step_select, src/decoding.cc:618-655) picks the same next beams as the oracle restatement
    orc_beam_select - which is what the device kernel is tested against on the GPU - and reports
    exactly the ended primaries as finished; the last step finishes every primary."""
    import sys
    sys.path.insert(0, HERE)
    import _oracle as O
    L = _lib.lib()
    L.ct2x_host_finalize_score.restype = C.c_float
    rng = np.random.default_rng(3)
    vocab, eos = 40, 2
    for beam in (1, 2, 4, 7):
        nc = 2 * beam
        for trial in range(200):
            words = rng.integers(0, vocab, (1, nc)).astype(np.int32)
            words[rng.random((1, nc)) < 0.4] = eos
            plen = int(rng.integers(0, 3))
            step = int(rng.integers(0, 5))
            ids = (rng.integers(0, beam, (1, nc)) * vocab + words).astype(np.int32)
            scores = -np.sort(rng.random((1, nc)).astype(np.float32), axis=1)
            # candidates after the device's prefix handling are what the host sees
            cw, cs, co, ni, ns, pa = O.beam_select(scores, ids, beam, vocab, eos,
                                                   [list(range(3, 3 + plen))], step)
            fin = (C.c_int32 * beam)()
            nxt = (C.c_int32 * beam)()
            w = np.ascontiguousarray(cw[0])
            n = L.ct2x_host_beam_step_select(beam, step, 0, eos, plen,
                                             w.ctypes.data_as(C.POINTER(C.c_int32)), fin, nxt)
            assert n >= 0
            assert [w[j] for j in nxt] == ni.tolist()
            assert [cs[0][j] for j in nxt] == ns.tolist()
            ended = [k for k in range(beam) if w[k] == eos and step >= plen]
            assert list(fin)[:n] == ended
            n_last = L.ct2x_host_beam_step_select(beam, step, 1, eos, plen,
                                                  w.ctypes.data_as(C.POINTER(C.c_int32)), fin, nxt)
            assert n_last == beam and list(fin) == list(range(beam))
    # finalize_hypothesis_score: score / length^penalty (src/decoding.cc:192-207)
    assert abs(L.ct2x_host_finalize_score(C.c_float(-6.0), C.c_float(3.0), C.c_float(1.0)) + 2.0) < 1e-6
    assert L.ct2x_host_finalize_score(C.c_float(-6.0), C.c_float(3.0), C.c_float(0.0)) == -6.0


REFERENCE = "/root/reference"


def _nm(path, flag):
    import subprocess
    out = subprocess.check_output(["nm", flag, path], text=True)
    return {line.split()[-1] for line in out.splitlines() if line.strip()}


@pytest.mark.skipif(not os.path.exists(os.path.join(REFERENCE, "include", "ctranslate2", "ops", "ops.h")),
                    reason="needs the reference headers (/root/reference)")
def test_integration_stubs_match_the_reference_headers(tmp_path):
    """The drop-in boundary, proven at link level. integration/ops_mi355_stub.cc (the Level-1
    stubs of INTEGRATION.md) compiles warning-free against the reference's REAL headers, and it
    defines, with identical mangled names, the Device::CUDA float symbols that the reference's own
    op front-ends (src/ops/*.cc compiled with the CUDA arm of DEVICE_DISPATCH enabled,
    -DCT2_WITH_CUDA) leave undefined - i.e. exactly what src/ops/*_gpu.cu provides today.
    Reference sources are only compiled where they lie; objects go to a temporary directory."""
    import subprocess
    inc = ["-I" + os.path.join(ROOT, "oracle", "shim"), "-I" + os.path.join(REFERENCE, "include"),
           "-I" + os.path.join(REFERENCE, "src"), "-I" + os.path.join(REFERENCE, "third_party"),
           "-I" + os.path.join(ROOT, "include")]
    stub = str(tmp_path / "stub.o")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-c"] + inc +
                          [os.path.join(ROOT, "integration", "ops_mi355_stub.cc"), "-o", stub])
    defined = _nm(stub, "--defined-only")
    checked = 0
    for op in ("quantize", "dequantize", "layer_norm", "rms_norm", "softmax", "rotary", "topk", "gather"):
        obj = str(tmp_path / (op + ".o"))
        subprocess.check_call(["g++", "-std=c++17", "-DCT2_WITH_CUDA", "-w", "-c"] + inc +
                              [os.path.join(REFERENCE, "src", "ops", op + ".cc"), "-o", obj])
        # undefined Device::CUDA (enumerator 1) member templates of namespace ops whose builtin
        # template arguments include float and no int16 (the float arm: "f", "fa", "af", "fi");
        # the half / bfloat16 arms and the integer payloads of Gather forward the same way
        needed = set()
        for sym in _nm(obj, "-u"):
            m = re.search(r"11ctranslate23ops.*ILNS_6DeviceE1E([a-z]+)EE", sym)
            if m and "f" in m.group(1) and "s" not in m.group(1):
                needed.add(sym)
        assert needed, op
        missing = needed - defined
        assert not missing, (op, sorted(missing))
        checked += len(needed)
    assert checked >= 9
    # primitives<Device::CUDA> / cross_device_primitives the front-ends reach for on this path
    for src, pattern in (("ops/gemm", r"10primitivesILNS_6DeviceE1EE4gemmIaiEE"),
                         ("decoding_utils", r"10primitivesILNS_6DeviceE1EE24penalize_previous_tokensIfEE"),
                         ("storage_view", r"23cross_device_primitivesILNS_6DeviceE(1ELS1_0|0ELS1_1)EE4copyIfEE")):
        obj = str(tmp_path / (os.path.basename(src) + ".o"))
        subprocess.check_call(["g++", "-std=c++17", "-DCT2_WITH_CUDA", "-w", "-c"] + inc +
                              [os.path.join(REFERENCE, "src", src + ".cc"), "-o", obj])
        needed = {sym for sym in _nm(obj, "-u") if re.search(pattern, sym)}
        assert needed, src
        assert needed <= defined, (src, sorted(needed - defined))


def test_host_sinusoidal_table_matches_the_reference():
    """The position table is host work (src/layers/common.cc:204-229 runs on the CPU even for
    GPU models). With an MKL runtime installed the backend uses VML as the reference does and the
    table must equal the reference's bit for bit (tests/golden/tap_golden.npz holds tables made by
    the real reference); with libm it may differ by one ulp in a few percent of the entries."""
    import ctypes as C
    from ctranslate2_amd import _lib
    L = _lib.lib()
    L.ct2x_host_math_backend.restype = C.c_char_p
    backend = L.ct2x_host_math_backend().decode()
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tap_golden.npz"))
    for depth in (32, 64, 256, 512):
        want = z["postable/%d" % depth]
        got = np.empty_like(want)
        assert L.ct2x_host_sinusoidal_table(C.c_int64(want.shape[0]), C.c_int64(depth),
                                            got.ctypes.data_as(C.c_void_p)) == 0
        if backend in ("mkl-vml", "vml-restated"):
            assert np.array_equal(got, want), depth
        else:
            assert np.mean(got == want) > 0.9
            np.testing.assert_allclose(got, want, rtol=2e-7, atol=1e-7)


def test_host_sin_cos_is_the_restated_vml():
    """The product's host sin / cos (host_math.cc: MKL VML's HA kernels restated) equal the oracle's
    restatement - which tests/test_oracle.py pins bit for bit against the MKL library itself - on
    table arguments, random arguments and a whole binade; beyond 10000 rad it says so."""
    import ctypes as C
    from ctranslate2_amd import _lib
    import _oracle as O
    L = _lib.lib()
    L.ct2x_host_math_backend.restype = C.c_char_p
    if L.ct2x_host_math_backend().decode() != "vml-restated":
        pytest.skip("CT2X_HOST_MATH selects another backend")
    rng = np.random.default_rng(3)
    xs = [rng.uniform(-10000, 10000, 1_000_000).astype(np.float32),
          (np.arange(1, 4001, dtype=np.float32)[:, None] *
           np.exp(-np.float32(0.036) * np.arange(256, dtype=np.float32))[None, :]).astype(np.float32).ravel(),
          np.arange(0x3F800000, 0x3F800000 + (1 << 20), dtype=np.uint32).view(np.float32),
          np.array([0.0, -0.0, 1e-30, np.pi, -np.pi, 10000.0, -10000.0], np.float32)]
    for x in xs:
        s, c = np.empty_like(x), np.empty_like(x)
        assert L.ct2x_host_sin_cos(C.c_int64(len(x)), x.ctypes.data_as(C.c_void_p),
                                   s.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p)) == 0
        so, co = O.vml_sin(x), O.vml_cos(x)
        assert np.array_equal(s.view(np.uint32), so.view(np.uint32))
        assert np.array_equal(c.view(np.uint32), co.view(np.uint32))


def test_host_rotary_tables_match_the_reference():
    """RotaryEmbeddings::initialize (src/layers/attention_layer.cc:254-330) is host work: plain,
    interleaved, Linear and Llama3 scaling. tests/golden/rotary_golden.npz holds the tables of the
    REAL reference (tests/golden/make_rotary_golden.py drives layers::RotaryEmbeddings through
    oracle/ref_capi.cc). With the MKL VML backend the tables must be identical bit for bit; with
    libm they may differ by one ulp in a few percent of the entries (and by sin / cos argument
    reduction at large angles: 2e-7 absolute)."""
    import ctypes as C
    import json
    from ctranslate2_amd import _lib
    L = _lib.lib()
    L.ct2x_host_math_backend.restype = C.c_char_p
    backend = L.ct2x_host_math_backend().decode()
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rotary_golden.npz"))
    cases = json.loads(bytes(z["meta"]).decode())
    assert any(kw.get("scaling_type") == 2 for kw in cases.values())
    for name, kw in cases.items():
        want_s, want_c = z[name + "/sin"], z[name + "/cos"]
        got_s, got_c = np.empty_like(want_s), np.empty_like(want_c)
        rc = L.ct2x_host_rotary_tables(
            C.c_int64(kw["dim"]), C.c_int(int(kw.get("interleave", False))),
            C.c_int(kw.get("scaling_type", -1)), C.c_float(kw.get("scaling_factor", 1.0)),
            C.c_float(kw.get("base", 10000.0)), C.c_float(kw.get("low_freq_factor", 1.0)),
            C.c_float(kw.get("high_freq_factor", 4.0)),
            C.c_int64(kw.get("original_max_position_embeddings", 0)), C.c_int64(kw["positions"]),
            got_s.ctypes.data_as(C.c_void_p), got_c.ctypes.data_as(C.c_void_p))
        assert rc == 0, _lib.last_error() if hasattr(_lib, "last_error") else rc
        if backend in ("mkl-vml", "vml-restated"):
            assert np.array_equal(got_s, want_s), name
            assert np.array_equal(got_c, want_c), name
        else:
            assert np.mean(got_s == want_s) > 0.9 and np.mean(got_c == want_c) > 0.9, name
            np.testing.assert_allclose(got_s, want_s, rtol=0, atol=3e-7)
            np.testing.assert_allclose(got_c, want_c, rtol=0, atol=3e-7)


def test_hot_kernels_do_not_spill():
    """The decode-path kernels must keep their working set in registers: a spill turns into HBM
    write traffic per thread (found once with the PMC WRITE_SIZE counter: 15 spilled VGPRs of the
    long-K streaming GEMM = 15 MB per launch, profiles/r02b_pmc_stream_gemm.json). Compiles the
    three hot translation units for gfx950 and reads the kernel metadata."""
    import re
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "ctranslate2_amd", "csrc")
    hot = {"gemm_s8_stream.hip": "stream_kernel", "gemm_s8.hip": "skinny_kernel",
           "attention_exact.hip": "attn_x_decode_kernel", "gemm_s8_big.hip": "big_kernel",
           # round 3: the 16-bit decode GEMM and the 16-bit memory K/V attention
           "gemm_h.hip": "gemm_h_rows_kernel", "attention_cross_h.hip": "xattn_h_kernel"}
    minimum = {"attention_cross_h.hip": 2}
    with tempfile.TemporaryDirectory() as tmp:
        for src, needle in hot.items():
            out = os.path.join(tmp, src + ".s")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17",
                                   "-ffp-contract=off", "-I" + os.path.join(root, "include"), "-mllvm",
                                   "-amdgpu-kernarg-preload-count=16", "-S", "--cuda-device-only", "-o", out,
                                   os.path.join(csrc, src)], stderr=subprocess.DEVNULL)
            text = open(out).read()
            checked = 0
            for block in text.split("  - .agpr_count:")[1:]:
                name = re.search(r"\.name:\s+(\S+)", block).group(1)
                if needle not in name:
                    continue
                checked += 1
                assert int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", block).group(1)) == 0, name
                assert int(re.search(r"\.vgpr_spill_count:\s+(\d+)", block).group(1)) == 0, name
            assert checked >= minimum.get(src, 5), (src, checked)
        # round 5, the decoder-stack launches: a spill RELOAD there is a vmcnt(0) wait that drains the
        # weight requests left in flight across a phase (45k -> 75k tokens/s at beam 4
        # came from removing them). One row per cluster: none at all; several rows: at most the
        # handful the build of the round ended with.
        out = os.path.join(tmp, "decoder_cluster.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17",
                               "-ffp-contract=off", "-I" + os.path.join(root, "include"), "-mllvm",
                               "-amdgpu-kernarg-preload-count=16", "-S", "--cuda-device-only", "-o", out,
                               os.path.join(csrc, "decoder_cluster.hip")], stderr=subprocess.DEVNULL)
        seen = {}
        for block in open(out).read().split("  - .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", block).group(1)
            seen[name] = (int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", block).group(1)),
                          int(re.search(r"\.vgpr_spill_count:\s+(\d+)", block).group(1)))
        one_row = [v for k, v in seen.items() if "decoder_cluster_kernel" in k]
        rows = [v for k, v in seen.items() if "decoder_cluster_beam_kernel" in k]
        assert len(one_row) == 2 and all(v == (0, 0) for v in one_row), seen
        assert len(rows) == 1 and rows[0][0] <= 64 and rows[0][1] <= 16, seen


def test_translator_from_memory_is_checked_on_the_host():
    """ct2x_translator_create_from_memory (Model::load on a ModelMemoryReader / the Level-1.5
    hand-over): the variable table is read on the host before anything touches a device - an
    unknown data type id, a missing target vocabulary and (here, without a GPU) the engine's own
    loud HIP error all come back as NULL + ct2x_last_error, never as a crash or a fallback."""
    L = _lib.lib()

    class Var(C.Structure):
        _fields_ = [("name", C.c_char_p), ("dtype", C.c_int32), ("rank", C.c_int32),
                    ("shape", C.POINTER(C.c_int64)), ("data", C.c_void_p), ("nbytes", C.c_uint64)]
    L.ct2x_translator_create_from_memory.restype = C.c_void_p
    L.ct2x_last_error.restype = C.c_char_p
    data = np.zeros(8, np.float32)
    shape = (C.c_int64 * 2)(2, 4)
    tokens = (C.c_char_p * 3)(b"<unk>", b"<s>", b"</s>")
    bad = Var(b"encoder/x", 7, 2, shape, data.ctypes.data, 32)
    h = L.ct2x_translator_create_from_memory(b"TransformerSpec", 1, 6, None, C.byref(bad), C.c_int64(1), None,
                                             C.c_int64(0), tokens, C.c_int64(3), 0, b"float32")
    assert not h and b"unknown data type id 7" in L.ct2x_last_error()
    # the byte count the caller states must be the shape's (a short buffer is never read past), no
    # negative rank or dimension, the binary version as Model::load checks it
    for var, version, text in ((Var(b"encoder/x", 0, 2, shape, data.ctypes.data, 16), 6, b"invalid payload size"),
                               (Var(b"encoder/x", 0, -1, shape, data.ctypes.data, 32), 6, b"invalid rank"),
                               (Var(b"encoder/x", 0, 2, (C.c_int64 * 2)(2, -4), data.ctypes.data, 32), 6, b"invalid shape"),
                               (Var(b"encoder/x", 0, 2, (C.c_int64 * 2)(1 << 30, 1 << 30), data.ctypes.data, 32), 6, b"invalid shape"),
                               (Var(b"encoder/x", 0, 2, shape, data.ctypes.data, 32), 7, b"Unsupported model binary version")):
        h = L.ct2x_translator_create_from_memory(b"TransformerSpec", 1, version, None, C.byref(var), C.c_int64(1), None,
                                                 C.c_int64(0), tokens, C.c_int64(3), 0, b"float32")
        assert not h and text in L.ct2x_last_error(), (text, L.ct2x_last_error())
    ok = Var(b"encoder/x", 0, 2, shape, data.ctypes.data, 32)
    h = L.ct2x_translator_create_from_memory(b"TransformerSpec", 1, 6, b"{not json", C.byref(ok), C.c_int64(1),
                                             None, C.c_int64(0), tokens, C.c_int64(3), 0, b"float32")
    assert not h and L.ct2x_last_error()
    import torch
    if not torch.cuda.is_available():
        h = L.ct2x_translator_create_from_memory(b"TransformerSpec", 1, 6, None, C.byref(ok), C.c_int64(1), None,
                                                 C.c_int64(0), tokens, C.c_int64(3), 0, b"float32")
        assert not h and b"HIP" in L.ct2x_last_error()


def test_streaming_gemm_plan_is_host_logic():
    """Which (n, k) get a persistent streaming form / a fragment-major weight copy is decided on the
    host (gemm_s8_stream.hip plan_stream; 256 CUs assumed without a device): the Llama-3-8B shapes
    do, small or ragged weights do not - and the answer is the same for the weight packer and the
    launcher (a packed weight must always find its kernel)."""
    from ctranslate2_amd import _lib
    L = _lib.lib()
    import ctypes as C
    L.ct2x_gemm_s8_stream_packable.restype = C.c_int
    q = lambda n, k, dual=0: L.ct2x_gemm_s8_stream_packable(C.c_int64(n), C.c_int64(k), C.c_int(dual))
    assert q(6144, 4096) and q(4096, 4096) and q(4096, 14336) and q(128256, 4096) and q(14336, 4096, 1)
    assert not q(512, 512) and not q(2048, 512)          # Transformer-base: cache resident, one-shot kernel
    assert not q(8208, 4096)                             # no even split into >= 192 workgroups
    assert not q(4096, 4160)                             # K not a multiple of 16 waves x 64 bytes
    assert not q(14336, 14336, 1)                        # gated pair at K = 14336: the ring would spill


def test_negative_dtw_against_the_reference():
    """ct2x_host_negative_dtw (Whisper word alignment) against the reference's own
    negative_dtw (src/dtw.cc) driven through oracle/ref_capi.cc: identical paths on random cost
    matrices, on matrices full of ties (the tie rule decides the path) and on degenerate shapes."""
    import ctypes as C
    import _refshim as R
    from ctranslate2_amd import _lib
    L, ref = _lib.lib(), R.lib()
    for f in (L.ct2x_host_negative_dtw, ref.ct2ref_negative_dtw):
        f.restype = C.c_int64
        f.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(0)
    cases = [rng.standard_normal((n, m)).astype(np.float32) for n, m in [(1, 1), (1, 9), (7, 1), (5, 8),
                                                                         (40, 150), (33, 33)]]
    cases += [np.zeros((6, 9), np.float32), rng.integers(0, 2, (12, 20)).astype(np.float32),
              np.eye(10, dtype=np.float32)]
    for x in cases:
        n, m = x.shape
        got, want = [], []
        for f, out in ((L.ct2x_host_negative_dtw, got), (ref.ct2ref_negative_dtw, want)):
            pi, pj = np.zeros(n + m, np.int64), np.zeros(n + m, np.int64)
            k = f(x.ctypes.data, n, m, pi.ctypes.data, pj.ctypes.data)
            assert k > 0
            out.extend(zip(pi[:k].tolist(), pj[:k].tolist()))
        assert got == want
        assert got[0] == (0, 0) and got[-1] == (n - 1, m - 1)


def test_model_bin_reader_survives_corrupt_files(tmp_path):
    """model.bin wire format (src/models/model.cc:594-661 <-> python/ctranslate2/specs/model_spec.py:
    382-415; SURVEY 8 f2 "reader / writer fuzzing"): truncated files, flipped header bytes, flipped
    payload bytes and planted 0xffffffff counts are either read or refused with an error - never a
    crash, a hang or an allocation of the planted size."""
    import ctypes as C
    import shutil
    from ctranslate2_amd import _lib
    L = _lib.lib()
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "models",
                       "aren-transliteration-i8")
    data = open(os.path.join(src, "model.bin"), "rb").read()
    d = str(tmp_path / "m")
    os.makedirs(d)
    for f in os.listdir(src):
        if f != "model.bin":
            shutil.copy(os.path.join(src, f), d)
    rng = np.random.default_rng(20260921)
    accepted = rejected = 0
    for it in range(240):
        b = bytearray(data)
        mode = it % 4
        if mode == 0:
            b = b[:int(rng.integers(0, len(b)))]
        elif mode == 1:
            for _ in range(int(rng.integers(1, 8))):
                b[int(rng.integers(0, 4096))] = int(rng.integers(0, 256))
        elif mode == 2:
            for _ in range(int(rng.integers(1, 8))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        else:
            p = int(rng.integers(0, len(b) - 8))
            b[p:p + 4] = (0xFFFFFFFF).to_bytes(4, "little")
        with open(os.path.join(d, "model.bin"), "wb") as f:
            f.write(bytes(b))
        n = L.ct2x_inspect_model(d.encode(), None, 0, None, None, None)
        if n < 0:
            rejected += 1
            assert _lib.last_error() if hasattr(_lib, "last_error") else True
            continue
        accepted += 1
        name = C.create_string_buffer(256)
        dt, rk, nb = C.c_int32(), C.c_int32(), C.c_int64()
        dims = (C.c_int64 * 8)()
        for i in range(min(n, 40)):
            L.ct2x_inspect_variable(d.encode(), C.c_int32(i), name, 256, C.byref(dt), C.byref(rk), dims,
                                    C.byref(nb))
    assert rejected >= 40 and accepted >= 40  # both outcomes exercised


def test_device_activation_source_matches_the_oracle_on_the_host(tmp_path):
    """ctranslate2_amd/csrc/activation_exact.h is what the HIP epilogues compile; built here with g++
    (-ffp-contract=off -mfma: mul, FMA and division round as on the device) it must return the bits
    of the oracle's own restatement - which tests/test_oracle.py::test_ref_dequantize_exact pins
    against the reference library - for every ops::ActivationType, over values that reach the
    clamps, the tiny-argument arm of tanh and both signs."""
    import ctypes as C
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    so = str(tmp_path / "activation_host.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-mfma", "-fopenmp", "-shared",
                           "-fPIC", "-o", so, os.path.join(here, "cpp", "activation_host.cc")])
    H = C.CDLL(so)
    # glibc's expf (the scalar tail of the CPU LogSoftMax), restated: every one of the 2^32 floats
    H.host_expf_glibc_exhaustive.restype = C.c_longlong
    assert H.host_expf_glibc_exhaustive() == 0
    import _oracle as O
    orc = O.lib()
    orc.orc_activation.restype = C.c_float
    orc.orc_activation.argtypes = [C.c_float, C.c_int]
    orc.orc_exp_avx2.restype = C.c_float
    orc.orc_exp_avx2.argtypes = [C.c_float]
    rng = np.random.default_rng(5)
    x = np.concatenate([
        rng.standard_normal(200000) * 3, rng.standard_normal(50000) * 30, rng.standard_normal(50000) * 1e-4,
        np.array([0.0, -0.0, 1e-30, -1e-30, 0.0004, -0.0004, 7.9, -7.9, 8.0, 88.4, -88.4, 100.0, -100.0]),
    ]).astype(np.float32)
    y = np.empty_like(x)
    for act in range(7):
        H.host_activation(x.ctypes.data_as(C.c_void_p), C.c_int64(x.size), C.c_int(act), y.ctypes.data_as(C.c_void_p))
        want = np.array([orc.orc_activation(float(v), act) for v in x[::37]], np.float32)
        got = y[::37]
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), act
    H.host_exp_avx2(x.ctypes.data_as(C.c_void_p), C.c_int64(x.size), y.ctypes.data_as(C.c_void_p))
    want = np.array([orc.orc_exp_avx2(float(v)) for v in x[::37]], np.float32)
    assert np.array_equal(y[::37].view(np.uint32), want.view(np.uint32))


def test_split_cross_attention_scratch_covers_a_shrinking_session():
    """ct2x_cross_attention_split_scratch_bytes (host arithmetic, attention_cross_h.hip): a decoding
    session is sized ONCE for its first batch and shrinks as hypotheses finish - and fewer
    (entry, head) units are cut into MORE chunks. The size must hold the chunk records of every
    smaller batch: records = units x chunks, chunks = as many as keep every workgroup resident
    (512 up to beam 5, 256 beyond), at most 8, chunks of at least 128 keys."""
    L = _lib.lib()
    L.ct2x_cross_attention_split_scratch_bytes.restype = C.c_int64
    i64 = C.c_int64

    def chunks(batch, beam, heads, s_max):
        resident = 512 if beam <= 5 else 256
        return max(1, min(resident // max(1, batch * heads), s_max // 128, 8))

    for beam in (1, 4, 5, 8):
        for heads in (4, 8, 20):
            for dh in (64, 128):
                for s_max in (100, 256, 1500, 4000):
                    for batch in (1, 3, 8, 17, 40, 200):
                        nbytes = int(L.ct2x_cross_attention_split_scratch_bytes(i64(batch), i64(beam), i64(heads),
                                                                                i64(dh), i64(s_max)))
                        tickets = batch * heads * 4
                        floats = (nbytes - tickets) // 4
                        assert nbytes >= tickets and (nbytes - tickets) % 4 == 0
                        if floats == 0:
                            # no scratch: the launcher then never splits, whatever the batch becomes
                            assert chunks(batch, beam, heads, s_max) == 1 or batch * heads >= (512 if beam <= 5 else 256) \
                                or s_max // 128 <= 1
                            continue
                        rec = beam * (dh + 2)
                        for smaller in range(1, batch + 1):
                            need = smaller * heads * chunks(smaller, beam, heads, s_max) * rec
                            assert need <= floats, (batch, smaller, beam, heads, dh, s_max, need, floats)
