"""Dev-time script (needs /root/reference): pulls the literal known-answer vectors of the
hot-path ops out of the reference's own C++ tests into tests/golden/ops_kat.json, so that the
CPU test-suite can pin the oracle (and the -m gpu suite the HIP kernels) against the reference's
golden values without /root/reference being present. Only numbers are extracted.

Sources: tests/ops_test.cc (GemmInt8 :683, QuantizeINT8 :936, QuantizeINT8ZeroRow :969, TopK :710,
TopKVariableDepth :726, TopKChangeK :746, TopPMask :772, SoftMax :791, LogSoftMax :807,
MaskedSoftMax :825, MaskedSoftMaxTriangular :841, LayerNorm :880, RMSNorm :920, the float Gemm
family :516-680, the activations :1015-1095) and tests/layers_test.cc (RotaryEmbedding :108).
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"


def test_body(text, name):
    m = re.search(r"TEST_P\(\w+,\s*%s\)\s*\{" % re.escape(name), text)
    assert m, name
    depth, i = 1, m.end()
    while depth:
        c = text[i]
        depth += (c == "{") - (c == "}")
        i += 1
    return text[m.end():i - 1]


def vectors(body):
    out = []
    for m in re.finditer(r"std::vector<(\w+)>\s*\{([^}]*)\}", body):
        kind, payload = m.group(1), m.group(2)
        vals = [v.strip() for v in payload.replace("\n", " ").split(",") if v.strip()]
        if kind in ("float",):
            def num(v):
                v = v.replace("inf", "INF")          # the TopPMask vectors use +-inf
                return float(v.rstrip("f").replace("INF", "inf"))
            out.append([num(v) for v in vals])
        else:
            out.append([int(v) for v in vals])
    return out


ops = open(os.path.join(REF, "tests/ops_test.cc")).read()
layers = open(os.path.join(REF, "tests/layers_test.cc")).read()
kat = {}

v = vectors(test_body(ops, "GemmInt8"))
kat["GemmInt8"] = {"a": v[0], "a_shape": [3, 8], "b": v[1], "b_shape": [8, 4], "c": v[2]}
v = vectors(test_body(ops, "QuantizeINT8"))
kat["QuantizeINT8"] = {"x": v[0], "shape": [2, 4], "scale": v[1], "q_round": v[2],
                       "q_round_shift_u8": v[3], "q_trunc": v[4]}
v = vectors(test_body(ops, "QuantizeINT8ZeroRow"))
kat["QuantizeINT8ZeroRow"] = {"x": v[0], "shape": [2, 4], "scale": v[1], "q_round": v[2], "q_trunc": v[3]}
v = vectors(test_body(ops, "TopK"))
kat["TopK"] = {"x": v[0], "shape": [2, 6], "k": 3, "values": v[1], "indices": v[2]}
v = vectors(test_body(ops, "SoftMax"))
kat["SoftMax"] = {"x": v[0], "shape": [2, 5], "y": v[1]}
v = vectors(test_body(ops, "LogSoftMax"))
kat["LogSoftMax"] = {"x": v[0], "shape": [2, 10], "y": v[1]}
v = vectors(test_body(ops, "MaskedSoftMax"))
kat["MaskedSoftMax"] = {"x": v[0], "shape": [2, 5], "lengths": v[1], "y": v[2]}
v = vectors(test_body(ops, "LayerNorm"))
kat["LayerNorm"] = {"gamma": v[0], "beta": v[1], "x": v[2], "shape": [2, 5], "y": v[3], "eps": 1e-5}
v = vectors(test_body(ops, "RMSNorm"))
kat["RMSNorm"] = {"gamma": v[0], "x": v[1], "shape": [2, 5], "y": v[2], "eps": 1e-6}
v = vectors(test_body(layers, "RotaryEmbedding"))
kat["RotaryEmbedding"] = {"x": v[0], "shape": [2, 4, 2, 6], "y": v[1], "offset": 2,
                          "base": 10000.0, "interleave": True}

# float Gemm with fused bias / residual / activation (tests/ops_test.cc:516-680): shared inputs
m = re.search(r"static const StorageView gemm_a\(.*?\);\s*static const StorageView gemm_b\(.*?\);"
              r"\s*static const StorageView gemm_y\(.*?\);", ops, re.S)
ga, gb, gy = vectors(m.group(0))
v = vectors(test_body(ops, "Gemm"))
gemm = {"a": ga, "a_shape": [4, 5], "b": gb, "b_shape": [5, 2], "y": gy, "y_shape": [4, 2],
        "a_b_plus_y": v[0]}
v = vectors(test_body(ops, "GemmBias"))
gemm["bias"] = v[0]
gemm["a_b_plus_bias_plus_y"] = v[1]
v = vectors(test_body(ops, "GemmResidual"))
gemm["residual"] = v[0]
gemm["residual_bias"] = v[1]
gemm["a_b_plus_bias_plus_y_plus_residual"] = v[2]
gemm["a_b_plus_y_plus_residual"] = v[3]
v = vectors(test_body(ops, "GemmGELU"))
gemm["gelu_bias"] = v[0]
gemm["gelu_a_b_plus_bias"] = v[3]
gemm["gelu_a_b"] = v[4]
kat["GemmFloat"] = gemm
v = vectors(test_body(ops, "TopKVariableDepth"))
kat["TopKVariableDepth"] = {"x": v[3], "shape": [2, 4], "k": 3, "values": v[4], "indices": v[5]}
v = vectors(test_body(ops, "TopKChangeK"))
kat["TopKChangeK"] = {"x": v[0], "shape": [2, 6], "values_k2": v[1], "indices_k2": v[2],
                      "values_k3": v[3], "indices_k3": v[4]}
v = vectors(test_body(ops, "TopPMask"))
kat["TopPMask"] = {"x": v[0], "shape": [2, 5], "p": 0.97, "kept": [[1, 1, 1, 0, 1], [1, 1, 0, 0, 1]]}
v = vectors(test_body(ops, "MaskedSoftMaxTriangular"))
kat["MaskedSoftMaxTriangular"] = {"x": v[0], "shape": [2, 2, 3, 3], "lengths": v[1], "y": v[2]}
acts = {}
for name, act_id in (("ReLU", 0), ("GELUTanh", 1), ("Swish", 2), ("GELU", 3), ("GELUSigmoid", 4),
                     ("Tanh", 5), ("Sigmoid", 6)):
    v = vectors(test_body(ops, name))
    acts[name] = {"act": act_id, "x": v[0], "y": v[1]}
kat["Activations"] = acts

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ops_kat.json")
with open(out, "w") as f:
    json.dump(kat, f, indent=1)
print("wrote", out, {k: len(json.dumps(v)) for k, v in kat.items()})
