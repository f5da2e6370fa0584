"""Driver for the reference's OWN C++ test-suite on this backend: oracle/_ref/ct2_reftests is
tests/*.cc of the reference compiled unmodified (googletest stand-in: tests/integration/
shim_gtest) and linked with the Level-1 library, so its suites exist for Device::CPU and
Device::CUDA (= the MI355X through libct2_mi355.so).

  python tests/integration/run_reftests.py '<gtest filter>'  -> one JSON line
{"ran": n, "passed": [...], "failed": [...], "skipped": [...], "crashed": [...]}

A test that reaches one of the 16-bit arms the Level-1 library leaves undefined kills the process
(lazy binding); the driver records it as "crashed" and reruns the rest.
"""
import json
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
BINARY = os.path.join(ROOT, "oracle", "_ref", "ct2_reftests")


def data_dir():
    """The layout the reference's tests expect (tests/test_utils.cc: <data>/models/v2/...) over the
    two fixtures this repository carries."""
    d = tempfile.mkdtemp(prefix="ct2x_reftests_")
    os.makedirs(os.path.join(d, "models", "v2"))
    for name in ("aren-transliteration", "aren-transliteration-i8"):
        os.symlink(os.path.join(ROOT, "tests", "golden", "models", name),
                   os.path.join(d, "models", "v2", name))
    return d


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else "*"
    positive, _, negative = flt.partition("-")
    env = dict(os.environ, CT2_USE_MKL="1", CT2X_S8_ACTIVATIONS="1")
    d = data_dir()
    passed, failed, skipped, crashed = [], [], [], []
    details = {}
    for _ in range(40):
        neg = ":".join(x for x in [negative] + passed + failed + skipped + crashed if x)
        p = subprocess.run([BINARY, d, "--gtest_filter=%s-%s" % (positive or "*", neg)],
                           capture_output=True, text=True, env=env, timeout=900)
        current = None
        block = []
        for line in p.stdout.splitlines():
            m = re.match(r"\[ RUN      \] (.*)", line)
            if m:
                current, block = m.group(1), []
                continue
            m = re.match(r"\[\s+(OK|FAILED|SKIPPED)\s+\] (\S+)$", line)
            if m and current == m.group(2):
                {"OK": passed, "FAILED": failed, "SKIPPED": skipped}[m.group(1)].append(current)
                if m.group(1) == "FAILED":
                    details[current] = "\n".join(block)[-600:]
                current = None
                continue
            block.append(line)
        if p.returncode in (0, 1) and current is None:
            break
        if current is None:
            raise SystemExit("test binary failed outside a test: rc=%d\n%s" % (p.returncode, p.stderr[-2000:]))
        crashed.append(current)
        details[current] = (p.stderr or "")[-400:]
    print(json.dumps({"ran": len(passed) + len(failed) + len(skipped) + len(crashed), "passed": passed,
                      "failed": failed, "skipped": skipped, "crashed": crashed, "details": details}))


if __name__ == "__main__":
    main()
