"""Builds tests/cpp/api_example.cc (the C++ host API, include/ct2_mi355.hpp) with plain g++
against the in-tree libct2_mi355.so and runs it."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIBDIR = os.path.join(ROOT, "ctranslate2_amd")
BINARY = os.path.join(HERE, "cpp", "build", "api_example")


def build():
    src = os.path.join(HERE, "cpp", "api_example.cc")
    hdrs = [os.path.join(ROOT, "include", n) for n in ("ct2_mi355.hpp", "ct2_mi355.h")]
    lib = os.path.join(LIBDIR, "libct2_mi355.so")
    newest = max(os.path.getmtime(p) for p in [src, lib] + hdrs)
    if not os.path.exists(BINARY) or os.path.getmtime(BINARY) < newest:
        os.makedirs(os.path.dirname(BINARY), exist_ok=True)
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror",
                               "-I" + os.path.join(ROOT, "include"), src, "-o", BINARY,
                               "-L" + LIBDIR, "-lct2_mi355", "-Wl,-rpath," + LIBDIR])
    return BINARY


def run(*args):
    p = subprocess.run([build()] + [str(a) for a in args], capture_output=True, text=True, timeout=300)
    return p.returncode, p.stdout, p.stderr
