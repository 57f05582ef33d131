"""AddressSanitizer + UndefinedBehaviorSanitizer over the CPU side (the reference has no sanitizer
runs): the oracle's C code on exact-size heap buffers and on every golden model file up to the last
cache row, and the GPU-less part of the C++17 host layer.  Each driver is compiled from the sources
with -fsanitize=address,undefined and must exit 0 with no sanitizer report."""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-static-libasan", "-static-libubsan",
       "-fno-omit-frame-pointer", "-g", "-O1"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
           KX_LOG_LEVEL="3")


def _clean(out):
    assert "AddressSanitizer" not in out and "runtime error" not in out, out[-3000:]


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
def test_oracle_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "oracle_san")
    subprocess.run(["gcc", "-std=c99", "-fopenmp", *SAN, "-I" + os.path.join(ROOT, "oracle"),
                    os.path.join(ROOT, "tests", "san", "oracle_san_check.c"),
                    os.path.join(ROOT, "oracle", "kx_oracle.c"), "-lm", "-o", exe], check=True)
    g = os.path.join(ROOT, "tests", "golden")
    models = ["fi:" + os.path.join(g, "g3_tiny_fp32.bin"), "qi:" + os.path.join(g, "g4_tiny_q8.bin"),
              "qh:" + os.path.join(g, "g6_llama3_q8.bin"), "bh:" + os.path.join(g, "g7_qwen2_fp32.bin"),
              "qh:" + os.path.join(g, "g7_qwen2_q8.bin"), "qh:" + os.path.join(g, "g8_llama3_tied_q8.bin")]
    r = subprocess.run([exe, *models], env=ENV, capture_output=True, text=True, timeout=600)
    _clean(r.stdout + r.stderr)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert r.stdout.count("model ") == len(models) and "kernels ok" in r.stdout


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_host_layer_under_asan_and_ubsan(tmp_path):
    pkg = os.path.join(ROOT, "llm-inference-engine_amd")
    lib = os.path.join(pkg, "lib")
    assert os.path.exists(os.path.join(lib, "libkx_hip.so")), "run __graft_entry__.build() first"
    srcs = sorted(glob.glob(os.path.join(pkg, "host", "source", "**", "*.cpp"), recursive=True))
    exe = str(tmp_path / "host_san")
    objs = []
    jobs = []
    for i, src in enumerate(srcs + [os.path.join(ROOT, "tests", "san", "host_san_check.cpp")]):
        obj = str(tmp_path / f"o{i}.o")
        objs.append(obj)
        jobs.append(subprocess.Popen(["g++", "-std=c++17", *SAN, "-I" + os.path.join(ROOT, "include"),
                                      "-I" + os.path.join(pkg, "host", "include"),
                                      "-I" + os.path.join(pkg, "host", "source"), "-c", src, "-o", obj]))
        if len(jobs) >= 8:
            assert jobs.pop(0).wait() == 0
    for j in jobs:
        assert j.wait() == 0
    subprocess.run(["g++", *SAN, *objs, "-L" + lib, "-lkx_hip", "-Wl,-rpath," + lib, "-o", exe], check=True)
    r = subprocess.run([exe], env=ENV, capture_output=True, text=True, timeout=600)
    _clean(r.stdout + r.stderr)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "0 failure(s)" in r.stdout
