"""the host-side C++ (TIFF codecs, prediction-JSON reader, packed-mask epilogue, ring simplifier, region
predicates, contour tracer) rebuilt with g++ AddressSanitizer + UndefinedBehaviorSanitizer — no HIP objects — and the
CPU tests that exercise it, corrupt / truncated inputs included (tests/test_host_fuzz.py), re-run against that build:
``make -C treedetection_amd/csrc asan-test``. The sanitizers abort the child process on the first finding."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_cpp_under_asan_ubsan():
    if not shutil.which("g++") or not shutil.which("make"):
        pytest.skip("no g++ / make")
    lib = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(lib) or not os.path.exists(lib):
        pytest.skip("libasan not installed")
    if os.environ.get("TD_HOST_LIB"):
        pytest.skip("already inside the sanitizer run")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "treedetection_amd", "csrc"), "asan-test"], capture_output=True, text=True,
                       timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail
