"""CPU: the product's host-side PLY reader / writer (csrc/host/ply.cpp) under AddressSanitizer + UndefinedBehavior
Sanitizer, fed well-formed and malformed files (truncated payload or header, bad magic, absurd or negative counts,
list properties, missing columns, big-endian, empty): every call returns a status, nothing reads out of bounds."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


def test_ply_reader_is_clean_under_asan_ubsan(tmp_path):
    cxx = shutil.which("g++")
    if cxx is None or not os.path.isdir("/opt/rocm/include"):
        pytest.skip("needs g++ and the HIP headers (type definitions only)")
    exe = str(tmp_path / "ply_san")
    build = subprocess.run([cxx, "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                            "-fno-omit-frame-pointer", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                            "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "helpers", "ply_san_main.cpp"),
                            "-o", exe, "-lpthread"], capture_output=True, text=True, timeout=600)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("toolchain without sanitizer runtimes")
    assert build.returncode == 0, build.stderr[-2000:]
    work = tmp_path / "files"
    work.mkdir()
    run = subprocess.run([exe, str(work)], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert run.returncode == 0, (run.stdout[-500:], run.stderr[-3000:])
    assert "failures 0" in run.stdout
    assert "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr, run.stderr[-3000:]
