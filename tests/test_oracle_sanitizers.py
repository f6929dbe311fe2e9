"""CPU: the oracle's C restatement under AddressSanitizer + UndefinedBehaviorSanitizer (the
reference has no sanitizer coverage at all; GPU sanitizers are not available on the pool, so the CPU checker is the
part that can be checked).  Forward and backward of a small scene with culled, off-screen, huge and degenerate splats
at odd resolutions."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


def test_oracle_forward_and_backward_are_clean_under_asan_ubsan(tmp_path):
    cc = shutil.which("gcc")
    if cc is None:
        pytest.skip("no gcc")
    exe = str(tmp_path / "oracle_san")
    src = [os.path.join(ROOT, "oracle", "lcgs_oracle.c"), os.path.join(ROOT, "oracle", "lcgs_oracle_bwd.c"),
           os.path.join(ROOT, "tests", "helpers", "oracle_san_main.c")]
    build = subprocess.run([cc, "-O1", "-g", "-std=c11", "-ffp-contract=off", "-fno-omit-frame-pointer",
                            "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-DORC_REAL=float"] + src +
                           ["-o", exe, "-lm"], capture_output=True, text=True, timeout=300)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("toolchain without sanitizer runtimes")
    assert build.returncode == 0, build.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert run.returncode == 0, (run.stdout[-500:], run.stderr[-3000:])
    assert "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr, run.stderr[-3000:]
    assert run.stdout.count("num_rendered") == 3
