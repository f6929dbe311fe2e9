"""`-m gpu`: the fused-frame and backward parity tests once more in a process whose library fills every fresh workspace
allocation with garbage (LCGS_POISON=1) instead of the zeros a fresh hipMalloc usually holds: a kernel that reads what
no kernel wrote fails here instead of passing by luck."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_parity_suites_with_poisoned_workspace():
    env = dict(os.environ, LCGS_POISON="1")
    files = ["tests/test_gpu_fused.py", "tests/test_gpu_backward.py", "tests/test_gpu_train.py",
             "tests/test_gpu_sh_degrees.py", "tests/test_gpu_random_sweep.py", "tests/test_gpu_comm.py",
             "tests/test_gpu_lod.py", "tests/test_gpu_ingest.py"]
    res = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"] + files,
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
