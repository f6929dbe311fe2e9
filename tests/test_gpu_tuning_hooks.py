"""`-m gpu`: the opt-in tuning hook LCGS_GRAPH=1 (the fused frame captured once and replayed as a hipGraph, per-call
parameters read from device memory) stays correct.  It is not the default (measured: no gain, the short kernels are
GPU-latency-bound, not host-bound) but must still produce the reference frame, so the fused-frame parity suite runs once
more under it."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

FILES = ["tests/test_gpu_fused.py", "tests/test_gpu_backward.py"]


@pytest.mark.parametrize("hook", [{"LCGS_GRAPH": "1"}], ids=["hipgraph"])
def test_parity_suites_under_tuning_hook(hook):
    env = dict(os.environ, **hook)
    res = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"] + FILES,
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
