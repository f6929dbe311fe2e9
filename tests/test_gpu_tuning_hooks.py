"""`-m gpu`: the two opt-in tuning hooks stay correct -- LCGS_GRAPH=1 (the fused frame captured once and replayed as
a hipGraph, per-call parameters read from device memory) and LCGS_RENDER_VARIANT=a (one wave64 per tile instead of
one workgroup per tile).  Neither is the default (measured: no gain); both must still produce the reference frame,
so the fused-frame and backward parity suites run once more under each."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

FILES = ["tests/test_gpu_fused.py", "tests/test_gpu_backward.py", "tests/test_gpu_sh_degrees.py"]


@pytest.mark.parametrize("hook", [{"LCGS_GRAPH": "1"}, {"LCGS_RENDER_VARIANT": "a"}], ids=["hipgraph", "wave_per_tile"])
def test_parity_suites_under_tuning_hook(hook):
    env = dict(os.environ, **hook)
    res = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"] + FILES,
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
