"""CPU: the parts of bench.py that can be checked without a GPU -- the byte model of SURVEY 8(d), the view poses of the
multi-GPU configuration, the command line contract, and that it refuses to run (instead of falling back) without a
device."""
import json
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_algorithmic_byte_model_matches_survey_example():
    # bicycle 1080p, V = 2.5 M, L = 12 M, G = 8160 -> about 4.2 GB per frame
    b = bench.algorithmic_bytes(6131954, 2_500_000, 12_000_000, 8160, 1920, 1080)
    assert 4.0e9 < b < 4.4e9
    # linear in every unit
    b2 = bench.algorithmic_bytes(2 * 6131954, 2 * 2_500_000, 2 * 12_000_000, 2 * 8160, 1920, 2 * 1080)
    assert abs(b2 - 2 * b) < 1e-6 * b


def test_view_poses_are_the_base_pose_rotated_about_world_up():
    base_pos, base_tgt, base_up = (np.array(x, dtype=np.float64) for x in bench.view_pose(0))
    assert np.allclose(base_up, [0, -1, 0])  # colmap world-up (app/main.cpp:193)
    for k in range(8):
        pos, target, up = (np.array(x, dtype=np.float64) for x in bench.view_pose(k))
        assert np.allclose(up, base_up)
        # a rotation about the up axis: heights along it and distances from it are kept, for eye and target
        assert np.isclose(pos[1], base_pos[1]) and np.isclose(target[1], base_tgt[1])
        assert np.isclose(np.hypot(pos[0], pos[2]), np.hypot(base_pos[0], base_pos[2]))
        assert np.isclose(np.linalg.norm(pos - target), np.linalg.norm(base_pos - base_tgt))
    assert not np.allclose(bench.view_pose(1)[0], bench.view_pose(0)[0])
    assert np.allclose(bench.view_pose(8)[0], bench.view_pose(0)[0])  # 8 x 45 degrees


def test_bench_refuses_to_run_without_a_gpu():
    import torch

    if torch.cuda.is_available():
        return  # (on the GPU box the real thing is exercised by the driver)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode != 0
    assert "GPU" in (res.stderr + res.stdout)
    assert not any(line.startswith("{") and "metric" in line for line in res.stdout.splitlines())


def test_committed_default_bench_line_has_the_contract_fields():
    import glob

    path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_default.json")))[-1]  # the newest round's
    d = json.loads(open(path).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["config"]["workload"] and "model" not in d["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None
    if "r01" not in os.path.basename(path):  # round 2 on: the whole-frame roofline is reported three ways, side by side
        fr = d["frame_roofline"]
        assert {"survey_model", "own_algorithmic"} <= set(fr) and fr["peak"] == 8000.0
        assert fr["own_algorithmic"]["bytes"] == sum(fr["own_algorithmic"]["per_stage_bytes"].values())
        assert "moving_camera" in d and "file_order" in d
