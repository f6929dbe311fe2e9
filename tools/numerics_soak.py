"""CPU-only soak of oracle/numerics.py's per-pixel bound: the random draws of tests/test_gpu_random_sweep.py::_draw (the frames
the `-m gpu` soak holds the HIP renderer to, bit for bit) rendered under every numerics variant; every pixel of every variant
must lie inside the bound derived from the parity oracle's own evaluations.
    python tools/numerics_soak.py [first_seed] [draws] -> a summary line per failing draw, then the distribution."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import Oracle, numerics  # noqa: E402
from test_gpu_random_sweep import _draw  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
draws = int(sys.argv[2]) if len(sys.argv) > 2 else 200
o = Oracle("f32")
worst, may_move, over, failed, pixels = [], [], {k: 0 for k in numerics.VARIANTS}, [], 0
for seed in range(first, first + draws):
    rng, scene, W, H, pose, fov, bg, sm = _draw(seed)
    cam = o.lookat(*pose, width=W, height=H, fov=fov)
    rep, _ = numerics.report(scene, cam, bg=bg, scale_modifier=sm)
    if rep["classes"].get("pixels", 0) == 0:
        continue
    n = rep["classes"]["pixels"]
    pixels += n
    may_move.append(rep["classes"]["pixels_that_may_move_over_1e_4"] / n)
    for name, v in rep["variants"].items():
        over[name] += v["pixels_over_1e-4"]
        worst.append(v["worst_ratio_diff_to_bound"])
        if not v["all_explained"]:
            failed.append((seed, name, v["unexplained_pixels"], v["max_unexplained_excess"], v["worst_ratio_diff_to_bound"]))
            print("UNEXPLAINED", seed, name, json.dumps(v), flush=True)
worst = np.array(worst)
print(json.dumps({"draws": draws, "first_seed": first, "pixels": pixels, "failed": failed,
                  "pixels_over_1e-4_by_variant": over,
                  "worst_ratio_diff_to_bound": {"median": float(np.median(worst)), "p99": float(np.percentile(worst, 99)),
                                                "max": float(worst.max())},
                  "fraction_of_frame_that_may_move_over_1e-4": {"median": float(np.median(may_move)),
                                                                "p90": float(np.percentile(may_move, 90)),
                                                                "max": float(np.max(may_move))}}, indent=1))
