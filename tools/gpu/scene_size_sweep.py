"""How the headline figures move with the SCENE SIZE (the workload sweep of bench.py --sweep moves the pair count at two nearly
equal splat counts, so it cannot tell its constant from its per-splat coefficient): the bicycle stand-in's generator at
0.75 M ... 24 M splats, 1920x1080, the headline pose.  gpurun -- python3 tools/gpu/scene_size_sweep.py
Per size: on-screen splats, reference pairs, sorted pairs, forward frames/s (strictly in order), forward+backward Msplats/s
(dense rows), per-stage times."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import luisacomputegaussiansplatting_amd as L  # noqa: E402

W, H = 1920, 1080
POSE = ([-3, -0.5, 2.3], [0, 0, 0.5], [0, -1, 0])  # app/main.cpp:195-197
KEYS = ("pos", "scale", "rotq", "sh", "opacity")
SIZES = [int(x) for x in os.environ.get("LCGS_SIZES", "750000,1500000,3000000,6131954,12000000,24000000").split(",")]
dev = torch.device("cuda", 0)
cam = L.get_lookat_cam(*POSE, width=W, height=H)
img = torch.zeros(3, H, W, device=dev)
dL = torch.randn(3, H, W, device=dev)


def timed(fn, n, reps=3):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / n)
    return float(np.median(out))


print(f"# bicycle stand-in generator (synth_scene(1, 2001, P)), {W}x{H}, library ingest (spatial order); forward: 50 frames in order x 3,")
print("# forward+backward: 15 steps x 3, dense rows")
print("#        P  on-screen V  reference pairs  sorted pairs  forward fps    ms   fwd+bwd Msplats/s    ms   stages (ms)")
rows = []
for P in SIZES:
    scene = L.synth_scene(1, 2001, P)
    r = L.Renderer(L.Context(0))
    r.upload_scene(scene)
    del scene
    for _ in range(3):
        r.forward(cam, img, sync=True)
    st = r.frame_stats()
    r.set_profiling(True)
    r.forward(cam, img, sync=True)
    stages = {k: round(v, 3) for k, v in r.stage_times().items() if v >= 0.01}
    r.set_profiling(False)
    r.forward(cam, img, sync=True)
    t_f = timed(lambda: r.forward(cam, img, sync=False), 50)
    r.ctx.synchronize()
    g = {k: torch.zeros(P, w, device=dev) if w > 1 else torch.zeros(P, device=dev)
         for k, w in (("pos", 3), ("scale", 3), ("rotq", 4), ("sh", 48), ("opacity", 1))}
    r.forward(cam, img, keep_state=True, sync=True)

    def step():
        r.forward(cam, img, keep_state=True, sync=False)
        r.backward(dL, *[g[k] for k in KEYS])
    t_b = timed(step, 15)
    r.ctx.synchronize()
    rows.append((P, st["num_visible"], t_f * 1e3))
    print(f"{P:10d}  {st['num_visible']:11d}  {st['num_rendered']:15d}  {st['num_pairs']:12d}  {1 / t_f:11.1f}  {t_f * 1e3:.3f}  "
          f"{P / t_b / 1e6:17.1f}  {t_b * 1e3:.3f}   {stages}", flush=True)
    del r, g
    torch.cuda.empty_cache()
# (the generator keeps the on-screen share at 39 %, so P and V move together: one slope, between consecutive sizes)
print("# forward ms per added M splats between consecutive sizes: " +
      ", ".join(f"{(t1 - t0) / ((p1 - p0) / 1e6):.3f}" for (p0, _, t0), (p1, _, t1) in zip(rows, rows[1:])))
