"""Host time to enqueue one asynchronous frame (the launches of lcgs_render_forward) against the GPU's frame time:
gpurun -- python tools/gpu/enqueue_cost.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import luisacomputegaussiansplatting_amd as L  # noqa: E402
from bench import view_pose  # noqa: E402

dev = torch.device("cuda", 0)
P, W, H = int(os.environ.get("P", 6131954)), 1920, 1080
s0 = torch.cuda.Stream(device=dev)
r = L.Renderer(L.Context(0, s0.cuda_stream))
r.upload_scene(L.synth_scene(1, 2001, P))
cam = L.get_lookat_cam(*view_pose(0), width=W, height=H)
img = torch.zeros(3, H, W, device=dev)
r.forward(cam, img, sync=True)
for _ in range(20):
    r.forward(cam, img, sync=False)
r.ctx.synchronize()
N = 200
t0 = time.perf_counter()
for _ in range(N):
    r.forward(cam, img, sync=False)
t1 = time.perf_counter()
r.ctx.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e6 * (t1 - t0) / N:.1f} us per frame on the host; {1e6 * (t2 - t0) / N:.1f} us per frame on the GPU")
