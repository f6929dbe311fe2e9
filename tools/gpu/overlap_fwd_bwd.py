"""Experiment: views of a gradient-accumulation batch alternating between two contexts, so that view j+1's forward overlaps
view j's backward.  gpurun -- python tools/gpu/overlap_fwd_bwd.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import luisacomputegaussiansplatting_amd as L  # noqa: E402
from bench import view_pose  # noqa: E402

dev = torch.device("cuda", 0)
P, W, H, B = 6131954, 1920, 1080, 4
scene = L.synth_scene(1, 2001, P)
s0, s1 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
r0 = L.Renderer(L.Context(0, s0.cuda_stream))
r0.upload_scene(scene)
d = r0.scene_tensors()
KEYS = ("pos", "scale", "rotq", "sh", "opacity")
r1 = L.Renderer(L.Context(0, s1.cuda_stream))
r1.bind_scene(*[d[k] for k in KEYS])
cams = [L.get_lookat_cam(*view_pose(k), width=W, height=H) for k in range(8)]
imgs = [torch.zeros(3, H, W, device=dev) for _ in range(2)]
dL = torch.randn(3, H, W, device=dev)
grads = {"pos": torch.zeros(P, 3, device=dev), "scale": torch.zeros(P, 3, device=dev), "rotq": torch.zeros(P, 4, device=dev),
         "sh": torch.zeros(P, 48, device=dev), "opacity": torch.zeros(P, device=dev)}
torch.cuda.synchronize()
for r in (r0, r1):
    for c in cams:
        r.forward(c, imgs[0], keep_state=True, sync=True)  # sizes the buffers / hints


def step(two, step_no):
    ev_prev = None
    for j in range(B):
        k = j & 1 if two else 0
        r, st = (r0, s0) if k == 0 else (r1, s1)
        r.forward(cams[(step_no * B + j) % 8], imgs[k], keep_state=True, sync=False)
        if ev_prev is not None and two:
            st.wait_event(ev_prev)
        r.backward(dL, *[grads[x] for x in KEYS], accumulate=j > 0)
        if two:
            ev_prev = torch.cuda.Event()
            ev_prev.record(st)
    if two:
        s0.wait_event(ev_prev)


for two in (False, True, False, True):
    for i in range(3):
        step(two, i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    N = 10
    for i in range(N):
        step(two, i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / (N * B)
    print("two contexts" if two else "one context ", f"{ms:.3f} ms per view, {P / ms / 1e3:.1f} Msplats/s", flush=True)
