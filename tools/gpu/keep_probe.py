import torch, numpy as np, sys, time
sys.path.insert(0, ".")
import luisacomputegaussiansplatting_amd as L
from bench import view_pose
scene = L.synth_scene(1, 2001, 6131954)
r = L.Renderer(L.Context(0))
r.upload_scene(scene)
cam = L.get_lookat_cam(*view_pose(0), width=1920, height=1080)
img = torch.zeros(3, 1080, 1920, device="cuda:0")
for keep in (False, True):
    r.forward(cam, img, keep_state=keep, sync=True)
    r.set_profiling(True)
    acc = {}
    for _ in range(10):
        r.forward(cam, img, keep_state=keep, sync=True)
        for k, v in r.stage_times().items():
            acc.setdefault(k, []).append(v)
    r.set_profiling(False)
    print("keep", keep, {k: round(float(np.median(v)), 4) for k, v in acc.items()})
    torch.cuda.synchronize()
    for _ in range(10):
        r.forward(cam, img, keep_state=keep, sync=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        r.forward(cam, img, keep_state=keep, sync=False)
    torch.cuda.synchronize()
    print("keep", keep, "ms/frame", round((time.perf_counter() - t0) * 10, 4))
