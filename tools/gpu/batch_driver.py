"""A camera batch (lcgs_render_forward_batch) of the bicycle stand-in on its own, for kernel timelines:
   rocprofv3 --kernel-trace ... -- python3 tools/gpu/batch_driver.py [frames] [fit]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import luisacomputegaussiansplatting_amd as L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
mode = sys.argv[2] if len(sys.argv) > 2 else "batch"
W, H = 1920, 1080
dev = torch.device("cuda", 0)
side = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(side)
r = L.Renderer(L.Context(0, side.cuda_stream))
r.upload_scene(L.synth_scene(1, 2001, 6131954))
pose = ([-3, -0.5, 2.3], [0, 0, 0.5], [0, -1, 0])
cam = L.get_lookat_cam(*pose, width=W, height=H)
imgs = [torch.zeros(3, H, W, device=dev) for _ in range(2)]
r.forward(cam, imgs[0], sync=True)
if mode == "batch":
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r.forward_batch([cam] * n, [imgs[i & 1] for i in range(n)])
        torch.cuda.synchronize()
        print(f"batch of {n}: {n / (time.perf_counter() - t0):.1f} frames/s", flush=True)
else:
    P = 6131954
    d = r.scene_tensors()
    g = [torch.zeros_like(d[k]) for k in ("pos", "scale", "rotq", "sh", "opacity")]
    tgt = torch.zeros(3, H, W, device=dev)
    losses = torch.zeros(8, device=dev)
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n // 4):
            r.fit_views([cam] * 4, [tgt] * 4, *g, losses)
        torch.cuda.synchronize()
        print(f"fit_views 4 x {n // 4}: {P * (n // 4) * 4 / (time.perf_counter() - t0) / 1e6:.1f} Msplats/s", flush=True)
