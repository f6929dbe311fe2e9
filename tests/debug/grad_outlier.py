"""Diagnostic (not a test): where does the gradient error of one soak draw come from?  python tests/debug/grad_outlier.py <seed>"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import luisacomputegaussiansplatting_amd as L
from oracle import Oracle
from test_gpu_random_sweep import _draw
from gpu_util import DEV, dev, upload_scene

seed = int(sys.argv[1])
rng, scene, W, H, pose, fov, bg, sm = _draw(seed)
o32, o64 = Oracle("f32"), Oracle("f64")
cam = L.get_lookat_cam(*pose, width=W, height=H); cam.fov = fov
d = upload_scene(scene)
r = L.Renderer(L.Context(0)); r.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"])
img = torch.zeros(3, H, W, device=DEV)
n = r.forward(cam, img, bg=bg, scale_modifier=sm, keep_state=True, sync=True)
dL = rng.normal(size=(3, H, W)).astype(np.float32)
g = {k: torch.zeros_like(d[k]) for k in ("pos", "scale", "rotq", "sh", "opacity")}
r.backward(dev(dL), *[g[k] for k in g]); r.ctx.synchronize()
ref32 = o32.render_backward_full(scene, o32.lookat(*pose, width=W, height=H, fov=fov), dL, bg=bg, scale_modifier=sm)
ref64 = o64.render_backward_full(scene, o64.lookat(*pose, width=W, height=H, fov=fov), dL, bg=bg, scale_modifier=sm)
fw = o32.render(scene, o32.lookat(*pose, width=W, height=H, fov=fov), bg=bg, scale_modifier=sm, ambig_eps=1e-5)
print("seed", seed, "P", scene["pos"].shape[0], "WxH", W, H, "fov", fov, "n", n, "ambig px", int(fw["ambig"].sum()),
      "img maxdiff", float(np.abs(img.cpu().numpy() - fw["img"]).max()))
rel = lambda a, b: np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
for k in g:
    a = g[k].cpu().numpy().astype(np.float64); b32 = ref32[k].astype(np.float64).reshape(a.shape); b64 = ref64[k].astype(np.float64).reshape(a.shape)
    e = np.abs(a - b64).reshape(a.shape[0], -1).sum(1); e32 = np.abs(b32 - b64).reshape(a.shape[0], -1).sum(1)
    top = np.argsort(-e)[:4]
    print(k, "gpu-vs-f64 %.2e  f32orc-vs-f64 %.2e  |g| %.3e" % (rel(a, b64), rel(b32, b64), np.linalg.norm(b64)),
          "top err splats", [(int(t), "%.2e" % e[t], "%.2e" % e32[t], int(fw["radii"][t])) for t in top])
