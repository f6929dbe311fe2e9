"""Tile-list length distribution of the bench workload (tuning aid, not part of the product path)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import luisacomputegaussiansplatting_amd as L
from bench import view_pose, P_BICYCLE

W, H = 1920, 1080
scene = L.synth_scene(1, 2001, P_BICYCLE)
dev = torch.device("cuda", 0)
d = {k: torch.from_numpy(scene[k]).to(dev) for k in ("pos", "scale", "rotq", "sh", "opacity")}
r = L.Renderer(L.Context(0))
r.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"])
cam = L.get_lookat_cam(*view_pose(0), width=W, height=H)
img = torch.zeros(3, H, W, device=dev)
n = r.forward(cam, img, keep_state=True, sync=True)
st = r.frame_stats()
G = ((W + 15) // 16) * ((H + 15) // 16)
lst = torch.zeros(st["num_pairs"], dtype=torch.int32, device=dev)
rng = torch.zeros(2 * G, dtype=torch.int32, device=dev)
r.last_lists(lst, rng)
rr = rng.cpu().numpy().view(np.uint32).reshape(G, 2)
ln = (rr[:, 1] - rr[:, 0]).astype(np.int64)
print("tiles", G, "pairs", ln.sum(), "mean", ln.mean())
for q in (50, 90, 99, 99.9, 100):
    print("pct", q, np.percentile(ln, q))
print("top 10", np.sort(ln)[-10:])
nc = r.ctx_n_contrib() if hasattr(r, "ctx_n_contrib") else None
