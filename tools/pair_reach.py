"""How many (splat, tile) pairs of the bench frame reach no pixel of their tile?  (tuning aid, not product path)

The fused path already prunes each splat's tile rect to the axis-aligned box of its alpha >= 1/255 ellipse
(gs_math.hpp tight_rect).  This tool counts what an exact ellipse-vs-tile test per pair would remove on top.
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import luisacomputegaussiansplatting_amd as L
from bench import view_pose, P_BICYCLE

W, H = 1920, 1080
P = int(sys.argv[1]) if len(sys.argv) > 1 else P_BICYCLE
scene = L.synth_scene(1, 2001, P)
dev = torch.device("cuda", 0)
d = {k: torch.from_numpy(scene[k]).to(dev) for k in ("pos", "scale", "rotq", "sh", "opacity")}
ctx = L.Context(0)
r = L.Renderer(ctx)
r.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"])
cam = L.get_lookat_cam(*view_pose(0), width=W, height=H)
img = torch.zeros(3, H, W, device=dev)
n = r.forward(cam, img, keep_state=True, sync=True)
st = r.frame_stats()
GX, GY = (W + 15) // 16, (H + 15) // 16
G = GX * GY
lst = torch.zeros(st["num_pairs"], dtype=torch.int32, device=dev)
rng = torch.zeros(2 * G, dtype=torch.int32, device=dev)
r.last_lists(lst, rng)

# pixel means / conics from the stage-level operators (allocate_tiles rewrites them in place)
prj, spl, shp = L.GSProjector(), L.GSTileSplatter(), L.SHProcessor()
for op in (prj, spl, shp):
    op.create(ctx)
z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=dev)
Lcap = 20_000_000
color, means, covs, depth = z(P, 3), z(P, 2), z(P, 3), z(P)
accel = L.GSTileSplatterAccelProxy(z(P, dt=torch.int32), z(P, dt=torch.int32), z(Lcap, dt=torch.int64),
                                   z(Lcap, dt=torch.int32), z(Lcap, dt=torch.int64), z(Lcap, dt=torch.int32),
                                   z(2 * G, dt=torch.int32))
radii, img_s = z(P, dt=torch.int32), z(3, H, W)
prj.forward(L.GSProjectorInputProxy(P, d["pos"], d["scale"], d["rotq"], 1.0),
            L.GSProjectorOutputProxy(means, covs, depth), cam)
n_ref = spl.forward(accel, L.GSTileSplatterInputProxy(P, (0.0, 0.0, 0.0), means, depth, covs, color, d["opacity"]),
                    L.GSSplatForwardOutputProxy(H, W, img_s, radii))
torch.cuda.synchronize()

rr = rng.view(G, 2).long()
ln = rr[:, 1] - rr[:, 0]
tile_of = torch.repeat_interleave(torch.arange(G, device=dev), ln)
idx = lst.long()
mx, my = means[idx, 0], means[idx, 1]
a, b, c = covs[idx, 0], covs[idx, 1], covs[idx, 2]
op = d["opacity"].view(-1)[idx]
x0 = (tile_of % GX).float() * 16
y0 = (tile_of // GX).float() * 16
lox, hix = x0 - mx, x0 + 15 - mx
loy, hiy = y0 - my, y0 + 15 - my
inside = (lox <= 0) & (hix >= 0) & (loy <= 0) & (hiy >= 0)
q = lambda dx, dy: 0.5 * (a * dx * dx + c * dy * dy) + b * dx * dy
def edge_x(X):
    dy = torch.minimum(torch.maximum(-b * X / c, loy), hiy)
    return q(X, dy)
def edge_y(Y):
    dx = torch.minimum(torch.maximum(-b * Y / a, lox), hix)
    return q(dx, Y)
qmin = torch.minimum(torch.minimum(edge_x(lox), edge_x(hix)), torch.minimum(edge_y(loy), edge_y(hiy)))
qmin = torch.where(inside, torch.zeros_like(qmin), qmin)
live = qmin <= torch.log(255.0 * op)
tot = idx.numel()
print(f"reference num_rendered {n_ref}  fused pairs {tot}  ({tot / n_ref:.3f} of reference)")
print(f"pairs reaching a pixel (continuous bound): {int(live.sum())}  = {live.float().mean().item():.3f} of fused pairs")
dead_per_tile = torch.zeros(G, device=dev).index_add_(0, tile_of, (~live).float())
print("dead pairs per tile: mean %.1f of mean list %.1f" % (dead_per_tile.mean().item(), ln.float().mean().item()))
