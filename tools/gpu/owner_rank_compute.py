"""What ONE rank of an N-rank ownership step COMPUTES, measured on one GPU (no transport): gpurun -- python3 tools/gpu/owner_rank_compute.py

DESIGN 7's N = 8 figures are a model (compute 1.4-1.7 ms against the exchange).  The exchange cannot be measured on a one-GPU
box; the rank's own work can, because none of it depends on the peers' GPUs: rank `me` of N
  1. projects ITS P/N rows for every one of the step's N views (lcgs_owner_project x N, one read-back of the counts),
  2. renders ITS view from the records of all N owners (here: produced beforehand by projecting every owner's range for that
     view -- exactly the bytes the messages would carry, in owner order) and differentiates it (lcgs_owner_render /
     lcgs_owner_render_backward),
  3. turns the 2-D gradients of its rows on each of the N views into parameter gradients (lcgs_owner_backward x N; the
     incoming 2-D rows are random here: their values do not change the work),
  4. applies Adam to its P/N rows.
Bicycle stand-in, the C5 views (bench.view_pose), 1920x1080.  Prints ms per phase and per step for N = 1, 2, 4, 8 and, beside
it, the replicated-scene step's compute (fused forward + backward, dense Adam on all rows)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import luisacomputegaussiansplatting_amd as L  # noqa: E402
import luisacomputegaussiansplatting_amd.multi_gpu as mg  # noqa: E402
from bench import view_pose  # noqa: E402

P = int(os.environ.get("LCGS_DRIVER_SPLATS", "6131954"))
W, H = 1920, 1080
STEPS = int(os.environ.get("LCGS_DRIVER_STEPS", "10"))
KEYS = ("pos", "scale", "rotq", "sh", "opacity")
dev = torch.device("cuda", 0)

scene = L.synth_scene(1, 2001, P)
r = L.Renderer(L.Context(0))
r.upload_scene(scene)
act = r.scene_tensors()
raw = {"pos": act["pos"].clone(), "scale": torch.log(act["scale"]), "rotq": act["rotq"].clone(), "sh": act["sh"].clone(),
       "opacity": torch.log(act["opacity"] / (1 - act["opacity"]).clamp_min(1e-6))}
# rates of ZERO: the optimiser does all of its work and the scene stays what it is, so every step projects the same rows
lr = {"pos": 0.0, "sh_dc": 0.0, "sh_rest": 0.0, "opacity": 0.0, "scale": 0.0, "rot": 0.0}
eng = mg.HipEngine(r, raw=raw, activated=act, lr=lr)
grads = {k: torch.zeros_like(act[k]) for k in KEYS}
img = torch.zeros(3, H, W, device=dev)
dL = torch.randn(3, H, W, device=dev)
cams = [L.get_lookat_cam(*view_pose(k), width=W, height=H) for k in range(8)]


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


print(f"# P = {P}, {W}x{H}; ms, median of {STEPS} steps after 2 warm-up steps (every phase between device synchronisations, so the")
print("# phases do not overlap: their sum is an upper bound of the step)")
# ---- the replicated-scene step's compute, for scale: fused forward + backward of one view, dense Adam on every row
rep = []
for step in range(1, STEPS + 3):
    a = timed(lambda: eng.forward_backward(cams[0], dL, grads))
    b = timed(lambda: eng.adam(grads, step))
    rep.append((a, b))
rep = np.median(np.array(rep[2:]), axis=0)
print(f"replicated scene, any N : forward+backward {rep[0]:.3f}  dense Adam on P rows {rep[1]:.3f}  -> {rep.sum():.3f} + the exchange")

for N in (1, 2, 4, 8):
    me = N // 2  # a rank in the middle of the row order
    first, count = L.api.owner_rows(P, N, me)
    # what the view's renderer would RECEIVE: every owner's records for view `me`, in owner order (untimed)
    rows_in, recs_in = [], []
    for o in range(N):
        f_o, c_o = L.api.owner_rows(P, N, o)
        rw, rc = r.owner_project(0, cams[me], f_o, c_o, keep_state=False)
        rows_in.append(rw.clone())
        recs_in.append(rc.clone())
    rows_in, recs_in = torch.cat(rows_in).contiguous(), torch.cat(recs_in).contiguous()
    g2d_out = torch.zeros(int(rows_in.shape[0]), r.OWNER_GRAD_FLOATS, device=dev)
    mine = r.owner_project_all(cams[:N], first, count, keep_state=True)  # (sizes of the incoming 2-D gradient rows)
    # (room for the rank's whole range: the per-splat kernel walks the slot's own row count)
    g2d_in = [torch.randn(count, r.OWNER_GRAD_FLOATS, device=dev) * 1e-3 for _ in mine]
    out = []
    for step in range(1, STEPS + 3):
        t1 = timed(lambda: r.owner_project_all(cams[:N], first, count, keep_state=True))
        t2 = timed(lambda: r.owner_render(cams[me], rows_in, recs_in, img, keep_state=True))
        t3 = timed(lambda: r.owner_render_backward(dL, g2d_out))

        def back():
            for v in range(N):
                r.owner_backward(v, g2d_in[v], *[grads[k] for k in KEYS], accumulate=v > 0)
        t4 = timed(back)
        t5 = timed(lambda: eng.adam(grads, step, rows=(first, count)))
        out.append((t1, t2, t3, t4, t5))
    m = np.median(np.array(out[2:]), axis=0)
    # the projections alone, as a step without read-back issues them: 20 calls back to back, no synchronisation in between
    import ctypes as C

    lib = L.api.load_library()
    outs = [(torch.empty(count, dtype=torch.int32, device=dev), torch.empty(count, 12, device=dev)) for _ in range(N)]
    rows_p = (C.c_void_p * N)(*[o[0].data_ptr() for o in outs])
    recs_p = (C.c_void_p * N)(*[o[1].data_ptr() for o in outs])
    arr = (L.api.Camera * N)(*cams[:N])

    def burst():
        for _ in range(20):
            assert lib.lcgs_owner_project_views(r.ctx._h, 0, N, arr, C.c_float(1.0), first, count, 1, rows_p, recs_p) == 0
    burst()
    back_to_back = min(timed(burst) for _ in range(3)) / 20
    print(f"ownership, N = {N} (rank {me}: {count} rows, {int(rows_in.shape[0])} on its screen): project x{N} {m[0]:.3f}  "
          f"view frame {m[1]:.3f}  render-backward {m[2]:.3f}  rows' backward x{N} {m[3]:.3f}  Adam on P/N rows {m[4]:.3f}  "
          f"-> {m.sum():.3f} + the exchange | project x{N} back to back, no read-back: {back_to_back:.3f}")
