"""The ownership step at N = 1 through RCCL (lcgs_owner_step_forward / _backward / _finish, no read-back from the second
step on) + Adam on the own rows, on the bicycle stand-in: the driver of tools/gpu/owner_step_timeline.sh.  Marks every step
for rocprofv3 (--marker-trace): step_begin / forward_exit / backward_exit / finish_exit / adam_exit."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import luisacomputegaussiansplatting_amd as L  # noqa: E402
import luisacomputegaussiansplatting_amd.multi_gpu as mg  # noqa: E402

try:
    _roctx = ctypes.CDLL("librocprofiler-sdk-roctx.so")
    mark = lambda s: _roctx.roctxMarkA(s.encode())
except OSError:
    mark = lambda s: None

P = int(os.environ.get("LCGS_DRIVER_SPLATS", "6131954"))
W, H = 1920, 1080
steps = int(os.environ.get("LCGS_DRIVER_STEPS", "8"))
use_async = os.environ.get("LCGS_DRIVER_ASYNC", "1") == "1"
scene = L.synth_scene(1, 2001, P)
r = L.Renderer(L.Context(0))
r.upload_scene(scene)
act = r.scene_tensors()
KEYS = ("pos", "scale", "rotq", "sh", "opacity")
raw = {"pos": act["pos"].clone(), "scale": torch.log(act["scale"]), "rotq": act["rotq"].clone(), "sh": act["sh"].clone(),
       "opacity": torch.log(act["opacity"] / (1 - act["opacity"]).clamp_min(1e-6))}
lr = {"pos": 1.6e-6, "sh_dc": 2.5e-4, "sh_rest": 1.25e-5, "opacity": 5e-3, "scale": 5e-4, "rot": 1e-4}
eng = mg.HipEngine(r, raw=raw, activated=act, lr=lr)
grads = {k: torch.zeros_like(act[k]) for k in KEYS}
coll = mg.RcclCollective(r.ctx, 0, 1)
comm = coll.comm
if use_async:
    comm.owner_step_set_async(True)
cam = L.get_lookat_cam([-3, -0.5, 2.3], [0, 0, 0.5], [0, -1, 0], width=W, height=H)
img = torch.zeros(3, H, W, device="cuda:0")
dL = torch.randn(3, H, W, device="cuda:0")
rows = L.api.owner_rows(P, 1, 0)
times = []
for step in range(1, steps + 1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mark("step_begin")
    comm.owner_step_forward([cam], img)
    mark("forward_exit")
    comm.owner_step_backward(dL, grads)
    mark("backward_exit")
    redo = comm.owner_step_finish() if use_async else False
    mark("finish_exit")
    assert not redo
    eng.adam(grads, step, rows=rows)
    mark("adam_exit")
    torch.cuda.synchronize()
    times.append((time.perf_counter() - t0) * 1e3)
print("ms per step:", " ".join(f"{t:.3f}" for t in times), f"| median of the last {steps - 2}: {np.median(times[2:]):.3f}")
coll.close()
