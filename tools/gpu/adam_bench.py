"""Times the dense optimiser step alone (lcgs_adam_step on 6.13 M degree-3 splats): gpurun -- python tools/gpu/adam_bench.py
LCGS_ADAM_VARIANT selects experimental kernels (train.hip)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import luisacomputegaussiansplatting_amd as L  # noqa: E402

P = int(os.environ.get("P", 6131954))
dev = torch.device("cuda", 0)
r = L.Renderer(L.Context(0))
KEYS = ("pos", "scale", "rotq", "sh", "opacity")
shape = {"pos": (P, 3), "scale": (P, 3), "rotq": (P, 4), "sh": (P, 48), "opacity": (P,)}
mk = lambda: {k: torch.rand(*shape[k], device=dev) * 0.1 + 0.1 for k in KEYS}
g, raw, m, v = mk(), mk(), mk(), mk()
act = {"pos": raw["pos"], "scale": torch.exp(raw["scale"]), "rotq": raw["rotq"].clone(), "sh": raw["sh"],
       "opacity": torch.sigmoid(raw["opacity"])}
lr = {"pos": 1e-4, "sh_dc": 1e-3, "sh_rest": 1e-4, "opacity": 1e-2, "scale": 1e-3, "rot": 1e-3}
r.bind_scene(*[act[k] for k in KEYS])
for i in range(5):
    r.adam_step(g, raw, m, v, act, i + 1, lr)
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 30
for i in range(N):
    r.adam_step(g, raw, m, v, act, i + 6, lr)
r.ctx.synchronize()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) * 1e3 / N
gb = P * 59 * 4 * 7 / 1e9
print(os.environ.get("LCGS_ADAM_VARIANT", "0"), f"{ms:.3f} ms per dense step, {gb / ms:.2f} TB/s of 7 x 236 B/splat",
      "checksum", float(raw["sh"].double().sum()))
