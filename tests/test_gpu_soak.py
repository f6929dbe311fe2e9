"""`-m gpu`, opt-in (LCGS_SOAK=<draws>): many more seeded random frames than the default sweep, forward through BOTH scene
paths (caller-bound arrays in the given order; context-owned upload in the library's default spatial order) and backward,
against the oracle.  One process, one context per draw; a failure names its seed."""
import os

import numpy as np
import pytest
import torch

from gpu_util import DEV, assert_image_parity, dev, upload_scene
from test_gpu_random_sweep import _draw

pytestmark = pytest.mark.gpu
N = int(os.environ.get("LCGS_SOAK", "0"))


@pytest.mark.skipif(N <= 0, reason="opt-in: set LCGS_SOAK=<number of draws>")
def test_soak_random_frames(lcgs, oracle, oracle64):
    flipped_total, worst_ratio = 0, 0.0
    for seed in range(1000, 1000 + N):
        rng, scene, W, H, pose, fov, bg, sm = _draw(seed)
        P = scene["pos"].shape[0]
        cam = lcgs.get_lookat_cam(*pose, width=W, height=H)
        cam.fov = fov
        ocam = oracle.lookat(*pose, width=W, height=H, fov=fov)
        ref = oracle.render(scene, ocam, bg=bg, scale_modifier=sm, ambig_eps=1e-5)
        for owned in (False, True):
            r = lcgs.Renderer(lcgs.Context(0))
            if owned:
                r.upload_scene(scene)  # context-owned, spatial order
            else:
                d = upload_scene(scene)
                r.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"])
            img = torch.full((3, H, W), -1.0, device=DEV)
            radii = torch.full((P,), -7, dtype=torch.int32, device=DEV)
            n = r.forward(cam, img, bg=bg, scale_modifier=sm, radii=radii, sync=True)
            assert n == ref["num_rendered"], f"seed {seed} owned={owned}"
            rad = radii.cpu().numpy()
            perm = r.permutation() if owned else None
            if perm is not None:
                back = np.empty_like(rad)
                back[perm.cpu().numpy().astype(np.int64)] = rad
                rad = back
            assert np.array_equal(rad, ref["radii"]), f"seed {seed} owned={owned}"
            if n:
                try:
                    _, flipped = assert_image_parity(img.cpu().numpy(), ref)
                except AssertionError as e:
                    raise AssertionError(f"seed {seed} owned={owned}: {e}")
                flipped_total += flipped
            # a second frame on the same context (previous-frame schedule, hints) must repeat the first
            img2 = torch.full((3, H, W), -1.0, device=DEV)
            assert r.forward(cam, img2, bg=bg, scale_modifier=sm, sync=True) == n
            assert torch.equal(img, img2), f"seed {seed} owned={owned}: second frame differs"
        if seed % 4 == 0 and ref["num_rendered"] > 0:  # every fourth draw: gradients too
            d = upload_scene(scene)
            r = lcgs.Renderer(lcgs.Context(0))
            r.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"])
            r.forward(cam, img, bg=bg, scale_modifier=sm, keep_state=True, sync=True)
            dL = rng.normal(size=(3, H, W)).astype(np.float32)
            g = {k: torch.full_like(d[k], 3.0) for k in ("pos", "scale", "rotq", "sh", "opacity")}
            r.backward(dev(dL), g["pos"], g["scale"], g["rotq"], g["sh"], g["opacity"])
            r.ctx.synchronize()
            # the yardstick of test_gpu_random_sweep.py: the f64 oracle, and the BASELINE tolerance or -- on ill-conditioned
            # draws (needles, giants: every third seed), where f32 itself is only good to a few 1e-3 -- a small multiple of
            # the f32 oracle's own error (3x here: the kernels use v_rcp_f32 and FMAs, and this net is cast wide for gross
            # errors, not to characterise precision; the worst ratio is printed)
            ref32 = oracle.render_backward_full(scene, ocam, dL, bg=bg, scale_modifier=sm)
            ref64 = oracle64.render_backward_full(scene, oracle64.lookat(*pose, width=W, height=H, fov=fov), dL, bg=bg,
                                                  scale_modifier=sm)
            rel = lambda x, y: np.linalg.norm(x - y) / max(np.linalg.norm(y), 1e-30)
            # Screen-filling giants (radius > 64 px: every third seed plants some next to the camera) are left out of the
            # norms: their geometry gradients are sums of ~1e5 cancelling terms in which the render-backward's v_rcp_f32 /
            # v_exp_f32 (1e-7 per term, by design: DESIGN.md 5) are amplified to several 1e-3 where the f32 oracle's IEEE
            # divide and libm exp stay at several 1e-4 (tests/debug/grad_outlier.py shows one such splat carrying all of
            # the excess).  They still have to be finite.
            keep = ref["radii"] <= 64
            for k in g:
                a = g[k].cpu().numpy().astype(np.float64)
                assert np.isfinite(a).all(), (seed, k)
                a = a.reshape(P, -1)[keep].ravel()
                b32 = ref32[k].astype(np.float64).reshape(P, -1)[keep].ravel()
                b64 = ref64[k].astype(np.float64).reshape(P, -1)[keep].ravel()
                assert rel(a, b64) <= max(1e-3, 3.0 * rel(b32, b64)), (seed, k, rel(a, b64), rel(b32, b64))
                if rel(a, b64) > 1e-3:
                    worst_ratio = max(worst_ratio, rel(a, b64) / max(rel(b32, b64), 1e-30))
    print(f"[soak] {N} draws, {flipped_total} threshold-flipped pixels in total; worst gradient error beyond 1e-3 = "
          f"{worst_ratio:.2f} x the f32 oracle's own error against f64")
