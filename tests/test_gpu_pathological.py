"""`-m gpu`: inputs a trained scene should never contain but a drop-in must survive: non-finite geometry, degenerate
scales and rotations, opacities outside (0, 1)."""
import numpy as np
import pytest
import torch

from conftest import make_scene
from gpu_util import DEV, assert_image_parity, upload_scene

pytestmark = pytest.mark.gpu
POSE = ([-3, -0.5, 2.3], [0, 0, 0.5], [0, 0, 1])


def _render(lcgs, scene, W, H, keep=False):
    r = lcgs.Renderer(lcgs.Context(0))
    d = upload_scene(scene)
    r.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"])
    img = torch.full((3, H, W), 0.25, device=DEV)
    n = r.forward(lcgs.get_lookat_cam(*POSE, width=W, height=H), img, bg=(0.25, 0.25, 0.25), keep_state=keep, sync=True)
    return r, d, img, n


def test_degenerate_but_finite_splats_match_the_oracle(lcgs, oracle):
    rng = np.random.default_rng(5)
    scene = make_scene(rng, 3000, log_scale=(-3.8, 0.7))
    scene["opacity"][0:20] = 0.0
    scene["opacity"][20:40] = -0.5
    scene["opacity"][40:60] = 7.0          # alpha is capped at 0.99 (shader.cpp:258)
    scene["scale"][60:80] = 0.0            # only the 0.3 px low-pass filter remains
    scene["scale"][80:83] = 50.0           # covers the whole frame
    scene["scale"][83:90, 0] = 1e-12
    scene["rotq"][90:100] *= 1e-3          # un-normalised quaternions are used as they are
    W, H = 256, 192
    r, d, img, n = _render(lcgs, scene, W, H)
    ref = oracle.render(scene, oracle.lookat(*POSE, width=W, height=H), bg=(0.25, 0.25, 0.25), ambig_eps=1e-5)
    assert n == ref["num_rendered"]
    assert_image_parity(img.cpu().numpy(), ref)


def test_non_finite_geometry_is_invisible_and_harmless(lcgs):
    rng = np.random.default_rng(6)
    base = make_scene(rng, 2000, log_scale=(-3.8, 0.7))
    bad = {k: v.copy() for k, v in base.items()}
    bad["pos"][0] = np.nan
    bad["pos"][1, 2] = np.inf
    bad["scale"][2] = np.nan
    bad["scale"][3] = np.inf
    bad["rotq"][4] = 0.0
    bad["rotq"][5] = np.nan
    bad["opacity"][6] = np.nan
    bad["sh"][7] = np.nan                  # colour NaN: clamp(NaN) -- the splat must not poison its neighbours' pixels
    bad["pos"][7] = [100.0, 100.0, -100.0]  # ... and it is parked off screen
    good_idx = np.arange(8, 2000)
    good = {k: np.ascontiguousarray(v[good_idx]) for k, v in base.items()}
    W, H = 256, 192
    _, _, img_bad, n_bad = _render(lcgs, bad, W, H, keep=True)
    _, _, img_good, n_good = _render(lcgs, good, W, H)
    assert torch.isfinite(img_bad).all()
    assert n_bad >= n_good
    # a zero quaternion gives a zero covariance (a filter-sized dot): visible and legal; everything else is gone
    diff = (img_bad - img_good).abs()
    assert (diff > 1e-4).float().mean().item() < 0.01
    # the backward of such a frame stays finite for every well-formed splat
    r, d, img, n = _render(lcgs, bad, W, H, keep=True)
    g = {k: torch.zeros_like(d[k]) for k in ("pos", "scale", "rotq", "sh", "opacity")}
    r.backward(torch.ones(3, H, W, device=DEV), g["pos"], g["scale"], g["rotq"], g["sh"], g["opacity"])
    r.ctx.synchronize()
    for k, t in g.items():
        assert torch.isfinite(t[8:]).all(), k
