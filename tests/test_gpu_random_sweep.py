"""`-m gpu`: randomised small frames (sizes, resolutions that are not multiples of 16, fields of view, scale
distributions, poses, backgrounds, scale modifiers) through the fused frame and its backward, against the oracle.
Every draw is seeded: a failure names its seed."""
import numpy as np
import pytest
import torch

from conftest import make_scene
from gpu_util import (DEV, assert_image_parity, assert_parity_vs_libm_expf, assert_parity_vs_numerics_variants, check_gradients, dev,
                      upload_scene)

pytestmark = pytest.mark.gpu


def _draw(seed):
    rng = np.random.default_rng(1000 + seed)
    P = int(rng.integers(1, 4000))
    W, H = int(rng.integers(17, 420)), int(rng.integers(17, 300))
    scene = make_scene(rng, P, spread=float(rng.uniform(0.2, 1.5)),
                       log_scale=(float(rng.uniform(-5.5, -2.0)), float(rng.uniform(0.2, 1.2))))
    if seed % 3 == 0:  # anisotropic needles and a few giants
        scene["scale"][:, 0] *= 8.0
        scene["scale"][: max(1, P // 50)] *= 25.0
    ang, elev, dist = rng.uniform(0, 2 * np.pi), rng.uniform(-0.6, 0.9), rng.uniform(0.3, 6.0)
    pos = [dist * np.cos(ang) * np.cos(elev), dist * np.sin(ang) * np.cos(elev), 0.5 + dist * np.sin(elev)]
    pose = (pos, [0.0, 0.0, 0.5], [0.0, 0.0, 1.0])
    return rng, scene, W, H, pose, float(rng.uniform(20.0, 110.0)), tuple(rng.uniform(0, 1, 3).tolist()), \
        float(rng.uniform(0.5, 1.5))


@pytest.mark.parametrize("seed", range(12))
def test_random_forward_frames(lcgs, oracle, seed):
    rng, scene, W, H, pose, fov, bg, sm = _draw(seed)
    P = scene["pos"].shape[0]
    cam = lcgs.get_lookat_cam(*pose, width=W, height=H)
    cam.fov = fov
    ocam = oracle.lookat(*pose, width=W, height=H, fov=fov)
    r = lcgs.Renderer(lcgs.Context(0))
    d = upload_scene(scene)
    r.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"])
    img = torch.full((3, H, W), -1.0, device=DEV)
    radii = torch.full((P,), -7, dtype=torch.int32, device=DEV)
    n = r.forward(cam, img, bg=bg, scale_modifier=sm, radii=radii, sync=True)
    ref = oracle.render(scene, ocam, bg=bg, scale_modifier=sm, ambig_eps=1e-5)
    assert n == ref["num_rendered"], f"seed {seed}"
    assert np.array_equal(radii.cpu().numpy(), ref["radii"]), f"seed {seed}"
    if n:
        assert_image_parity(img.cpu().numpy(), ref)
        assert_parity_vs_libm_expf(img.cpu().numpy(), oracle, scene, ocam, bg=bg, scale_modifier=sm)
        if seed % 4 == 0:  # (every draw runs on the CPU in tests/test_oracle_numerics.py; here the HIP frame is the one held)
            from oracle import numerics

            rep, _ = numerics.report(scene, ocam, bg=bg, scale_modifier=sm, img=img.cpu().numpy())
            assert all(v["all_explained"] for v in rep["variants"].values()), rep


@pytest.mark.parametrize("seed", range(5))
def test_random_backward_frames(lcgs, oracle, oracle64, seed):
    rng, scene, W, H, pose, fov, bg, sm = _draw(100 + seed)
    P = scene["pos"].shape[0]
    cam = lcgs.get_lookat_cam(*pose, width=W, height=H)
    cam.fov = fov
    ocam = oracle.lookat(*pose, width=W, height=H, fov=fov)
    dL = rng.normal(size=(3, H, W)).astype(np.float32)
    r = lcgs.Renderer(lcgs.Context(0))
    d = upload_scene(scene)
    r.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"])
    img = torch.zeros(3, H, W, device=DEV)
    n = r.forward(cam, img, bg=bg, scale_modifier=sm, keep_state=True, sync=True)
    if n == 0:
        pytest.skip("nothing on screen for this draw")
    g = {k: torch.full_like(d[k], 3.0) for k in ("pos", "scale", "rotq", "sh", "opacity")}
    r.backward(dev(dL), g["pos"], g["scale"], g["rotq"], g["sh"], g["opacity"])
    r.ctx.synchronize()
    ref32 = oracle.render_backward_full(scene, ocam, dL, bg=bg, scale_modifier=sm)
    ref64 = oracle64.render_backward_full(scene, oracle64.lookat(*pose, width=W, height=H, fov=fov), dL, bg=bg,
                                          scale_modifier=sm)
    radii = oracle.render(scene, ocam, bg=bg, scale_modifier=sm)["radii"]
    check_gradients(g, ref32, ref64, P, radii, f"seed {seed}")  # every row, the f64 oracle: gpu_util.check_gradients
