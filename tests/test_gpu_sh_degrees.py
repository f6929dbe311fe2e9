"""`-m gpu`: SH degrees below 3 (sh_preprocessor.cpp:91-147 gates each band on the level; the reference's app only ever
passes 3) through the fused frame and its backward -- the code paths without the staged 192-byte rows, the colour
Jacobian or the f16 copy."""
import numpy as np
import pytest
import torch

from conftest import make_scene
from gpu_util import DEV, assert_image_parity, dev

pytestmark = pytest.mark.gpu
POSE = ([-3, -0.5, 2.3], [0, 0, 0.5], [0, 0, 1])


@pytest.mark.parametrize("deg", [0, 1, 2])
def test_lower_sh_degrees_forward_and_backward(lcgs, oracle, deg):
    rng = np.random.default_rng(50 + deg)
    scene = make_scene(rng, 6000, log_scale=(-3.8, 0.7))
    feat = (deg + 1) ** 2 * 3
    scene["sh"] = np.ascontiguousarray(scene["sh"][:, :feat])
    W, H = 200, 150
    cam = lcgs.get_lookat_cam(*POSE, width=W, height=H)
    ocam = oracle.lookat(*POSE, width=W, height=H)
    r = lcgs.Renderer(lcgs.Context(0))
    d = {k: torch.from_numpy(v).to(DEV) for k, v in scene.items()}
    r.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"], sh_degree=deg)
    img = torch.zeros(3, H, W, device=DEV)
    n = r.forward(cam, img, bg=(0.1, 0.0, 0.2), keep_state=True, sync=True)
    ref = oracle.render(scene, ocam, bg=(0.1, 0.0, 0.2), sh_deg=deg, ambig_eps=1e-5)
    assert n == ref["num_rendered"]
    assert_image_parity(img.cpu().numpy(), ref)
    dL = rng.normal(size=(3, H, W)).astype(np.float32)
    g = {k: torch.full_like(d[k], 5.0) for k in ("pos", "scale", "rotq", "sh", "opacity")}
    r.backward(dev(dL), g["pos"], g["scale"], g["rotq"], g["sh"], g["opacity"])
    r.ctx.synchronize()
    gref = oracle.render_backward_full(scene, ocam, dL, bg=(0.1, 0.0, 0.2), sh_deg=deg)
    for k in g:
        a, b = g[k].cpu().numpy().astype(np.float64).ravel(), gref[k].astype(np.float64).ravel()
        assert np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30) <= 1e-3, (deg, k)
    if deg < 3:
        with pytest.raises(lcgs.LcgsError):
            r.use_half_sh(True)  # the f16 copy exists for degree 3 only
