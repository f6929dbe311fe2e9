"""`-m gpu`: the opt-in footprint (LOD) cull of the fused frame (lcgs_set_lod; SURVEY 8f rank 4, doc/roadmap.md:8) against
the oracle with the same rule restated (oracle.set_lod_min_radius).  Never the default: with the threshold at 0 the frame
is the reference's, bit for bit."""
import numpy as np
import pytest
import torch

from conftest import make_scene
from gpu_util import DEV, assert_image_parity, upload_scene

pytestmark = pytest.mark.gpu

POSE = ([-3, -0.5, 2.3], [0, 0, 0.5], [0, 0, 1])
KEYS = ("pos", "scale", "rotq", "sh", "opacity")


@pytest.mark.parametrize("min_radius", [4, 6])  # (3 is the smallest radius the reference's formula yields)
def test_lod_cull_matches_the_oracle_rule(lcgs, oracle, min_radius):
    rng = np.random.default_rng(31)
    P, W, H = 60000, 640, 480
    scene = make_scene(rng, P, log_scale=(-5.0, 1.0))  # many sub-pixel splats
    d = upload_scene(scene)
    r = lcgs.Renderer(lcgs.Context(0))
    r.bind_scene(*[d[k] for k in KEYS])
    cam = lcgs.get_lookat_cam(*POSE, width=W, height=H)
    ocam = oracle.lookat(*POSE, width=W, height=H)
    full, radii0 = torch.zeros(3, H, W, device=DEV), torch.zeros(P, dtype=torch.int32, device=DEV)
    n_full = r.forward(cam, full, radii=radii0)
    r.set_lod(min_radius)
    img, radii = torch.zeros(3, H, W, device=DEV), torch.full((P,), -7, dtype=torch.int32, device=DEV)
    n = r.forward(cam, img, radii=radii, keep_state=True)
    oracle.set_lod_min_radius(min_radius)
    try:
        orc = oracle.render(scene, ocam, ambig_eps=1e-5)
    finally:
        oracle.set_lod_min_radius(0)
    assert n == orc["num_rendered"] and 0 < n < n_full
    assert np.array_equal(radii.cpu().numpy(), orc["radii"])
    r0 = radii0.cpu().numpy()
    dropped = (r0 > 0) & (r0 < min_radius)
    assert dropped.any() and not radii.cpu().numpy()[dropped].any()
    assert np.array_equal(radii.cpu().numpy()[~dropped], r0[~dropped])
    max_clear, _ = assert_image_parity(img.cpu().numpy(), orc)
    assert max_clear <= 1e-4
    assert (img - full).abs().max().item() > 1e-4  # it is a different image: a quality / speed trade, hence opt-in
    # culled splats receive exact zero gradients
    g = {k: torch.full_like(d[k], 9.0) for k in KEYS}
    r.backward(torch.randn(3, H, W, device=DEV), *[g[k] for k in KEYS])
    r.ctx.synchronize()
    idx = torch.from_numpy(np.nonzero(dropped)[0]).to(DEV)
    assert all((g[k][idx] == 0).all() for k in KEYS)
    # camera batches (sibling context) follow the setting; 0 restores the reference's frame bit for bit
    pair = [torch.zeros(3, H, W, device=DEV) for _ in range(2)]
    r.forward_batch([cam, cam], pair)
    r.ctx.synchronize()
    assert torch.equal(pair[0], img) and torch.equal(pair[1], img)
    r.set_lod(0)
    again = torch.zeros(3, H, W, device=DEV)
    assert r.forward(cam, again) == n_full
    assert torch.equal(again, full)


def test_lod_argument_check(lcgs):
    r = lcgs.Renderer(lcgs.Context(0))
    with pytest.raises(lcgs.LcgsError):
        r.set_lod(-1)
