"""`-m gpu`: the cull kernel's conservative screen test (phase 1 of k_cull_compact) never rejects a splat the full
projection would keep.  A frame rendered WITHOUT a radii output uses the two-phase kernel; the same frame WITH a
radii output sends every splat through the full projection.  Both must produce the same sorted lists, counts and
image, bit for bit -- on scenes built to sit on the bound: splats hugging the screen edges from outside, long
needles pointing at the screen, un-normalised quaternions, negative scales, scale modifiers, skewed cameras."""
import numpy as np
import pytest
import torch

from conftest import make_scene
from gpu_util import DEV, upload_scene

pytestmark = pytest.mark.gpu


def _edge_scene(rng, P):
    """Splats on a wide shell around the view axis: most centres fall outside the frustum by 0 .. a few radii."""
    sc = make_scene(rng, P, spread=2.5, log_scale=(-2.5, 1.2))
    k = P // 4
    sc["scale"][:k, 0] *= rng.uniform(5, 60, k).astype(np.float32)          # needles
    sc["rotq"][k:2 * k] *= rng.uniform(0.3, 2.5, (k, 1)).astype(np.float32)  # |q| != 1 scales the rotation matrix
    sc["scale"][2 * k:2 * k + k // 2] *= -1.0                                # sign of a scale does not matter to Sigma
    sc["opacity"][:] = np.clip(sc["opacity"], 0.02, 1.0)
    return sc


def _skewed(lcgs, cam):
    d = cam.to_dict()
    r, u, f = (np.asarray(d[k], np.float64) for k in ("right", "up", "front"))
    d["right"] = (1.15 * r + 0.2 * f).tolist()   # neither unit length nor orthogonal to front
    d["up"] = (0.9 * u - 0.15 * f + 0.05 * r).tolist()
    return lcgs.Camera.from_dict(d)


def _lists(lcgs, r, cam, W, H, with_radii, scale_modifier):
    img = torch.full((3, H, W), -1.0, device=DEV)
    radii = torch.full((r.P,), -7, dtype=torch.int32, device=DEV) if with_radii else None
    n = r.forward(cam, img, scale_modifier=scale_modifier, radii=radii, keep_state=True, sync=True)
    st = r.frame_stats()
    G = ((W + 15) // 16) * ((H + 15) // 16)
    lst = torch.zeros(max(1, st["num_pairs"]), dtype=torch.int32, device=DEV)
    rng_ = torch.zeros(2 * G, dtype=torch.int32, device=DEV)
    if st["num_pairs"] > 0:
        r.last_lists(lst, rng_)
    return n, st, img, lst, rng_


@pytest.mark.parametrize("seed,res,scale_modifier,skew", [(1, (640, 360), 1.0, False), (2, (333, 517), 2.5, False),
                                                         (3, (1920, 1080), 0.4, False), (4, (640, 480), 1.0, True),
                                                         (5, (97, 61), 6.0, True)])
def test_two_phase_cull_equals_full_projection(lcgs, seed, res, scale_modifier, skew):
    rng = np.random.default_rng(seed)
    P = 120_000
    scene = _edge_scene(rng, P)
    d = upload_scene(scene)
    W, H = res
    for ang in (0.0, 1.1, 2.7):
        eye = [-4.0 * np.cos(ang), 4.0 * np.sin(ang), 1.0]
        cam = lcgs.get_lookat_cam(eye, [0, 0, 0.5], [0, 0, 1], width=W, height=H)
        if skew:
            cam = _skewed(lcgs, cam)
        r = lcgs.Renderer(lcgs.Context(0))
        r.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"])
        r.P = P
        a = _lists(lcgs, r, cam, W, H, False, scale_modifier)
        b = _lists(lcgs, r, cam, W, H, True, scale_modifier)
        assert a[0] == b[0], (seed, ang, a[0], b[0])                  # the reference's num_rendered
        assert a[1]["num_pairs"] == b[1]["num_pairs"] and a[1]["num_visible"] == b[1]["num_visible"], (a[1], b[1])
        assert a[0] > 0 and a[1]["num_visible"] < P                   # the frame draws, and something was culled
        assert torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])    # sorted lists and tile ranges
        assert torch.equal(a[2], b[2])


def test_screen_bound_with_extreme_values(lcgs):
    """Huge / tiny / non-finite inputs: the bound overflows to inf or NaN and must then keep the splat a candidate."""
    rng = np.random.default_rng(9)
    P = 20_000
    scene = _edge_scene(rng, P)
    scene["scale"][0:50] = 1e18
    scene["scale"][50:100] = 1e-30
    scene["pos"][100:150] *= 1e6
    scene["rotq"][150:200] *= 1e10
    scene["rotq"][200:220] = 0.0
    scene["pos"][220:230] = np.nan
    scene["scale"][230:240] = np.inf
    scene["rotq"][240:250] = np.nan
    d = upload_scene(scene)
    W, H = 512, 288
    cam = lcgs.get_lookat_cam([-4.0, 0.0, 1.0], [0, 0, 0.5], [0, 0, 1], width=W, height=H)
    r = lcgs.Renderer(lcgs.Context(0))
    r.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"])
    r.P = P
    a = _lists(lcgs, r, cam, W, H, False, 1.0)
    b = _lists(lcgs, r, cam, W, H, True, 1.0)
    assert a[0] == b[0] and a[1]["num_pairs"] == b[1]["num_pairs"] and a[1]["num_visible"] == b[1]["num_visible"]
    assert torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])
    assert torch.equal(a[2].nan_to_num(7.0), b[2].nan_to_num(7.0))
