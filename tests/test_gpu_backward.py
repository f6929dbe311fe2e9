"""`-m gpu`: lcgs_render_backward against the f32 oracle backward (BASELINE tolerance: 1e-3 relative)."""
import numpy as np
import pytest
import torch

from conftest import make_scene
from gpu_util import DEV, dev, upload_scene

pytestmark = pytest.mark.gpu

POSE = ([-3, -0.5, 2.3], [0, 0, 0.5], [0, 0, 1])


def _grads(lcgs, oracle, scene, W, H, bg=(0.1, 0.2, 0.3), scale_modifier=1.0, seed=0):
    P = scene["pos"].shape[0]
    cam = lcgs.get_lookat_cam(*POSE, width=W, height=H)
    ocam = oracle.lookat(*POSE, width=W, height=H)
    dL = np.random.default_rng(seed).normal(size=(3, H, W)).astype(np.float32)
    r = lcgs.Renderer(lcgs.Context(0))
    d = upload_scene(scene)
    r.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"])
    img = torch.zeros(3, H, W, device=DEV)
    r.forward(cam, img, bg=bg, scale_modifier=scale_modifier, keep_state=True, sync=True)
    z = lambda *s: torch.full(s, 7.0, device=DEV)  # sentinel: the backward must overwrite everything
    g = {"pos": z(P, 3), "scale": z(P, 3), "rotq": z(P, 4), "sh": z(P, 48), "opacity": z(P)}
    r.backward(dev(dL), g["pos"], g["scale"], g["rotq"], g["sh"], g["opacity"])
    r.ctx.synchronize()
    ref = oracle.render_backward_full(scene, ocam, dL, bg=bg, scale_modifier=scale_modifier)
    return {k: v.cpu().numpy() for k, v in g.items()}, ref


def _check(got, ref):
    for name in ("pos", "scale", "rotq", "sh", "opacity"):
        a, b = got[name].astype(np.float64), ref[name].astype(np.float64)
        assert np.isfinite(a).all(), name
        rel = np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
        assert rel <= 1e-3, f"{name}: relative L2 error {rel:.2e}"
        tol = 1e-3 * np.abs(b) + 1e-3 * np.abs(b).max()
        frac_bad = float((np.abs(a - b) > tol).mean())
        assert frac_bad <= 1e-3, f"{name}: {frac_bad:.2e} of the elements off by more than 1e-3"
        # rows the oracle leaves at exactly zero (splats that never reached a pixel) must be exactly zero here too
        zero_rows = np.all(b.reshape(b.shape[0], -1) == 0, axis=1)
        assert not a.reshape(a.shape[0], -1)[zero_rows].any(), f"{name}: non-zero gradient on a row the oracle leaves at 0"


@pytest.mark.parametrize("P,res", [(300, (64, 48)), (5000, (160, 120)), (40000, (400, 300))])
def test_backward_matches_oracle(lcgs, oracle, P, res):
    rng = np.random.default_rng(P)
    scene = make_scene(rng, P, log_scale=(-3.6, 0.7))
    if P > 1000:
        scene["pos"][:50] = rng.normal(0, 0.3, (50, 3)) + POSE[0]
    got, ref = _grads(lcgs, oracle, scene, res[0], res[1])
    _check(got, ref)


def test_backward_culled_splats_get_exact_zeros(lcgs, oracle):
    rng = np.random.default_rng(5)
    scene = make_scene(rng, 2000, log_scale=(-3.6, 0.7))
    scene["pos"][:100] = np.array(POSE[0]) - 2.0 * (np.array(POSE[1]) - np.array(POSE[0]))
    scene["opacity"][100:150] = 1e-4
    got, ref = _grads(lcgs, oracle, scene, 128, 96, scale_modifier=1.2)
    _check(got, ref)
    for name in ("pos", "scale", "rotq", "sh", "opacity"):
        assert np.all(got[name][:150] == 0), name


def test_backward_requires_forward_state(lcgs):
    rng = np.random.default_rng(6)
    scene = make_scene(rng, 100)
    r = lcgs.Renderer(lcgs.Context(0))
    d = upload_scene(scene)
    r.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"])
    img = torch.zeros(3, 32, 32, device=DEV)
    r.forward(lcgs.get_lookat_cam(*POSE, width=32, height=32), img, keep_state=False)
    z = lambda *s: torch.zeros(*s, device=DEV)
    with pytest.raises(lcgs.LcgsError) as e:
        r.backward(z(3, 32, 32), z(100, 3), z(100, 3), z(100, 4), z(100, 48), z(100))
    assert e.value.status == 8  # LCGS_ERR_STATE


def test_backward_of_an_empty_frame_is_all_zeros(lcgs):
    """Nothing reaches the screen (every splat behind the camera): forward leaves the image alone, backward zero-fills."""
    rng = np.random.default_rng(9)
    scene = make_scene(rng, 700)
    scene["pos"][:] = np.array(POSE[0]) - 3.0 * (np.array(POSE[1]) - np.array(POSE[0]))
    r = lcgs.Renderer(lcgs.Context(0))
    d = upload_scene(scene)
    r.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"])
    img = torch.full((3, 48, 64), -1.0, device=DEV)
    assert r.forward(lcgs.get_lookat_cam(*POSE, width=64, height=48), img, keep_state=True) == 0
    assert (img == -1.0).all()
    g = {k: torch.full_like(d[k], 9.0) for k in ("pos", "scale", "rotq", "sh", "opacity")}
    r.backward(torch.randn(3, 48, 64, device=DEV), g["pos"], g["scale"], g["rotq"], g["sh"], g["opacity"])
    r.ctx.synchronize()
    assert all((t == 0).all() for t in g.values())
    # and a second backward on the same frame state is allowed
    r.backward(torch.randn(3, 48, 64, device=DEV), g["pos"], g["scale"], g["rotq"], g["sh"], g["opacity"])
    r.ctx.synchronize()
    assert all((t == 0).all() for t in g.values())


def test_backward_accumulate_sums_the_views_of_a_batch(lcgs, oracle):
    """lcgs_render_backward_accumulate: the second view's dense gradients are ADDED to the first view's (no zero-fill):
    the arrays then hold the sum the oracle gives for the two views."""
    rng = np.random.default_rng(19)
    P, W, H = 20000, 320, 240
    scene = make_scene(rng, P, log_scale=(-3.8, 0.7))
    poses = [POSE, ([2.5, 1.5, 1.0], [0, 0, 0.5], [0, 0, 1])]
    d = upload_scene(scene)
    r = lcgs.Renderer(lcgs.Context(0))
    r.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"])
    img = torch.zeros(3, H, W, device=DEV)
    keys = ("pos", "scale", "rotq", "sh", "opacity")
    g = {k: torch.full_like(d[k], 7.0) for k in keys}
    dLs = [np.random.default_rng(i).normal(size=(3, H, W)).astype(np.float32) for i in range(2)]
    for j, pose in enumerate(poses):
        r.forward(lcgs.get_lookat_cam(*pose, width=W, height=H), img, keep_state=True, sync=False)
        r.backward(dev(dLs[j]), *[g[k] for k in keys], accumulate=j > 0)
    r.ctx.synchronize()
    refs = [oracle.render_backward_full(scene, oracle.lookat(*p, width=W, height=H), dL) for p, dL in zip(poses, dLs)]
    for k in keys:
        a = g[k].cpu().numpy().astype(np.float64).ravel()
        b = sum(ref[k].astype(np.float64).ravel() for ref in refs)
        assert np.linalg.norm(a - b) / np.linalg.norm(b) <= 1e-3, k
    with pytest.raises(ValueError):
        r.backward(dev(dLs[0]), *[g[k] for k in keys], compact=True, accumulate=True)
