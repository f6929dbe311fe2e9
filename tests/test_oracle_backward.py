"""The backward oracle (oracle/lcgs_oracle_bwd.c).  The reference has no backward; the oracle is pinned by
(a) the reference's own -- unused -- dL/dSH helpers (sh.hpp:37-165) through the committed golden vectors, and
(b) central finite differences of the f64 build of the forward restatement."""
import os

import numpy as np

from conftest import make_scene

POSE = ([-3, -0.5, 2.3], [0, 0, 0.5], [0, 0, 1])


def test_sh_gradient_matches_reference_helpers(oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, "sh_bands.npz"))
    n = g["dirs"].shape[0]
    cam = oracle.lookat(*POSE, width=64, height=64)
    campos = np.array(cam.position, np.float32)
    scene = {"pos": campos + g["dirs"] * np.float32(4.0), "scale": np.full((n, 3), 0.01, np.float32),
             "rotq": np.tile(np.array([1, 0, 0, 0], np.float32), (n, 1)), "sh": np.zeros((n, 48), np.float32)}
    for deg in range(4):
        sc = dict(scene)
        sc["sh"] = np.zeros((n, (deg + 1) ** 2 * 3), np.float32)  # raw colour = 0.5: clamp not saturated
        out = oracle.preprocess_backward(sc, cam, np.ones(n, np.int32), np.zeros((n, 2)), np.zeros((n, 3)),
                                         g["dL_dcolor"], sh_deg=deg)
        expect = g[f"dL_dsh_deg{deg}"][:, : (deg + 1) ** 2, :].reshape(n, -1)
        assert np.allclose(out["sh"], expect, rtol=2e-6, atol=2e-7), f"degree {deg}"


def test_backward_matches_finite_differences_f64(oracle64):
    """Smooth mode (alpha-skip / T-stop disabled in forward AND backward): the image is a smooth function of the
    parameters, so central differences validate every analytic Jacobian to ~1e-6."""
    o = oracle64
    rng = np.random.default_rng(1)
    P = 24
    scene = {k: v.astype(np.float64) for k, v in make_scene(rng, P, spread=0.35, log_scale=(-2.2, 0.4)).items()}
    scene["opacity"] = np.clip(scene["opacity"], 0.05, 0.9)
    W, H = 40, 28
    cam = o.lookat(*POSE, width=W, height=H)
    wt = rng.normal(size=(3, H, W))
    bg = (0.2, 0.3, 0.1)
    o.set_smooth(True)
    try:
        loss = lambda sc: float((o.render(sc, cam, bg=bg, scale_modifier=1.1)["img"] * wt).sum())
        g = o.render_backward_full(scene, cam, wt, bg=bg, scale_modifier=1.1)
        assert g["num_rendered"] > 0
        for name in ("pos", "scale", "rotq", "sh", "opacity"):
            flat, gf = scene[name].reshape(-1), g[name].reshape(-1)
            scale_g = np.abs(gf).max()
            for i in rng.choice(flat.size, min(25, flat.size), replace=False):
                h = 1e-6 * max(1.0, abs(flat[i]))
                old = flat[i]
                flat[i] = old + h
                lp = loss(scene)
                flat[i] = old - h
                lm = loss(scene)
                flat[i] = old
                fd = (lp - lm) / (2 * h)
                assert abs(fd - gf[i]) <= 2e-5 * max(abs(fd), abs(gf[i])) + 1e-7 * scale_g, (name, i, fd, gf[i])
    finally:
        o.set_smooth(False)


def test_backward_thresholds_gate_the_sums(oracle):
    """With the hard thresholds active the gradient of a culled / never-contributing splat is exactly zero."""
    rng = np.random.default_rng(2)
    scene = make_scene(rng, 300, log_scale=(-3.3, 0.6))
    scene["pos"][:10] = np.array(POSE[0]) - 2.0 * (np.array(POSE[1]) - np.array(POSE[0]))  # behind the camera
    scene["opacity"][10:20] = 1e-4  # alpha < 1/255 everywhere
    cam = oracle.lookat(*POSE, width=64, height=48)
    g = oracle.render_backward_full(scene, cam, rng.normal(size=(3, 48, 64)).astype(np.float32))
    for name in ("pos", "scale", "rotq", "sh", "opacity"):
        assert np.all(g[name][:10] == 0)
    assert np.all(g["opacity"][10:20] == 0) and np.all(g["scale"][10:20] == 0) and np.all(g["sh"][10:20] == 0)
    assert np.abs(g["pos"][20:]).max() > 0
