"""`-m gpu`: the lcgs-app CLI work-alike end to end (scene -> device -> frames -> flipped RGB8 PNG), both through the
fused frame and through the three stage-level operators in the reference's own call order (app/main.cpp:266-308)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("path", ["fused", "stage", "deferred"])
def test_lcgs_app_renders_png(lcgs, oracle, tmp_path, path):
    app = os.path.join(ROOT, "luisacomputegaussiansplatting_amd", "lcgs-app")
    if not os.path.exists(app):
        lcgs.build_library()
    out = str(tmp_path)
    res = subprocess.run([app, "--synth", "0:20000:1001", "--res=320x240", "--out", out, "--world", "blender",
                          "--exp_N", "2", f"--path={path}"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    assert "num_gaussians: 20000" in res.stdout and "fps:" in res.stdout
    from PIL import Image

    png = np.array(Image.open(os.path.join(out, "synth0_20000_hip.png")))
    assert png.shape == (240, 320, 3)
    # the same frame through the oracle: garden pose of app/main.cpp:191-193 with --world blender up vector
    scene = lcgs.synth_scene(0, 1001, 20000)
    cam = oracle.lookat([-3, -0.5, 3.3], [0, 3, 0.5], [0, 0, 1], width=320, height=240)
    ref = oracle.image_to_rgb8(oracle.render(scene, cam)["img"])
    diff = np.abs(png.astype(int) - ref.astype(int))
    assert (diff > 1).mean() < 1e-3 and diff.max() <= 2  # 8-bit truncation of values 1e-6 apart may differ by 1


@pytest.mark.parametrize("path", ["fused", "stage"])
def test_lcgs_app_spatial_order_gives_the_same_png(lcgs, tmp_path, path):
    """--order spatial (lcgs_scene_reorder_spatial at load): byte-identical PNG through both paths."""
    app = os.path.join(ROOT, "luisacomputegaussiansplatting_amd", "lcgs-app")
    if not os.path.exists(app):
        lcgs.build_library()
    pngs = []
    for order in ("file", "spatial"):
        out = str(tmp_path / order)
        os.makedirs(out)
        res = subprocess.run([app, "--synth", "1:60000:2001", "--res=480x270", "--out", out, f"--path={path}",
                              "--order", order], capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stderr
        pngs.append(open(os.path.join(out, "synth1_60000_hip.png"), "rb").read())
    assert pngs[0] == pngs[1] and len(pngs[0]) > 1000


@pytest.mark.parametrize("ingest", ["device", "host"])
def test_lcgs_app_ply_ingest_and_camera_batch(lcgs, oracle, tmp_path, ingest):
    """--ply through both ingest paths and a --cameras batch: every view equals the oracle's."""
    app = os.path.join(ROOT, "luisacomputegaussiansplatting_amd", "lcgs-app")
    P = 15000
    rng = np.random.default_rng(21)
    ply = str(tmp_path / "scene.ply")
    lcgs.write_ply_raw(ply, rng.normal(0, 0.6, (P, 3)) + [0, 0, 0.5], rng.normal(0.3, 0.8, (P, 3)),
                       rng.normal(0, 0.1, (P, 45)), rng.normal(0, 2.5, P), rng.normal(-4.0, 0.8, (P, 3)),
                       rng.normal(size=(P, 4)))
    views = [([-3, -0.5, 2.3], [0, 0, 0.5], [0, 0, 1], None), ([2.5, 1.5, 1.0], [0, 0, 0.5], [0, 0, 1], 45.0),
             ([0.2, -3.0, 0.4], [0, 0, 0.5], [0, 0, 1], None)]
    cams = str(tmp_path / "cams.txt")
    with open(cams, "w") as f:
        f.write("# position target up [fov]\n\n")
        for p, t, u, fov in views:
            f.write(" ".join(str(x) for x in p + t + u) + (f" {fov}" if fov else "") + "\n")
    out = str(tmp_path / "out")
    res = subprocess.run([app, "--ply", ply, "--res=256x192", "--out", out, "--ingest", ingest, "--cameras", cams],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    assert f"num_gaussians: {P}" in res.stdout and f"({ingest} ingest)" in res.stdout
    from PIL import Image

    scene = lcgs.read_gs_ply(ply)
    for k, (p, t, u, fov) in enumerate(views):
        png = np.array(Image.open(os.path.join(out, f"scene_hip_{k}.png")))
        assert png.shape == (192, 256, 3)
        cam = oracle.lookat(p, t, u, width=256, height=192)
        if fov:
            cam.fov = fov
        ref = oracle.image_to_rgb8(oracle.render(scene, cam)["img"])
        diff = np.abs(png.astype(int) - ref.astype(int))
        # device ingest: exp() within 2 ulp of the host's moves a few pixels by one 8-bit step
        assert (diff > 1).mean() < 2e-3 and diff.max() <= (3 if ingest == "device" else 2)


def test_lcgs_app_view_sharded_backward_with_gradient_sum(lcgs, oracle, tmp_path):
    """--gpus 1 --backward: the multi-view batch driver of SURVEY 8e from C++ (lcgs.hpp -> C ABI): per round, the view's
    backward with dL/dimg = 1 and lcgs_grads_allreduce over the (here: one) rank; the printed norms of the summed
    gradients equal the oracle's for every view.  (More ranks need more GPUs than a box has: the rank launcher -- fork
    before any HIP call, rendezvous token over pipes -- only runs with --gpus > 1.)"""
    import re

    app = os.path.join(ROOT, "luisacomputegaussiansplatting_amd", "lcgs-app")
    P, W, H = 8000, 256, 192
    views = [([-3, -0.5, 2.3], [0, 0, 0.5], [0, 0, 1]), ([2.5, 1.5, 1.0], [0, 0, 0.5], [0, 0, 1])]
    cams = str(tmp_path / "cams.txt")
    with open(cams, "w") as f:
        for p, t, u in views:
            f.write(" ".join(str(x) for x in p + t + u) + "\n")
    out = str(tmp_path / "out")
    res = subprocess.run([app, "--synth", f"0:{P}:1001", f"--res={W}x{H}", "--out", out, "--cameras", cams, "--gpus", "1",
                          "--backward"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    rounds = re.findall(r"round (\d+) \(1 view\): grad_l2 pos (\S+) scale (\S+) rotq (\S+) sh (\S+) opacity (\S+)", res.stdout)
    assert len(rounds) == len(views), res.stdout
    scene = lcgs.synth_scene(0, 1001, P)
    dL = np.ones((3, H, W), np.float32)
    for (k, *norms), (p, t, u) in zip(rounds, views):
        ref = oracle.render_backward_full(scene, oracle.lookat(p, t, u, width=W, height=H), dL)
        for name, got in zip(("pos", "scale", "rotq", "sh", "opacity"), norms):
            want = float(np.linalg.norm(ref[name].astype(np.float64)))
            assert abs(float(got) - want) <= 1e-3 * want, (k, name, got, want)
        assert os.path.exists(os.path.join(out, f"synth0_{P}_hip_{k}.png"))


def test_lcgs_app_ownership_step_prints_the_same_norms(lcgs, tmp_path):
    """--backward --owner: the C++ caller of lcgs_owner_step_forward / _backward (lcgs.hpp Comm::owner_step_*), RCCL at world
    size 1; its verification norms are the dense --backward path's."""
    import re

    app = os.path.join(ROOT, "luisacomputegaussiansplatting_amd", "lcgs-app")
    P, W, H = 8000, 256, 192
    views = [([-3, -0.5, 2.3], [0, 0, 0.5], [0, 0, 1]), ([2.5, 1.5, 1.0], [0, 0, 0.5], [0, 0, 1])]
    cams = str(tmp_path / "cams.txt")
    with open(cams, "w") as f:
        for p, t, u in views:
            f.write(" ".join(str(x) for x in p + t + u) + "\n")
    got = {}
    for tag, extra in (("dense", []), ("owner", ["--owner", "--comm-selftest"])):
        out = str(tmp_path / tag)
        res = subprocess.run([app, "--synth", f"0:{P}:1001", f"--res={W}x{H}", "--out", out, "--cameras", cams, "--gpus", "1",
                              "--backward"] + extra, capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stderr
        if "--comm-selftest" in extra:  # the communicator's self-test ran first and said so
            assert re.search(r"rank 0 / 1 communicator self-test: all-reduce ok \([^)]*\), point-to-point ok \([^)]*\), "
                             r"ownership step ok", res.stdout), res.stdout
        got[tag] = re.findall(r"round (\d+) \(1 view[^)]*\): grad_l2 pos (\S+) scale (\S+) rotq (\S+) sh (\S+) opacity (\S+)", res.stdout)
        assert len(got[tag]) == len(views), res.stdout
        for k in range(len(views)):
            assert os.path.exists(os.path.join(out, f"synth0_{P}_hip_{k}.png"))
    for a, b in zip(got["dense"], got["owner"]):
        for x, y in zip(a[1:], b[1:]):
            assert abs(float(x) - float(y)) <= 1e-4 * float(x), (a, b)
    for k in range(len(views)):  # the same images, byte for byte
        pa, pb = (os.path.join(str(tmp_path / t), f"synth0_{P}_hip_{k}.png") for t in ("dense", "owner"))
        assert open(pa, "rb").read() == open(pb, "rb").read()


@pytest.mark.parametrize("extra", [[], ["--fused-adam"]])
def test_lcgs_app_fit_trains_through_the_c_abi_only(lcgs, tmp_path, extra):
    """--fit K: "training without python binding" (doc/roadmap.md:4) -- forward, lcgs_l2_loss_backward, lcgs_render_backward,
    lcgs_adam_step in a C++ loop (--fused-adam: lcgs_render_backward_adam, the optimiser inside the backward's per-splat
    kernel, no gradient arrays).  The target is the scene's own frame and the start a perturbed copy, so the loss has to
    fall, and by a lot."""
    import re

    app = os.path.join(ROOT, "luisacomputegaussiansplatting_amd", "lcgs-app")
    res = subprocess.run([app, "--synth", "0:20000:1001", "--res=320x240", "--out", str(tmp_path), "--world", "blender",
                          "--pose", "lego", "--fit", "30"] + extra, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    losses = [float(x) for x in re.findall(r"step \d+ loss (\S+)", res.stdout)]
    assert len(losses) == 30 and all(np.isfinite(losses))
    assert losses[0] > 1e-5, "the perturbation must be visible in the first loss"
    assert losses[-1] < 0.25 * losses[0], (losses[0], losses[-1])
    assert min(losses[-5:]) <= min(losses[:5])
    assert os.path.exists(os.path.join(str(tmp_path), "synth0_20000_hip.png"))


def test_lcgs_app_fit_over_several_views(lcgs, tmp_path):
    """--fit K --cameras f: every optimiser step covers all views of the file through lcgs_fit_views (a view's forward beside
    the previous view's backward); the mean loss over the views has to fall like the single-view one does."""
    import re

    app = os.path.join(ROOT, "luisacomputegaussiansplatting_amd", "lcgs-app")
    cams = str(tmp_path / "cams.txt")
    with open(cams, "w") as f:
        for p in ([-3, -0.5, 2.3], [2.5, 1.5, 1.0], [0.2, -3.0, 0.4]):
            f.write(" ".join(str(x) for x in p + [0, 0, 0.5] + [0, 0, 1]) + "\n")
    res = subprocess.run([app, "--synth", "0:20000:1001", "--res=320x240", "--out", str(tmp_path), "--cameras", cams, "--fit",
                          "30"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    assert "30 steps of 3 view(s)" in res.stdout
    losses = [float(x) for x in re.findall(r"step \d+ loss (\S+)", res.stdout)]
    assert len(losses) == 30 and all(np.isfinite(losses))
    assert losses[0] > 1e-5 and losses[-1] < 0.35 * losses[0], (losses[0], losses[-1])


def test_l2_loss_backward_matches_torch(lcgs):
    import torch

    r = lcgs.Renderer(lcgs.Context(0))
    H, W = 37, 53
    g = torch.Generator(device="cuda:0").manual_seed(1)
    img = torch.rand(3, H, W, device="cuda:0", generator=g)
    tgt = torch.rand(3, H, W, device="cuda:0", generator=g)
    dL, loss = torch.empty_like(img), torch.full((1,), 9.0, device="cuda:0")
    r.l2_loss_backward(img, tgt, dL, loss)
    r.ctx.synchronize()
    torch.cuda.synchronize()
    ref = ((img.double() - tgt.double()) ** 2).mean()
    assert abs(float(loss) - float(ref)) <= 1e-6 * float(ref)
    assert torch.allclose(dL, (2.0 * (img - tgt) / img.numel()), rtol=1e-6, atol=1e-12)
