"""`-m gpu`: scene ingest with the de-interleave and the activations on the device against the
host restatement of read_gs_ply (app/gaussians.cpp:75-171), itself pinned against the reference's happly reader in
test_oracle_golden.py."""
import os

import numpy as np
import pytest
import torch

from gpu_util import DEV

pytestmark = pytest.mark.gpu


def _ulp_diff(a, b):
    ai = a.view(np.int32).astype(np.int64)
    bi = b.view(np.int32).astype(np.int64)
    return np.abs(ai - bi)


def _compare(dev_scene, host_scene):
    # pure copies and the IEEE divide / square root: bit-identical
    for k in ("pos", "sh", "rotq"):
        assert np.array_equal(dev_scene[k].view(np.uint32), host_scene[k].reshape(dev_scene[k].shape).view(np.uint32)), k
    # exp(): device libm vs host libm
    assert _ulp_diff(dev_scene["scale"], host_scene["scale"].reshape(-1, 3)).max() <= 2
    assert _ulp_diff(dev_scene["opacity"], host_scene["opacity"].reshape(-1)).max() <= 2


def test_device_ingest_matches_host_reader_on_golden_ply(lcgs, golden_dir):
    path = os.path.join(golden_dir, "tiny_scene.ply")
    host = lcgs.read_gs_ply(path)
    r = lcgs.Renderer(lcgs.Context(0))
    assert r.load_ply(path, order="file") == host["pos"].shape[0]
    _compare(r.download_scene(), host)
    assert r.permutation() is None


def test_load_ply_keeps_the_scene_in_spatial_order_by_default(lcgs, tmp_path):
    """lcgs_scene_load_ply's default (lcgs_set_ingest_order: LCGS_ORDER_SPATIAL): the context's arrays are the file's rows
    through lcgs_scene_permutation, the frame equals the file-order frame, per-splat outputs follow the new order; a second
    re-order composes the permutation; binding caller arrays drops it."""
    rng = np.random.default_rng(77)
    P = 40000
    path = str(tmp_path / "s.ply")
    lcgs.write_ply_raw(path, rng.normal(0, 0.9, (P, 3)) + [0, 0, 0.5], rng.normal(0.3, 0.8, (P, 3)),
                       rng.normal(0, 0.1, (P, 45)), rng.normal(0, 2.5, P), rng.normal(-4.0, 0.8, (P, 3)),
                       rng.normal(size=(P, 4)))
    cam = lcgs.get_lookat_cam([-3, -0.5, 2.3], [0, 0, 0.5], [0, 0, 1], width=320, height=240)
    rf = lcgs.Renderer(lcgs.Context(0))
    rf.load_ply(path, order="file")
    file_scene = rf.download_scene()
    img_f, rad_f = torch.zeros(3, 240, 320, device=DEV), torch.zeros(P, dtype=torch.int32, device=DEV)
    n_f = rf.forward(cam, img_f, radii=rad_f)
    rs = lcgs.Renderer(lcgs.Context(0))
    assert rs.load_ply(path) == P  # the default
    for rep in range(2):
        perm = rs.permutation()
        assert perm is not None
        pn = perm.cpu().numpy().astype(np.int64)
        assert np.array_equal(np.sort(pn), np.arange(P))
        got = rs.download_scene()
        for k in ("pos", "scale", "rotq", "sh", "opacity"):
            assert np.array_equal(got[k], file_scene[k][pn]), (rep, k)
        t = rs.scene_tensors()  # aliases of the context's arrays
        assert np.array_equal(t["pos"].cpu().numpy(), got["pos"]) and t["sh"].shape == (P, 48)
        img_s, rad_s = torch.zeros(3, 240, 320, device=DEV), torch.zeros(P, dtype=torch.int32, device=DEV)
        assert rs.forward(cam, img_s, radii=rad_s) == n_f
        assert torch.equal(img_s, img_f)
        assert np.array_equal(rad_s.cpu().numpy(), rad_f.cpu().numpy()[pn])
        if rep == 0:
            rs.reorder_scene_spatial()  # once more: already sorted, the kept permutation is the composition
    d = {k: torch.from_numpy(file_scene[k]).to(DEV) for k in file_scene}
    rs.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"])
    assert rs.permutation() is None


@pytest.mark.parametrize("P", [0, 1, 63, 64, 65, 100003])
def test_device_ingest_sizes_and_render(lcgs, tmp_path, P):
    rng = np.random.default_rng(P + 5)
    path = str(tmp_path / "s.ply")
    rot = rng.normal(size=(P, 4)).astype(np.float32)
    lcgs.write_ply_raw(path, rng.normal(0, 0.6, (P, 3)) + [0, 0, 0.5], rng.normal(0.3, 0.8, (P, 3)),
                       rng.normal(0, 0.1, (P, 45)), rng.normal(0, 2.5, P), rng.normal(-4.0, 0.8, (P, 3)), rot)
    host = lcgs.read_gs_ply(path)
    r = lcgs.Renderer(lcgs.Context(0))
    assert r.load_ply(path, order="file") == P
    if P == 0:
        assert lcgs.Renderer(lcgs.Context(0)).load_ply(path) == 0  # (and the default order on an empty file)
        return
    _compare(r.download_scene(), host)
    # the loaded scene renders; against the host-loaded scene the image differs only through the <= 2 ulp of exp()
    cam = lcgs.get_lookat_cam([-3, -0.5, 2.3], [0, 0, 0.5], [0, 0, 1], width=160, height=120)
    img = torch.zeros(3, 120, 160, device=DEV)
    n = r.forward(cam, img)
    r2 = lcgs.Renderer(lcgs.Context(0))
    r2.upload_scene(host)
    img2 = torch.zeros(3, 120, 160, device=DEV)
    n2 = r2.forward(cam, img2)
    assert abs(n - n2) <= max(2, n2 // 1000)
    assert (img - img2).abs().max().item() < 5e-3 and (img - img2).abs().mean().item() < 1e-5


def test_upload_keeps_the_scene_in_spatial_order_by_default_too(lcgs, oracle):
    """lcgs_scene_upload (host arrays -> context-owned copies) follows the same ingest order as lcgs_scene_load_ply; the
    oracle frame of the ORIGINAL arrays is the yardstick for both orders."""
    from conftest import make_scene
    from gpu_util import assert_image_parity

    rng = np.random.default_rng(4)
    scene = make_scene(rng, 50000, spread=1.5, log_scale=(-4.0, 0.8))
    pose = ([-3, -0.5, 2.3], [0, 0, 0.5], [0, 0, 1])
    cam = lcgs.get_lookat_cam(*pose, width=400, height=300)
    orc = oracle.render(scene, oracle.lookat(*pose, width=400, height=300), ambig_eps=1e-5)
    imgs = {}
    for order in (None, "file"):
        r = lcgs.Renderer(lcgs.Context(0))
        r.upload_scene(scene, order=order)
        img, radii = torch.zeros(3, 300, 400, device=DEV), torch.zeros(50000, dtype=torch.int32, device=DEV)
        assert r.forward(cam, img, radii=radii) == orc["num_rendered"]
        perm = r.permutation()
        assert (perm is None) == (order == "file")
        rad = radii.cpu().numpy()
        if perm is not None:
            back = np.empty_like(rad)
            back[perm.cpu().numpy().astype(np.int64)] = rad
            rad = back
        assert np.array_equal(rad, orc["radii"])
        assert_image_parity(img.cpu().numpy(), orc)
        imgs[order] = img
    assert torch.equal(imgs[None], imgs["file"])


def test_device_ingest_falls_back_for_ascii(lcgs, tmp_path):
    path = str(tmp_path / "a.ply")
    names = (["x", "y", "z", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(45)] + ["opacity"] +
             [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)])
    rng = np.random.default_rng(2)
    rows = rng.normal(0, 0.5, (7, len(names))).astype(np.float32)
    with open(path, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex 7\n" + "".join(f"property float {n}\n" for n in names) + "end_header\n")
        for row in rows:
            f.write(" ".join(repr(float(v)) for v in row) + "\n")
    host = lcgs.read_gs_ply(path)
    r = lcgs.Renderer(lcgs.Context(0))
    assert r.load_ply(path, order="file") == 7
    dev = r.download_scene()
    for k in ("pos", "sh", "rotq", "scale", "opacity"):
        assert np.array_equal(dev[k], host[k].reshape(dev[k].shape))  # same host code path


def test_device_ingest_errors(lcgs, tmp_path):
    r = lcgs.Renderer(lcgs.Context(0))
    with pytest.raises(lcgs.LcgsError):
        r.load_ply(str(tmp_path / "missing.ply"))
    bad = str(tmp_path / "bad.ply")
    open(bad, "w").write("ply\nformat binary_little_endian 1.0\nelement vertex 5\nproperty float x\nend_header\n")
    with pytest.raises(lcgs.LcgsError):
        r.load_ply(bad)


def test_spatial_reorder_keeps_the_image_and_permutes_the_outputs(lcgs, oracle):
    """lcgs_scene_reorder_spatial: the context's scene becomes old[perm] (perm a permutation, Morton-sorted), the frame
    is bit-identical, per-splat outputs follow the new order, and the oracle agrees on the permuted scene."""
    import torch
    from conftest import make_scene
    from gpu_util import DEV, assert_image_parity, upload_scene

    rng = np.random.default_rng(21)
    P = 50000
    scene = make_scene(rng, P, spread=3.0, log_scale=(-4.2, 0.7))  # the camera sits inside the cloud: most splats are off screen
    scene["pos"][100:130] = np.array([1e4, -2e5, 3e3], np.float32)  # far outliers must not flatten the grid
    scene["pos"][200:220] = scene["pos"][200]                        # coincident splats keep their file order
    pose = ([-3, -0.5, 2.3], [0, 0, 0.5], [0, 0, 1])
    W, H = 640, 360
    cam = lcgs.get_lookat_cam(*pose, width=W, height=H)
    r = lcgs.Renderer(lcgs.Context(0))
    d = upload_scene(scene)
    r.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"])
    img0 = torch.zeros(3, H, W, device=DEV)
    rad0 = torch.zeros(P, dtype=torch.int32, device=DEV)
    n0 = r.forward(cam, img0, radii=rad0, keep_state=True, sync=True)
    g0 = {k: torch.zeros_like(d[k]) for k in d}
    dL = torch.randn(3, H, W, device=DEV)
    r.backward(dL, g0["pos"], g0["scale"], g0["rotq"], g0["sh"], g0["opacity"])

    perm = r.reorder_scene_spatial().long()
    assert torch.equal(torch.sort(perm).values, torch.arange(P, device=DEV))
    back = r.download_scene()
    for k in ("pos", "scale", "rotq", "sh", "opacity"):
        assert np.array_equal(back[k].reshape(P, -1), scene[k].reshape(P, -1)[perm.cpu().numpy()]), k
    p20 = perm[(perm >= 200) & (perm < 220)]
    assert torch.equal(p20, torch.arange(200, 220, device=DEV))  # stable: equal keys stay in file order
    # for the caller's untouched arrays nothing changed
    assert np.array_equal(d["pos"].cpu().numpy(), scene["pos"])

    img1 = torch.zeros(3, H, W, device=DEV)
    rad1 = torch.zeros(P, dtype=torch.int32, device=DEV)
    n1 = r.forward(cam, img1, radii=rad1, keep_state=True, sync=True)
    assert n1 == n0 and torch.equal(rad1, rad0[perm])
    # same per-pixel blend sequence, hence bit-identical -- except where two splats of exactly equal depth overlap: they
    # are blended in splat order (as in the reference), which is what changed.  None in this scene.
    assert torch.equal(img1, img0)
    # other views, without the radii request (the cull pass's two-phase form): same frame, same counts as file order
    st1 = r.frame_stats()
    for pose2, res in ((pose, (W, H)), (([0.5, 0.2, 0.6], [3, 2, 0.4], [0, 0, 1]), (333, 201))):
        cam2 = lcgs.get_lookat_cam(*pose2, width=res[0], height=res[1])
        a = torch.zeros(3, res[1], res[0], device=DEV)
        b = torch.full((3, res[1], res[0]), -1.0, device=DEV)
        fresh = lcgs.Renderer(lcgs.Context(0))
        fresh.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"])  # file order, no bounds
        na = fresh.forward(cam2, a, keep_state=True, sync=True)  # (both per tile: num_pairs depends on the list granularity)
        nb = r.forward(cam2, b, keep_state=True, sync=True)
        sa, sb_ = fresh.frame_stats(), r.frame_stats()
        assert na == nb and float((a - b).abs().max()) <= 1e-6
        assert all(sa[k] == sb_[k] for k in ("num_visible", "num_rendered", "num_pairs")), (sa, sb_)
    assert r.frame_stats()["num_visible"] < st1["num_visible"]  # (the second pose sits inside the cloud)
    r.forward(cam, img1, keep_state=True, sync=True)
    g1 = {k: torch.zeros_like(d[k]) for k in d}
    r.backward(dL, g1["pos"], g1["scale"], g1["rotq"], g1["sh"], g1["opacity"])
    for k in g0:
        a, b = g1[k].reshape(P, -1).double(), g0[k].reshape(P, -1)[perm].double()
        assert (a - b).norm() <= 1e-4 * b.norm(), k
    # locality: the on-screen splats now come in runs -- far fewer visible/invisible switches along the index
    sw0 = int((rad0[1:] > 0).ne(rad0[:-1] > 0).sum())
    sw1 = int((rad1[1:] > 0).ne(rad1[:-1] > 0).sum())
    assert sw1 * 4 < sw0, (sw0, sw1)
    # and the oracle, given the permuted scene, agrees with the frame
    ps = {k: np.ascontiguousarray(scene[k][perm.cpu().numpy()]) for k in scene}
    orc = oracle.render(ps, oracle.lookat(*pose, width=W, height=H), ambig_eps=1e-5)
    assert orc["num_rendered"] == n1 and np.array_equal(orc["radii"], rad1.cpu().numpy())
    assert_image_parity(img1.cpu().numpy(), orc)


def test_spatial_reorder_with_non_finite_and_degenerate_inputs(lcgs):
    """Non-finite positions (kept out of the box, sorted into cell 0), a scene collapsed to one point (zero-sized box)
    and P = 1: the permutation stays a bijection and the frame stays the file-order frame (to the last bit or two: see
    below)."""
    import torch
    from conftest import make_scene
    from gpu_util import DEV, upload_scene

    pose = ([-3, -0.5, 2.3], [0, 0, 0.5], [0, 0, 1])
    W, H = 200, 120
    cam = lcgs.get_lookat_cam(*pose, width=W, height=H)
    rng = np.random.default_rng(31)
    cases = []
    s = make_scene(rng, 5000)
    s["pos"][10] = np.nan
    s["pos"][500, 1] = np.inf
    s["pos"][900, 2] = -np.inf
    s["pos"][1200] = 3e38
    cases.append(s)
    s = make_scene(rng, 3000)
    s["pos"][:] = np.array([0.1, 0.2, 0.5], np.float32)  # every splat at one point: sigma = 0
    cases.append(s)
    cases.append(make_scene(rng, 1))
    for scene in cases:
        P = scene["pos"].shape[0]
        d = upload_scene(scene)
        r = lcgs.Renderer(lcgs.Context(0))
        r.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"])
        a = torch.full((3, H, W), 0.25, device=DEV)
        na = r.forward(cam, a, sync=True)
        perm = r.reorder_scene_spatial().long()
        assert torch.equal(torch.sort(perm).values, torch.arange(P, device=DEV))
        b = torch.full((3, H, W), 0.25, device=DEV)
        nb = r.forward(cam, b, sync=True)
        # (two splats of exactly equal depth keep their splat order, as in the reference -- and the splat order is what
        #  changed: where such a pair overlaps, the blend order, hence the last bit, may differ)
        assert na == nb and float((a - b).abs().max()) <= 1e-6, P
        back = r.download_scene()
        pn = perm.cpu().numpy()
        assert np.array_equal(back["pos"], scene["pos"][pn], equal_nan=True)
        assert np.array_equal(back["sh"].reshape(P, -1), scene["sh"].reshape(P, -1)[pn])


def _plane_scene(rng, P):
    """P overlapping splats on the plane x = 0, seen head-on by an axis-aligned camera: every view depth is exactly 5.0"""
    from conftest import make_scene

    scene = make_scene(rng, P, spread=0.25, log_scale=(-2.6, 0.3))
    scene["pos"][:, 0] = 0.0
    scene["pos"][:, 2] += 0.0
    scene["opacity"][:] = rng.uniform(0.3, 0.8, P).astype(np.float32)
    return scene


PLANE_POSE = ([-5.0, 0.0, 0.5], [0.0, 0.0, 0.5], [0.0, 0.0, 1.0])


def test_equal_depths_blend_in_file_order_after_the_spatial_reorder(lcgs, oracle):
    """The reference's stable sort on (tile, depth bits) blends splats of exactly equal depth in ascending FILE index
    (gs_tile_splatter/impl.cpp:135-143).  A context-owned scene is re-ordered along a Morton curve by default; the pass
    behind the depth sort must put every run of equal depths back into file order -- here ALL depths are equal and the
    splats overlap heavily, so any other order gives a visibly different image."""
    from gpu_util import assert_image_parity

    rng = np.random.default_rng(91)
    scene = _plane_scene(rng, 900)
    W, H = 320, 240
    cam = lcgs.get_lookat_cam(*PLANE_POSE, width=W, height=H)
    orc = oracle.render(scene, oracle.lookat(*PLANE_POSE, width=W, height=H), ambig_eps=1e-5)
    assert orc["num_rendered"] > 3000
    imgs = {}
    for order in ("file", None):
        r = lcgs.Renderer(lcgs.Context(0))
        r.upload_scene(scene, order=order)
        img = torch.zeros(3, H, W, device=DEV)
        assert r.forward(cam, img) == orc["num_rendered"]
        assert r.frame_stats()["equal_depth_unresolved"] == 0
        assert_image_parity(img.cpu().numpy(), orc)
        imgs[order] = img
        if order is None:  # and through the camera-batch sibling, which borrows the permutation
            pair = [torch.zeros(3, H, W, device=DEV) for _ in range(2)]
            r.forward_batch([cam, cam], pair)
            r.ctx.synchronize()
            assert torch.equal(pair[0], img) and torch.equal(pair[1], img)
            # binding the context's own arrays again keeps the permutation (and with it the order of equal depths)
            t = r.scene_tensors()
            r.bind_scene(t["pos"], t["scale"], t["rotq"], t["sh"], t["opacity"])
            assert r.permutation() is not None
            again = torch.zeros(3, H, W, device=DEV)
            r.forward(cam, again)
            assert torch.equal(again, img)
    assert torch.equal(imgs["file"], imgs[None])
    # the order matters here: the same splats in reversed file order are a different image
    rev = {k: np.ascontiguousarray(v[::-1]) for k, v in scene.items()}
    r = lcgs.Renderer(lcgs.Context(0))
    r.upload_scene(rev, order="file")
    other = torch.zeros(3, H, W, device=DEV)
    r.forward(cam, other)
    assert (other - imgs["file"]).abs().max().item() > 1e-2


@pytest.mark.parametrize("P", [6000, 20000])
def test_equal_depth_runs_beyond_the_lds_cap_are_exact_too(lcgs, oracle, P):
    """A coplanar sheet seen head-on: 6 000 / 20 000 splats at exactly ONE depth -- a single run far beyond what a
    workgroup ranks through LDS (4096).  Until round 3 such a run kept the context's (Morton) order and was only reported
    (lcgs_frame_stats.equal_depth_unresolved); now its workgroup radix-sorts it on the file indices through global scratch:
    the re-ordered scene's frame is the oracle's bit for bit, its per-tile lists are the file-order scene's, and the
    counter stays 0 (gs_tile_splatter/impl.cpp:135-143: a stable sort is exact for any run length)."""
    from gpu_util import assert_image_parity

    rng = np.random.default_rng(92 + P)
    scene = _plane_scene(rng, P)
    W, H = 256, 192
    cam = lcgs.get_lookat_cam(*PLANE_POSE, width=W, height=H)
    orc = oracle.render(scene, oracle.lookat(*PLANE_POSE, width=W, height=H))
    out = {}
    for order in (None, "file"):
        r = lcgs.Renderer(lcgs.Context(0))
        r.upload_scene(scene, order=order)
        img = torch.zeros(3, H, W, device=DEV)
        assert r.forward(cam, img) == orc["num_rendered"]
        st = r.frame_stats()
        assert st["num_visible"] > 4096 and st["equal_depth_unresolved"] == 0
        assert_image_parity(img.cpu().numpy(), orc)
        lst = torch.zeros(st["num_pairs"], dtype=torch.int32, device=DEV)
        rng_ = torch.zeros(st["num_tiles"] * 2, dtype=torch.int32, device=DEV)
        r.last_lists(lst, rng_)
        lst = lst.cpu().numpy().astype(np.int64)
        if r.permutation() is not None:
            lst = r.permutation().cpu().numpy().astype(np.int64)[lst]
        out[order] = (lst, rng_.cpu().numpy())
        if order is None:  # a second frame on the same context (scratch reused), and the camera-batch sibling
            again = torch.zeros(3, H, W, device=DEV)
            r.forward(cam, again)
            pair = [torch.zeros(3, H, W, device=DEV) for _ in range(2)]
            r.forward_batch([cam, cam], pair)
            r.ctx.synchronize()
            assert torch.equal(again, img) and torch.equal(pair[0], img) and torch.equal(pair[1], img)
    assert np.array_equal(out[None][0], out["file"][0]) and np.array_equal(out[None][1], out["file"][1])


def _frame_lists_in_file_indices(lcgs, scene, cam, order):
    """per-tile lists of one fused frame, every entry as the splat's FILE index"""
    r = lcgs.Renderer(lcgs.Context(0))
    r.upload_scene(scene, order=order)
    img = torch.zeros(3, cam.height, cam.width, device=DEV)
    # a first frame that sees a sliver of the scene: the next one runs with launch sizes hinted far below its survivor
    # count (the strided paths of the sort chain and of the equal-depth pass)
    away = lcgs.get_lookat_cam([-5.0, 3.2, 0.5], [0.0, 9.0, 0.5], [0.0, 0.0, 1.0], width=cam.width, height=cam.height)
    few = r.forward(away, img)
    n = r.forward(cam, img)
    assert few * 4 < n
    st = r.frame_stats()
    lst = torch.zeros(max(st["num_pairs"], 1), dtype=torch.int32, device=DEV)
    rng_ = torch.zeros(st["num_tiles"] * 2, dtype=torch.int32, device=DEV)
    r.last_lists(lst, rng_)
    lst = lst[:st["num_pairs"]].cpu().numpy().astype(np.int64)
    perm = r.permutation()
    if perm is not None:
        lst = perm.cpu().numpy().astype(np.int64)[lst]
    return n, st, lst, rng_.cpu().numpy(), img


@pytest.mark.parametrize("levels", [20000, 1500, 60, 7])
def test_equal_depth_runs_of_every_length_keep_the_file_order(lcgs, levels):
    """Depths quantised to `levels` planes in a 60 K-splat scene: runs of 2-3 equal depths (put right inside the last
    radix pass), of dozens (listed, ranked by a workgroup), of thousands (many of them crossing the sort's 2048-key
    chunks) and -- 7 levels -- beyond the LDS cap (radix-sorted through global scratch).  The re-ordered scene must yield the file-order scene's per-tile lists
    entry for entry (as file indices), and the same image bit for bit."""
    from conftest import make_scene

    rng = np.random.default_rng(400 + levels)
    P = 60000
    scene = make_scene(rng, P, spread=1.2, log_scale=(-4.0, 0.3))
    x = scene["pos"][:, 0]
    scene["pos"][:, 0] = (np.round((x - x.min()) / (x.max() - x.min()) * (levels - 1)) / (levels - 1) * 2.0 - 1.0) \
        .astype(np.float32)  # the camera looks along +x: the view depth is x + 5, equal x = equal depth bits
    W, H = 320, 240
    cam = lcgs.get_lookat_cam(*PLANE_POSE, width=W, height=H)
    n_f, st_f, list_f, rng_f, img_f = _frame_lists_in_file_indices(lcgs, scene, cam, "file")
    n_s, st_s, list_s, rng_s, img_s = _frame_lists_in_file_indices(lcgs, scene, cam, None)
    assert n_f == n_s and st_f["num_pairs"] == st_s["num_pairs"] > 20000
    assert np.array_equal(rng_f, rng_s)
    # (7 levels: ~8.5 K splats per depth, beyond the 4096 a workgroup ranks through LDS -- sorted through global scratch)
    assert st_s["equal_depth_unresolved"] == 0
    assert np.array_equal(list_f, list_s)
    assert torch.equal(img_f, img_s)
