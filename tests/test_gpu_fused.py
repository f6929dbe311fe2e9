"""`-m gpu`: the fused one-submission frame (lcgs_render_forward) against the oracle."""
import numpy as np
import pytest
import torch

from conftest import make_scene
from gpu_util import DEV, assert_image_parity, assert_parity_vs_libm_expf, assert_parity_vs_numerics_variants, dev, upload_scene

pytestmark = pytest.mark.gpu

POSE = ([-3, -0.5, 2.3], [0, 0, 0.5], [0, 0, 1])


def _render_both(lcgs, oracle, scene, W, H, bg=(0.1, 0.2, 0.3), pose=POSE, scale_modifier=1.0, check_lists=True,
                 vs_libm=False):
    P = scene["pos"].shape[0]
    cam = lcgs.get_lookat_cam(*pose, width=W, height=H)
    ocam = oracle.lookat(*pose, width=W, height=H)
    r = lcgs.Renderer(lcgs.Context(0))
    d = upload_scene(scene)
    r.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"])
    img = torch.full((3, H, W), -1.0, device=DEV)
    radii = torch.full((P,), -7, dtype=torch.int32, device=DEV)
    n = r.forward(cam, img, bg=bg, scale_modifier=scale_modifier, radii=radii, keep_state=False, sync=True)
    orc = oracle.render(scene, ocam, bg=bg, scale_modifier=scale_modifier, ambig_eps=1e-5)
    assert n == orc["num_rendered"]
    if P:
        assert np.array_equal(radii.cpu().numpy(), orc["radii"])
    if n == 0:
        assert (img == -1.0).all()  # image untouched, gs_tile_splatter/impl.cpp:109
        return r, orc, None
    st = r.frame_stats()
    assert st["num_rendered"] == n and 0 < st["num_pairs"] <= n
    if check_lists:
        # Frames that keep no backward state list their pairs per block of 2 x 2 tiles (CamParams::list_shift); the per-tile lists
        # inspected below are those of a frame that keeps it -- the same view again, which must also give the same image bit for bit.
        # (round 6: a keep-state frame may use per-block lists too -- its renderer then writes the backward per-tile lists of
        # its own --, so the inspection frame ASKS for the reference's granularity)
        img_k = torch.full((3, H, W), -1.0, device=DEV)
        r.set_list_policy("tile")
        assert r.forward(cam, img_k, bg=bg, scale_modifier=scale_modifier, keep_state=True, sync=True) == n
        assert r.frame_stats()["list_shift"] == 0
        assert torch.equal(img_k, img), "per-tile lists and per-block lists render different images"
        assert r.frame_stats()["num_pairs"] >= st["num_pairs"]
        st = r.frame_stats()
        # Per tile, the fused list is the reference list (stable sort on (tile<<32|depth), index-order ties) with
        # only never-contributing entries pruned: same relative order, and every pruned (tile, splat) pair has
        # alpha < 1/255 (or power > 0) at every pixel of the tile.
        m, dd, c = oracle.project(scene["pos"], scene["scale"], scene["rotq"], ocam, scale_modifier=scale_modifier)
        mp, conic, tiles, rad = oracle.allocate_tiles(W, H, dd, m, c)
        offs = oracle.inclusive_sum(tiles)
        k, v = oracle.copy_with_keys(W, H, mp, offs, rad, dd)
        ks, vs = oracle.sort_pairs(k, v)
        gx, gy = (W + 15) // 16, (H + 15) // 16
        G = gx * gy
        ranges = oracle.get_ranges(ks, G)
        Lp = st["num_pairs"]
        d_list = torch.zeros(Lp, dtype=torch.int32, device=DEV)
        d_ranges = torch.zeros(2 * G, dtype=torch.int32, device=DEV)
        r.last_lists(d_list, d_ranges)
        fl = d_list.cpu().numpy().view(np.uint32)
        fr = d_ranges.cpu().numpy().view(np.uint32).reshape(G, 2)
        assert int((fr[:, 1] - fr[:, 0]).sum()) == Lp
        removed = []
        for t in range(G):
            o_l = vs[ranges[t, 0]:ranges[t, 1]]
            f_l = fl[fr[t, 0]:fr[t, 1]]
            if f_l.size == 0:
                if o_l.size:
                    removed.append((np.full(o_l.size, t), o_l))
                continue
            sorter = np.argsort(o_l, kind="stable")
            pos = sorter[np.clip(np.searchsorted(o_l, f_l, sorter=sorter), 0, o_l.size - 1)]
            assert np.array_equal(o_l[pos], f_l), f"tile {t}: fused list has entries the reference list lacks"
            assert (np.diff(pos.astype(np.int64)) > 0).all(), f"tile {t}: order differs from the reference"
            keep = np.zeros(o_l.size, bool)
            keep[pos] = True
            if (~keep).any():
                removed.append((np.full(int((~keep).sum()), t), o_l[~keep]))
        if removed:
            rt = np.concatenate([x[0] for x in removed])
            rs = np.concatenate([x[1] for x in removed])
            if rt.size > 20000:
                sel = np.random.default_rng(0).choice(rt.size, 20000, replace=False)
                rt, rs = rt[sel], rs[sel]
            ys, xs = np.mgrid[0:16, 0:16]
            pxs = (rt % gx)[:, None, None] * 16 + xs[None]
            pys = (rt // gx)[:, None, None] * 16 + ys[None]
            dx = mp[rs, 0][:, None, None] - pxs.astype(np.float32)
            dy = mp[rs, 1][:, None, None] - pys.astype(np.float32)
            cx, cy, cz = (conic[rs, i][:, None, None] for i in range(3))
            power = np.float32(-0.5) * (cx * dx * dx + cz * dy * dy) - cy * dx * dy
            alpha = np.minimum(np.float32(0.99), scene["opacity"][rs][:, None, None] * np.exp(power))
            contributes = (power <= 0) & (alpha >= np.float32(1.0 / 255.0))
            assert not contributes.any(), "a pruned (tile, splat) pair would have contributed"
    stats = assert_image_parity(img.cpu().numpy(), orc)
    if vs_libm:  # ... and against the oracle with a standard exp (libm's expf) in the blend
        assert_parity_vs_libm_expf(img.cpu().numpy(), oracle, scene, ocam, bg=bg, scale_modifier=scale_modifier)
    return r, orc, stats


@pytest.mark.parametrize("P,res", [(1, (64, 64)), (255, (64, 64)), (256, (100, 72)), (257, (333, 201)),
                                   (30011, (800, 800)), (120000, (1920, 1080))])
def test_fused_forward_matches_oracle(lcgs, oracle, P, res):
    rng = np.random.default_rng(P)
    scene = make_scene(rng, P, log_scale=(-4.2, 0.8))
    if P > 1000:
        scene["pos"][:100] = rng.normal(0, 0.3, (100, 3)) + POSE[0]
        scene["scale"][100:110] *= 60.0
    _render_both(lcgs, oracle, scene, res[0], res[1])


def test_fused_equal_depth_ties_keep_index_order(lcgs, oracle):
    """Duplicate splats (identical depth bits) must stay in splat-index order inside every tile."""
    rng = np.random.default_rng(3)
    scene = make_scene(rng, 2000, log_scale=(-3.5, 0.5))
    for k in ("pos", "scale", "rotq"):
        scene[k][1000:] = scene[k][:1000]  # geometry duplicated, colours/opacities differ
    _render_both(lcgs, oracle, scene, 160, 120)


def test_fused_all_culled_and_empty(lcgs, oracle):
    rng = np.random.default_rng(9)
    scene = make_scene(rng, 500)
    scene["pos"][:] = np.array(POSE[0]) - 3.0 * (np.array(POSE[1]) - np.array(POSE[0]))
    _render_both(lcgs, oracle, scene, 64, 48)
    empty = {k: v[:0] for k, v in scene.items()}
    r = lcgs.Renderer(lcgs.Context(0))
    d = upload_scene(empty)
    r.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"])
    img = torch.full((3, 48, 64), -1.0, device=DEV)
    assert r.forward(lcgs.get_lookat_cam(*POSE, width=64, height=48), img) == 0
    assert (img == -1.0).all()


def test_fused_scale_modifier_and_colmap_pose(lcgs, oracle):
    rng = np.random.default_rng(12)
    scene = make_scene(rng, 20000, log_scale=(-4.4, 0.8))
    pose = ([-3, -0.5, 3.3], [0, 3, 0.5], [0, -1, -1])  # the garden pose of app/main.cpp:191-193
    _render_both(lcgs, oracle, scene, 640, 360, bg=(0, 0, 0), pose=pose, scale_modifier=0.7)


def test_fused_pair_buffer_growth(lcgs, oracle):
    """Pairs beyond the workspace capacity: the frame is redone with larger buffers (the reference has a
    fixed 20M-pair buffer and no check, app/main.cpp:245)."""
    rng = np.random.default_rng(13)
    scene = make_scene(rng, 6000, log_scale=(-1.0, 0.2))  # big splats: L >> max(4 * P, 2^22)
    _, orc, _ = _render_both(lcgs, oracle, scene, 1920, 1080)
    assert orc["num_rendered"] > (1 << 22)


def test_fused_is_deterministic_and_reentrant(lcgs, oracle):
    rng = np.random.default_rng(14)
    scene = make_scene(rng, 50000, log_scale=(-4.2, 0.8))
    r = lcgs.Renderer(lcgs.Context(0))
    d = upload_scene(scene)
    r.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"])
    imgs = []
    for k in range(4):
        ang = 0.3 * (k % 2)
        cam = lcgs.get_lookat_cam([-3 * np.cos(ang), -0.5 + 3 * np.sin(ang), 2.3], [0, 0, 0.5], [0, 0, 1], width=640,
                                  height=480)
        img = torch.zeros(3, 480, 640, device=DEV)
        r.forward(cam, img, sync=(k < 2))
        imgs.append(img)
    r.ctx.synchronize()
    assert torch.equal(imgs[0], imgs[2]) and torch.equal(imgs[1], imgs[3])
    assert not torch.equal(imgs[0], imgs[1])


def test_synth_stand_in_scenes(lcgs, oracle):
    """BASELINE config 2 (nerf_blender_chair, 800x800, forward, pixel diff vs the oracle): the real scene where
    LCGS_CHAIR_PLY points at it, else the chair-like stand-in (300k splats) -- whole-image parity, bit for bit."""
    from conftest import baseline_scene

    scene, data = baseline_scene(lcgs, "chair")
    r, orc, stats = _render_both(lcgs, oracle, scene, 800, 800, bg=(0, 0, 0), check_lists=True, vs_libm=True)
    assert orc["num_rendered"] > (1_000_000 if data == "synthetic" else 100_000)
    # the same frame against the reference's LIKELY numerics (oracle/numerics.py): every pixel inside its bound
    img = torch.zeros(3, 800, 800, device=DEV)
    r.forward(lcgs.get_lookat_cam(*POSE, width=800, height=800), img, sync=True)
    assert_parity_vs_numerics_variants(img.cpu().numpy(), scene, oracle.lookat(*POSE, width=800, height=800))


@pytest.mark.parametrize("P,spread", [(5000, 0.05), (30000, 0.04), (70000, 0.03)])
def test_fused_long_tile_lists(lcgs, oracle, P, spread):
    """Per-tile lists far beyond the staging depth of the renderer and of any per-tile scratch (4096+ entries per tile),
    up to tens of thousands of entries on one tile."""
    rng = np.random.default_rng(P)
    scene = make_scene(rng, P, spread=spread, log_scale=(-3.6, 0.5))
    scene["opacity"] *= 0.05  # keep transmittance alive deep into the lists
    _, orc, _ = _render_both(lcgs, oracle, scene, 96, 64)
    assert orc["num_rendered"] > 3 * P


def test_fused_tile_list_of_one_depth(lcgs, oracle):
    """Every splat at the same position: all depth keys equal, index order must survive untouched."""
    rng = np.random.default_rng(77)
    scene = make_scene(rng, 700, log_scale=(-3.0, 0.5))
    scene["pos"][:] = scene["pos"][0]
    scene["opacity"] *= 0.02
    _render_both(lcgs, oracle, scene, 128, 96)


def test_forward_batch_equals_single_frames(lcgs, oracle):
    """lcgs_render_forward_batch (two frames in flight on sibling workspaces) == the same views rendered one by one."""
    rng = np.random.default_rng(31)
    scene = make_scene(rng, 60000, log_scale=(-4.2, 0.8))
    r = lcgs.Renderer(lcgs.Context(0))
    d = upload_scene(scene)
    r.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"])
    cams, sizes = [], [(640, 480), (320, 240), (640, 480), (801, 333), (640, 480)]
    for k, (w, h) in enumerate(sizes):
        ang = 0.5 * k
        cams.append(lcgs.get_lookat_cam([-3 * np.cos(ang), -0.5 + 3 * np.sin(ang), 2.3], [0, 0, 0.5], [0, 0, 1], width=w, height=h))
    singles = []
    for cam, (w, h) in zip(cams, sizes):
        img = torch.zeros(3, h, w, device=DEV)
        r.forward(cam, img, bg=(0.2, 0.1, 0.0), sync=True)
        singles.append(img)
    for rep in range(2):  # second round: the sibling context is warm
        imgs = [torch.full((3, h, w), -1.0, device=DEV) for (w, h) in sizes]
        r.forward_batch(cams, imgs, bg=(0.2, 0.1, 0.0))
        r.ctx.synchronize()
        for a, b in zip(imgs, singles):
            assert torch.equal(a, b)
    # a batch of one, and an empty batch
    one = [torch.zeros(3, 480, 640, device=DEV)]
    r.forward_batch(cams[:1], one, bg=(0.2, 0.1, 0.0))
    r.forward_batch([], [])
    r.ctx.synchronize()
    assert torch.equal(one[0], singles[0])


def test_half_precision_sh_is_opt_in_and_close(lcgs, oracle):
    """lcgs_scene_use_half_sh: f16 coefficient rows for the colour pass -- close to, but by
    construction not within 1e-4 of, the f32 frame; switching it off restores the f32 frame bit for bit."""
    rng = np.random.default_rng(41)
    scene = make_scene(rng, 50000, log_scale=(-4.2, 0.8))
    r = lcgs.Renderer(lcgs.Context(0))
    d = upload_scene(scene)
    r.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"])
    cam = lcgs.get_lookat_cam(*POSE, width=640, height=480)
    ref = torch.zeros(3, 480, 640, device=DEV)
    n_ref = r.forward(cam, ref)
    r.use_half_sh(True)
    half = torch.zeros(3, 480, 640, device=DEV)
    n_half = r.forward(cam, half)
    assert n_half == n_ref  # geometry is untouched
    err = (half - ref).abs()
    assert 0.0 < err.max().item() < 5e-3 and err.mean().item() < 2e-4
    # against the oracle: the f16 path is the f32 algorithm on coefficients rounded to f16 (round-to-nearest-even, what
    # numpy's astype does) and widened again -- so the oracle fed those coefficients must agree within the 1e-4 bar
    scene_h = dict(scene)
    scene_h["sh"] = scene["sh"].astype(np.float16).astype(np.float32)
    orc = oracle.render(scene_h, oracle.lookat(*POSE, width=640, height=480), ambig_eps=1e-5)
    assert n_half == orc["num_rendered"]
    max_clear, _ = assert_image_parity(half.cpu().numpy(), orc)
    assert max_clear <= 1e-4
    # camera batches read the same f16 copy
    imgs = [torch.zeros(3, 480, 640, device=DEV) for _ in range(3)]
    r.forward_batch([cam] * 3, imgs)
    r.ctx.synchronize()
    assert all(torch.equal(i, half) for i in imgs)
    r.use_half_sh(False)
    again = torch.zeros(3, 480, 640, device=DEV)
    r.forward(cam, again)
    assert torch.equal(again, ref)
    # a re-bind drops the copy
    r.use_half_sh(True)
    r.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"])
    r.forward(cam, again)
    assert torch.equal(again, ref)


def test_resolution_and_view_changes_on_one_context(lcgs, oracle):
    """One context through a sequence of frames that change resolution (tile count), view and keep_state, with a
    backward in the middle: every frame must equal the same frame from a fresh context (workspace alternation,
    previous-frame tile schedule, zeroed copies and counters all carry state from frame to frame)."""
    rng = np.random.default_rng(77)
    scene = make_scene(rng, 30000, log_scale=(-4.0, 0.8))
    d = upload_scene(scene)
    seq = [((640, 480), 0.0, False), ((320, 200), 0.4, False), ((640, 480), 0.4, True), ((1000, 700), 0.9, False),
           ((320, 200), 0.0, True), ((320, 200), 0.0, False), ((1000, 700), 0.2, False)]

    def cam_for(res, ang):
        return lcgs.get_lookat_cam([-3 * np.cos(ang), -0.5 + 3 * np.sin(ang), 2.3], [0, 0, 0.5], [0, 0, 1], width=res[0],
                                   height=res[1])

    r = lcgs.Renderer(lcgs.Context(0))
    r.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"])
    outs = []
    for k, (res, ang, keep) in enumerate(seq):
        img = torch.full((3, res[1], res[0]), -1.0, device=DEV)
        r.forward(cam_for(res, ang), img, keep_state=keep, sync=(k % 2 == 0))
        if keep:
            g = [torch.zeros_like(d[key]) for key in ("pos", "scale", "rotq", "sh", "opacity")]
            r.backward(torch.ones(3, res[1], res[0], device=DEV), *g)
        outs.append(img)
    r.ctx.synchronize()
    for (res, ang, keep), img in zip(seq, outs):
        fresh = lcgs.Renderer(lcgs.Context(0))
        fresh.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"])
        ref = torch.zeros(3, res[1], res[0], device=DEV)
        fresh.forward(cam_for(res, ang), ref, sync=True)
        assert torch.equal(img, ref), (res, ang, keep)


def test_stream_switch_with_frames_in_flight(lcgs, oracle):
    """lcgs_set_stream while asynchronous frames are queued: the frames before and after the switch (NULL stream,
    two user streams, back) all equal the frame of a fresh context."""
    rng = np.random.default_rng(78)
    scene = make_scene(rng, 40000, log_scale=(-4.0, 0.8))
    d = upload_scene(scene)
    cam = lcgs.get_lookat_cam([-3.0, -0.5, 2.3], [0, 0, 0.5], [0, 0, 1], width=800, height=600)
    s1, s2 = torch.cuda.Stream(device=DEV), torch.cuda.Stream(device=DEV)
    torch.cuda.synchronize()
    r = lcgs.Renderer(lcgs.Context(0))
    r.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"])
    order = (None, s1, s1, s2, None, s2)
    outs = [torch.full((3, 600, 800), -1.0, device=DEV) for _ in order]
    torch.cuda.synchronize()  # the fills ran on torch's stream, not on the ones used below
    for st, img in zip(order, outs):
        r.ctx.set_stream(0 if st is None else st.cuda_stream)
        r.forward(cam, img, sync=False)
    r.ctx.synchronize()
    torch.cuda.synchronize()
    fresh = lcgs.Renderer(lcgs.Context(0))
    fresh.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"])
    ref = torch.zeros(3, 600, 800, device=DEV)
    fresh.forward(cam, ref, sync=True)
    for k, img in enumerate(outs):
        assert torch.equal(img, ref), k


def test_asynchronous_overflow_is_reported_at_the_next_sync(lcgs, oracle, monkeypatch):
    """A frame enqueued without synchronisation that needs more pairs than the workspace holds: its lists are
    truncated, the next synchronising call says so (LCGS_ERR_CAPACITY) and grows the workspace; rendering the frame
    again gives the full image."""
    rng = np.random.default_rng(13)
    scene = make_scene(rng, 6000, log_scale=(-1.0, 0.2))  # big splats: L >> max(4 * P, 2^22)
    # (per-tile lists: with the per-block lists of frames that keep no backward state these 6000 splats fit the initial
    #  workspace; the two tests below overflow it in that mode too, with four times the splats and no CPU oracle to wait for)
    monkeypatch.setenv("LCGS_COARSE_LISTS", "0")
    r = lcgs.Renderer(lcgs.Context(0))
    d = upload_scene(scene)
    r.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"])
    cam = lcgs.get_lookat_cam(*POSE, width=1920, height=1080)
    img = torch.zeros(3, 1080, 1920, device=DEV)
    r.forward(cam, img, sync=False)
    with pytest.raises(lcgs.LcgsError) as e:
        r.ctx.synchronize()
    assert e.value.status == 5  # LCGS_ERR_CAPACITY
    n = r.forward(cam, img, sync=True)
    ref = oracle.render(scene, oracle.lookat(*POSE, width=1920, height=1080), ambig_eps=1e-5)
    assert n == ref["num_rendered"] and n > (1 << 22)
    assert_image_parity(img.cpu().numpy(), ref)


@pytest.mark.parametrize("second_sync", [False, True])
@pytest.mark.parametrize("coarse", [False, True], ids=["tile-lists", "block-lists"])
def test_asynchronous_overflow_is_not_forgotten_when_a_later_frame_fits(lcgs, second_sync, coarse, monkeypatch):
    """The overflow record is sticky on the device: an asynchronous frame that was truncated is still reported after a
    later frame (here a small one that fits) has rewritten the per-frame counters -- by lcgs_synchronize, and by a later
    synchronising lcgs_render_forward."""
    rng = np.random.default_rng(13)
    monkeypatch.setenv("LCGS_COARSE_LISTS", "1" if coarse else "0")
    scene = make_scene(rng, 24000 if coarse else 6000, log_scale=(-1.0, 0.2))  # big splats: L >> max(4 * P, 2^22) at 1080p
    r = lcgs.Renderer(lcgs.Context(0))
    d = upload_scene(scene)
    r.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"])
    big = torch.zeros(3, 1080, 1920, device=DEV)
    small = torch.zeros(3, 64, 64, device=DEV)
    cam_big = lcgs.get_lookat_cam(*POSE, width=1920, height=1080)
    cam_small = lcgs.get_lookat_cam(*POSE, width=64, height=64)
    r.forward(cam_big, big, sync=False)  # truncated
    with pytest.raises(lcgs.LcgsError) as e:
        if second_sync:
            r.forward(cam_small, small, sync=True)  # fits; must still report the earlier frame
        else:
            r.forward(cam_small, small, sync=False)
            r.ctx.synchronize()
    assert e.value.status == 5 and "truncated" in str(e.value)
    r.ctx.synchronize()  # reported once, then clear
    fresh = lcgs.Renderer(lcgs.Context(0))
    fresh.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"])
    ref_small, ref_big = torch.zeros_like(small), torch.zeros_like(big)
    fresh.forward(cam_small, ref_small, sync=True)
    fresh.forward(cam_big, ref_big, sync=True)
    assert torch.equal(small, ref_small)  # the frame that fitted is complete
    r.forward(cam_big, big, sync=False)  # the workspace has been grown: the big frame now fits asynchronously
    r.ctx.synchronize()
    assert torch.equal(big, ref_big)


def test_asynchronous_overflow_inside_a_camera_batch(lcgs):
    """Same through lcgs_render_forward_batch: both workspaces (the context's and its sibling's) report and grow."""
    rng = np.random.default_rng(13)
    scene = make_scene(rng, 24000, log_scale=(-1.0, 0.2))  # (per-block lists: a quarter of the pairs per big splat)
    d = upload_scene(scene)
    cam = lcgs.get_lookat_cam(*POSE, width=1920, height=1080)
    good = lcgs.Renderer(lcgs.Context(0))
    good.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"])
    ref = torch.zeros(3, 1080, 1920, device=DEV)
    good.forward(cam, ref, sync=True)
    r = lcgs.Renderer(lcgs.Context(0))
    r.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"])
    imgs = [torch.zeros(3, 1080, 1920, device=DEV) for _ in range(4)]
    errors = 0
    for attempt in range(4):
        r.forward_batch([cam] * 4, imgs)
        try:
            r.ctx.synchronize()
            break
        except lcgs.LcgsError as e:
            assert e.status == 5
            errors += 1
    assert 1 <= errors <= 3
    assert all(torch.equal(i, ref) for i in imgs)
