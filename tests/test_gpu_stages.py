"""`-m gpu`: each stage-level C-ABI entry point (one per reference operator method) against the oracle.
Per-splat stages and all integer work are compared BIT FOR BIT; images within 1e-4 L-inf."""
import numpy as np
import pytest
import torch

from conftest import make_scene
from gpu_util import DEV, assert_image_parity, dev

pytestmark = pytest.mark.gpu

POSE = ([-3, -0.5, 2.3], [0, 0, 0.5], [0, 0, 1])


def _make_ops(lcgs):
    ctx = lcgs.Context(0)
    sh, pr, ts = lcgs.SHProcessor(), lcgs.GSProjector(), lcgs.GSTileSplatter()
    sh.create(ctx)
    pr.create(ctx)
    ts.create(ctx)
    return ctx, sh, pr, ts


@pytest.fixture(scope="module")
def ops(lcgs):
    return _make_ops(lcgs)


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_processor(lcgs, oracle, ops, deg):
    ctx, sh, _, _ = ops
    rng = np.random.default_rng(deg)
    P = 100003
    scene = make_scene(rng, P)
    feat = (deg + 1) ** 2 * 3
    cam = lcgs.get_lookat_cam(*POSE, width=800, height=800)
    d_color = torch.zeros(P, 3, device=DEV)
    shs = np.ascontiguousarray(scene["sh"][:, :feat])
    sh.process(lcgs.GPUPointsProxy(P, 3, dev(scene["pos"])), cam, dev(shs), d_color, 3, deg)
    ctx.synchronize()
    expect = oracle.sh_process(np.array(cam.position, np.float32), scene["pos"], shs, deg=deg)
    assert np.array_equal(d_color.cpu().numpy(), expect)


@pytest.mark.parametrize("use_focal", [True, False])
def test_projector(lcgs, oracle, ops, use_focal):
    ctx, _, pr, _ = ops
    rng = np.random.default_rng(5)
    P = 70001
    scene = make_scene(rng, P, spread=1.5, log_scale=(-4.0, 1.0))
    scene["pos"][:300] = rng.normal(0, 0.4, (300, 3)) + POSE[0]
    cam = lcgs.get_lookat_cam(*POSE, width=1920, height=1080)
    ocam = oracle.lookat(*POSE, width=1920, height=1080)
    # sentinel-filled outputs: culled splats must stay untouched (gs_projector/shader.cpp:121)
    means = torch.full((P, 2), 7.0, device=DEV)
    covs = torch.full((P, 3), 5.0, device=DEV)
    depth = torch.full((P,), 9.0, device=DEV)
    pr.forward(lcgs.GSProjectorInputProxy(P, dev(scene["pos"]), dev(scene["scale"]), dev(scene["rotq"]), 1.5),
               lcgs.GSProjectorOutputProxy(means, covs, depth), cam, use_focal)
    ctx.synchronize()
    init = (np.full((P, 2), 7, np.float32), np.full(P, 9, np.float32), np.full((P, 3), 5, np.float32))
    m, d, c = oracle.project(scene["pos"], scene["scale"], scene["rotq"], ocam, scale_modifier=1.5,
                             use_focal=use_focal, init=init)
    assert (d == 9).any() and (d != 9).any()
    assert np.array_equal(means.cpu().numpy(), m)
    assert np.array_equal(depth.cpu().numpy(), d)
    assert np.array_equal(covs.cpu().numpy(), c)


def _tile_splat(lcgs, oracle, ops, scene, W, H, bg, use_focal=True, cap_slack=1.0):
    ctx, sh, pr, ts = ops
    P = scene["pos"].shape[0]
    cam = lcgs.get_lookat_cam(*POSE, width=W, height=H)
    ocam = oracle.lookat(*POSE, width=W, height=H)
    d = {k: dev(v) for k, v in scene.items()}
    color = torch.zeros(P, 3, device=DEV)
    means = torch.zeros(P, 2, device=DEV)
    covs = torch.zeros(P, 3, device=DEV)
    depth = torch.zeros(P, device=DEV)
    sh.process(lcgs.GPUPointsProxy(P, 3, d["pos"]), cam, d["sh"], color, 3, 3)
    pr.forward(lcgs.GSProjectorInputProxy(P, d["pos"], d["scale"], d["rotq"], 1.0),
               lcgs.GSProjectorOutputProxy(means, covs, depth), cam, use_focal)
    # oracle side
    o_color = oracle.sh_process(np.array(cam.position, np.float32), scene["pos"], scene["sh"])
    m, dd, c = oracle.project(scene["pos"], scene["scale"], scene["rotq"], ocam, use_focal=use_focal)
    mp, conic, tiles, radii = oracle.allocate_tiles(W, H, dd, m, c, use_focal=use_focal)
    offs = oracle.inclusive_sum(tiles)
    keys, vals = oracle.copy_with_keys(W, H, mp, offs, radii, dd)
    ks, vs = oracle.sort_pairs(keys, vals)
    G = ((W + 15) // 16) * ((H + 15) // 16)
    ranges = oracle.get_ranges(ks, G)
    img, fT, nc, amb = oracle.render_forward(W, H, bg, ranges, vs, mp, conic, scene["opacity"], o_color, ambig_eps=1e-5)
    L = int(offs[-1]) if P else 0
    cap = max(1, int(L * cap_slack) + 7)
    i64 = lambda n: torch.zeros(n, dtype=torch.int64, device=DEV)
    i32 = lambda n: torch.zeros(n, dtype=torch.int32, device=DEV)
    # the unsorted pair buffers start as GARBAGE: the reference zero-fills them every frame (impl.cpp:117-118); the library
    # only does when a slot would otherwise stay unwritten (a NaN covariance) -- either way [0, L) must be the oracle's
    junk64 = torch.full((cap,), 0x5A5A5A5A5A5A5A5A, dtype=torch.int64, device=DEV)
    junk32 = torch.full((cap,), 0x5A5A5A5A, dtype=torch.int32, device=DEV)
    accel = lcgs.GSTileSplatterAccelProxy(i32(P), i32(P), junk64, junk32, i64(cap), i32(cap), i32(2 * G))
    target = torch.full((3, H, W), -1.0, device=DEV)
    out = lcgs.GSSplatForwardOutputProxy(H, W, target, i32(P), torch.zeros(H, W, device=DEV), i32(H * W))
    inp = lcgs.GSTileSplatterInputProxy(P, tuple(bg), means, depth, covs, color, d["opacity"])
    n = ts.forward(accel, inp, out, use_focal)
    ctx.synchronize()
    assert n == L and ts.num_rendered == L
    u32 = lambda t: t.cpu().numpy().view(np.uint32)
    assert np.array_equal(u32(accel.tiles_touched), tiles)
    assert np.array_equal(u32(accel.point_offsets), offs)
    assert np.array_equal(out.radii.cpu().numpy(), radii)
    vis = dd >= np.float32(0.2)
    assert np.array_equal(means.cpu().numpy()[vis], mp[vis])   # NDC -> pixel, in place
    assert np.array_equal(covs.cpu().numpy()[vis], conic[vis], equal_nan=True)  # cov -> conic, in place
    if L > 0:
        assert np.array_equal(accel.point_list_keys_unsorted.cpu().numpy().view(np.uint64)[:L], keys)
        assert np.array_equal(u32(accel.point_list_unsorted)[:L], vals)
        assert np.array_equal(accel.point_list_keys.cpu().numpy().view(np.uint64)[:L], ks)
        assert np.array_equal(u32(accel.point_list)[:L], vs)
        assert np.array_equal(u32(accel.ranges).reshape(G, 2), ranges)
        assert_image_parity(target.cpu().numpy(), {"img": img, "ambig": amb})
        # (bit for bit since round 3: the blend's exp is one defined sequence of binary32 operations on both sides)
        assert np.array_equal(u32(out.n_contrib).reshape(H, W), nc)
        assert np.array_equal(out.final_T.cpu().numpy().reshape(H, W), fT.reshape(H, W))
    else:
        assert (target == -1.0).all()  # image untouched (impl.cpp:109)
    return L


@pytest.mark.parametrize("sort", ["literal", "splats"])
@pytest.mark.parametrize("res", [(800, 800), (100, 72), (333, 201), (16, 16)])
def test_tile_splatter_chain(lcgs, oracle, ops, res, sort, monkeypatch):
    """Every buffer GSTileSplatter::forward leaves behind, entry for entry, through both ways the splatter produces the
    sorted pairs: the reference's six-pass sort of the unsorted pairs ("literal", what frames below 4 M pairs take) and
    sort-before-duplicate ("splats": the splats by depth, their pairs re-emitted in that order, two passes on the tile bits
    -- what large frames take); LCGS_STAGE_SORT forces either."""
    monkeypatch.setenv("LCGS_STAGE_SORT", sort)
    ops = _make_ops(lcgs)  # (the hook is read when a context is created)
    rng = np.random.default_rng(res[0])
    scene = make_scene(rng, 30011, log_scale=(-4.2, 0.8))
    scene["pos"][:100] = rng.normal(0, 0.3, (100, 3)) + POSE[0]
    scene["scale"][100:110] *= 60.0  # rects that cover the whole grid
    L = _tile_splat(lcgs, oracle, ops, scene, res[0], res[1], (0.1, 0.2, 0.3))
    assert L > 0 or res == (16, 16)


@pytest.mark.parametrize("sort", ["literal", "splats"])
def test_tile_splatter_nan_covariance_keeps_the_references_zero_filled_pairs(lcgs, oracle, ops, sort, monkeypatch):
    """A splat whose covariance is NaN gets radius 0 but claims a tile (allocate_tiles has no such test,
    shader.cpp:102-163); copy_with_keys skips it (radius <= 0, :41-42), so its pair slots keep the BufferFiller's zeros
    (impl.cpp:117-118): key 0 / value 0 = splat 0 in tile 0 at depth 0.  The stage-level path reproduces that literally
    -- the zero-fill is skipped only in frames WITHOUT such a splat -- down to the sorted lists and the image."""
    monkeypatch.setenv("LCGS_STAGE_SORT", sort)  # ("splats" is overridden by the zero-filled slots: they exist nowhere else)
    ops = _make_ops(lcgs)  # (the hook is read when a context is created)
    rng = np.random.default_rng(78)
    scene = make_scene(rng, 6000, log_scale=(-4.0, 0.7))
    scene["scale"][[17, 2500, 5999], 1] = np.nan
    L = _tile_splat(lcgs, oracle, ops, scene, 320, 240, (0.2, 0.1, 0.0))
    assert L > 0


def test_tile_splatter_nonfocal(lcgs, oracle, ops):
    rng = np.random.default_rng(77)
    scene = make_scene(rng, 5000, log_scale=(-4.0, 0.7))
    _tile_splat(lcgs, oracle, ops, scene, 320, 240, (0, 0, 0), use_focal=False)


def test_tile_splatter_all_culled_and_capacity(lcgs, oracle, ops):
    rng = np.random.default_rng(8)
    scene = make_scene(rng, 1000)
    scene["pos"][:, :] = np.array(POSE[0]) - 3.0 * (np.array(POSE[1]) - np.array(POSE[0]))  # behind the camera
    assert _tile_splat(lcgs, oracle, ops, scene, 64, 64, (0.5, 0.5, 0.5)) == 0
    scene = make_scene(rng, 4000)
    with pytest.raises(lcgs.LcgsError) as e:
        _tile_splat(lcgs, oracle, ops, scene, 256, 256, (0, 0, 0), cap_slack=0.5)
    assert e.value.status == 5  # LCGS_ERR_CAPACITY (the reference silently overruns, app/main.cpp:245)


@pytest.mark.parametrize("n", [0, 1, 63, 1024, 1025, 4097, 1_000_003])
def test_inclusive_sum(lcgs, oracle, ops, n):
    ctx = ops[0]
    rng = np.random.default_rng(n)
    x = rng.integers(0, 50, n).astype(np.uint32)
    if n > 10:
        x[3] = 2**31  # u32 wrap-around
        x[7] = 2**31
    d_in = dev(x.view(np.int32)) if n else torch.zeros(0, dtype=torch.int32, device=DEV)
    d_out = torch.zeros(n, dtype=torch.int32, device=DEV)
    ctx.inclusive_sum(d_in, d_out, n)
    ctx.synchronize()
    assert np.array_equal(d_out.cpu().numpy().view(np.uint32), oracle.inclusive_sum(x))


@pytest.mark.parametrize("n,bits", [(0, 64), (1, 64), (4095, 64), (4096, 64), (4097, 45), (300_007, 45), (2_000_003, 64),
                                    (100_000, 13), (100_000, 8), (100_000, 1)])
def test_sort_pairs(lcgs, oracle, ops, n, bits):
    ctx = ops[0]
    rng = np.random.default_rng(n + bits)
    keys = rng.integers(0, 2**63, n, dtype=np.uint64)
    if bits < 64:
        keys &= np.uint64((1 << bits) - 1)
    if n > 100:
        keys[: n // 3] = keys[n // 3: 2 * (n // 3)]  # duplicates: stability matters
    vals = np.arange(n, dtype=np.uint32)
    dk, dv = dev(keys.view(np.int64)), dev(vals.view(np.int32))
    ok, ov = torch.zeros(n, dtype=torch.int64, device=DEV), torch.zeros(n, dtype=torch.int32, device=DEV)
    ctx.sort_pairs(dk, ok, dv, ov, n, 0, bits)
    ctx.synchronize()
    ks, vs = oracle.sort_pairs(keys, vals)
    assert np.array_equal(ok.cpu().numpy().view(np.uint64), ks)
    assert np.array_equal(ov.cpu().numpy().view(np.uint32), vs)
    assert np.array_equal(dk.cpu().numpy().view(np.uint64), keys), "inputs must be preserved"


def test_blend_exp_on_the_device_is_the_oracles_bit_for_bit(lcgs, oracle):
    """The exp of the compositing loop (shader.cpp:258) is a build-defined sequence of binary32 operations.  What the
    DEVICE evaluates (v_fma_f32 / v_pk_fma_f32 / v_lshl_add_u32) must be the oracle's C version bit for bit over the
    whole domain [-86, 0]: every 61st binary32 of the blend's range [-6, 0], every 1021st beyond, and the edges."""
    near = np.arange(0x80000000, np.float32(-6.0).view(np.uint32), 61, dtype=np.uint64).astype(np.uint32)
    far = np.arange(np.float32(-6.0).view(np.uint32), np.float32(-86.0).view(np.uint32), 1021,
                    dtype=np.uint64).astype(np.uint32)
    edge = np.array([0.0, -0.0, -86.0, -5.5412636, -1e-30, -1e-45], np.float32).view(np.uint32)
    x = np.concatenate([near, far, edge]).view(np.float32)
    ctx = lcgs.Context(0)
    dx, dout = dev(x), torch.zeros(x.size, device=DEV)
    ctx.blend_exp(dx, dout, x.size)
    got = dout.cpu().numpy()
    ref = oracle.blend_exp(x)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), int((got.view(np.uint32) != ref.view(np.uint32)).sum())


def test_deferred_stage_mode_gives_the_exact_modes_image_and_never_changes_a_result(lcgs, oracle, ops):
    """lcgs_set_stage_mode(LCGS_STAGES_DEFERRED): the reference's call pattern (process, forward, forward -- back to back,
    app/main.cpp:266-308) renders the fused frame from the 3-D arrays: same image, radii and num_rendered as the exact mode
    bit for bit, the intermediate buffers are not written.  A splatter call that does NOT match what was recorded (here:
    another colour buffer), a flush, or lcgs_synchronize run the recorded operators, so results never depend on the mode."""
    ctx, sh, pr, ts = ops
    rng = np.random.default_rng(31)
    P, W, H = 40000, 640, 400
    scene = make_scene(rng, P, log_scale=(-4.0, 0.8))
    scene["pos"][:100] = rng.normal(0, 0.3, (100, 3)) + POSE[0]
    cam = lcgs.get_lookat_cam(*POSE, width=W, height=H)
    d = {k: dev(v) for k, v in scene.items()}
    G = ((W + 15) // 16) * ((H + 15) // 16)
    i64 = lambda n: torch.zeros(n, dtype=torch.int64, device=DEV)
    i32 = lambda n: torch.zeros(n, dtype=torch.int32, device=DEV)
    bg = (0.1, 0.2, 0.3)

    def run(mode, other_color=False):
        ctx.set_stage_mode(mode)
        color = torch.full((P, 3), -5.0, device=DEV)
        means, covs, depth = (torch.full(s_, -5.0, device=DEV) for s_ in ((P, 2), (P, 3), (P,)))
        sh.process(lcgs.GPUPointsProxy(P, 3, d["pos"]), cam, d["sh"], color, 3, 3)
        pr.forward(lcgs.GSProjectorInputProxy(P, d["pos"], d["scale"], d["rotq"], 1.25),
                   lcgs.GSProjectorOutputProxy(means, covs, depth), cam, True)
        cap = 4_000_000
        accel = lcgs.GSTileSplatterAccelProxy(i32(P), i32(P), i64(cap), i32(cap), i64(cap), i32(cap), i32(2 * G))
        target = torch.full((3, H, W), -1.0, device=DEV)
        radii = i32(P)
        out = lcgs.GSSplatForwardOutputProxy(H, W, target, radii, None, None)
        col_in = color
        if other_color:  # the splatter is handed a colour buffer the SH operator did not write: no match
            col_in = torch.rand(P, 3, device=DEV)
        n = ts.forward(accel, lcgs.GSTileSplatterInputProxy(P, bg, means, depth, covs, col_in, d["opacity"]), out, True)
        ctx.synchronize()
        ctx.set_stage_mode("exact")
        return n, target, radii, color, means, accel, col_in

    n_e, img_e, rad_e, col_e, means_e, accel_e, _ = run("exact")
    n_d, img_d, rad_d, col_d, means_d, accel_d, _ = run("deferred")
    assert n_d == n_e > 100000
    assert torch.equal(img_d, img_e) and torch.equal(rad_d, rad_e)
    # the fused frame did not produce the intermediates ...
    assert float(col_d.max()) == -5.0 and float(means_d.max()) == -5.0 and int(accel_d.point_list.abs().max()) == 0
    assert float(col_e.min()) >= 0.0 and int(accel_e.point_list.abs().max()) > 0
    # ... and a call that does not match runs the recorded operators first: the exact mode's buffers and image
    n_m, img_m, rad_m, col_m, means_m, accel_m, col_in = run("deferred", other_color=True)
    ctx.set_stage_mode("exact")
    assert torch.equal(col_m, col_e) and torch.equal(means_m, means_e) and torch.equal(rad_m, rad_e) and n_m == n_e
    assert not torch.equal(img_m, img_e)  # (it was rendered with the other colours, as asked)
    # a flush / synchronise alone materialises what was recorded
    ctx.set_stage_mode("deferred")
    color = torch.full((P, 3), -5.0, device=DEV)
    sh.process(lcgs.GPUPointsProxy(P, 3, d["pos"]), cam, d["sh"], color, 3, 3)
    ctx.synchronize()
    assert torch.equal(color, col_e)
    # a stream switch between record and flush: the recorded operator runs on the OLD stream, in its order, before the switch
    color2 = torch.full((P, 3), -5.0, device=DEV)
    sh.process(lcgs.GPUPointsProxy(P, 3, d["pos"]), cam, d["sh"], color2, 3, 3)
    old = torch.cuda.current_stream(0).cuda_stream
    side = torch.cuda.Stream(device=DEV)
    ctx.set_stream(side.cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(color2, col_e)
    ctx.set_stream(old)
    ctx.set_stage_mode("exact")
