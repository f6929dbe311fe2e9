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


# ---- round 4: a scene the CONTEXT owns keeps {position, extent bound} rows (16 bytes a splat) and phase 1 reads those
# instead of position + scale + rotation; arrays bound by the caller take the full-input path.  Same frame, bit for bit.
def _owned_and_bound(lcgs, scene):
    own = lcgs.Renderer(lcgs.Context(0))
    own.upload_scene(scene, order="file")  # context-owned, the given order: rows line up with the caller-bound renderer's
    d = upload_scene(scene)
    ref = lcgs.Renderer(lcgs.Context(0))
    ref.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"])
    ref.P = own.P
    return own, ref, d


@pytest.mark.parametrize("seed,res,scale_modifier,skew", [(11, (640, 360), 1.0, False), (12, (333, 517), 2.5, False),
                                                         (13, (1920, 1080), 0.4, True), (14, (97, 61), -6.0, True)])
def test_bound_rows_equal_full_inputs(lcgs, seed, res, scale_modifier, skew):
    rng = np.random.default_rng(seed)
    P = 120_000 + seed  # (a ragged last chunk)
    scene = _edge_scene(rng, P)
    own, ref, _ = _owned_and_bound(lcgs, scene)
    W, H = res
    for ang in (0.0, 1.1, 2.7):
        cam = lcgs.get_lookat_cam([-4.0 * np.cos(ang), 4.0 * np.sin(ang), 1.0], [0, 0, 0.5], [0, 0, 1], width=W, height=H)
        if skew:
            cam = _skewed(lcgs, cam)
        a = _lists(lcgs, own, cam, W, H, False, scale_modifier)   # bound rows
        b = _lists(lcgs, ref, cam, W, H, False, scale_modifier)   # position + scale + rotation
        c = _lists(lcgs, own, cam, W, H, True, scale_modifier)    # radii asked for: phase 2 on everything
        for x in (b, c):
            assert a[0] == x[0] and a[1]["num_pairs"] == x[1]["num_pairs"] and a[1]["num_visible"] == x[1]["num_visible"]
            assert torch.equal(a[3], x[3]) and torch.equal(a[4], x[4]) and torch.equal(a[2], x[2])
        assert a[0] > 0 and a[1]["num_visible"] < P


def test_bound_rows_with_extreme_values_and_in_a_batch(lcgs):
    rng = np.random.default_rng(19)
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
    scene["scale"][250:260, 1] = np.nan
    own, ref, _ = _owned_and_bound(lcgs, scene)
    W, H = 512, 288
    cam = lcgs.get_lookat_cam([-4.0, 0.0, 1.0], [0, 0, 0.5], [0, 0, 1], width=W, height=H)
    for sm in (1.0, 0.0, 3.0):
        a = _lists(lcgs, own, cam, W, H, False, sm)
        b = _lists(lcgs, ref, cam, W, H, False, sm)
        assert a[0] == b[0] and a[1]["num_pairs"] == b[1]["num_pairs"] and a[1]["num_visible"] == b[1]["num_visible"]
        assert torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])
        assert torch.equal(a[2].nan_to_num(7.0), b[2].nan_to_num(7.0))
    # the camera-batch sibling borrows the rows
    cam2 = lcgs.get_lookat_cam([0.0, -4.0, 1.5], [0, 0, 0.5], [0, 0, 1], width=W, height=H)
    pair = [torch.zeros(3, H, W, device=DEV) for _ in range(4)]
    own.forward_batch([cam, cam2, cam, cam2], pair)
    own.ctx.synchronize()
    for cam_k, img_k in zip([cam, cam2], pair[:2]):
        one = torch.zeros(3, H, W, device=DEV)
        ref.forward(cam_k, one, sync=True)
        assert torch.equal(one.nan_to_num(7.0), img_k.nan_to_num(7.0))
    assert torch.equal(pair[0].nan_to_num(7.0), pair[2].nan_to_num(7.0)) and torch.equal(pair[1].nan_to_num(7.0), pair[3].nan_to_num(7.0))


def test_bound_rows_follow_the_scene(lcgs):
    """The rows are derived data: an optimiser step on the context's own arrays drops them (frames read the arrays again),
    binding the arrays again rebuilds them, a spatial re-order rebuilds them in the new order."""
    rng = np.random.default_rng(21)
    P = 60_000
    scene = make_scene(rng, P, spread=2.0)
    W, H = 400, 300
    cam = lcgs.get_lookat_cam([-4.0, 0.5, 1.0], [0, 0, 0.5], [0, 0, 1], width=W, height=H)
    own, ref, d = _owned_and_bound(lcgs, scene)
    keys = ("pos", "scale", "rotq", "sh", "opacity")
    act = own.scene_tensors()

    def same_frame():
        a, b = torch.zeros(3, H, W, device=DEV), torch.zeros(3, H, W, device=DEV)
        na, nb = own.forward(cam, a, sync=True), ref.forward(cam, b, sync=True)
        assert na == nb > 0 and torch.equal(a, b)
        return a

    first = same_frame()
    # 1. Adam in place on the context's arrays (and the same step on the caller-bound copy): positions and scales move
    grads = {k: torch.from_numpy(rng.normal(size=tuple(act[k].shape)).astype(np.float32)).to(DEV) for k in keys}
    lr = {"pos": 0.05, "sh_dc": 0.0, "sh_rest": 0.0, "opacity": 0.0, "scale": 0.2, "rot": 0.05}
    for r_, arr in ((own, act), (ref, d)):
        raw = {"pos": arr["pos"], "scale": torch.log(arr["scale"]), "rotq": arr["rotq"].clone(), "sh": arr["sh"],
               "opacity": torch.log(arr["opacity"] / (1 - arr["opacity"]))}
        m = {k: torch.zeros_like(raw[k]) for k in keys}
        v = {k: torch.zeros_like(raw[k]) for k in keys}
        r_.adam_step(grads, raw, m, v, arr, 1, lr)
    moved = same_frame()
    assert not torch.equal(moved, first)
    # 2. the arrays bound again: rows rebuilt from the moved scene
    own.bind_scene(*[act[k] for k in keys])
    assert torch.equal(same_frame(), moved)
    # 3. written behind the library's back, then bound again (the documented way to say "the scene changed")
    act["pos"][: P // 2] += 0.3
    d["pos"][: P // 2] += 0.3
    own.bind_scene(*[act[k] for k in keys])
    assert not torch.equal(same_frame(), moved)
    # 4. a spatial re-order: new arrays, new rows; the image does not change
    before = same_frame()
    own.reorder_scene_spatial()
    after = torch.zeros(3, H, W, device=DEV)
    own.forward(cam, after, sync=True)
    assert torch.equal(after, before)


def test_declared_static_arrays_of_the_caller(lcgs):
    """lcgs_scene_declare_static: caller-bound arrays get the rows on request; the frames (fused; the three stage operators in
    deferred mode beside a standing declaration) stay the full-input frames bit for bit; the declaration is keyed to the pointers, repeated after a change,
    and dropped by the library's own writers."""
    rng = np.random.default_rng(31)
    P = 90_001
    scene = _edge_scene(rng, P)
    d = upload_scene(scene)
    W, H = 640, 360
    cam = lcgs.get_lookat_cam([-4.0, 0.3, 1.0], [0, 0, 0.5], [0, 0, 1], width=W, height=H)
    plain = lcgs.Renderer(lcgs.Context(0))
    plain.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"])
    plain.P = P
    ref = _lists(lcgs, plain, cam, W, H, False, 1.0)
    r = lcgs.Renderer(lcgs.Context(0))
    r.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"])
    r.P = P
    r.declare_static(d["pos"], d["scale"], d["rotq"])
    a = _lists(lcgs, r, cam, W, H, False, 1.0)
    assert a[0] == ref[0] > 0 and torch.equal(a[3], ref[3]) and torch.equal(a[4], ref[4]) and torch.equal(a[2], ref[2])
    # other arrays bound: the rows belong to the declared ones and are not used
    d2 = upload_scene({k: (v + (0.25 if k == "pos" else 0.0)).astype(np.float32) for k, v in scene.items()})
    r.bind_scene(d2["pos"], d2["scale"], d2["rotq"], d2["sh"], d2["opacity"])
    plain.bind_scene(d2["pos"], d2["scale"], d2["rotq"], d2["sh"], d2["opacity"])
    b, refb = _lists(lcgs, r, cam, W, H, False, 1.0), _lists(lcgs, plain, cam, W, H, False, 1.0)
    assert b[0] == refb[0] and torch.equal(b[2], refb[2]) and not torch.equal(b[2], a[2])
    # back to the declared arrays, changed in place: the declaration is repeated
    r.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"])
    plain.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"])
    d["pos"][: P // 3] -= 0.4
    d["scale"][P // 2:] *= 1.5
    r.declare_static(d["pos"], d["scale"], d["rotq"])
    c, refc = _lists(lcgs, r, cam, W, H, False, 1.0), _lists(lcgs, plain, cam, W, H, False, 1.0)
    assert c[0] == refc[0] and torch.equal(c[3], refc[3]) and torch.equal(c[2], refc[2]) and not torch.equal(c[2], a[2])
    # the stage operators in deferred mode render the fused frame from the caller's arrays (with radii: every splat is projected)
    shp, prj, spl = lcgs.SHProcessor(), lcgs.GSProjector(), lcgs.GSTileSplatter()
    for op in (shp, prj, spl):
        op.create(r.ctx)
    z = lambda *s_, dt=torch.float32: torch.zeros(*s_, dtype=dt, device=DEV)
    G = ((W + 15) // 16) * ((H + 15) // 16)
    color, means, covs, depth = z(P, 3), z(P, 2), z(P, 3), z(P)
    Lcap = 4_000_000
    accel = lcgs.GSTileSplatterAccelProxy(z(P, dt=torch.int32), z(P, dt=torch.int32), z(Lcap, dt=torch.int64), z(Lcap, dt=torch.int32),
                                          z(Lcap, dt=torch.int64), z(Lcap, dt=torch.int32), z(2 * G, dt=torch.int32))
    radii_s, img_s = z(P, dt=torch.int32), z(3, H, W)
    r.ctx.set_stage_mode("deferred")
    shp.process(lcgs.GPUPointsProxy(P, 3, d["pos"]), cam, d["sh"], color, 3, 3)
    prj.forward(lcgs.GSProjectorInputProxy(P, d["pos"], d["scale"], d["rotq"], 1.0), lcgs.GSProjectorOutputProxy(means, covs, depth), cam)
    n = spl.forward(accel, lcgs.GSTileSplatterInputProxy(P, (0.0, 0.0, 0.0), means, depth, covs, color, d["opacity"]),
                    lcgs.GSSplatForwardOutputProxy(H, W, img_s, radii_s))
    r.ctx.set_stage_mode("exact")
    assert n == refc[0]
    fused = torch.zeros(3, H, W, device=DEV)
    plain.forward(cam, fused, sync=True)
    assert torch.equal(img_s, fused)
    # withdrawn: frames as before
    r.declare_static()
    e = _lists(lcgs, r, cam, W, H, False, 1.0)
    assert e[0] == refc[0] and torch.equal(e[2], refc[2])


def test_in_place_writes_need_scene_modified_and_the_guard_sees_them(lcgs):
    """Renderer.scene_tensors() hands out writable aliases of arrays the context derives rows from: a write through them is
    invisible to the library.  lcgs_debug_verify_derived (the test-mode guard) counts the stale rows; lcgs_scene_modified is
    the documented notification, after which the frame is the moved scene's."""
    rng = np.random.default_rng(41)
    P = 70_003
    scene = _edge_scene(rng, P)
    W, H = 640, 360
    cam = lcgs.get_lookat_cam([-4.0, 0.3, 1.0], [0, 0, 0.5], [0, 0, 1], width=W, height=H)
    own = lcgs.Renderer(lcgs.Context(0))
    own.upload_scene(scene, order="file")
    assert own.verify_derived() == 0
    act = own.scene_tensors()
    act["pos"][: P // 2] += 0.35        # behind the library's back
    act["scale"][P // 3:] *= 1.7
    torch.cuda.synchronize()
    stale = own.verify_derived()
    assert stale >= P // 2, stale       # the guard sees every moved row
    own.scene_modified()
    assert own.verify_derived() == 0
    ref = lcgs.Renderer(lcgs.Context(0))
    d = {k: act[k].clone() for k in act}
    ref.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"])
    a, b = torch.zeros(3, H, W, device=DEV), torch.zeros(3, H, W, device=DEV)
    assert own.forward(cam, a, sync=True) == ref.forward(cam, b, sync=True) > 0
    assert torch.equal(a, b)


def test_optimiser_step_through_another_context_drops_the_rows(lcgs):
    """Context B binds the arrays context A owns and runs lcgs_adam_step on them as `activated`: A's derived rows go with the
    step (every live context is looked at), so A's next frame is the moved scene's."""
    rng = np.random.default_rng(43)
    P = 60_001
    scene = _edge_scene(rng, P)
    W, H = 640, 360
    cam = lcgs.get_lookat_cam([-4.0, 0.3, 1.0], [0, 0, 0.5], [0, 0, 1], width=W, height=H)
    keys = ("pos", "scale", "rotq", "sh", "opacity")
    A = lcgs.Renderer(lcgs.Context(0))
    A.upload_scene(scene, order="file")
    act = A.scene_tensors()
    before = torch.zeros(3, H, W, device=DEV)
    A.forward(cam, before, sync=True)
    B = lcgs.Renderer(lcgs.Context(0))
    B.bind_scene(*[act[k] for k in keys])
    grads = {k: torch.from_numpy(rng.normal(size=tuple(act[k].shape)).astype(np.float32)).to(DEV) for k in keys}
    lr = {"pos": 0.05, "sh_dc": 0.0, "sh_rest": 0.0, "opacity": 0.0, "scale": 0.2, "rot": 0.05}
    raw = {"pos": act["pos"], "scale": torch.log(act["scale"].abs() + 1e-12), "rotq": act["rotq"].clone(), "sh": act["sh"],
           "opacity": torch.log(act["opacity"] / (1 - act["opacity"]).clamp_min(1e-6))}
    m = {k: torch.zeros_like(raw[k]) for k in keys}
    v = {k: torch.zeros_like(raw[k]) for k in keys}
    B.adam_step(grads, raw, m, v, act, 1, lr)
    B.ctx.synchronize()
    assert A.verify_derived() == 0      # nothing derived is in use any more (the rows were dropped, not left stale)
    ref = lcgs.Renderer(lcgs.Context(0))
    d = {k: act[k].clone() for k in act}
    ref.bind_scene(*[d[k] for k in keys])
    a, b = torch.zeros(3, H, W, device=DEV), torch.zeros(3, H, W, device=DEV)
    assert A.forward(cam, a, sync=True) == ref.forward(cam, b, sync=True) > 0
    assert torch.equal(a, b) and not torch.equal(a, before)


def test_writers_on_another_host_thread_post_to_the_owner_instead_of_touching_it(lcgs):
    """The threading contract (context.hpp foreign_writes; round-5 advisor): a context is used by one host thread, and a writer
    on ANOTHER thread -- an optimiser step through its own context B on arrays context A renders, or B's lcgs_scene_modified
    -- never mutates A: it compares with what A published and posts a flag that A's thread honours at its next use.  Thread B
    steps the arrays 20 times while thread A keeps rendering them (those frames read arrays in flux: no result is asserted,
    only that nothing faults); afterwards A, untouched by any call of its own, renders the final arrays' frame -- its rows
    were declared stale, not left in use -- and after B's scene_modified A's kept frame state is gone."""
    import threading

    rng = np.random.default_rng(47)
    P = 50_001
    scene = _edge_scene(rng, P)
    W, H = 480, 272
    cam = lcgs.get_lookat_cam([-4.0, 0.3, 1.0], [0, 0, 0.5], [0, 0, 1], width=W, height=H)
    keys = ("pos", "scale", "rotq", "sh", "opacity")
    A = lcgs.Renderer(lcgs.Context(0))
    A.upload_scene(scene, order="file")
    act = A.scene_tensors()
    B = lcgs.Renderer(lcgs.Context(0))
    B.bind_scene(*[act[k] for k in keys])
    grads = {k: torch.from_numpy(rng.normal(size=tuple(act[k].shape)).astype(np.float32)).to(DEV) for k in keys}
    lr = {"pos": 0.01, "sh_dc": 0.0, "sh_rest": 0.0, "opacity": 0.0, "scale": 0.02, "rot": 0.01}
    raw = {"pos": act["pos"], "scale": torch.log(act["scale"].abs() + 1e-12), "rotq": act["rotq"].clone(), "sh": act["sh"],
           "opacity": torch.log(act["opacity"] / (1 - act["opacity"]).clamp_min(1e-6))}
    m = {k: torch.zeros_like(raw[k]) for k in keys}
    v = {k: torch.zeros_like(raw[k]) for k in keys}
    errors = []
    go = threading.Barrier(2)

    def writer():
        try:
            go.wait()
            for step in range(1, 21):
                B.adam_step(grads, raw, m, v, act, step, lr)
            B.ctx.synchronize()
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    def viewer():
        try:
            img = torch.zeros(3, H, W, device=DEV)
            go.wait()
            for _ in range(40):
                A.forward(cam, img, sync=True)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    ts = [threading.Thread(target=writer), threading.Thread(target=viewer)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    torch.cuda.synchronize()
    assert A.verify_derived() == 0      # A uses no derived rows any more: they were declared stale by B's first step
    ref = lcgs.Renderer(lcgs.Context(0))
    d = {k: act[k].clone() for k in act}
    ref.bind_scene(*[d[k] for k in keys])
    a, b = torch.zeros(3, H, W, device=DEV), torch.zeros(3, H, W, device=DEV)
    assert A.forward(cam, a, keep_state=True, sync=True) == ref.forward(cam, b, sync=True) > 0
    assert torch.equal(a, b)
    # binding the own arrays again rebuilds the rows (and settles the flag); same frame
    A.bind_scene(*[act[k] for k in keys])
    A.forward(cam, a, keep_state=True, sync=True)
    assert torch.equal(a, b) and A.verify_derived() == 0
    # B writes behind the library's back and says so: A's kept frame state must not survive (a backward would mix frames)
    th = threading.Thread(target=lambda: (act["pos"].add_(0.01), torch.cuda.synchronize(), B.scene_modified()))
    th.start()
    th.join()
    g = {k: torch.zeros_like(act[k]) for k in keys}
    with pytest.raises(lcgs.LcgsError):
        A.backward(torch.zeros(3, H, W, device=DEV), *[g[k] for k in keys])
    ref.bind_scene(*[act[k].clone() for k in keys])
    assert A.forward(cam, a, sync=True) == ref.forward(cam, b, sync=True) and torch.equal(a, b)


def test_destroy_returns_every_byte(lcgs):
    """create / upload / frame (forward + backward state) / destroy in a loop: free device memory comes back each time
    (round 4 leaked the 16-byte cull rows of every context that owned a scene)."""
    rng = np.random.default_rng(47)
    P = 400_000
    scene = make_scene(rng, P)
    W, H = 640, 360
    cam = lcgs.get_lookat_cam([-3.0, 0.2, 1.0], [0, 0, 0.5], [0, 0, 1], width=W, height=H)

    def once():
        r = lcgs.Renderer(lcgs.Context(0))
        r.upload_scene(scene)
        img = torch.zeros(3, H, W, device=DEV)
        r.forward(cam, img, keep_state=True, sync=True)
        r.ctx.close()
        del r, img

    once()  # first use pays for one-off allocations of the runtime
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(4):
        once()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < (2 << 20), f"{(free0 - free1) / 2**20:.1f} MiB lost over four create/destroy rounds (16 B x P = {16 * P / 2**20:.1f} MiB a round)"


def test_empty_frame_still_clears_what_the_backward_relies_on(lcgs):
    """A keep_state frame that draws nothing returns before its tiles -- but the host has noted the 2-D gradient rows and the
    backward's counters as cleared by that launch: the gradients of the empty frame must be exact zeros, also right after a
    step that filled them."""
    rng = np.random.default_rng(53)
    P = 30_000
    scene = make_scene(rng, P)
    W, H = 320, 240
    r = lcgs.Renderer(lcgs.Context(0))
    r.upload_scene(scene)
    d = r.scene_tensors()
    g = {k: torch.zeros_like(d[k]) for k in ("pos", "scale", "rotq", "sh", "opacity")}
    img = torch.zeros(3, H, W, device=DEV)
    dL = torch.randn(3, H, W, device=DEV)
    see = lcgs.get_lookat_cam([-3.0, 0.2, 1.0], [0, 0, 0.5], [0, 0, 1], width=W, height=H)
    away = lcgs.get_lookat_cam([-3.0, 0.2, 1.0], [-9.0, 0.4, 1.5], [0, 0, 1], width=W, height=H)
    assert r.forward(see, img, keep_state=True, sync=True) > 0
    r.backward(dL, g["pos"], g["scale"], g["rotq"], g["sh"], g["opacity"])
    r.ctx.synchronize()
    assert float(g["pos"].abs().sum()) > 0
    assert r.forward(away, img, keep_state=True, sync=True) == 0
    r.backward(dL, g["pos"], g["scale"], g["rotq"], g["sh"], g["opacity"])
    r.ctx.synchronize()
    for k in g:
        assert not g[k].any(), k
