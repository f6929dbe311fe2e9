"""`-m gpu`: the frame in two halves (csrc/abi_owner.cpp; DESIGN.md 7b, splat ownership).  With VIRTUAL ranks on the one GPU:
every owner projects its row range for a view, the view's renderer draws the concatenated records, and the owners turn their
share of the 2-D gradients into parameter gradients -- the image must be the fused frame's (and the oracle's) bit for bit, the
gradients the ordinary backward's."""
import numpy as np
import pytest
import torch

from conftest import make_scene
from gpu_util import DEV, assert_image_parity, upload_scene

pytestmark = pytest.mark.gpu

KEYS = ("pos", "scale", "rotq", "sh", "opacity")
POSE = ([-3, -0.5, 2.3], [0, 0, 0.5], [0, 0, 1])
W, H = 320, 240


def _ranges(P, world):
    import luisacomputegaussiansplatting_amd.multi_gpu as mg

    return [mg.owner_range(P, world, r) for r in range(world)]


def _rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


def _two_halves(lcgs, r, cam, spans, dL, grads, accumulate=False, bg=(0.1, 0.2, 0.3)):
    """one view through the two halves on renderer r: returns the image"""
    parts = [r.owner_project(o, cam, first, count) for o, (first, count) in enumerate(spans)]
    rows = torch.cat([p[0] for p in parts])
    recs = torch.cat([p[1] for p in parts])
    assert bool((rows[1:] > rows[:-1]).all())
    img = torch.full((3, H, W), -1.0, device=DEV)
    r.owner_render(cam, rows, recs, img, bg=bg, keep_state=True)
    g2d = torch.zeros(rows.shape[0], r.OWNER_GRAD_FLOATS, device=DEV)
    r.owner_render_backward(dL, g2d)
    at = 0
    for o, p in enumerate(parts):
        n = int(p[0].shape[0])
        r.owner_backward(o, g2d[at:at + n].contiguous(), *[grads[k] for k in KEYS], accumulate=accumulate)
        at += n
    r.ctx.synchronize()
    return img, int(rows.shape[0])


@pytest.mark.parametrize("world", [1, 2, 3])
@pytest.mark.parametrize("reordered", [False, True])
def test_two_halves_equal_the_fused_frame_and_its_backward(lcgs, oracle, world, reordered):
    rng = np.random.default_rng(40 + world)
    P = 30001
    scene = make_scene(rng, P, log_scale=(-4.0, 0.8))
    scene["pos"][100:200] = scene["pos"][300:400]  # exactly equal depths across what will be different owners' rows
    scene["pos"][P - 150:P - 50] = scene["pos"][300:400]
    cam = lcgs.get_lookat_cam(*POSE, width=W, height=H)
    bg = (0.1, 0.2, 0.3)
    # the ordinary path: one context, the whole scene
    ref = lcgs.Renderer(lcgs.Context(0))
    if reordered:
        ref.upload_scene(scene)  # context-owned: spatial order, equal depths restored to file order by the tie pass
        act = ref.scene_tensors()
    else:
        act = upload_scene(scene)
        ref.bind_scene(*[act[k] for k in KEYS])
    img_ref = torch.full((3, H, W), -1.0, device=DEV)
    n_ref = ref.forward(cam, img_ref, bg=bg, keep_state=True, sync=True)
    dL = torch.from_numpy(rng.normal(size=(3, H, W)).astype(np.float32)).to(DEV)
    g_ref = {k: torch.full_like(act[k], 7.0) for k in KEYS}
    ref.backward(dL, *[g_ref[k] for k in KEYS])
    ref.ctx.synchronize()
    orc = oracle.render(scene, oracle.lookat(*POSE, width=W, height=H), bg=bg)
    assert n_ref == orc["num_rendered"] > 0
    assert_image_parity(img_ref.cpu().numpy(), orc)
    # the two halves, with `world` virtual owners, on another context holding the same arrays (same order, same permutation)
    r = lcgs.Renderer(lcgs.Context(0))
    if reordered:
        r.upload_scene(scene)
        act2 = r.scene_tensors()
        assert all(torch.equal(act2[k], act[k]) for k in KEYS)
    else:
        r.bind_scene(*[act[k] for k in KEYS])
        act2 = act
    g = {k: torch.full_like(act2[k], 5.0) for k in KEYS}
    img, n_rows = _two_halves(lcgs, r, cam, _ranges(P, world), dL, g, bg=bg)
    assert n_rows == ref.frame_stats()["num_visible"]
    assert torch.equal(img, img_ref), f"{int((img != img_ref).any(0).sum())} pixels differ from the fused frame"
    for k in KEYS:
        zero_ref = (g_ref[k].reshape(P, -1) == 0).all(1)
        assert torch.equal((g[k].reshape(P, -1) == 0).all(1), zero_ref), k  # the same rows are exact zeros
        assert _rel(g[k], g_ref[k]) <= 1e-4, (k, _rel(g[k], g_ref[k]))  # (float-atomic order of the 2-D sums: two runs of ONE path differ by ~2e-5)


def test_views_accumulate_and_empty_ranges_are_harmless(lcgs):
    rng = np.random.default_rng(9)
    P = 12000
    scene = make_scene(rng, P, log_scale=(-4.0, 0.8))
    scene["pos"][:4000] += np.array([0.0, 0.0, 50.0], np.float32)  # the first owner's rows are nowhere near the screen
    act = upload_scene(scene)
    cams = [lcgs.get_lookat_cam([-3 * np.cos(a), -0.5 + 3 * np.sin(a), 2.3], [0, 0, 0.5], [0, 0, 1], width=W, height=H)
            for a in (0.0, 0.7)]
    dLs = [torch.randn(3, H, W, device=DEV) for _ in cams]
    ref = lcgs.Renderer(lcgs.Context(0))
    ref.bind_scene(*[act[k] for k in KEYS])
    img = torch.zeros(3, H, W, device=DEV)
    g_ref = {k: torch.zeros_like(act[k]) for k in KEYS}
    for j, (cam, dL) in enumerate(zip(cams, dLs)):
        ref.forward(cam, img, keep_state=True, sync=True)
        ref.backward(dL, *[g_ref[k] for k in KEYS], accumulate=j > 0)
    ref.ctx.synchronize()
    # two owners ((0, 4000): off screen; the rest), two views; each (owner, view) pair its own slot
    r = lcgs.Renderer(lcgs.Context(0))
    r.bind_scene(*[act[k] for k in KEYS])
    spans = [(0, 4000), (4000, P - 4000)]
    g = {k: torch.full_like(act[k], 3.0) for k in KEYS}
    for j, (cam, dL) in enumerate(zip(cams, dLs)):
        parts = [r.owner_project(2 * j + o, cam, f, c) for o, (f, c) in enumerate(spans)]
        assert parts[0][0].numel() == 0  # nothing of the first range reaches the screen
        rows, recs = torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])
        r.owner_render(cam, rows, recs, img, keep_state=True)
        g2d = torch.zeros(rows.shape[0], r.OWNER_GRAD_FLOATS, device=DEV)
        r.owner_render_backward(dL, g2d)
        r.owner_backward(2 * j + 0, g2d[:0], *[g[k] for k in KEYS], accumulate=j > 0)
        r.owner_backward(2 * j + 1, g2d, *[g[k] for k in KEYS], accumulate=j > 0)
    r.ctx.synchronize()
    for k in KEYS:
        assert float(g[k][:4000].abs().max()) == 0.0  # zero-filled by the first view, never touched again
        assert _rel(g[k], g_ref[k]) <= 1e-4, (k, _rel(g[k], g_ref[k]))


def test_owner_step_of_the_package_on_the_hip_engine(lcgs):
    """multi_gpu.ViewParallelTrainer(mode="owner") with HipEngine at world size 1 (a single-rank process group on gloo): the
    protocol's local path -- every message stays on the rank -- must give the parameters of an ordinary step."""
    import os
    import socket

    import torch.distributed as dist

    import luisacomputegaussiansplatting_amd.multi_gpu as mg

    rng = np.random.default_rng(3)
    P = 20000
    scene = make_scene(rng, P, log_scale=(-4.0, 0.8))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        lr = {"pos": 1.6e-3, "sh_dc": 2.5e-2, "sh_rest": 1.25e-3, "opacity": 5e-2, "scale": 5e-3, "rot": 1e-2}
        cams = [lcgs.get_lookat_cam(*POSE, width=W, height=H)]
        dL = torch.randn(3, H, W, device=DEV)
        results = {}
        for mode in ("local", "owner"):
            act = upload_scene(scene)
            raw = {"pos": act["pos"], "scale": torch.log(act["scale"]), "rotq": act["rotq"].clone(), "sh": act["sh"],
                   "opacity": torch.log(act["opacity"] / (1 - act["opacity"]))}
            r = lcgs.Renderer(lcgs.Context(0))
            eng = mg.HipEngine(r, raw=raw, activated=act, lr=lr)
            grads = {k: torch.zeros_like(act[k]) for k in KEYS}
            coll = mg.TorchCollective(dist, 0, 1) if mode == "owner" else None
            tr = mg.ViewParallelTrainer(eng, coll, cams, grads, mode=mode)
            for _ in range(2):
                tr.step(dL)
            r.ctx.synchronize()
            results[mode] = {k: raw[k].clone() for k in KEYS}
            if mode == "owner":
                assert coll.last_stats["bytes_sent"] == 0 and coll.last_stats["on_screen_rows_received"] > 0
        for k in KEYS:
            a, b = results["owner"][k], results["local"][k]
            # Adam's first steps move every touched parameter by ~lr whatever the gradient's size: a last-bit difference in a
            # gradient near zero can flip a sign -- compare in units of the learning rate
            unit = max(lr.get({"rotq": "rot"}.get(k, k), lr["sh_dc"]), 1e-12)
            d = (a - b).abs() / unit
            assert float((d > 0.05).float().mean()) < 0.01, (k, float(d.max()))
    finally:
        dist.destroy_process_group()


def test_the_two_backward_entry_points_do_not_mix(lcgs):
    rng = np.random.default_rng(1)
    scene = make_scene(rng, 5000)
    act = upload_scene(scene)
    r = lcgs.Renderer(lcgs.Context(0))
    r.bind_scene(*[act[k] for k in KEYS])
    cam = lcgs.get_lookat_cam(*POSE, width=W, height=H)
    img = torch.zeros(3, H, W, device=DEV)
    dL = torch.ones(3, H, W, device=DEV)
    g = {k: torch.zeros_like(act[k]) for k in KEYS}
    g2d = torch.zeros(5000, r.OWNER_GRAD_FLOATS, device=DEV)
    with pytest.raises(lcgs.LcgsError) as e:  # no owner frame yet
        r.owner_render_backward(dL, g2d)
    assert e.value.status == 8  # LCGS_ERR_STATE
    rows, recs = r.owner_project(0, cam, 0, 5000)
    r.owner_render(cam, rows, recs, img, keep_state=True)
    with pytest.raises(lcgs.LcgsError):  # a frame drawn from received records is not differentiated by the ordinary call
        r.backward(dL, *[g[k] for k in KEYS])
    r.owner_render_backward(dL, g2d[:rows.shape[0]])
    r.forward(cam, img, keep_state=True, sync=True)  # an ordinary frame again: the ordinary backward again
    r.backward(dL, *[g[k] for k in KEYS])
    with pytest.raises(lcgs.LcgsError):
        r.owner_render_backward(dL, g2d)
    with pytest.raises(lcgs.LcgsError):  # rows outside the scene
        r.owner_project(1, cam, 4000, 2000)
    r.ctx.synchronize()


def test_every_view_of_a_step_with_one_read_back(lcgs):
    """owner_project_all: N asynchronous projections (view v -> slot v) + lcgs_owner_counts = the N synchronous calls; a slot
    whose count is still on the device refuses the backward."""
    rng = np.random.default_rng(77)
    P = 30_000
    scene = make_scene(rng, P, spread=2.0)
    r = lcgs.Renderer(lcgs.Context(0))
    r.upload_scene(scene)  # context-owned: the cull pass of the range reads the bound rows
    poses = [([-3, -0.5, 2.3], [0, 0, 0.5], [0, 0, 1]), ([0.5, -3.2, 1.5], [0, 0, 0.5], [0, 0, 1]),
             ([60.0, 0.0, 1.0], [80.0, 0.0, 1.0], [0, 0, 1])]  # (the third looks away: nothing on screen)
    cams = [lcgs.get_lookat_cam(*p, width=W, height=H) for p in poses]
    first, count = 5_000, 20_000
    one_by_one = [r.owner_project(v, cam, first, count) for v, cam in enumerate(cams)]
    together = r.owner_project_all(cams, first, count)
    assert len(together) == len(cams)
    for (ra, qa), (rb, qb) in zip(one_by_one, together):
        assert ra.shape == rb.shape and torch.equal(ra, rb) and torch.equal(qa, qb)
    assert together[0][0].numel() > 0 and together[2][0].numel() == 0
    # more views than the context has lanes (lcgs_owner_project_views runs them side by side, two at a time): a lane's second
    # and third pipeline reuse its scratch behind the first
    more = cams + [lcgs.get_lookat_cam([3.0 * np.cos(a), 3.0 * np.sin(a), 1.8], [0, 0, 0.5], [0, 0, 1], width=W, height=H)
                   for a in np.linspace(0.3, 5.5, 8)]
    one_by_one = [r.owner_project(v, cam, first, count) for v, cam in enumerate(more)]
    for _ in range(2):  # (twice: the second call finds the lanes' buffers and streams in place)
        together = r.owner_project_all(more, first, count)
        for (ra, qa), (rb, qb) in zip(one_by_one, together):
            assert ra.shape == rb.shape and torch.equal(ra, rb) and torch.equal(qa, qb)
    assert sum(int(t[0].numel() > 0) for t in together) >= 8
    together = r.owner_project_all(cams, first, count)  # (slots 0..2 as the rest of the test expects them)
    # the backward of a slot needs its count on the host
    import ctypes as C

    import luisacomputegaussiansplatting_amd.api as api

    lib = api.load_library()
    rows = torch.empty(count, dtype=torch.int32, device=DEV)
    recs = torch.empty(count, 12, dtype=torch.float32, device=DEV)
    assert lib.lcgs_owner_project(r.ctx._h, C.c_int(0), C.byref(cams[0]), C.c_float(1.0), C.c_int(first), C.c_int(count),
                                  C.c_int(1), api._ptr(rows), api._ptr(recs), None) == 0
    g = {k: torch.zeros(P, *s, device=DEV) for k, s in (("pos", (3,)), ("scale", (3,)), ("rotq", (4,)), ("sh", (48,)), ("opacity", ()))}
    g2d = torch.zeros(count, 12, device=DEV)
    with pytest.raises(api.LcgsError):
        r.owner_backward(0, g2d, *[g[k] for k in KEYS], accumulate=False)
    n = (C.c_int * 1)()
    assert lib.lcgs_owner_counts(r.ctx._h, C.c_int(0), C.c_int(1), n) == 0 and n[0] == together[0][0].numel()
    r.owner_backward(0, g2d[:n[0]].contiguous(), *[g[k] for k in KEYS], accumulate=False)
    r.ctx.synchronize()


# ---------------------------------------------------------------------------------------------------------------------
# The ownership step WITH its transport through the C ABI (lcgs_owner_step_forward / _backward, host/comm.cpp)
# ---------------------------------------------------------------------------------------------------------------------
def _reference_views(lcgs, scene, cams, dLs, bg):
    """the ordinary path: every view's fused frame + the gradients of all views summed (file-order arrays)"""
    act = upload_scene(scene)
    ref = lcgs.Renderer(lcgs.Context(0))
    ref.bind_scene(*[act[k] for k in KEYS])
    imgs, g = [], {k: torch.zeros_like(act[k]) for k in KEYS}
    vis = []
    for j, (cam, dL) in enumerate(zip(cams, dLs)):
        img = torch.full((3, H, W), -1.0, device=DEV)
        assert ref.forward(cam, img, bg=bg, keep_state=True, sync=True) > 0
        vis.append(ref.frame_stats()["num_visible"])
        ref.backward(dL, *[g[k] for k in KEYS], accumulate=j > 0)
        imgs.append(img)
    ref.ctx.synchronize()
    return imgs, g, vis


def _padded(n):
    return n + n // 4 + 1024  # comm.cpp padded_rows (before the clip to the owner's range)


@pytest.mark.parametrize("self_p2p,async_steps", [(False, False), (True, False), (False, True), (True, True)])
def test_owner_step_through_rccl_at_world_size_one(lcgs, monkeypatch, self_p2p, async_steps):
    """One rank owns everything: the step is the fused frame + its backward.  With LCGS_OWNER_SELF_P2P=1 the rank's own share
    travels through ncclSend / ncclRecv to itself -- RCCL's point-to-point path, the one N > 1 ranks use.  async_steps: from the
    second step on nothing is read back (lcgs_owner_step_set_async): padded messages sized from the previous step's counts,
    the true counts on the device, one verdict behind the step (lcgs_owner_step_finish)."""
    if self_p2p:
        monkeypatch.setenv("LCGS_OWNER_SELF_P2P", "1")
    rng = np.random.default_rng(71)
    P = 40_003
    scene = make_scene(rng, P, log_scale=(-4.0, 0.8))
    cams = [lcgs.get_lookat_cam(*POSE, width=W, height=H)]
    dLs = [torch.from_numpy(rng.normal(size=(3, H, W)).astype(np.float32)).to(DEV)]
    bg = (0.1, 0.2, 0.3)
    imgs, g_ref, vis = _reference_views(lcgs, scene, cams, dLs, bg)
    act = upload_scene(scene)
    r = lcgs.Renderer(lcgs.Context(0))
    r.bind_scene(*[act[k] for k in KEYS])
    comm = lcgs.Comm(r.ctx, 0, 1)
    try:
        g = {k: torch.full_like(act[k], 9.0) for k in KEYS}
        if async_steps:
            comm.owner_step_set_async(True)
        for step in range(3):  # repeatedly: buffers re-used, slots re-used; async: the first step reads back, the others do not
            img = torch.full((3, H, W), -1.0, device=DEV)
            if async_steps:
                assert comm.owner_step(cams, img, dLs[0], g, bg=bg) == 0
            else:
                comm.owner_step_forward(cams, img, bg=bg)
                comm.owner_step_backward(dLs[0], g)
            r.ctx.synchronize()
            assert torch.equal(img, imgs[0])
            for k in KEYS:
                assert _rel(g[k], g_ref[k]) <= 1e-4, k
        st = comm.stats()
        assert st["touched_rows"] == vis[0]
        rows = min(P, _padded(vis[0])) if async_steps else vis[0]
        want = rows * (4 + 48 + 48) if self_p2p else 0
        assert st["bytes_sent"] == want and st["bytes_received"] == want, (st, want)
    finally:
        comm.close()


@pytest.mark.parametrize("world,reordered,async_steps", [(2, False, False), (3, True, False), (8, False, False),
                                                         (2, True, True), (3, False, True), (8, False, True)])
def test_owner_step_with_n_ranks_in_process(lcgs, world, reordered, async_steps):
    """N contexts on the one GPU, one host thread each, joined by the in-process loopback transport: the SAME C code path as
    the RCCL step (message table, offsets, slot state, stream ordering) with N > 1 participants.  Every rank's image is its
    view's fused frame bit for bit; every rank's rows hold the gradients of all N views summed; the byte counts are the
    design's (4 + 48) out as an owner, 48 back as a renderer.  async_steps: the steps behind the first read nothing back --
    padded per-owner segments, true counts on the device, positions instead of compacted rows -- and give the same."""
    import threading

    rng = np.random.default_rng(80 + world)
    P = 50_007
    scene = make_scene(rng, P, log_scale=(-4.0, 0.8))
    scene["pos"][100:200] = scene["pos"][300:400]  # equal depths across owners' ranges
    cams = [lcgs.get_lookat_cam([-3 * np.cos(a), -0.5 + 3 * np.sin(a), 2.3], [0, 0, 0.5], [0, 0, 1], width=W, height=H)
            for a in np.linspace(0.0, 1.4, world)]
    dLs = [torch.from_numpy(rng.normal(size=(3, H, W)).astype(np.float32)).to(DEV) for _ in cams]
    bg = (0.1, 0.2, 0.3)
    imgs_ref, g_ref, vis = _reference_views(lcgs, scene, cams, dLs, bg)
    group = lcgs.api.LoopbackGroup(world)
    out, errors = [None] * world, []

    def rank_main(me):
        try:
            side = torch.cuda.Stream(device=DEV)
            with torch.cuda.stream(side):
                r = lcgs.Renderer(lcgs.Context(0, side.cuda_stream))
                if reordered:
                    r.upload_scene(scene)  # spatial order: the ranks own ranges of the RE-ORDERED rows
                    act = r.scene_tensors()
                    perm = r.permutation().long()
                else:
                    act = upload_scene(scene)
                    r.bind_scene(*[act[k] for k in KEYS])
                    perm = None
                comm = lcgs.Comm(r.ctx, me, world, loopback=group)
                g = {k: torch.full_like(act[k], 9.0) for k in KEYS}
                img = torch.full((3, H, W), -1.0, device=DEV)
                if async_steps:
                    comm.owner_step_set_async(True)
                for _ in range(3 if async_steps else 2):
                    if async_steps:
                        assert comm.owner_step(cams, img, dLs[me], g, bg=bg) == 0
                    else:
                        comm.owner_step_forward(cams, img, bg=bg)
                        comm.owner_step_backward(dLs[me], g)
                r.ctx.synchronize()
                side.synchronize()
                out[me] = (img, g, comm.stats(), perm)
                comm.close()
        except Exception as e:  # noqa: BLE001
            errors.append((me, repr(e)))

    torch.cuda.synchronize()
    threads = [threading.Thread(target=rank_main, args=(me,)) for me in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    assert all(not t.is_alive() for t in threads), "a rank hangs"
    group.close()
    sent = received = 0
    for me in range(world):
        img, g, st, perm = out[me]
        assert torch.equal(img, imgs_ref[me]), f"rank {me}: {int((img != imgs_ref[me]).any(0).sum())} pixels differ"
        assert st["touched_rows"] == vis[me]
        first, count = lcgs.api.owner_rows(P, world, me)
        for k in KEYS:
            mine = g[k][first:first + count]
            ref = (g_ref[k][perm] if perm is not None else g_ref[k])[first:first + count]  # (row r of a re-ordered scene = file row perm[r])
            assert _rel(mine, ref) <= 1e-4, (me, k, _rel(mine, ref))
            rest = torch.cat([g[k][:first], g[k][first + count:]])
            assert bool((rest == 9.0).all()), (me, k)  # rows of other owners are not touched
        sent += st["bytes_sent"]
        received += st["bytes_received"]
    assert sent == received
    # every on-screen row of every view travels once as (4 + 48) bytes and its gradient once as 48, except an owner's own view
    table_bytes = world * (world - 1) * world * 4
    slack = (sum(vis) // 4 + 1024 * world * world) * 100 if async_steps else 0  # padded messages: <= 1.25 x + 1024 rows each
    assert (sent - table_bytes) % 4 == 0 and sent - table_bytes <= sum(vis) * 100 + slack
    assert sent - table_bytes >= (sum(vis) * 100) * (world - 1) // world // 2  # (views see the ranges unevenly; not an exact law)


def test_owner_step_without_read_back_is_repeated_by_every_rank_when_a_message_was_clipped(lcgs):
    """Three in-process ranks, steps without read-back.  Step 1 (reads back) looks AWAY from the scene: its table is all but
    empty.  Step 2 turns to the scene: every message is sized for step 1's counts and clipped -- the flag is raised on the
    device, max-reduced, and lcgs_owner_step_finish tells EVERY rank to repeat the step (that repetition reads its sizes
    back); the result is the ordinary one.  Step 3 runs without read-back again, sized by step 2's table."""
    import threading

    world = 3
    rng = np.random.default_rng(91)
    P = 45_001
    scene = make_scene(rng, P, log_scale=(-4.0, 0.8))
    cams = [lcgs.get_lookat_cam([-3 * np.cos(a), -0.5 + 3 * np.sin(a), 2.3], [0, 0, 0.5], [0, 0, 1], width=W, height=H)
            for a in np.linspace(0.0, 1.0, world)]
    away = [lcgs.get_lookat_cam([-3 * np.cos(a), -0.5 + 3 * np.sin(a), 2.3], [-9 * np.cos(a), 6 * np.sin(a), 8.0], [0, 0, 1],
                                width=W, height=H) for a in np.linspace(0.0, 1.0, world)]
    dLs = [torch.from_numpy(rng.normal(size=(3, H, W)).astype(np.float32)).to(DEV) for _ in cams]
    bg = (0.1, 0.2, 0.3)
    imgs_ref, g_ref, vis = _reference_views(lcgs, scene, cams, dLs, bg)
    group = lcgs.api.LoopbackGroup(world)
    out, errors = [None] * world, []

    def rank_main(me):
        try:
            side = torch.cuda.Stream(device=DEV)
            with torch.cuda.stream(side):
                r = lcgs.Renderer(lcgs.Context(0, side.cuda_stream))
                act = upload_scene(scene)
                r.bind_scene(*[act[k] for k in KEYS])
                comm = lcgs.Comm(r.ctx, me, world, loopback=group)
                comm.owner_step_set_async(True)
                g = {k: torch.full_like(act[k], 9.0) for k in KEYS}
                img = torch.full((3, H, W), -1.0, device=DEV)
                redos = [comm.owner_step(away, img, dLs[me], g, bg=bg),   # reads back (first step): nearly nothing on screen
                         comm.owner_step(cams, img, dLs[me], g, bg=bg)]   # sized by that: clipped, repeated
                r.ctx.synchronize()
                first = (img.clone(), {k: g[k].clone() for k in KEYS})
                redos.append(comm.owner_step(cams, img, dLs[me], g, bg=bg))  # sized by step 2's table: fits
                r.ctx.synchronize()
                side.synchronize()
                out[me] = (first, img, g, redos)
                comm.close()
        except Exception as e:  # noqa: BLE001
            errors.append((me, repr(e)))

    torch.cuda.synchronize()
    threads = [threading.Thread(target=rank_main, args=(me,)) for me in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    assert all(not t.is_alive() for t in threads), "a rank hangs"
    group.close()
    for me in range(world):
        (img2, g2), img3, g3, redos = out[me]
        assert redos == [0, 1, 0], (me, redos)
        first, count = lcgs.api.owner_rows(P, world, me)
        for img, g in ((img2, g2), (img3, g3)):
            assert torch.equal(img, imgs_ref[me]), me
            for k in KEYS:
                assert _rel(g[k][first:first + count], g_ref[k][first:first + count]) <= 1e-4, (me, k)
