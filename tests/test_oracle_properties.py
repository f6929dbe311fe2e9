"""Size-independent properties and edge cases of the oracle's stages (the reference tests none of this:
test/test_radix_sort_alignment.cpp:10-14 is a placeholder)."""
import numpy as np

from conftest import make_scene


def test_sort_is_stable_and_sorted(oracle):
    rng = np.random.default_rng(11)
    n = 50000
    keys = (rng.integers(0, 37, n).astype(np.uint64) << np.uint64(32)) | rng.integers(0, 5, n).astype(np.uint64)
    vals = np.arange(n, dtype=np.uint32)
    ks, vs = oracle.sort_pairs(keys, vals)
    assert np.all(ks[1:] >= ks[:-1])
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(vs, vals[order]) and np.array_equal(ks, keys[order])


def test_sort_edge_cases(oracle):
    ks, vs = oracle.sort_pairs(np.zeros(0, np.uint64), np.zeros(0, np.uint32))
    assert ks.size == 0 and vs.size == 0
    ks, vs = oracle.sort_pairs(np.array([5], np.uint64), np.array([9], np.uint32))
    assert ks[0] == 5 and vs[0] == 9
    k = np.array([2**64 - 1, 0, 2**63, 1], np.uint64)
    ks, vs = oracle.sort_pairs(k, np.arange(4, dtype=np.uint32))
    assert list(vs) == [1, 3, 2, 0]


def test_inclusive_sum_wraps_like_u32(oracle):
    x = np.array([2**31, 2**31, 5], np.uint32)
    assert list(oracle.inclusive_sum(x)) == [2**31, 0, 5]
    assert oracle.inclusive_sum(np.zeros(0, np.uint32)).size == 0


def test_ranges_partition_the_list(oracle):
    rng = np.random.default_rng(5)
    tiles = np.sort(rng.integers(0, 40, 3000)).astype(np.uint64)
    keys = (tiles << np.uint64(32)) | rng.integers(0, 2**32, 3000).astype(np.uint64)
    ranges = oracle.get_ranges(keys, 41)
    for t in range(41):
        s, e = ranges[t]
        idx = np.nonzero(tiles == t)[0]
        if idx.size == 0:
            assert (s, e) == (0, 0)
        else:
            assert s == idx[0] and e == idx[-1] + 1


def test_near_cull_leaves_buffers_untouched(oracle):
    """gs_projector/shader.cpp:121 returns before any write."""
    rng = np.random.default_rng(1)
    scene = make_scene(rng, 64)
    cam = oracle.lookat([0, 0, 0], [0, 0, 1], [0, 1, 0], width=64, height=64)
    scene["pos"][:, 2] = np.linspace(-1, 1, 64)
    init = (np.full((64, 2), 7, np.float32), np.full(64, 9, np.float32), np.full((64, 3), 5, np.float32))
    m, d, c = oracle.project(scene["pos"], scene["scale"], scene["rotq"], cam, init=init)
    culled = scene["pos"][:, 2] < np.float32(0.2)
    assert culled.any() and (~culled).any()
    assert (m[culled] == 7).all() and (d[culled] == 9).all() and (c[culled] == 5).all()
    assert (d[~culled] == scene["pos"][~culled, 2]).all()


def test_empty_scene_and_all_culled(oracle):
    cam = oracle.lookat([0, 0, 0], [0, 0, 1], [0, 1, 0], width=32, height=32)
    rng = np.random.default_rng(2)
    scene = make_scene(rng, 16)
    scene["pos"][:, 2] = -5.0  # everything behind the camera
    r = oracle.render(scene, cam, bg=(0.2, 0.3, 0.4))
    assert r["num_rendered"] == 0 and (r["img"] == 0).all()  # image left untouched (impl.cpp:109)
    assert (r["radii"] == 0).all()


def test_render_is_linear_in_colour_and_bg(oracle):
    """img = bg*T + sum_i w_i c_i: linear in (bg, colours) for fixed geometry."""
    rng = np.random.default_rng(4)
    scene = make_scene(rng, 400)
    cam = oracle.lookat([-3, -0.5, 2.3], [0, 0, 0.5], [0, 0, 1], width=64, height=48)
    W, H = 64, 48
    m, d, c = oracle.project(scene["pos"], scene["scale"], scene["rotq"], cam)
    mp, conic, tiles, radii = oracle.allocate_tiles(W, H, d, m, c)
    offs = oracle.inclusive_sum(tiles)
    k, v = oracle.copy_with_keys(W, H, mp, offs, radii, d)
    ks, vs = oracle.sort_pairs(k, v)
    ranges = oracle.get_ranges(ks, 4 * 3)
    c1 = rng.random((400, 3)).astype(np.float32)
    c2 = rng.random((400, 3)).astype(np.float32)
    f = lambda col, bg: oracle.render_forward(W, H, bg, ranges, vs, mp, conic, scene["opacity"], col)[0]
    a = f(c1, [0.1, 0.2, 0.3]) + f(c2, [0.3, 0.1, 0.0])
    b = f(c1 + c2, [0.4, 0.3, 0.3])
    assert np.allclose(a, b, atol=2e-6)
