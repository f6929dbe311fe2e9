"""The product's per-splat math header (csrc/kernels/gs_math.hpp) executed on the host vs the CPU oracle,
bit for bit.  The header is __host__ __device__: the very same source is what the HIP kernels run, so an
operation-order slip shows up here, in the container without a GPU, before any GPU time is spent."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, make_scene

HELPER_SRC = os.path.join(ROOT, "tests", "helpers", "host_math.cpp")
HELPER_SO = os.path.join(ROOT, "tests", "helpers", "libhost_math.so")


@pytest.fixture(scope="module")
def host_math(lcgs):
    hdr = os.path.join(ROOT, "luisacomputegaussiansplatting_amd", "csrc", "kernels", "gs_math.hpp")
    if (not os.path.exists(HELPER_SO) or os.path.getmtime(HELPER_SO) < max(os.path.getmtime(HELPER_SRC),
                                                                          os.path.getmtime(hdr))):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-x", "hip", "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC",
                               "-shared", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), HELPER_SRC, "-o",
                               HELPER_SO])
    return C.CDLL(HELPER_SO)


def _run(host_math, lcgs, scene, cam, use_focal=True, scale_modifier=1.0, deg=3):
    P = scene["pos"].shape[0]
    f = lambda *s: np.zeros(s, np.float32)
    out = dict(color=f(P, 3), means_ndc=f(P, 2), depth=f(P), cov=f(P, 3), means_pix=f(P, 2), conic=f(P, 3),
               radii=np.zeros(P, np.int32), tiles=np.zeros(P, np.uint32), rects=np.zeros((P, 4), np.uint32))
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    feat = (deg + 1) ** 2 * 3
    sh = np.ascontiguousarray(scene["sh"][:, :feat])
    host_math.hm_preprocess(C.c_int(P), C.byref(cam), C.c_int(int(use_focal)), C.c_float(scale_modifier), C.c_int(deg),
                            p(scene["pos"]), p(scene["scale"]), p(scene["rotq"]), p(sh), *[p(out[k]) for k in
                            ("color", "means_ndc", "depth", "cov", "means_pix", "conic", "radii", "tiles", "rects")])
    return out


@pytest.mark.parametrize("use_focal", [True, False])
@pytest.mark.parametrize("res", [(1920, 1080), (800, 800), (100, 72)])
def test_gs_math_matches_oracle_bitwise(host_math, lcgs, oracle, use_focal, res):
    rng = np.random.default_rng(100 + res[0])
    P = 20000
    scene = make_scene(rng, P, spread=1.5, log_scale=(-4.0, 1.0))
    # stress: splats at/behind the camera, huge splats, tiny splats, denormal-ish scales
    scene["pos"][:200] = rng.normal(0, 0.4, (200, 3)) + [-3.0, -0.5, 2.3]
    scene["scale"][200:260] *= 200.0
    scene["scale"][260:300] *= 1e-6
    pos, tgt, up = [-3, -0.5, 2.3], [0, 0, 0.5], [0, 0, 1]
    cam = lcgs.get_lookat_cam(pos, tgt, up, width=res[0], height=res[1])
    ocam = oracle.lookat(pos, tgt, up, width=res[0], height=res[1])
    got = _run(host_math, lcgs, scene, cam, use_focal, scale_modifier=1.25)
    color = oracle.sh_process(ocam.position, scene["pos"], scene["sh"])
    m, d, c = oracle.project(scene["pos"], scene["scale"], scene["rotq"], ocam, scale_modifier=1.25, use_focal=use_focal)
    mp, conic, tiles, radii = oracle.allocate_tiles(res[0], res[1], d, m, c, use_focal=use_focal)
    vis = d >= np.float32(0.2)
    assert vis.any() and (~vis).any()
    assert np.array_equal(got["color"], color)
    assert np.array_equal(got["depth"], d)
    assert np.array_equal(got["means_ndc"][vis], m[vis])
    assert np.array_equal(got["cov"][vis], c[vis])
    assert np.array_equal(got["means_pix"][vis], mp[vis])
    assert np.array_equal(got["conic"][vis], conic[vis])
    assert np.array_equal(got["radii"], radii)
    assert np.array_equal(got["tiles"], tiles)


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_degrees(host_math, lcgs, oracle, deg):
    rng = np.random.default_rng(deg)
    scene = make_scene(rng, 3000)
    cam = lcgs.get_lookat_cam([-3, -0.5, 2.3], [0, 0, 0.5], [0, 0, 1], width=64, height=64)
    got = _run(host_math, lcgs, scene, cam, deg=deg)
    feat = (deg + 1) ** 2 * 3
    color = oracle.sh_process(np.array(cam.position, np.float32), scene["pos"], scene["sh"][:, :feat], deg=deg)
    assert np.array_equal(got["color"], color)


def _blend_exp_inputs():
    """Every 257th binary32 of [-6, 0] (the blend's range), a coarser sweep down to the domain's end, the edges."""
    near = np.arange(0x80000000, np.float32(-6.0).view(np.uint32), 257, dtype=np.uint64).astype(np.uint32)
    far = np.arange(np.float32(-6.0).view(np.uint32), np.float32(-86.0).view(np.uint32), 4099,
                    dtype=np.uint64).astype(np.uint32)
    edge = np.array([0.0, -0.0, -86.0, -5.5412636, -1e-30, -1e-45], np.float32).view(np.uint32)
    return np.concatenate([near, far, edge]).view(np.float32)


def test_blend_exp_header_matches_oracle_bitwise_and_is_an_exp(host_math, oracle):
    """The exp of the compositing loop is a build-defined sequence of binary32 operations (gs_math.hpp::blend_exp =
    oracle/lcgs_oracle.c::orc_blend_exp).  The header's host path and the oracle must agree bit for bit, and both must
    be the exponential: <= 2.73 ulp on [-6, 0] (exhaustively measured bound), <= 21 ulp on [-86, 0], exp(0) == 1."""
    x = _blend_exp_inputs()
    got = np.empty_like(x)
    host_math.hm_blend_exp(C.c_longlong(x.size), x.ctypes.data_as(C.c_void_p), got.ctypes.data_as(C.c_void_p))
    ref = oracle.blend_exp(x)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    true = np.exp(x.astype(np.float64))
    ulp = np.ldexp(1.0, np.floor(np.log2(true)).astype(np.int64) - 23)
    err = np.abs(ref.astype(np.float64) - true) / ulp
    near = x >= np.float32(-6.0)
    assert err[near].max() <= 2.73, err[near].max()
    assert err.max() <= 21.0, err.max()
    assert (ref[x == 0] == 1.0).all() and (x == 0).sum() >= 2
    # outside the domain: 0 below -86, NaN passes through (oracle only; the kernels never evaluate it there)
    out = oracle.blend_exp(np.array([-86.5, -1e30, -np.inf, np.nan], np.float32))
    assert out[:3].tolist() == [0.0, 0.0, 0.0] and np.isnan(out[3])
