"""Camera math known answers -- the values of the reference's own test/test_camera.cpp:48-144, applied to
the CPU oracle AND to the product's host camera code (both restate lcgs/include/lcgs/util/camera.h)."""
import json
import math
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def kat(golden_dir):
    with open(os.path.join(golden_dir, "camera_kat.json")) as f:
        return json.load(f)


def _check_cam(front, up, right, l2w, w2l, proj_fn, kat):
    eps = kat["tolerance"]
    assert np.allclose(front, kat["lookat"]["front"], atol=eps)  # test_camera.cpp:61-63
    assert abs(np.dot(up, right)) < eps and abs(np.dot(up, front)) < eps and abs(np.dot(right, front)) < eps
    local = np.array(kat["l2w_point"]["local"], np.float32)
    world = l2w @ local
    assert np.allclose(world[:3], kat["l2w_point"]["world"], atol=eps)  # :74-80
    assert np.allclose(w2l @ world, local, atol=eps)  # :82-86
    p = kat["proj"]
    tanx = math.tan(math.radians(p["fovx_deg"]) / 2)
    tany = math.tan(math.radians(p["fovy_deg"]) / 2)
    proj = proj_fn(tanx, tany, p["near"], p["far"])
    near = proj @ np.array([0, 0, p["near"], 1], np.float32)
    far = proj @ np.array([0, 0, p["far"], 1], np.float32)
    assert abs(near[2] / near[3]) < eps  # :103-106
    assert abs(far[2] / far[3] - 1.0) < eps  # :108-111
    pt = proj @ np.array(p["point"], np.float32)
    assert abs(pt[0] / pt[3] - p["point"][0] / tanx / 2.0) < eps  # :113-118
    assert abs(pt[1] / pt[3] - p["point"][1] / tany / 2.0) < eps


def test_oracle_camera_kat(oracle, kat):
    cam = oracle.lookat(kat["lookat"]["pos"], kat["lookat"]["target"], kat["lookat"]["up"])
    assert list(cam.position) == kat["lookat"]["pos"]
    _check_cam(np.array(cam.front), np.array(cam.up), np.array(cam.right), oracle.local_to_world(cam),
               oracle.world_to_local(cam), oracle.projection, kat)
    s = kat["special"]
    cam2 = oracle.lookat(s["pos"], s["target"], s["up"])
    # test_camera.cpp:129 expects (1,0,0); camera.h:79 (cross(front, world_up)) gives (-1,0,0) and the l2w
    # known answer above pins that handedness -- the restatement follows the reference CODE.
    assert np.allclose(list(cam2.right), s["right_per_reference_code"], atol=kat["tolerance"])
    assert np.allclose(list(cam2.right), -np.array(s["right_expected_by_reference_test"]), atol=kat["tolerance"])
    wp = oracle.local_to_world(cam2) @ np.array([0, 0, 1, 1], np.float32)
    assert np.allclose(wp[:3], np.array(cam2.position) + np.array(cam2.front), atol=kat["tolerance"])  # :131-141


def test_product_camera_kat(lcgs, kat):
    cam = lcgs.get_lookat_cam(kat["lookat"]["pos"], kat["lookat"]["target"], kat["lookat"]["up"])
    _check_cam(np.array(cam.front), np.array(cam.up), np.array(cam.right), lcgs.local_to_world_matrix(cam),
               lcgs.world_to_local_matrix(cam), lcgs.projection_matrix, kat)
    s = kat["special"]
    cam2 = lcgs.get_lookat_cam(s["pos"], s["target"], s["up"])
    assert np.allclose(list(cam2.right), s["right_per_reference_code"], atol=kat["tolerance"])
    assert (cam.fov, cam.aspect_ratio, cam.width, cam.height) == (60.0, 1.0, 512, 512)  # camera.h:21-24


def test_product_camera_matches_oracle_bitwise(lcgs, oracle):
    rng = np.random.default_rng(7)
    for _ in range(20):
        pos, target, up = rng.normal(size=(3, 3)).astype(np.float32)
        a = lcgs.get_lookat_cam(pos, target, up, width=1920, height=1080)
        b = oracle.lookat(pos, target, up, width=1920, height=1080)
        for k in ("position", "front", "up", "right"):
            assert list(getattr(a, k)) == list(getattr(b, k))
        assert a.aspect_ratio == b.aspect_ratio
        assert np.array_equal(lcgs.world_to_local_matrix(a), oracle.world_to_local(b))
        assert np.array_equal(lcgs.local_to_world_matrix(a), oracle.local_to_world(b))
