"""pytest configuration: the `gpu` marker and shared fixtures.

`-m "not gpu"` runs here (no GPU): oracle vs golden vectors, host logic, ABI surface, gloo multi-process.
`-m gpu` runs on an MI355X: parity of the HIP path against the oracle, through the C ABI.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device")


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _side_stream_under_the_graph_hook():
    """LCGS_GRAPH=1 (tests/test_gpu_tuning_hooks.py re-runs suites under it): the legacy NULL stream cannot be captured, and
    a Context created without a stream takes torch's CURRENT one -- so under the hook the whole session runs on a side
    stream; otherwise the hook would silently exercise the eager path."""
    if os.environ.get("LCGS_GRAPH") == "1" and _has_gpu():
        import torch

        torch.cuda.set_stream(torch.cuda.Stream(device=0))
    yield


@pytest.fixture(scope="session")
def oracle():
    from oracle import Oracle

    return Oracle("f32")


@pytest.fixture(scope="session")
def oracle64():
    from oracle import Oracle

    return Oracle("f64")


@pytest.fixture(scope="session")
def lcgs():
    import luisacomputegaussiansplatting_amd as L

    if not os.path.exists(L.library_path()):
        L.build_library()
    L.load_library()
    return L


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# BASELINE.json's scenes.  The real PLYs are release assets of the reference (README.md:26-29) and cannot be fetched offline;
# a box that has them points LCGS_<NAME>_PLY at the file and every full-size test (and bench.py) runs on the real scene
# without a code change -- `data: real` instead of the synthetic stand-in of SURVEY 8(d).
BASELINE_SCENES = {  # name -> (env var, stand-in kind, seed, splats)
    "lego": ("LCGS_LEGO_PLY", 0, 1001, 300_000),
    "chair": ("LCGS_CHAIR_PLY", 0, 1002, 300_000),
    "bicycle": ("LCGS_BICYCLE_PLY", 1, 2001, 6_131_954),
    "garden": ("LCGS_GARDEN_PLY", 1, 2002, 5_834_784),
}


def baseline_scene(L, name):
    """(scene dict in the activated layout of read_gs_ply, "real" | "synthetic") for one of BASELINE.json's scenes"""
    env, kind, seed, P = BASELINE_SCENES[name]
    path = os.environ.get(env, "")
    if path and os.path.exists(path):
        scene = L.read_gs_ply(path)
        scene.pop("sh_degree", None)
        print(f"[scene] {name}: REAL {path} ({scene['pos'].shape[0]} splats)")
        return scene, "real"
    return L.synth_scene(kind, seed, P), "synthetic"


def make_scene(rng, P, spread=0.6, center=(0.0, 0.0, 0.5), log_scale=(-3.8, 0.6)):
    """Small random scene in the activated layout of read_gs_ply (tests only)."""
    pos = (rng.normal(0, spread, (P, 3)) + np.asarray(center)).astype(np.float32)
    scale = np.exp(rng.normal(log_scale[0], log_scale[1], (P, 3))).astype(np.float32)
    rotq = rng.normal(size=(P, 4)).astype(np.float32)
    rotq /= np.linalg.norm(rotq, axis=1, keepdims=True)
    sh = rng.normal(0, 0.3, (P, 48)).astype(np.float32)
    sh[:, :3] += 0.5
    opacity = (1 / (1 + np.exp(-rng.normal(0, 2, P)))).astype(np.float32)
    return {"pos": pos, "scale": scale, "rotq": rotq.astype(np.float32), "sh": sh, "opacity": opacity}
