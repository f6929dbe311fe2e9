"""BASELINE config C1 at size (CPU only, `-m "not gpu"`): nerf_blender_lego stand-in, 300 000 splats, 800x800,
PLY -> frame -> PNG through the oracle (tests/c1_config.py; flow of app/main.cpp:166-339)."""
import os

import numpy as np

import c1_config as c1


def test_c1_lego_stand_in_on_the_cpu_path(lcgs, oracle, golden_dir, tmp_path):
    from PIL import Image

    ply = c1.write_stand_in_ply(lcgs, str(tmp_path / "lego_stand_in.ply"))
    assert os.path.getsize(ply) > c1.P_LEGO * 62 * 4  # 62 float properties per vertex (app/gaussians.cpp:93-135)
    rgb, res = c1.render_cpu(lcgs, oracle, ply)
    W, H = c1.RES
    assert rgb.shape == (H, W, 3) and rgb.dtype == np.uint8
    png = str(tmp_path / "lego_stand_in_cpu.png")
    Image.fromarray(rgb).save(png)
    assert np.array_equal(np.array(Image.open(png)), rgb)
    g = np.load(os.path.join(golden_dir, "c1_lego_800_thumb.npz"))
    assert res["num_rendered"] == int(g["num_rendered"]) and res["num_rendered"] > 1_000_000
    # the committed 100x100 block means of the frame (an 8-bit level of slack: libm exp may differ between hosts)
    assert np.abs(c1.thumbnail(rgb) - g["thumb"]).max() <= 1.0
    # module.cpp:31-35: the last tile row / column is never rasterised; PNG row 0 is image row H-1 (app/main.cpp:331)
    assert not rgb[:16].any() and not rgb[:, W - 16:].any()
    assert rgb[H // 2 - 100:H // 2 + 100, W // 2 - 100:W // 2 + 100].min(axis=2).mean() > 20  # the object is in view
