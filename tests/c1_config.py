"""BASELINE config C1 at size: the nerf_blender_lego stand-in (synth_object, seed 1001, 300 000 splats) as a 62-property
binary PLY -> read_gs_ply -> one 800x800 frame with `--world blender` -> flipped RGB8 PNG, i.e. the flow of the
reference's app/main.cpp:166-339, on the CPU through the oracle (the reference has no CPU path of its own:
CMakeLists.txt:23).  Test infrastructure (imports the oracle); run as a script it writes the PLY and the PNG:

    python tests/c1_config.py [out_dir]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

P_LEGO, SEED_LEGO, RES = 300_000, 1001, (800, 800)
# the reference's lego look-at (app/main.cpp:195-197; `lcgs-app --pose lego`) with the --world blender up vector (:199-202)
POSE = ([-3.0, -0.5, 2.3], [0.0, 0.0, 0.5], [0.0, 0.0, 1.0])


def write_stand_in_ply(lcgs, path, count=P_LEGO, seed=SEED_LEGO, kind=0):
    """The stand-in as the INRIA-layout PLY the reference's loader expects (raw columns: log-scales, opacity logits,
    f_dc / channel-major f_rest).  Every consumer (oracle side and GPU side) then reads the same file."""
    s = lcgs.synth_scene(kind, seed, count)
    sh = s["sh"].reshape(count, 16, 3)
    f_rest = np.ascontiguousarray(sh[:, 1:, :].transpose(0, 2, 1).reshape(count, 45))  # app/gaussians.cpp:120-135
    op = s["opacity"].astype(np.float64)
    lcgs.write_ply_raw(path, s["pos"], np.ascontiguousarray(sh[:, 0, :]), f_rest,
                       np.log(op / (1.0 - op)).astype(np.float32), np.log(s["scale"].astype(np.float64)).astype(np.float32),
                       s["rotq"])
    return path


def render_cpu(lcgs, oracle, ply_path, res=RES, pose=POSE):
    """read_gs_ply (the product's host reader, pinned against the reference's happly) -> oracle frame -> RGB8 rows in
    PNG order (app/main.cpp:323-335).  Returns (rgb8 HxWx3, oracle result dict)."""
    scene = lcgs.read_gs_ply(ply_path)
    scene.pop("sh_degree", None)
    out = oracle.render(scene, oracle.lookat(*pose, width=res[0], height=res[1]), ambig_eps=1e-5)
    return oracle.image_to_rgb8(out["img"]), out


def thumbnail(rgb, cells=100):
    """cells x cells block means of an RGB8 image (the committed regression pin of the C1 frame)."""
    H, W, _ = rgb.shape
    assert H % cells == 0 and W % cells == 0
    return rgb.reshape(cells, H // cells, cells, W // cells, 3).astype(np.float64).mean(axis=(1, 3)).astype(np.float32)


if __name__ == "__main__":
    import time

    from PIL import Image

    import luisacomputegaussiansplatting_amd as L
    from oracle import Oracle

    out_dir = sys.argv[1] if len(sys.argv) > 1 else "out"
    os.makedirs(out_dir, exist_ok=True)
    ply = write_stand_in_ply(L, os.path.join(out_dir, "lego_stand_in.ply"))
    o = Oracle("f32")
    t0 = time.perf_counter()
    rgb, res = render_cpu(L, o, ply)
    dt = time.perf_counter() - t0
    png = os.path.join(out_dir, "lego_stand_in_cpu.png")
    Image.fromarray(rgb).save(png)
    print(f"num_gaussians: {P_LEGO}\nnum_rendered: {res['num_rendered']}\nexp time: {dt * 1e3:.1f} ms on {o.get_threads()} "
          f"host threads (PLY read included)\nresult saved in {png}")
    if len(sys.argv) > 2 and sys.argv[2] == "--update-golden":
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", "c1_lego_800_thumb.npz"), thumb=thumbnail(rgb),
                            num_rendered=np.int64(res["num_rendered"]))
