"""What the render-backward's waves actually walk on the bench frame (measuring aid, round 5).

Needs a library built with -DLCGS_BWD_STATS (make OUT=... OBJDIR=... EXTRA=-DLCGS_BWD_STATS) in the package's place:
    cp gpurun_in/liblcgs_bwdstats.so luisacomputegaussiansplatting_amd/liblcgs_hip.so && python tools/gpu/bwd_stats.py
Prints the counters of ONE forward(keep_state) + backward of the mip360_bicycle stand-in at 1080p (view 0 and, with
--views N, the first N of the eight C5 views): (entry, strip) pairs walked, how many pass the wave-level candidate
test, lane occupancy of those that do, the distribution of per-strip list lengths, and what a rotation ("systolic")
walk would add in fill + drain steps (63 per strip and tile).
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import luisacomputegaussiansplatting_amd as L  # noqa: E402
from bench import P_BICYCLE, view_pose  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--splats", type=int, default=P_BICYCLE)
    ap.add_argument("--views", type=int, default=1)
    ap.add_argument("--res", default="1920x1080")
    args = ap.parse_args()
    W, H = (int(x) for x in args.res.split("x"))
    lib = L.load_library()
    if not hasattr(lib, "lcgs_debug_bwd_stats"):
        sys.exit("this liblcgs_hip.so was not built with -DLCGS_BWD_STATS")
    side = torch.cuda.Stream()
    torch.cuda.set_stream(side)
    r = L.Renderer(L.Context(0, side.cuda_stream))
    r.upload_scene(L.synth_scene(1, 2001, args.splats))
    d = r.scene_tensors()
    g = {k: torch.zeros_like(d[k]) for k in ("pos", "scale", "rotq", "sh", "opacity")}
    img = torch.zeros(3, H, W, device="cuda")
    dL = torch.randn(3, H, W, device="cuda")
    out = (C.c_ulonglong * 32)()
    for k in range(args.views):
        cam = L.get_lookat_cam(*view_pose(k), width=W, height=H)
        r.forward(cam, img, keep_state=True, sync=True)  # sizes the buffers
        r.forward(cam, img, keep_state=True, sync=True)
        torch.cuda.synchronize()
        lib.lcgs_debug_bwd_stats(out, 1)
        r.backward(dL, g["pos"], g["scale"], g["rotq"], g["sh"], g["opacity"])
        r.ctx.synchronize()
        torch.cuda.synchronize()
        lib.lcgs_debug_bwd_stats(out, 1)
        s = [int(x) for x in out]
        st = r.frame_stats()
        walked, passed, lanes, strips, rounds, tiles, staged, fetched = s[:8]
        hist = s[8:24]
        rec = {
            "view": k, "num_pairs": int(st["num_pairs"]) if isinstance(st, dict) else None,
            "tiles": tiles, "staging_rounds": rounds, "list_entries_staged": staged, "entries_fetched(mask!=0)": fetched,
            "entry_strip_walked": walked, "pass_wave_test": passed, "pass_fraction": round(passed / max(walked, 1), 4),
            "lanes_blending": lanes, "lane_occupancy_of_passed": round(lanes / max(64 * passed, 1), 4),
            "strips_with_work": strips, "mean_walked_per_strip": round(walked / max(strips, 1), 1),
            "rotation_fill_drain_steps(63/strip)": 63 * strips,
            "rotation_overhead_vs_walked": round(63 * strips / max(walked, 1), 4),
            "strip_length_histogram(<=2^b)": {str(1 << b): hist[b] for b in range(16) if hist[b]},
            # sub-block streams: if a wave walked one independent entry stream per sub-block of its strip (lanes of a
            # sub-block see only the entries that blend into it), its iteration count is the LONGEST stream's length
            "entries_blending_anywhere": s[28],
            "sum_longest_stream": {"8x4_halves": s[24], "16x2_halves": s[25], "4x4_blocks": s[26], "16x1_rows": s[27]},
            "longest_stream_vs_blending": {k: round(v / max(s[28], 1), 4) for k, v in
                                           (("8x4_halves", s[24]), ("16x2_halves", s[25]), ("4x4_blocks", s[26]), ("16x1_rows", s[27]))},
            "sum_streams": {"8x4_halves": s[29], "4x4_blocks": s[30], "16x1_rows": s[31]},
        }
        print(json.dumps(rec))


if __name__ == "__main__":
    main()
