"""`python bench.py --sweep` -- how the headline figures move with the WORKLOAD (round 6; outside the bench line).

Every figure of the bench line is one operating point: the bicycle stand-in at scale_modifier 1, 1920x1080 (V = 2.39 M
on-screen splats, 12.98 M reference pairs).  A real scene of the same splat count can carry 2-3 x the pairs.  This sweep moves
the pair count WITHOUT a new scene: scale_modifier (the reference's own frame parameter, app/main.cpp:269) scales every
footprint, i.e. the pairs roughly quadratically; two stand-ins (bicycle, garden) x three resolutions x five modifiers.
Per point: the frame's counts (visible splats V, reference pairs, sorted pairs L, list entries the renderer's workgroups
stage), forward frames/s (three list-granularity policies: the default's decision, forced per-tile, forced per-block),
per-stage times, forward+backward Msplats/s, and whether the pair workspace had to grow inside a timed loop.
Then a non-negative least-squares fit  t = a + b P + c V + d L + e staged + f examined + g pixels  of the forward time, the points that
sit >= 1.3 x off it, and the points where the default's granularity decision loses to a forced one.
Two small-raster points are also rendered by the CPU oracle and compared bit for bit (the extremes of the modifier range).

Writes profiles/r06_workload_sweep.json (or --sweep-out)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SCENES = {"bicycle": (1, 2001, 6_131_954, ([-3, -0.5, 2.3], [0, 0, 0.5], [0, -1, 0])),   # app/main.cpp:195-197
          "garden": (1, 2002, 5_834_784, ([-3, -0.5, 3.3], [0, 3, 0.5], [0, -1, -1]))}    # app/main.cpp:191-193
RESOLUTIONS = ((800, 800), (1920, 1080), (3840, 2160))
MODIFIERS = (0.5, 1.0, 1.5, 2.0, 3.0)
KEYS = ("pos", "scale", "rotq", "sh", "opacity")


def _timed(torch, fn, n, reps=3):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            fn(i)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / n)
    return float(np.median(out))


def run(out_path, frames=30, steps=12, quick=False):
    import torch

    import luisacomputegaussiansplatting_amd as L
    from bench import renderer_entries

    dev = "cuda:0"
    points, oracle_checks = [], []
    for name, (kind, seed, P, pose) in SCENES.items():
        if quick:
            P = P // 20
        scene = L.synth_scene(kind, seed, P)
        rs = {}
        for mode in ("auto", "0", "1"):  # the list granularity is a per-context policy read at creation (test hook)
            if mode == "auto":
                os.environ.pop("LCGS_COARSE_LISTS", None)
            else:
                os.environ["LCGS_COARSE_LISTS"] = mode
            rs[mode] = L.Renderer(L.Context(0))
            rs[mode].upload_scene(scene)
        os.environ.pop("LCGS_COARSE_LISTS", None)
        grads = {k: torch.zeros(P, w, device=dev) if w > 1 else torch.zeros(P, device=dev)
                 for k, w in (("pos", 3), ("scale", 3), ("rotq", 4), ("sh", 48), ("opacity", 1))}
        for (W, H) in RESOLUTIONS:
            img = torch.zeros(3, H, W, device=dev)
            dL = torch.randn(3, H, W, device=dev)
            cam = L.get_lookat_cam(*pose, width=W, height=H)
            for sm in MODIFIERS:
                pt = {"scene": name, "P": P, "width": W, "height": H, "scale_modifier": sm}
                fps = {}
                for mode, r in rs.items():
                    grew = False
                    for _ in range(3):  # synchronising frames: buffers sized, hints and the granularity decision settled
                        r.forward(cam, img, scale_modifier=sm, sync=True)
                    if mode == "auto":
                        st = r.frame_stats()
                        pt.update(V=st["num_visible"], L_ref=st["num_rendered"], L_sorted=st["num_pairs"],
                                  list_shift=st["list_shift"], tiles=st["num_tiles"],
                                  staged=renderer_entries(torch, r, st, W, H, dev))
                        r.set_profiling(True)
                        r.forward(cam, img, scale_modifier=sm, sync=True)
                        pt["stage_ms"] = {k: round(v, 4) for k, v in r.stage_times().items()}
                        r.set_profiling(False)
                        r.forward(cam, img, scale_modifier=sm, sync=True)
                    try:
                        t = _timed(torch, lambda i: r.forward(cam, img, scale_modifier=sm, sync=False), frames)
                        r.ctx.synchronize()
                    except L.LcgsError as e:  # the sticky overflow record: a frame of the timed loop was truncated
                        grew = True
                        pt.setdefault("errors", []).append(f"{mode}: {e}"[:200])
                        r.forward(cam, img, scale_modifier=sm, sync=True)
                        t = _timed(torch, lambda i: r.forward(cam, img, scale_modifier=sm, sync=False), frames)
                        r.ctx.synchronize()
                    fps[mode] = 1.0 / t
                    if grew:
                        pt["pair_workspace_grew_inside_a_timed_loop"] = True
                pt["forward_ms"] = round(1e3 / fps["auto"], 4)
                pt["forward_fps"] = {"default": round(fps["auto"], 1), "per_tile_lists": round(fps["0"], 1),
                                     "per_block_lists": round(fps["1"], 1)}
                best = max(fps["0"], fps["1"])
                pt["default_vs_best_forced"] = round(fps["auto"] / best, 4)
                r = rs["auto"]
                # forward + backward, dense rows (keep_state frames list per tile)
                r.forward(cam, img, scale_modifier=sm, keep_state=True, sync=True)
                pt["L_keep_state"] = r.frame_stats()["num_pairs"]
                # what the compositing loop EXAMINES: per pixel the list position of its last contributor (the kept state),
                # summed -- small footprints occlude less, so a frame of fewer pairs can walk more of them
                ncon = torch.zeros(H, W, dtype=torch.int32, device=dev)
                r.last_state(None, ncon)
                pt["examined"] = int(ncon.long().sum().item())
                del ncon

                def step(i):
                    r.forward(cam, img, scale_modifier=sm, keep_state=True, sync=False)
                    r.backward(dL, *[grads[k] for k in KEYS])

                t = _timed(torch, step, steps)
                r.ctx.synchronize()
                pt["fwd_bwd_ms"] = round(t * 1e3, 4)
                pt["fwd_bwd_msplats"] = round(P / t / 1e6, 1)
                points.append(pt)
                print(f"[sweep] {name} {W}x{H} sm {sm}: V {pt['V']} L_ref {pt['L_ref']} sorted {pt['L_sorted']} "
                      f"(shift {pt['list_shift']}) -> {pt['forward_fps']} fps, fwd+bwd {pt['fwd_bwd_msplats']} Msplats/s",
                      file=sys.stderr, flush=True)
        # ---- the two extreme small-raster points against the CPU oracle, bit for bit
        if name == "bicycle":
            from oracle import Oracle

            o = Oracle("f32")
            o.set_threads(0)
            W, H = RESOLUTIONS[0]
            cam = L.get_lookat_cam(*pose, width=W, height=H)
            ocam = o.lookat(*pose, width=W, height=H)
            img = torch.zeros(3, H, W, device=dev)
            for sm in (MODIFIERS[0], MODIFIERS[-1]):
                n = rs["auto"].forward(cam, img, scale_modifier=sm, sync=True)
                ref = o.render(scene, ocam, scale_modifier=sm)
                oracle_checks.append({"scene": name, "width": W, "height": H, "scale_modifier": sm,
                                      "num_rendered": int(n), "num_rendered_equal": bool(n == ref["num_rendered"]),
                                      "bit_identical": bool(np.array_equal(img.cpu().numpy(), ref["img"]))})
        del rs, grads, scene
        torch.cuda.empty_cache()

    # ---- the fit: forward time against what the frame has to touch
    from scipy.optimize import nnls

    feats = ("const", "P", "V", "L_sorted", "staged", "examined", "pixels")
    A = np.array([[1.0, p["P"], p["V"], p["L_sorted"], p["staged"], p["examined"], p["width"] * p["height"]] for p in points])
    y = np.array([p["forward_ms"] for p in points])
    scale = A.max(axis=0)
    coef, _ = nnls(A / scale, y)
    coef = coef / scale
    pred = A @ coef
    for p, q in zip(points, pred):
        p["model_ms"] = round(float(q), 4)
        p["measured_over_model"] = round(float(p["forward_ms"] / q), 3)
    model = {"form": "forward_ms = a + b P + c V + d L_sorted + e staged + f examined + g pixels  (non-negative least squares; "
                     "examined = sum over pixels of the list position of the last contributor)",
             "coefficients": {f: float(c) for f, c in zip(feats, coef)},
             "per_million": {f: round(float(c) * 1e6, 5) for f, c in zip(feats[1:], coef[1:])},
             "constant_ms": round(float(coef[0]), 4),
             "rms_relative_residual": round(float(np.sqrt(np.mean((y / pred - 1.0) ** 2))), 4),
             "worst_measured_over_model": round(float((y / pred).max()), 3),
             "best_measured_over_model": round(float((y / pred).min()), 3)}
    findings = {
        "points_off_the_model_by_1p3": [{k: p[k] for k in ("scene", "width", "height", "scale_modifier", "forward_ms", "model_ms",
                                                           "measured_over_model")}
                                        for p in points if p["measured_over_model"] >= 1.3 or p["measured_over_model"] <= 1 / 1.3],
        "default_granularity_loses_by_more_than_3pct": [{k: p[k] for k in ("scene", "width", "height", "scale_modifier", "L_sorted",
                                                                          "list_shift", "forward_fps", "default_vs_best_forced")}
                                                        for p in points if p["default_vs_best_forced"] < 0.97],
        "pair_workspace_grew_inside_a_timed_loop": [{k: p[k] for k in ("scene", "width", "height", "scale_modifier")}
                                                    for p in points if p.get("pair_workspace_grew_inside_a_timed_loop")],
    }
    base = next(p for p in points if p["scene"] == "bicycle" and (p["width"], p["height"]) == (1920, 1080) and p["scale_modifier"] == 1.0)
    table = []
    for p in points:
        if p["scene"] == "bicycle" and (p["width"], p["height"]) == (1920, 1080):
            table.append({"scale_modifier": p["scale_modifier"], "pairs_vs_headline": round(p["L_ref"] / base["L_ref"], 2),
                          "forward_fps": p["forward_fps"]["default"], "fwd_bwd_msplats": p["fwd_bwd_msplats"]})
    out = {"what": __doc__.split("\n\n")[0], "quick": quick, "frames_per_timing": frames, "steps_per_timing": steps,
           "points": points, "model": model, "findings": findings, "bicycle_1080p_by_pair_count": table,
           "oracle_checks": oracle_checks}
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    with open(out_path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({"model": model, "findings": findings, "bicycle_1080p_by_pair_count": table,
                      "oracle_checks": oracle_checks}, indent=1))
    return out


if __name__ == "__main__":
    run(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r06_workload_sweep.json"),
        quick=os.environ.get("LCGS_SWEEP_QUICK") == "1")
