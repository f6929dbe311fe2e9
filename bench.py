#!/usr/bin/env python3
"""bench.py -- forward fps @1080p on the mip360_bicycle stand-in (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one view: the fused forward frame (SH colour -> projection ->
tile keys -> sort -> per-tile compositing) of a 6,131,954-splat scene at 1920x1080, scene resident in HBM.
With N > 1 every rank renders its own view of the (replicated) scene -- the path shards across views with no
data-path collective (weak scaling); `value` = frames all ranks rendered / max-over-ranks time.

The real mip360_bicycle_30000.ply is a GitHub release asset of the reference and is not available offline;
the workload is the deterministic synthetic stand-in `synth_unbounded` (seed 2001) with the same splat count
unless --ply points at the real file.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import math
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

P_BICYCLE = 6_131_954
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
# What a plain read stream actually sustains on this part: 16- and 12-byte-per-lane streams over a 1 GiB table, measured
# with tools/microbench/fetch_calib.hip (profiles/r02_fetch_calibration.txt: 6.32 / 6.55 TB/s).  Reported BESIDE the
# spec peak, never instead of it.
HBM_ATTAINABLE_GBS = 6550.0


def kernel_sources_hash() -> str:
    """sha256 over the device sources (csrc/kernels/*) and the ABI files that order the launches (csrc/abi_*): profile
    artefacts under profiles/ carry the hash of the sources they were measured on, and a figure read back from them is only
    used while it still matches -- a changed kernel must not keep reporting last round's counters."""
    import glob
    import hashlib

    h = hashlib.sha256()
    base = os.path.join(ROOT, "luisacomputegaussiansplatting_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(base, "kernels", "*")) + glob.glob(os.path.join(base, "abi_*"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def library_hash() -> str:
    """sha256 of the liblcgs_hip.so this process loads: the profile artefacts record it too, so counters are tied to the
    binary they were measured on, not only to the sources."""
    import hashlib

    import luisacomputegaussiansplatting_amd as L

    try:
        return hashlib.sha256(open(L.library_path(), "rb").read()).hexdigest()[:16]
    except OSError:
        return "unreadable"


def read_sq_file(path):
    """profiles/rNN_pmc_sq*.txt (profiles/pmc_summary.py): ({kernel: {counter: per-launch mean}}, header dict)"""
    import ast

    kernels, head = {}, {}
    for ln in open(path).read().splitlines():
        if ln.startswith("#"):
            parts = ln[1:].split()
            if len(parts) >= 2:
                head[parts[0]] = parts[-1]
        elif "{" in ln:
            kernels[ln.split()[0]] = ast.literal_eval(ln[ln.index("{"):])
    return kernels, head


def valu_issue_from_counters(c):
    """How busy a kernel keeps the SIMDs' vector ALUs, from SQ counters of ONE rocprofv3 pass (clock-independent: both sides
    are cycle counts of the same launches).  Kernel cycles = GRBM_GUI_ACTIVE / 8 (rocprofv3 sums the 8 XCDs).
    SQ_ACTIVE_INST_VALU counts QUAD-cycles (MI355X_MICROARCH.md) in which a wave has a VALU instruction in issue; measured
    here it equals SQ_INSTS_VALU to 0.1 % on the forward renderer -- one quad-cycle per wave64 instruction -- while the
    32-lane ALU is occupied 2 cycles by a full-rate instruction (v_add / v_mul / v_mov: the guide's 157 TFLOP/s f32 peak),
    so two waves' instructions overlap and `active` can reach 2 per SIMD cycle.
      frac = 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x kernel cycles) / 2
    is therefore the MEASURED share of the VALU's peak instruction rate -- a lower bound of how busy the ALU is, because
    instructions that are not full-rate (v_fma with three register operands, v_cmp, v_cndmask, v_min / v_max: 4.1-4.4
    cycles in tools/microbench/issue_rates.hip; transcendental and permlane 8) hold it longer than the 2 cycles assumed.
    The wave-cycle split (issuing / ready but not issued / parked at a wait or barrier) is measured too."""
    need = ("SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES")
    if not c or any(k not in c or not c[k] for k in need):
        return None
    cycles = c["GRBM_GUI_ACTIVE"] / 8.0
    active = 4.0 * c["SQ_ACTIVE_INST_VALU"] / (1024.0 * cycles)
    out = {"frac": round(active / 2.0, 4), "valu_active_per_simd_cycle": round(active, 4), "kernel_cycles": round(cycles),
           "basis": "measured: SQ_ACTIVE_INST_VALU (quad-cycles) over GRBM_GUI_ACTIVE, at the 2-cycle full-rate floor"}
    wc = c["SQ_WAVE_CYCLES"]
    split = {}
    for key, name in (("SQ_ACTIVE_INST_ANY", "issuing"), ("SQ_WAIT_INST_ANY", "ready_but_not_issued"), ("SQ_WAIT_ANY", "parked")):
        if key in c:
            split[name] = round(c[key] / wc, 4)
    out["wave_cycles"] = split
    out["counters_per_launch"] = {k: c[k] for k in sorted(c)}
    return out


def view_pose(k: int):
    """C5 views: the lego/bicycle pose of app/main.cpp:195-197 rotated about world-up (colmap: (0,-1,0)) by k*45 deg."""
    pos = np.array([-3.0, -0.5, 2.3])
    tgt = np.array([0.0, 0.0, 0.5])
    up = np.array([0.0, -1.0, 0.0])
    a = math.radians(45.0 * k)
    c, s = math.cos(a), math.sin(a)
    # rotation about the y axis (world-up is -y)
    R = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    return (R @ pos).tolist(), (R @ tgt).tolist(), up.tolist()


def algorithmic_bytes(P, V, L, G, W, H):
    """Per-frame algorithmic HBM bytes, SURVEY 8(d) / BASELINE.md 3 (n = radix passes over the live key bits)."""
    b = 32 + max(1, math.ceil(math.log2(max(G, 2))))
    n = math.ceil(b / 8)
    return (236 * P + 48 * V + 8 * P + (8 * P + 12 * L) + (8 + 24 * n) * L + (8 * L + 8 * G) + 40 * L + 12 * W * H)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--splats", type=int, default=P_BICYCLE)
    ap.add_argument("--res", type=str, default="1920x1080")
    ap.add_argument("--ply", type=str, default=os.environ.get("LCGS_BICYCLE_PLY", ""))
    ap.add_argument("--garden-ply", type=str, default=os.environ.get("LCGS_GARDEN_PLY", ""),
                    help="BASELINE config C4 (mip360_garden, 1920x1080, forward+backward): the real file; the leg runs when it "
                         "exists (or with --c4 on the stand-in) and is reported beside the bicycle figures as `c4_garden`")
    ap.add_argument("--c4", action="store_true", help="run the C4 leg on the garden stand-in when --garden-ply is absent")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-backward", action="store_true")
    ap.add_argument("--no-train-step", action="store_true")
    ap.add_argument("--no-batch", action="store_true")
    ap.add_argument("--no-stage-path", action="store_true")
    ap.add_argument("--half-sh", action="store_true", help="also time the opt-in f16 SH colour pass")
    ap.add_argument("--lod", type=int, default=0, help="also time the opt-in footprint cull (lcgs_set_lod) at this radius")
    ap.add_argument("--no-spatial", action="store_true", help="skip the file-order legs (the scene is kept in spatial order)")
    ap.add_argument("--no-moving-camera", action="store_true", help="skip the moving-camera forward leg")
    ap.add_argument("--grad-transport", choices=("f32", "f16"), default="f32",
                    help="N > 1, --collective rccl: the gradient all-reduce's wire format (f16 is opt-in: half the bytes, "
                         "about sqrt(N) x 5e-4 relative error; never the default)")
    ap.add_argument("--leg-timeout", type=float, default=600.0,
                    help="N > 1: seconds the legs after the forward measurement may take before the line is printed without them")
    ap.add_argument("--start-timeout", type=float, default=900.0,
                    help="N > 1: seconds the process group's creation and the forward measurement (the first collectives this "
                         "library ever issues on a new node) may take before rank 0 prints an error line instead of hanging")
    ap.add_argument("--sweep", action="store_true",
                    help="instead of the bench line: the workload sweep (tools/workload_sweep.py) -- scale_modifier x scene x "
                         "resolution, counts / frames/s / forward+backward per point, a fitted time model and what sits off it; "
                         "writes profiles/r06_workload_sweep.json (--sweep-out)")
    ap.add_argument("--sweep-out", type=str, default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles",
                                                                  "r06_workload_sweep.json"))
    ap.add_argument("--collective", choices=("rccl", "torch"), default="rccl",
                    help="N > 1 gradient collective: the library's own RCCL path (lcgs_comm C ABI, default) or "
                         "torch.distributed's (cross-check)")
    return ap.parse_args()


def leg_moving_camera(L, r, cam, img, args, rank, world, W, H, timed):
    # ---- moving camera: the eight C5 poses (base pose rotated about world-up by k x 45 deg) cycled INSIDE the timed
    # loop, a different view every frame: the previous frame's tile schedule is stale, V / L change from frame to frame
    # (launch sizes and buffer hints come from whichever frame synchronised last).  Beside `value`, never instead of it.
    cams8 = [L.get_lookat_cam(*view_pose(k), width=W, height=H) for k in range(8)]
    per_view = []
    for c in cams8:  # one synchronising frame per view: sizes the pair buffers for the largest of them
        r.forward(c, img, sync=True)
        st8 = r.frame_stats()
        per_view.append({"visible_splats": st8["num_visible"], "tile_pairs_sorted": st8["num_pairs"]})
    el_m = timed(lambda i: r.forward(cams8[(i + rank) % 8], img, sync=False), args.steps, max(args.warmup, 8))
    # the same eight views one at a time, each repeated: what a static camera gives on THESE views (the headline
    # pose is view 0 only), so that the cost of motion is separated from the cost of the other views
    el_each = []
    for c in cams8:
        el_each.append(timed(lambda i, c=c: r.forward(c, img, sync=False), max(8, args.steps // 4), 3) /
                       max(8, args.steps // 4))
    static_mean_ms = 1e3 * sum(el_each) / len(el_each)
    moving = {"value": round(world * args.steps / el_m, 2), "unit": "frames/s",
              "ms_per_step": round(el_m * 1e3 / args.steps, 4), "views": 8,
              "same_views_static_ms_per_step": round(static_mean_ms, 4),
              "motion_overhead": round(el_m * 1e3 / args.steps / static_mean_ms - 1.0, 4),
              "per_view": per_view}
    r.forward(cam, img, sync=True)  # back to the headline view (hints, schedule)

    return moving


def leg_camera_batch(torch, r, cam, img, args, world, W, H, dev, dist, red_dev, barrier):
    # ---- the same K frames as one camera batch (lcgs_render_forward_batch: two frames in flight on sibling
    # workspaces, so one frame's latency-bound sort chain overlaps the other's bandwidth- and VALU-bound kernels).
    # Reported beside `value` (which stays the strictly in-order figure), never instead of it.
    # (four target images in rotation: frames that may be in flight together never share one)
    imgs = [img] + [torch.zeros(3, H, W, device=dev) for _ in range(3)]
    cams_k = [cam] * args.steps
    imgs_k = [imgs[i & 3] for i in range(args.steps)]
    r.forward_batch([cam] * max(4, args.warmup), [imgs[i & 3] for i in range(max(4, args.warmup))])
    barrier()
    t0 = time.perf_counter()
    r.forward_batch(cams_k, imgs_k)
    barrier()
    el_p = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el_p], device=red_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el_p = float(t.item())
    pipelined = {"api": "lcgs_render_forward_batch", "frames_in_flight": 2, "value": round(world * args.steps / el_p, 2),
                 "unit": "frames/s", "ms_per_step": round(el_p * 1e3 / args.steps, 4),
                 "images_equal": bool(all(torch.equal(imgs[0], x) for x in imgs[1:]))}

    return pipelined


def leg_stage_path(torch, L, ctx, d, cam, img, args, world, P, W, H, dev, barrier):
    # ---- the drop-in boundary itself: the reference's three operators in its own call order (app/main.cpp:266-308)
    # on the same frame -- SHProcessor.process, GSProjector.forward, GSTileSplatter.forward (which synchronises once
    # per frame for num_rendered, like impl.cpp:106-107).  Informational: `value` is the fused frame.
    shp, prj, spl = L.SHProcessor(), L.GSProjector(), L.GSTileSplatter()
    for op in (shp, prj, spl):
        op.create(ctx)
    z = lambda *sh_, dt=torch.float32: torch.zeros(*sh_, dtype=dt, device=dev)
    Lcap = 20_000_000  # app/main.cpp:245
    G_ = ((W + 15) // 16) * ((H + 15) // 16)
    color, means, covs, depth = z(P, 3), z(P, 2), z(P, 3), z(P)
    accel = L.GSTileSplatterAccelProxy(z(P, dt=torch.int32), z(P, dt=torch.int32), z(Lcap, dt=torch.int64),
                                       z(Lcap, dt=torch.int32), z(Lcap, dt=torch.int64), z(Lcap, dt=torch.int32),
                                       z(2 * G_, dt=torch.int32))
    radii_s, img_s = z(P, dt=torch.int32), z(3, H, W)

    def stage_frame():
        shp.process(L.GPUPointsProxy(P, 3, d["pos"]), cam, d["sh"], color, 3, 3)
        prj.forward(L.GSProjectorInputProxy(P, d["pos"], d["scale"], d["rotq"], 1.0),
                    L.GSProjectorOutputProxy(means, covs, depth), cam)
        return spl.forward(accel, L.GSTileSplatterInputProxy(P, (0.0, 0.0, 0.0), means, depth, covs, color, d["opacity"]),
                           L.GSSplatForwardOutputProxy(H, W, img_s, radii_s))
    n_stage = 0
    for _ in range(max(1, args.warmup)):
        n_stage = stage_frame()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        stage_frame()
    barrier()
    el_s = time.perf_counter() - t0
    stage_path = {"value": round(world * args.steps / el_s, 2), "unit": "frames/s",
                  "ms_per_step": round(el_s * 1e3 / args.steps, 4), "num_rendered": int(n_stage),
                  "max_abs_diff_vs_fused": float((img_s - img).abs().max().item()),
                  "mode": "LCGS_STAGES_EXACT: every operator runs at once, every buffer of the reference is produced"}
    # the same three calls, same loop, with lcgs_set_stage_mode(LCGS_STAGES_DEFERRED): process / forward are recorded, the
    # splatter renders the fused frame from the 3-D arrays (intermediates not written; same image, radii, num_rendered)
    exact_img = img_s.clone()
    ctx.set_stage_mode("deferred")
    img_s.zero_()
    n_def = 0
    for _ in range(max(1, args.warmup)):
        n_def = stage_frame()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        stage_frame()
    barrier()
    el_d = time.perf_counter() - t0
    ctx.set_stage_mode("exact")
    stage_path["deferred"] = {"value": round(world * args.steps / el_d, 2), "unit": "frames/s",
                              "ms_per_step": round(el_d * 1e3 / args.steps, 4), "num_rendered": int(n_def),
                              "image_equal_to_exact_mode": bool(torch.equal(img_s, exact_img)),
                              "mode": "LCGS_STAGES_DEFERRED (opt-in): same calls, the splatter renders the fused frame"}
    del accel, color, means, covs, depth, radii_s, img_s, exact_img

    return stage_path


def leg_half_sh(r, cam, img, args, world, barrier):
    # ---- opt-in f16 SH coefficients for the colour pass (outside the 1e-4 bar, never `value`)
    ref_img = img.clone()
    r.use_half_sh(True)
    for _ in range(args.warmup):
        r.forward(cam, img, sync=False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r.forward(cam, img, sync=False)
    barrier()
    el_h = time.perf_counter() - t0
    half_sh = {"value": round(world * args.steps / el_h, 2), "unit": "frames/s",
               "max_abs_diff_vs_f32": float((img - ref_img).abs().max().item())}
    r.use_half_sh(False)
    r.forward(cam, img, sync=True)

    return half_sh


def leg_lod(r, cam, img, args, world, timed):
    # ---- opt-in footprint (LOD) cull (changes the image, never `value`)
    ref_img = img.clone()
    r.set_lod(args.lod)
    n_lod = r.forward(cam, img, sync=True)
    st_lod = r.frame_stats()
    el_l = timed(lambda i: r.forward(cam, img, sync=False), args.steps, args.warmup)
    lod = {"min_radius_px": args.lod, "value": round(world * args.steps / el_l, 2), "unit": "frames/s",
           "num_rendered": int(n_lod), "visible_splats": st_lod["num_visible"], "tile_pairs_sorted": st_lod["num_pairs"],
           "max_abs_diff_vs_unculled": float((img - ref_img).abs().max().item()),
           "mean_abs_diff_vs_unculled": float((img - ref_img).abs().mean().item())}
    r.set_lod(0)
    r.forward(cam, img, sync=True)
    del ref_img

    return lod


def renderer_entries(torch, r, stats, W, H, dev):
    """List entries the renderer's workgroups stage in the frame just rendered.  A frame that keeps no backward state lists its
    pairs per block of 2 x 2 tiles (fewer pairs through duplication / partition / ranges) and every tile's workgroup walks its
    block's list: the renderer then reads each block's list once per tile of the block."""
    if os.environ.get("LCGS_COARSE_LISTS", "auto") == "0" or stats.get("list_shift", 1) == 0:
        return int(stats["num_pairs"])
    gx, gy = (W + 15) // 16, (H + 15) // 16
    rng = torch.zeros(2 * gx * gy, dtype=torch.int32, device=dev)
    r.last_lists(None, rng)
    bx, by = (gx + 1) // 2, (gy + 1) // 2
    rr = rng[:2 * bx * by].view(by, bx, 2).long()
    lens = rr[..., 1] - rr[..., 0]
    # (the library decides per context from the pair count; per-block ranges fill the first bx * by slots and leave the rest zero)
    per_block = bx * by < gx * gy and int(lens.sum().item()) == int(stats["num_pairs"]) and not bool(rng[2 * bx * by:].any().item())
    if not per_block:
        return int(stats["num_pairs"])
    tx = torch.full((bx,), 2, device=dev, dtype=torch.long)
    ty = torch.full((by,), 2, device=dev, dtype=torch.long)
    if gx % 2: tx[-1] = 1
    if gy % 2: ty[-1] = 1
    return int((lens * ty[:, None] * tx[None, :]).sum().item())


def rooflines(r, stats, acc, data, P, W, H, ms_per_step):
    """the dominant kernel's roofline (HBM figures + the measured VALU issue fraction) and the whole frame three ways"""
    V, Lref, Lp, G = stats["num_visible"], stats["num_rendered"], stats["num_pairs"], stats["num_tiles"]
    # Algorithmic bytes per launch of each stage = per-unit figure x units
    stage_bytes = {
        # a context-owned scene (this one): one 16-byte {position, extent bound} row per splat, then scale / rotation / opacity
        # of the candidates (~V), slab slots; caller-bound arrays (LCGS_CULL_BOUND=0: the A/B hook): pos/scale/rotq of every splat
        "cull_compact": (16 * P + 32 * V + 16 * V) if os.environ.get("LCGS_CULL_BOUND", "1") != "0" else (40 * P + 4 * V + 16 * V),
        "build_records": (4 + 44 + 192) * V + 48 * V,
        # first pass from the 16-byte slab slots (+ dense index / rect writes), three 20-byte passes, and -- re-ordered
        # scenes -- the equal-depth pass (keys, about a third of the values)
        "depth_sort": (16 + 20 + 3 * 20) * V + (6 * V if r.permutation() is not None else 0),
        "expand": 12 * V + 8 * V + 8 * Lp,
        "tile_sort": (16 + 20) * Lp,  # the first pass's counts come from the emitter
        "ranges": 4 * Lp + 8 * G,
        # 4-byte id + 36 of the 48-byte record (+ its 8-byte tile rect when lists are per block) per list entry, each counted ONCE:
        # with per-block lists the (up to) four tiles of a block each stage the block's list -- `renderer_entries` -- and the
        # re-reads are served by L2 for the most part (PMC traffic ~1.85 x these bytes, profiles/r05_pmc_traffic.txt)
        "render": (48 if stats.get("renderer_entries", Lp) != Lp else 40) * Lp + 12 * W * H,
    }
    stage_kernel = {"render": "k_render_forward_b", "build_records": "k_build_records", "cull_compact": "k_cull_compact"}
    dominant = max(acc, key=acc.get) if acc else "render"
    dom_ms = acc.get(dominant, float("nan"))  # HIP events on the stream the kernel is launched on
    achieved = stage_bytes.get(dominant, 0) / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    traffic, pmc_all, pmc_source, profile_errors = None, None, None, []
    src_hash = kernel_sources_hash()
    same_workload = data == "synthetic" and P == P_BICYCLE and (W, H) == (1920, 1080)
    try:  # HBM bytes per launch from the committed rocprofv3 PMC passes of this same workload (profiles/, newest round)
        import glob

        pmc_path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[-1]
        pmc_source = os.path.relpath(pmc_path, ROOT)
        pmc_all = json.load(open(pmc_path))
        if pmc_all.get("kernel_sources_sha256") != src_hash:  # never silently reuse counters of other kernels
            profile_errors.append(f"{pmc_source} was measured on kernel sources {pmc_all.get('kernel_sources_sha256')}, the "
                                  f"library is built from {src_hash}: re-run tools/profile_round.sh (traffic not reported)")
            pmc_all = None
        k = pmc_all["kernels"].get(stage_kernel.get(dominant, "")) if pmc_all else None
        if k and same_workload:
            traffic = {"fetch_bytes_raw": k["fetch_bytes_raw"], "fetch_bytes_x2_corrected": k["fetch_bytes_x2"],
                       "write_bytes": k["write_bytes"], "source": pmc_source, "kernel_sources_sha256": src_hash}
    except Exception as e:  # noqa: BLE001
        profile_errors.append(f"PMC traffic: {type(e).__name__}: {e}")
        traffic = None
    # The dominant kernel is VALU-issue-bound, not HBM-bound: beside the mandatory HBM figures, its VALU issue
    # utilisation = wave-instructions per launch (rocprofv3 SQ_INSTS_VALU, profiles/rNN_pmc_sq.txt) x the average issue
    # cost of the compositing loop's instruction mix on this part (tools/microbench/issue_rates.hip,
    # profiles/r01_issue_rates.txt) / (1024 SIMDs x launch duration x 2.4 GHz).
    valu, sq_kernels, sq_issue_kernels, sq_src, sq_issue_src = None, {}, {}, None, None
    lib_hash = library_hash()
    try:
        import glob

        def newest(pattern, what):
            paths = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
            if not paths:
                return {}, None
            kernels, head = read_sq_file(paths[-1])
            rel = os.path.relpath(paths[-1], ROOT)
            if head.get("kernel_sources_sha256") != src_hash:
                profile_errors.append(f"{rel} was measured on kernel sources {head.get('kernel_sources_sha256')}, the library "
                                      f"is built from {src_hash}: re-run tools/profile_round.sh ({what} not reported)")
                return {}, rel
            if head.get("library_sha256") not in (None, lib_hash):
                profile_errors.append(f"{rel} was measured on liblcgs_hip.so {head.get('library_sha256')}, this process loaded "
                                      f"{lib_hash} (same sources, another build): figures kept, flagged")
            return kernels, rel

        sq_kernels, sq_src = newest("r*_pmc_sq.txt", "instruction counts")
        sq_issue_kernels, sq_issue_src = newest("r*_pmc_sq_issue.txt", "valu_issue")
        if same_workload and dominant == "render":
            # MEASURED issue fraction (round 4): SQ_ACTIVE_INST_VALU against the kernel's cycles, one rocprofv3 pass -- the
            # 2.8-cycle price-list estimate of rounds 1-3 is gone
            valu = valu_issue_from_counters(sq_issue_kernels.get("k_render_forward_b"))
            if valu is not None:
                valu["source"] = sq_issue_src
                # beside the measurement, the ESTIMATE of rounds 1-3 (labelled as such): the loop's instruction mix priced
                # with the per-kind issue costs of tools/microbench/issue_rates.hip (profiles/r01_issue_rates.txt), 2.8
                # cycles per instruction on average -- where the kernel sits if those prices hold inside the loop
                insts = valu["counters_per_launch"].get("SQ_INSTS_VALU")
                if insts:
                    valu["priced_estimate"] = {"avg_issue_cycles_per_instruction": 2.8,
                                               "frac": round(insts * 2.8 / (1024.0 * valu["kernel_cycles"]), 4),
                                               "basis": "estimate, not a counter: instruction mix x microbenchmark prices"}
    except Exception as e:  # noqa: BLE001
        profile_errors.append(f"SQ counters: {type(e).__name__}: {e}")
        valu = None
    # `bound` names what actually limits the kernel: the per-tile compositing loop is bound by VALU issue (valu_issue.frac
    # of the SIMDs' issue cycles), so achieved / peak / frac -- the HBM figures the contract asks for -- say how far it is
    # from the OTHER roof, not how well it is optimised.  Kernels that stream (build_records, cull_compact) report "hbm".
    bound = "valu" if dominant == "render" else "hbm"
    roofline = {"kernel": stage_kernel.get(dominant, dominant), "bound": bound, "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "hbm_frac": round(achieved / HBM_PEAK_GBS, 4), "attainable_peak": HBM_ATTAINABLE_GBS,
                "algorithmic_bytes_per_launch": stage_bytes.get(dominant, 0), "avg_launch_ms": round(dom_ms, 4),
                "valu_issue": valu, "kernel_sources_sha256": src_hash, "library_sha256": lib_hash,
                "note": "the dominant kernel (per-tile compositing) is VALU-issue bound, not HBM bound: `frac` is its HBM "
                        "fraction (the contract's figure), valu_issue.frac the roof it is at; see DESIGN.md 4"}
    # PMC bytes over algorithmic bytes: how much the kernel re-reads (1 = every byte once)
    roofline["traffic_ratio"] = (round((traffic["fetch_bytes_x2_corrected"] + traffic["write_bytes"]) / stage_bytes[dominant], 3)
                                 if traffic and stage_bytes.get(dominant) else None)
    if dominant == "render" and stats.get("renderer_entries", Lp) != Lp:
        roofline["list_entries"] = Lp
        roofline["entries_staged_by_tile_workgroups"] = stats["renderer_entries"]
        roofline["note"] += ("; pair lists are per block of 2 x 2 tiles: the bytes count every list entry once, the (up to) four tile "
                             "workgroups of a block each stage it (re-reads mostly served by L2)")
    if profile_errors:
        roofline["profile_errors"] = profile_errors
    # Whole frame, three ways, side by side (none of them is `roofline`, which is the dominant kernel's):
    #  * survey_model      SURVEY 8(d)'s byte model OF THE REFERENCE ALGORITHM (its L_ref pairs, its 64-bit-key sort passes)
    #                      / this frame's time: a work-equivalent rate, not bytes this implementation moves;
    #  * own_algorithmic   the bytes THIS implementation's kernels have to move (DESIGN.md 4 per-unit figures x the
    #                      measured P, V, L) / the frame time;
    #  * pmc               what the memory-side counters saw: FETCH_SIZE + WRITE_SIZE summed over one frame's launches
    #                      (profiles/, same workload), raw and with the gfx950 x2 FETCH_SIZE correction for wide reads.
    frame_bytes = algorithmic_bytes(P, V, Lref, G, W, H)
    frame_gbs = frame_bytes / (ms_per_step * 1e-3) / 1e9
    own_bytes = sum(stage_bytes.values())
    frame_views = {"survey_model": {"bytes": frame_bytes, "GB/s": round(frame_gbs, 1), "frac": round(frame_gbs / HBM_PEAK_GBS, 4),
                                    "frac_of_attainable": round(frame_gbs / HBM_ATTAINABLE_GBS, 4)},
                   "own_algorithmic": {"bytes": own_bytes, "GB/s": round(own_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                                       "frac": round(own_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                       "frac_of_attainable": round(own_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_ATTAINABLE_GBS, 4),
                                       "per_stage_bytes": stage_bytes}}
    if pmc_all and same_workload:
        fwd_kernels = pmc_all.get("forward_frame_launches")  # {kernel: launches per frame}
        if fwd_kernels:
            raw = sum(pmc_all["kernels"][k]["fetch_bytes_raw"] * n for k, n in fwd_kernels.items() if k in pmc_all["kernels"])
            wr = sum(pmc_all["kernels"][k]["write_bytes"] * n for k, n in fwd_kernels.items() if k in pmc_all["kernels"])
            sec = ms_per_step * 1e-3
            frame_views["pmc"] = {"fetch_bytes_raw": raw, "fetch_bytes_x2": 2 * raw, "write_bytes": wr,
                                  "GB/s_raw": round((raw + wr) / sec / 1e9, 1), "GB/s_x2": round((2 * raw + wr) / sec / 1e9, 1),
                                  "frac_raw": round((raw + wr) / sec / 1e9 / HBM_PEAK_GBS, 4),
                                  "frac_x2": round((2 * raw + wr) / sec / 1e9 / HBM_PEAK_GBS, 4),
                                  "frac_x2_of_attainable": round((2 * raw + wr) / sec / 1e9 / HBM_ATTAINABLE_GBS, 4),
                                  "source": pmc_source}
    # north_star: ">= 1000 frames/s at >= 60 % of the HBM roofline, evidenced by rocprof HBM GB/s".  ONE answer, by the bytes
    # the counters saw (x2-corrected fetch + write) where they are available for these sources, else by this
    # implementation's own algorithmic bytes -- never by the survey model, which prices the reference's algorithm.
    by = frame_views.get("pmc", {}).get("frac_x2")
    basis = "PMC bytes (FETCH_SIZE x2 + WRITE_SIZE)" if by is not None else "own algorithmic bytes (no PMC file for these sources)"
    by = by if by is not None else frame_views["own_algorithmic"]["frac"]
    frame_views["target_60pct_hbm"] = (f"{'met' if by >= 0.6 else 'not met'}: {by:.2f} of 8 TB/s by {basis}; the frame's time goes to "
                                       "VALU-issue-bound compositing and latency-bound sort launches, not to bandwidth")

    return {"roofline": roofline, "frame_views": frame_views, "V": V, "Lref": Lref, "Lp": Lp, "pmc_all": pmc_all,
            "pmc_source": pmc_source, "same_workload": same_workload, "src_hash": src_hash,
            "sq_issue_kernels": sq_issue_kernels, "sq_issue_src": sq_issue_src}


def leg_cpu_baseline_and_parity(out, scene, r, cam, img, P, W, H):
    # ---- CPU baseline: the oracle (CPU restatement of the reference) on this box's host cores, rank 0, N = 1 only
    from oracle import Oracle

    o = Oracle("f32")
    o.set_threads(0)
    ocam = o.lookat(*view_pose(0), width=W, height=H)
    t0 = time.perf_counter()
    frames = 0
    while True:
        ref = o.render(scene, ocam)
        frames += 1
        el = time.perf_counter() - t0
        if el > 12.0 or frames >= 8:
            break
    out["cpu_baseline"] = {"value": round(frames / el, 4), "unit": "frames/s", "cores": o.get_threads(), "kind": "port",
                           "sample": f"{frames} full frame(s) of the same workload ({P} splats, {W}x{H}) in {el:.1f} s"}
    n_rendered = r.forward(cam, img, sync=True)  # the frame the oracle is compared with
    gi = img.cpu().numpy()
    diff = np.abs(gi - ref["img"]).max(axis=0)
    # the blend's exp is one defined sequence of binary32 operations in the kernels and in the oracle (round 3), so
    # the two frames are expected to be EQUAL; anything else is reported, never hidden behind a tolerance
    out["parity"] = {"num_rendered_equal": bool(ref["num_rendered"] == n_rendered),
                     "bit_identical": bool(np.array_equal(gi, ref["img"])),
                     "pixels_different": int((diff > 0).sum()),
                     "pixels_over_1e-4": int((diff > 1e-4).sum()), "max_abs_diff": float(diff.max())}
    # ... and the distance to a STANDARD exp in the blend (the reference says `exp(power)`, gs_tile_splatter/shader.cpp:259;
    # north_star's bar: 1e-4 per-pixel L-inf): the same frame against the oracle run with libm's expf.  Pixels beyond 1e-4
    # are threshold flips (an ulp of exp moves `alpha < 1/255` or `T < 1e-4`); `all_flagged` says every one of them is
    # marked threshold-ambiguous (within 1e-5 relative) by the libm oracle itself.
    o.set_blend_exp(True)
    ref_libm = o.render(scene, ocam, ambig_eps=1e-5)
    o.set_blend_exp(False)
    with np.errstate(invalid="ignore"):
        dl = np.abs(gi.astype(np.float64) - ref_libm["img"].astype(np.float64)).max(axis=0)
    amb = ref_libm["ambig"].astype(bool)
    over = dl > 1e-4
    out["parity"]["vs_libm_expf"] = {
        "pixels_over_1e-4": int(over.sum()), "max_abs_diff": float(dl.max()),
        "all_flagged": bool((~over | amb).all()), "ambiguous_pixels": int(amb.sum()),
        "max_abs_diff_unflagged": float(dl[~amb].max()) if (~amb).any() else 0.0, "pixels": int(dl.size)}
    # ... and to the reference's LIKELY numerics (round 6): the same frame against the oracle built with FMA contraction
    # (what a CUDA JIT does by default), with reciprocal-multiply division, rsqrt forms, right-to-left sums -- samples of
    # such a compiler's choices, not replicas (oracle/numerics.py).  `classes` says how many pixels of the frame carry a
    # decision inside its rounding window (threshold / depth order / rect-radius) and how many may move beyond 1e-4 at
    # all; per variant: how many do, how far, and whether EVERY pixel of the frame stays inside its bound.
    try:
        from oracle import numerics

        rep, _ = numerics.report(scene, ocam, img=gi)
        vs = rep["variants"]
        out["parity"]["vs_contracted"] = dict(vs["contracted"], classes=rep["classes"])
        out["parity"]["vs_other_numerics"] = {
            k: {q: v[q] for q in ("pixels_over_1e-4", "max_abs_diff", "unexplained_pixels", "all_explained",
                                  "worst_ratio_diff_to_bound", "radii_differ")}
            for k, v in vs.items() if k != "contracted"}
    except Exception as e:  # a reporting leg: never costs the line
        out["parity"]["vs_contracted"] = {"error": repr(e)}


def leg_gradients(S):
    """forward + backward (+ the gradient collective at N > 1), the multi-view steps, BASELINE C4, the training-step variants and
    the ownership step.  Fills out["fwd_bwd"] / ["train_step"] / ["c4_garden"]; returns (collective, timed_steps) for the legs
    behind it.  N > 1: whatever a leg throws goes into the line (leg_errors), N = 1: it raises."""
    (args, dist, dev, P, W, H, r, cam, img, d, ctx, rank, world, out, timed, barrier, leg, KEYS, mg, L, torch, local_rank, side, Lp, V, pmc_all, same_workload, pmc_source, sq_issue_kernels, sq_issue_src, src_hash) = (S.args, S.dist, S.dev, S.P, S.W, S.H, S.r, S.cam, S.img, S.d, S.ctx, S.rank, S.world, S.out, S.timed, S.barrier, S.leg, S.KEYS, S.mg, S.L, S.torch, S.local_rank, S.side, S.Lp, S.V, S.pmc_all, S.same_workload, S.pmc_source, S.sq_issue_kernels, S.sq_issue_src, S.src_hash)
    timed_steps = None
    coll = None
    try:
        if dist is not None and os.environ.get("LCGS_BENCH_INJECT_LEG_FAILURE") == "1":  # (test hook)
            raise RuntimeError("injected failure of the gradient legs")
        if not args.no_backward:
            # pos 3 | scale 3 | rotq 4 | sh 48 | opacity 1 in one buffer, every array starting on a 16-byte boundary
            # (lcgs_render_backward wants that of dL/drotq; an odd P would break it otherwise)
            gbuf = torch.zeros(59 * P + 16, device=dev)
            o0 = 0
            views = {}
            for name, width in (("pos", 3), ("scale", 3), ("rotq", 4), ("sh", 48), ("opacity", 1)):
                views[name] = gbuf[o0:o0 + width * P].view(P, width) if width > 1 else gbuf[o0:o0 + P]
                o0 = (o0 + width * P + 3) & ~3
            dL = torch.randn(3, H, W, device=dev)
            if dist is not None:
                if args.collective == "rccl":
                    def exchange(payload):  # rank 0's 128-byte rendezvous token to everybody, over the process group
                        t = torch.zeros(128, dtype=torch.uint8, device=dev)
                        if payload is not None:
                            t.copy_(torch.frombuffer(bytearray(payload), dtype=torch.uint8))
                        dist.broadcast(t, 0)
                        return t.cpu().numpy().tobytes()
                    # the library's own communicator (ncclCommInitRank under lcgs_comm_create): when ANY rank cannot create it
                    # the gradient legs run over torch.distributed's instead -- a figure for the metric all the same, and the
                    # line says why
                    try:
                        coll = mg.RcclCollective(ctx, rank, world, exchange)
                        create_err = None
                    except Exception as e:  # noqa: BLE001
                        coll, create_err = None, f"{type(e).__name__}: {e}"
                    made = torch.tensor([0 if coll is None else 1], device=dev, dtype=torch.int32)
                    if world > 1:
                        dist.all_reduce(made, op=dist.ReduceOp.MIN)
                    if int(made.item()) == 0:
                        if coll is not None:
                            coll.close()
                        out.setdefault("leg_errors", {})["comm_create"] = (create_err or "another rank could not create the communicator")[:300]
                        out["comm_selftest"] = {"ok": False, "every_rank_ok": False, "message": "the communicator was not created",
                                                "fallback": "gradient legs over torch.distributed"}
                        coll = mg.TorchCollective(dist, rank, world)
                        args.collective = "torch"
                if dist is not None and args.collective == "rccl":  # (created on every rank)
                    # ---- before anything distributed is timed: what the communicator says about ITSELF (lcgs_comm_selftest: a
                    # 1 KB all-reduce, zero- and one-byte messages to every peer in one group, an ownership step on a 10 000-splat
                    # scratch scene with and without read-back; 30 s per phase).  In the line whatever it says; when it fails the
                    # gradient legs are not attempted (a transport that cannot pass this would hang them)
                    st_self = coll.comm.selftest(timeout_s=30.0, check=False)
                    if os.environ.get("LCGS_BENCH_INJECT_SELFTEST_FAILURE") == "1":  # (test hook)
                        st_self.update(ok=False, p2p_ok=0, message="injected self-test failure")
                    oks = torch.tensor([1 if st_self["ok"] else 0], device=dev, dtype=torch.int32)
                    if world > 1:
                        dist.all_reduce(oks, op=dist.ReduceOp.MIN)
                    st_self["every_rank_ok"] = bool(int(oks.item()) == 1)
                    out["comm_selftest"] = st_self
                    r.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"])  # (the self-test put the binding back; the rows too)
                    if not st_self["every_rank_ok"]:
                        # the library's communicator cannot be trusted with the timed legs: the gradient legs run over
                        # torch.distributed's own communicator instead (a figure for the metric all the same), and the line says
                        # so.  A communicator with a phase stuck inside RCCL is left alone (destroying it would wait for it).
                        tm = torch.tensor([1 if st_self["timed_out"] else 0], device=dev, dtype=torch.int32)
                        if world > 1:
                            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                        if int(tm.item()) == 0:
                            coll.close()
                        else:
                            coll.comm.abandon()  # (its destructor would wait for the stuck stream at interpreter exit)
                        st_self["fallback"] = "gradient legs over torch.distributed (the library's communicator failed its self-test)"
                        out.setdefault("leg_errors", {})["comm_selftest"] = st_self["message"][:300]
                        coll = mg.TorchCollective(dist, rank, world)
                        args.collective = "torch"
                    elif args.grad_transport != "f32":
                        coll.comm.set_transport(args.grad_transport)
                else:
                    coll = mg.TorchCollective(dist, rank, world)
            engine = mg.HipEngine(r, raw=None, activated=d, lr=None)  # gradients only: no optimiser state
            tr_sum = mg.ViewParallelTrainer(engine, coll, [cam], views, mode="allreduce" if coll is not None else "local")
            tr_local = mg.ViewParallelTrainer(engine, None, [cam], views, mode="local")
            warm = max(1, args.warmup)

            def timed_steps(collective, compact=False):
                engine.compact_rows = compact
                tr = tr_sum if collective else tr_local
                el_ = timed(lambda i: tr.step(dL, optimise=False), args.steps, warm)
                engine.compact_rows = False
                return el_

            el = timed_steps(True)
            # N > 1: the same steps without the gradient collective, so that its share is visible
            el_local = timed_steps(False) if dist is not None else None
            # N = 1: the same step with compact gradient rows (lcgs_render_backward_compact: one row per on-screen splat,
            # no zero-fill) -- reported beside the dense figure, which stays `value` (a sum over views needs per-splat rows)
            el_compact = timed_steps(False, compact=True) if dist is None else None
            r.set_profiling(True)
            r.forward(cam, img, keep_state=True, sync=True)
            Lk = r.frame_stats()["num_pairs"]  # the keep-state frame's own PER-TILE lists: what k_render_backward walks
            r.backward(dL, *[views[k] for k in KEYS])
            bwd_stages = r.stage_times()
            r.set_profiling(False)
            out["fwd_bwd"] = {"metric": "fwd+bwd Msplats/s", "value": round(world * P * args.steps / el / 1e6, 1),
                              "unit": "Msplats/s", "ms_per_step": round(el * 1e3 / args.steps, 4),
                              "grad_allreduce_bytes_per_gpu": 59 * 4 * P if world > 1 else 0,
                              "xgmi_bytes_sent_per_gpu": mg.allreduce_bus_bytes_per_gpu(P, world),
                              "collective": coll.name if coll is not None else None,
                              "backward_stages_ms": {k: round(v, 4) for k, v in bwd_stages.items()}}
            # ---- the metric's second half has a dominant kernel of its own: k_render_backward.  Algorithmic bytes per launch
            # (render half of the backward): 40 L -- L = the pairs of the frame that KEPT the backward's state, per
            # tile (7.53 M on the stand-in), not the forward-only frame's per-block pairs (4.47 M) that rounds up to 5 priced it
            # with -- (list entry + 36-byte record re-gathered) + 20 W H (final_T,
            # n_contrib, dL/dimg) + 40 V (2-D gradient rows, read-modify-write); time = HIP events around the kernel on the
            # context's stream (profiling mode: alone, not beside the zero-fill); traffic and the VALU issue fraction from the
            # committed rocprofv3 passes of this workload, refused when measured on other sources.
            rb_ms = bwd_stages.get("render_backward", 0.0)
            rb_bytes = 40 * Lk + 20 * W * H + 40 * V
            if rb_ms > 0:
                rb_gbs = rb_bytes / (rb_ms * 1e-3) / 1e9
                rb_traffic = None
                kk = pmc_all["kernels"].get("k_render_backward") if (pmc_all and same_workload) else None
                if kk:
                    rb_traffic = {"fetch_bytes_raw": kk["fetch_bytes_raw"], "fetch_bytes_x2_corrected": kk["fetch_bytes_x2"],
                                  "write_bytes": kk["write_bytes"], "source": pmc_source}
                rb_valu = valu_issue_from_counters(sq_issue_kernels.get("k_render_backward")) if same_workload else None
                if rb_valu is not None:
                    rb_valu["source"] = sq_issue_src
                out["fwd_bwd"]["roofline"] = {
                    "kernel": "k_render_backward", "bound": "valu", "achieved": round(rb_gbs, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(rb_gbs / HBM_PEAK_GBS, 4), "traffic": rb_traffic,
                    "algorithmic_bytes_per_launch": rb_bytes, "list_entries": Lk, "avg_launch_ms": round(rb_ms, 4),
                    "traffic_ratio": (round((rb_traffic["fetch_bytes_x2_corrected"] + rb_traffic["write_bytes"]) / rb_bytes, 3)
                                      if rb_traffic else None),
                    "valu_issue": rb_valu, "kernel_sources_sha256": src_hash,
                    "note": "the step's dominant kernel: VALU-issue-bound (per-pixel terms and nine 64-lane sums per list entry and strip); "
                            "`frac` is its HBM fraction, valu_issue.frac the measured share of SIMD cycles issuing VALU work"}
            if dist is not None:
                out["fwd_bwd"]["note"] = ("gradient collective: lcgs_grads_allreduce (RCCL, chunked behind the backward's "
                                          "slices)" if args.collective == "rccl" else "gradient collective: torch.distributed")
                out["fwd_bwd"]["grad_transport"] = args.grad_transport if args.collective == "rccl" else "f32"
                if isinstance(coll, mg.RcclCollective):
                    # what RCCL itself saw: the communicator's size (ncclCommInitRank's, via lcgs_comm_info) and what the last
                    # gradient sum of the timed loop issued / moved (lcgs_comm_get_stats) -- so that the line certifies by itself
                    # that N ranks exchanged gradients through the library's communicator
                    tr_sum.step(dL, optimise=False)
                    st_c = coll.comm.stats()
                    out["fwd_bwd"]["rccl_ranks"] = coll.comm.info()[1]
                    out["fwd_bwd"]["collective_groups"] = st_c["collective_groups"]
                    out["fwd_bwd"]["rccl_bytes_sent_last_step"] = st_c["bytes_sent"]
            if el_compact is not None:
                out["fwd_bwd"]["compact_rows"] = {"value": round(P * args.steps / el_compact / 1e6, 1), "unit": "Msplats/s",
                                                  "ms_per_step": round(el_compact * 1e3 / args.steps, 4)}
            if el_local is not None:
                out["fwd_bwd"]["without_collective"] = {"value": round(world * P * args.steps / el_local / 1e6, 1),
                                                        "unit": "Msplats/s",
                                                        "ms_per_step": round(el_local * 1e3 / args.steps, 4)}
            if not args.no_moving_camera:
                # every rank a different one of the eight C5 views at every step (view_of_rank), collective included
                cams8 = [L.get_lookat_cam(*view_pose(k), width=W, height=H) for k in range(8)]
                tr_mov = mg.ViewParallelTrainer(engine, coll, cams8, views, mode="allreduce" if coll is not None else "local")
                el_mv = timed(lambda i: tr_mov.step(dL, optimise=False), args.steps, max(warm, 8))
                out["fwd_bwd"]["moving_camera"] = {"value": round(world * P * args.steps / el_mv / 1e6, 1), "unit": "Msplats/s",
                                                   "ms_per_step": round(el_mv * 1e3 / args.steps, 4), "views": 8}
                r.forward(cam, img, sync=True)
                if dist is not None:
                    # the answer to a collective that costs several views' worth of compute: B views per GPU and step,
                    # their gradients accumulated (lcgs_render_backward_accumulate), ONE gradient sum per step
                    B = 4
                    tr_acc = mg.ViewParallelTrainer(engine, coll, cams8, views, mode="allreduce", views_per_step=B)
                    acc_steps = max(2, args.steps // B)
                    el_acc = timed(lambda i: tr_acc.step(dL, optimise=False), acc_steps, 2)
                    out["fwd_bwd"]["views_per_gpu_and_step_4"] = {
                        "value": round(world * P * B * acc_steps / el_acc / 1e6, 1), "unit": "Msplats/s",
                        "ms_per_step": round(el_acc * 1e3 / acc_steps, 4), "views_per_step": world * B,
                        "note": "one gradient collective per step of 4 views per GPU"}
                    r.forward(cam, img, sync=True)
                # ---- a multi-view step on each GPU (4 views, L2 loss against a target image, gradients summed; + the
                # collective when N > 1): the views one after the other, and through lcgs_fit_views, which lets a view's
                # forward run beside the previous view's backward
                B = 4
                tgt = torch.rand(3, H, W, device=dev)
                dLb = torch.empty(3, H, W, device=dev)
                losses = torch.zeros(B, device=dev)
                gl = [views[k] for k in KEYS]

                def views_one_by_one(i):
                    for j in range(B):
                        r.forward(cams8[(i * B + j + rank) % 8], img, keep_state=True, sync=False)
                        r.l2_loss_backward(img, tgt, dLb, losses[j:j + 1])
                        r.backward(dLb, *gl, accumulate=j > 0)
                    if coll is not None:
                        coll.allreduce_grads(views)

                def views_overlapped(i):
                    r.fit_views([cams8[(i * B + j + rank) % 8] for j in range(B)], [tgt] * B, *gl, losses)
                    if coll is not None:
                        coll.allreduce_grads(views)
                mv_steps = max(2, args.steps // B)
                el_seq = timed(views_one_by_one, mv_steps, 2)
                el_fit = timed(views_overlapped, mv_steps, 2)
                out["fwd_bwd"]["multi_view_step_4"] = {
                    "unit": "Msplats/s", "views_per_step": world * B,
                    "one_by_one": {"value": round(world * P * B * mv_steps / el_seq / 1e6, 1),
                                   "ms_per_step": round(el_seq * 1e3 / mv_steps, 4)},
                    "lcgs_fit_views": {"value": round(world * P * B * mv_steps / el_fit / 1e6, 1),
                                       "ms_per_step": round(el_fit * 1e3 / mv_steps, 4)}}
                r.forward(cam, img, sync=True)

            # ---- BASELINE config C4: mip360_garden, 1920x1080, one forward+backward step, its own camera
            # (app/main.cpp:191-193).  Runs on the real scene when LCGS_GARDEN_PLY / --garden-ply has it (data: real),
            # on the stand-in with --c4; N = 1 only (a second resident scene; nothing here shards differently).
            have_garden = bool(args.garden_ply) and os.path.exists(args.garden_ply)
            if dist is None and (have_garden or args.c4):
                ctx4 = L.Context(local_rank, side.cuda_stream)
                r4 = L.Renderer(ctx4)
                if have_garden:
                    r4.load_ply(args.garden_ply)
                else:
                    r4.upload_scene(L.synth_scene(1, 2002, 5_834_784))
                d4 = r4.scene_tensors()
                P4 = int(d4["pos"].shape[0])
                cam4 = L.get_lookat_cam([-3, -0.5, 3.3], [0, 3, 0.5], [0, -1, -1], width=1920, height=1080)
                img4, dL4 = torch.zeros(3, 1080, 1920, device=dev), torch.randn(3, 1080, 1920, device=dev)
                g4 = {k: torch.zeros_like(d4[k]) for k in KEYS}
                n4 = r4.forward(cam4, img4, keep_state=True, sync=True)

                def c4_step(i):
                    r4.forward(cam4, img4, keep_state=True, sync=False)
                    r4.backward(dL4, *[g4[k] for k in KEYS])

                el4 = timed(c4_step, args.steps, warm)
                st4 = r4.frame_stats()
                out["c4_garden"] = {"data": "real" if have_garden else "synthetic", "splats": P4, "num_rendered": int(n4),
                                    "visible_splats": st4["num_visible"], "tile_pairs_sorted": st4["num_pairs"],
                                    "value": round(P4 * args.steps / el4 / 1e6, 1), "unit": "Msplats/s",
                                    "ms_per_step": round(el4 * 1e3 / args.steps, 4)}
                del r4, ctx4, d4, g4, img4, dL4

            # ---- full training-style step: + the optimiser (gradients -> Adam on the raw parameters -> refreshed
            # activated arrays).  N = 1: dense, restricted to the on-screen splats, and on compact rows.
            # N > 1: "allreduce" (lcgs_grads_allreduce + a dense lcgs_adam_step on every rank) and "sharded"
            # (lcgs_adam_step_sharded: reduce-scatter -> Adam on the own rows -> all-gather of the activated arrays).
            if not args.no_train_step:
                # (on a copy of the scene: exp(log(s)) is not s to the last bit, and the parity block below compares the
                # frame of the pristine scene)
                act = {k: d[k].clone() for k in KEYS}
                raw = {"pos": act["pos"], "scale": torch.log(act["scale"]), "rotq": act["rotq"].clone(), "sh": act["sh"],
                       "opacity": torch.log(act["opacity"] / (1 - act["opacity"]))}
                lr = {"pos": 0.0, "sh_dc": 0.0, "sh_rest": 0.0, "opacity": 0.0, "scale": 0.0, "rot": 0.0}  # scene stays put
                eng2 = mg.HipEngine(r, raw=raw, activated=act, lr=lr)  # (binds `act` to the renderer)
                out["train_step"] = {}
                if dist is None:
                    modes = ("dense", "visible_only", "visible_only_compact", "visible_only_fused")
                else:
                    modes = ("allreduce", "sharded", "sparse")
                for mode in modes:
                    if dist is None:
                        def full_step(i, mode=mode):
                            compact = mode == "visible_only_compact"
                            r.forward(cam, img, keep_state=True, sync=False)
                            if mode == "visible_only_fused":  # lcgs_render_backward_adam: no gradient arrays at all
                                r.backward_adam(dL, raw, eng2.m, eng2.v, act, i + 1, lr)
                                return
                            r.backward(dL, *[views[k] for k in KEYS], compact=compact)
                            r.adam_step(views, raw, eng2.m, eng2.v, act, i + 1, lr, visible_only=(mode != "dense"),
                                        compact_grads=compact)
                    else:
                        tr = mg.ViewParallelTrainer(eng2, coll, [cam], views, mode=mode)

                        def full_step(i, tr=tr):
                            tr.step(dL, optimise=True)
                    el2 = timed(full_step, args.steps, 2)
                    out["train_step"][mode] = {"value": round(world * P * args.steps / el2 / 1e6, 1), "unit": "Msplats/s",
                                               "ms_per_step": round(el2 * 1e3 / args.steps, 4)}
                    if dist is not None:  # bytes one GPU sends per step: the dense modes by formula, the sparse step
                        # from the actual message sizes of its last step (lcgs_comm_get_stats)
                        if mode == "sparse":
                            st = getattr(coll, "last_stats", None) or {}
                            out["train_step"][mode].update({
                                "xgmi_bytes_sent_per_gpu": st.get("bytes_sent"), "touched_rows": st.get("touched_rows"),
                                "note": "reduce half = touched rows only (indices + 59 floats each, one message per owner); "
                                        "all-gather of the activated rows as in the sharded step"})
                        else:
                            out["train_step"][mode]["xgmi_bytes_sent_per_gpu"] = mg.allreduce_bus_bytes_per_gpu(P, world)
                if dist is not None:
                    # ---- splat ownership: nothing replicated, nothing all-gathered -- every rank owns P / N rows,
                    # projects them for every view of the step and exchanges 48-byte records / 48-byte 2-D gradients of
                    # on-screen rows with the views' renderers.  Transport: the library's own (lcgs_owner_step_forward / _backward:
                    # ncclSend / ncclRecv groups on the communicator's stream) with --collective rccl, torch.distributed
                    # point-to-point on the process group with --collective torch; a leg of its own, so a failure costs nothing else.
                    try:
                        cams_o = [L.get_lookat_cam(*view_pose(k), width=W, height=H) for k in range(8)]
                        ocoll = coll if isinstance(coll, mg.RcclCollective) else mg.TorchCollective(dist, rank, world)
                        tr_o = mg.ViewParallelTrainer(eng2, ocoll, cams_o, views, mode="owner")
                        el_o = timed(lambda i: tr_o.step(dL, optimise=True), args.steps, 2)
                        st_o = getattr(ocoll, "last_stats", None) or {}
                        out["train_step"]["owner"] = {
                            "value": round(world * P * args.steps / el_o / 1e6, 1), "unit": "Msplats/s",
                            "ms_per_step": round(el_o * 1e3 / args.steps, 4),
                            "transport": ocoll.name,
                            "xgmi_bytes_sent_per_gpu": st_o.get("bytes_sent"),
                            "on_screen_rows_rendered": st_o.get("on_screen_rows_received"),
                            "collective_groups": st_o.get("collective_groups"),
                            "steps_repeated": getattr(ocoll, "owner_redos", None),
                            "note": "splat ownership: 2-D records / 2-D gradients of on-screen rows travel, Adam on the own rows only, "
                                    "no all-gather; through the library's transport no call of a step reads anything back from the "
                                    "second step on (messages sized from the previous step's counts x 1.25 + 1024 rows, one "
                                    "max-reduced verdict behind the forward half: lcgs_owner_step_finish)"}
                    except Exception as e:  # noqa: BLE001
                        out.setdefault("leg_errors", {})["train_step.owner"] = f"{type(e).__name__}: {e}"[:400]
                eng2.close()
                r.bind_scene(d["pos"], d["scale"], d["rotq"], d["sh"], d["opacity"])
                del act, raw, eng2
    except Exception as e:  # noqa: BLE001 -- N > 1: whatever the hardware run throws belongs in the line
        if dist is None:
            raise
        leg["failed"] = True
        out.setdefault("leg_errors", {})["fwd_bwd / train_step"] = f"{type(e).__name__}: {e}"[:400]
    return coll, timed_steps


def main():
    args = parse_args()
    if args.sweep:  # a tool of its own output: never the bench line
        from tools import workload_sweep

        workload_sweep.run(args.sweep_out, quick=os.environ.get("LCGS_SWEEP_QUICK") == "1")
        return

    # The contract is ONE JSON line on stdout.  Libraries below print there too (RCCL's version banner at communicator
    # creation, gloo's "connected to N peer ranks"): from here on file descriptor 1 points at stderr, and the line is
    # written to the saved descriptor by emit().
    sys.stdout.flush()
    line_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch

    import luisacomputegaussiansplatting_amd as L
    import luisacomputegaussiansplatting_amd.multi_gpu as mg

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        # launched without torchrun: start the N ranks as children and exit with their code
        import subprocess

        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29517"), __file__] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, stdout=line_out))  # (the children's stdout is the real one)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: liblcgs_hip has no CPU path")
    # N > 1: the process group's creation and the barriers around the forward measurement are the first collectives of a run
    # -- on a new node, RCCL's first contact.  If they never return there is no measurement to print, but a line that says so
    # beats a silent hang: rank 0 prints it and every rank exits non-zero (the leg watchdog below takes over afterwards).
    early = None
    if world > 1:
        def never_started():
            if rank == 0:
                print(json.dumps({"metric": "forward fps @1080p, mip360_bicycle", "value": None, "unit": "frames/s", "n_gpus": world,
                                  "error": f"no forward measurement within {args.start_timeout} s: the process group's creation "
                                           "or the first barrier did not return (rank 0's view)"}), file=line_out, flush=True)
            else:
                time.sleep(20.0)  # (rank 0 prints the line: a launcher ends every rank as soon as the first one exits)
            os._exit(3)
        early = threading.Timer(args.start_timeout, never_started)
        early.daemon = True
        early.start()
    # a launcher that narrows each rank's visibility to one GPU leaves fewer devices than local ranks
    local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dist = None
    backend = os.environ.get("LCGS_BENCH_BACKEND", "nccl")
    # (LCGS_BENCH_FORCE_DIST=1: a rehearsal hook -- one rank, but through the process group, so that every N > 1 code
    #  path of this file, collectives included, runs on a one-GPU box)
    if world > 1 or os.environ.get("LCGS_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # (LCGS_BENCH_BACKEND=gloo: a second rehearsal hook -- several ranks on ONE GPU, which RCCL refuses: the launch,
        #  the per-rank views, the barriers and the max-over-ranks timing run as on a node; the gradient legs then fail
        #  on every rank ("duplicate GPU") and exercise the leg-error path below)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    W, H = (int(x) for x in args.res.lower().split("x"))
    dev = torch.device("cuda", local_rank)
    red_dev = dev if backend == "nccl" else torch.device("cpu")  # where the timing reductions live
    KEYS = mg.KEYS

    # ---- scene (replicated on every GPU), through the library's own ingest: the context owns the device arrays and keeps
    # them in spatial (Morton) order, its default for scenes it owns (lcgs_set_ingest_order; same images -- the blend
    # order is by depth).  The file-order figure is reported beside `value` (`file_order`).
    data = "synthetic"
    side = torch.cuda.Stream(device=dev)  # a dedicated (non-NULL) HIP stream for the context
    torch.cuda.set_stream(side)
    ctx = L.Context(local_rank, side.cuda_stream)
    r = L.Renderer(ctx)
    if args.ply and os.path.exists(args.ply):
        r.load_ply(args.ply)  # records -> GPU -> activated arrays (lcgs_scene_load_ply)
        scene = L.read_gs_ply(args.ply)  # host copy, file order: the CPU baseline's input
        scene.pop("sh_degree", None)
        workload = os.path.basename(args.ply)
        data = "real"
    else:
        scene = L.synth_scene(1, 2001, args.splats)
        # host arrays -> context-owned device arrays (lcgs_scene_upload); LCGS_BENCH_FILE_ORDER=1 is a profiling hook:
        # every leg then runs on the file-order scene (the workload name says so)
        file_order = os.environ.get("LCGS_BENCH_FILE_ORDER") == "1"
        r.upload_scene(scene, order="file" if file_order else None)
        workload = f"mip360_bicycle stand-in: synth_unbounded(seed=2001, P={args.splats})"
    workload += " [ingest order: file (LCGS_BENCH_FILE_ORDER)]" if r.permutation() is None else \
                " [ingest order: spatial (library default)]"
    P = scene["pos"].shape[0]
    d = r.scene_tensors()  # torch views of the context's arrays (the order the library keeps them in)
    torch.cuda.synchronize(dev)
    cam = L.get_lookat_cam(*view_pose(rank), width=W, height=H)
    img = torch.zeros(3, H, W, device=dev)

    # first frame synchronises: sizes the pair buffers for this view
    n_rendered = r.forward(cam, img, sync=True)
    # (a context's first frame lists its pairs per tile; from the second on the library may switch to per-block lists for frames
    #  without backward state, by the pair count: the counters reported are those of the frames that are timed)
    n_rendered = r.forward(cam, img, sync=True)
    stats = r.frame_stats()
    stats["renderer_entries"] = renderer_entries(torch, r, stats, W, H, dev)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        r.forward(cam, img, sync=False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r.forward(cam, img, sync=False)
    barrier()
    elapsed = time.perf_counter() - t0
    # SURVEY 8d's protocol beside the contract's: the median of >= 50 frames on hipEvent pairs -- one event behind every
    # frame on the context's stream (torch's current stream here).  Its own loop, so that `value` above is not perturbed
    # by the event packets; reported beside `value`, never instead of it.
    n_ev = max(50, args.steps)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(n_ev + 1)]
    marks[0].record()
    for i in range(n_ev):
        r.forward(cam, img, sync=False)
        marks[i + 1].record()
    barrier()
    frame_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(n_ev))
    per_frame = {"frames": n_ev, "median_ms": round(frame_ms[n_ev // 2], 4), "min_ms": round(frame_ms[0], 4),
                 "p90_ms": round(frame_ms[(n_ev * 9) // 10], 4), "frames_per_s_at_median": round(1e3 / frame_ms[n_ev // 2], 1),
                 "how": "hipEvent pairs between consecutive frames on the context's stream (SURVEY 8d)"}
    if dist is not None:
        t = torch.tensor([elapsed], device=red_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if early is not None:
        early.cancel()
    ms_per_step = elapsed * 1e3 / args.steps
    value = world * args.steps / elapsed

    # N > 1 bookkeeping: every timed leg carries its ordinal and the rank's failure flag through the max-reduction, so a
    # rank that failed alone and moved on can never have its reduction paired with another leg's on its peers (the
    # figures would silently mix): a mismatch raises on every rank that sees it.
    leg = {"no": 0, "failed": False}

    def reduce_leg(el):
        if dist is None:
            return el
        tt = torch.tensor([el, leg["no"], -leg["no"], 1.0 if leg["failed"] else 0.0], device=red_dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        if tt[1].item() != -tt[2].item():
            leg["failed"] = True
            raise RuntimeError(f"ranks are in different legs ({int(-tt[2].item())} .. {int(tt[1].item())}): a rank failed alone")
        if tt[3].item() > 0 and not leg["failed"]:
            leg["failed"] = True
            raise RuntimeError("another rank reported a leg error")
        return float(tt[0].item())

    def timed(fn, steps, warm):
        """warm untimed calls, then `steps` timed ones between barriers; max over ranks; seconds"""
        leg["no"] += 1
        for i in range(warm):
            fn(i)
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            fn(warm + i)
        barrier()
        return reduce_leg(time.perf_counter() - t0)

    moving = None if args.no_moving_camera else leg_moving_camera(L, r, cam, img, args, rank, world, W, H, timed)

    # ---- per-stage device times (HIP events on the context's stream), outside the timed region
    r.set_profiling(True)
    acc = {}
    reps = 10
    for _ in range(reps):
        r.forward(cam, img, sync=True)
        for k, v in r.stage_times().items():
            acc[k] = acc.get(k, 0.0) + v / reps
    r.set_profiling(False)

    pipelined = None if args.no_batch else leg_camera_batch(torch, r, cam, img, args, world, W, H, dev, dist, red_dev, barrier)
    stage_path = None if args.no_stage_path else leg_stage_path(torch, L, ctx, d, cam, img, args, world, P, W, H, dev, barrier)
    half_sh = leg_half_sh(r, cam, img, args, world, barrier) if args.half_sh else None
    lod = leg_lod(r, cam, img, args, world, timed) if args.lod > 0 else None

    rf = rooflines(r, stats, acc, data, P, W, H, ms_per_step)
    roofline, frame_views, V, Lref, Lp = rf["roofline"], rf["frame_views"], rf["V"], rf["Lref"], rf["Lp"]
    pmc_all, pmc_source, same_workload, src_hash = rf["pmc_all"], rf["pmc_source"], rf["same_workload"], rf["src_hash"]
    sq_issue_kernels, sq_issue_src = rf["sq_issue_kernels"], rf["sq_issue_src"]

    out = {
        "metric": "forward fps @1080p, mip360_bicycle", "value": round(value, 2), "unit": "frames/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": data,
        "config": {"workload": workload, "resolution": f"{W}x{H}", "splats": P, "visible_splats": V,
                   "tile_pairs_reference": Lref, "tile_pairs_sorted": Lp, "views_per_gpu": 1,
                   "parallelism": f"view-parallel x{world}"},
        "roofline": roofline,
        "frame_roofline": {"peak": HBM_PEAK_GBS, "attainable_peak": HBM_ATTAINABLE_GBS,
                           "attainable_peak_source": "tools/microbench/fetch_calib.hip: plain 12- / 16-byte-per-lane read streams "
                                                     "over 1 GiB run at 6.3-6.55 TB/s on this part "
                                                     "(profiles/r02_fetch_calibration.txt); 8 TB/s is the spec figure",
                           "unit": "GB/s", **frame_views},
        "per_frame_events": per_frame,
        "stages_ms": {k: round(v, 4) for k, v in acc.items()},
    }
    if stage_path is not None:
        out["stage_path"] = stage_path
    if half_sh is not None:
        out["half_sh"] = half_sh
    if lod is not None:
        out["lod"] = lod
    if pipelined is not None:
        out["camera_batch"] = pipelined
    if moving is not None:
        out["moving_camera"] = moving

    # ---- N > 1 insurance.  The legs below (RCCL communicator of the library, collectives, sharded optimiser) have only
    # ever run on one GPU (`LCGS_BENCH_FORCE_DIST=1`) and on gloo: on a real multi-GPU node a rank that fails alone
    # leaves the others waiting in a collective.  The forward measurement above is complete at this point, so a
    # watchdog prints the line with what has been measured (plus `error`) instead of hanging the whole run.
    printed = threading.Lock()
    watchdog = None

    def emit(error=None):
        if not printed.acquire(blocking=False):
            return
        if error is not None:
            out["error"] = error
        if rank == 0:
            print(json.dumps(out), file=line_out, flush=True)

    if dist is not None:
        def give_up():
            done = printed.locked()  # (the line is out: only the teardown is stuck)
            emit(f"the legs after the forward measurement did not finish within {args.leg_timeout} s on rank {rank}")
            sys.stdout.flush()
            # A timeout is a failure on EVERY rank (the line, with `error`, is on stdout for whoever wants to parse it) --
            # unless only an AUXILIARY leg hung: a run whose line was already complete (the teardown alone is stuck), or
            # one that has both halves of BASELINE's metric (forward frames/s in `value`, forward+backward Msplats/s in
            # `fwd_bwd.value`), ends with 0 so that a harness keyed on the exit code keeps the measured figures.
            headline = args.no_backward or ("value" in out.get("fwd_bwd", {}))
            os._exit(0 if done or headline else 3)
        watchdog = threading.Timer(args.leg_timeout, give_up)
        watchdog.daemon = True
        watchdog.start()

        # ... and a rank that DIES in one of those legs (not hangs: a fault inside a first-contact RCCL path) makes the launcher
        # send SIGTERM to the others: the survivors print what has been measured before they go, by the same exit-code rule
        import signal

        def terminated(signum, frame):  # noqa: ARG001
            done = printed.locked()
            emit(f"terminated by the launcher (signal {signum}) during the legs after the forward measurement: another rank failed")
            sys.stdout.flush()
            os._exit(0 if done or args.no_backward or ("value" in out.get("fwd_bwd", {})) else 3)
        try:
            signal.signal(signal.SIGTERM, terminated)
        except ValueError:  # (not the main thread)
            pass

    # ---- forward + backward (+ the RCCL sum of the dense per-splat gradients when N > 1): one training-style step per
    # view through the package's view-parallel protocol (luisacomputegaussiansplatting_amd.multi_gpu: the same
    # ViewParallelTrainer tests/test_distributed.py runs on gloo); Msplats/s = splats x views / time.
    # Same barrier / max-over-ranks timing.
    import types

    coll, timed_steps = leg_gradients(types.SimpleNamespace(
        args=args, dist=dist, dev=dev, P=P, W=W, H=H, r=r, cam=cam, img=img, d=d, ctx=ctx, rank=rank, world=world, out=out, timed=timed, barrier=barrier, leg=leg, KEYS=KEYS, mg=mg, L=L, torch=torch, local_rank=local_rank, side=side, Lp=Lp, V=V, pmc_all=pmc_all, same_workload=same_workload, pmc_source=pmc_source, sq_issue_kernels=sq_issue_kernels, sq_issue_src=sq_issue_src, src_hash=src_hash))

    # ---- the same frames with the splats in FILE order (caller-bound arrays in the order of the file: for this stand-in
    # i.i.d., the worst case -- a view's splats are scattered over every DRAM page).  Same splats, same image.
    try:
        if not args.no_spatial and dist is not None:
            # every rank that gets here learns whether ANY rank failed an earlier leg; then all of them skip this one
            # together (a lone survivor would otherwise wait in this leg's barriers for peers that are not coming)
            leg["no"] += 1
            try:
                reduce_leg(0.0)
            except RuntimeError:
                pass
        if not args.no_spatial and leg["failed"]:
            out.setdefault("leg_errors", {})["file_order"] = "skipped: a rank reported an error in an earlier leg"
        elif not args.no_spatial:
            ref_img = torch.empty_like(img)
            r.forward(cam, ref_img, sync=True)
            df = {k: torch.from_numpy(np.ascontiguousarray(scene[k], dtype=np.float32)).to(dev) for k in KEYS}
            r.bind_scene(*[df[k] for k in KEYS])
            n_fo = r.forward(cam, img, sync=True)
            fo = {"api": "lcgs_scene_bind of file-order arrays", "num_rendered_equal": bool(n_fo == n_rendered),
                  "image_equal": bool(torch.equal(img, ref_img))}
            el_s = timed(lambda i: r.forward(cam, img, sync=False), args.steps, args.warmup)
            fo["forward"] = {"value": round(world * args.steps / el_s, 2), "unit": "frames/s",
                             "ms_per_step": round(el_s * 1e3 / args.steps, 4)}
            if not args.no_backward:
                for compact in ((False, True) if dist is None else (False,)):
                    el_b = timed_steps(False, compact=compact)  # (per-view steps; no collective in this leg)
                    fo["fwd_bwd_compact_rows" if compact else "fwd_bwd"] = {
                        "value": round(world * P * args.steps / el_b / 1e6, 1), "unit": "Msplats/s",
                        "ms_per_step": round(el_b * 1e3 / args.steps, 4)}
            if not args.no_stage_path and dist is None:
                # the reference's three operators on the file-order arrays (what app/main.cpp:180-223 hands them)
                sp = leg_stage_path(torch, L, ctx, df, cam, img, args, world, P, W, H, dev, barrier)
                fo["stage_path"] = {"value": sp["value"], "unit": "frames/s", "ms_per_step": sp["ms_per_step"],
                                    "deferred": sp["deferred"]["value"], "max_abs_diff_vs_fused": sp["max_abs_diff_vs_fused"]}
            out["file_order"] = fo
            r.bind_scene(*[d[k] for k in KEYS])
            del df, ref_img
    except Exception as e:  # noqa: BLE001 -- N > 1: whatever the hardware run throws belongs in the line
        if dist is None:
            raise
        leg["failed"] = True
        out.setdefault("leg_errors", {})["file_order"] = f"{type(e).__name__}: {e}"[:400]

    # ---- CPU baseline: the oracle (CPU restatement of the reference) on this box's host cores, rank 0, N = 1 only
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        leg_cpu_baseline_and_parity(out, scene, r, cam, img, P, W, H)
    emit()
    if coll is not None:
        barrier()
        coll.close()
    if dist is not None:
        dist.destroy_process_group()
    if watchdog is not None:
        watchdog.cancel()


if __name__ == "__main__":
    main()
