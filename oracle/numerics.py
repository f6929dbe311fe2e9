"""How far does the frame move under the reference's LIKELY numerics, and is every moved pixel explained?
TEST INFRASTRUCTURE ONLY (same rule as the rest of oracle/).

The parity oracle (`Oracle("f32")`) evaluates the reference's expressions in source order with no contraction, IEEE
division and square root.  The reference's kernels are JIT-compiled by LuisaCompute (absent here); a CUDA JIT contracts
a*b+c by default and, under fast-math, forms a/b as a*rcp(b) and normalize / sqrt through rsqrt.  `render_variant` renders
the same frame under samples of those choices (VARIANTS: gcc's contraction pattern, reciprocal division, one-rounding
rsqrt, libm's expf, right-to-left sums -- samples, not replicas of any compiler).

A frame is a DISCONTINUOUS function of its per-splat values: a 1-ulp perturbation moves a few pixels by 1e-3 and the
rest by 1e-6.  `classify()` computes, from the checker's own evaluations (no reference frame needed), a per-pixel BOUND
of how far another valid f32 evaluation may land:
    bound = RSS_FACTOR x rss  +  flip  +  SENS_FLOOR
  rss   the continuous part, first order: every contributing entry k moves the pixel by at most T_k alpha_k |d power_k|
        (+ its colour's uncertainty); entries are independent, so the terms add as a root-sum-square.  |d power_k| is
        MEASURED, not assumed: the largest change of `power` at this pixel when the splat's record (pixel mean, conic) is
        replaced by any of the ENSEMBLE's evaluations of it -- the f64 twin and five numerics variants, signed, because an
        ill-conditioned splat's conic terms are each far larger than the power they cancel to -- plus the rounding of the
        pixel's own evaluation of gs_tile_splatter/shader.cpp:256 and of the mean at the size of its terms;
  flip  the discontinuous part: the sum, over the pixel's decisions that lie inside their rounding window, of what the
        decision going the other way would move the pixel by, each attributed to a class (`cls` bits, set when the move
        could exceed IMPACT_FLOOR):
          CLS_THRESHOLD  `power > 0` / `alpha < 1/255` / `T < 1e-4` (shader.cpp:256-265) within AMBIG_EPS + SENS_FACTOR x
                         the entry's own |d power| (for T: + the accumulated uncertainty of the factors in front, and every
                         ambiguous alpha skip in front): T alpha, resp. T;
          CLS_DEPTH      two adjacent contributors closer in depth than their depth windows (K_DEPTH ulp of the dot
                         product's terms, or the ensemble's spread): the global sort (gs_tile_splatter/impl.cpp:135-143)
                         may order them either way: T_i alpha_i alpha_j;
          CLS_RECT       a splat reaches the pixel through a tile that its radius `ceil(3 sqrt(lambda))` (shader.cpp:145-148)
                         or the rect's float -> uint edges (module.cpp:30-35) may or may not list: alpha.
`explain(img, other, cl)` holds a second frame against the bound: EVERY pixel must be within it, and the pixels beyond
1e-4 are attributed to the classes they carry.  Four of VARIANTS are not in the ensemble -- the independent check that the
measured uncertainties generalise: libm's expf, the THIRD grouping of every three- and four-term sum, and two combinations of
everything at once.  (Right-to-left sums were a hold-out until a 400-draw soak, tools/numerics_soak.py, showed what holding a
whole KIND of perturbation out costs: a screen-filling splat whose conic moves by 5e-4 under re-association and 1e-6 under
everything else flips an alpha skip far outside its window.  Each kind is now sampled once in the ensemble.)
"""
from __future__ import annotations

import numpy as np

from . import CLS_DEPTH, CLS_RECT, CLS_THRESHOLD, NUM_RCP_DIV, NUM_REASSOC, NUM_REASSOC2, NUM_RSQRT, Oracle

EPS32 = float(np.finfo(np.float32).eps)  # 2^-23
AMBIG_EPS = 1e-5   # the threshold window of rounds 4-5 (tests/gpu_util.py::LIBM_AMBIG_EPS): T carries every earlier factor's error
K_DEPTH = 2.0      # depth window per splat, in units of 2^-23 x the sum of |terms| of its dot product (the 4-term forward bound)
K_RECT = 4.0       # rounding window of the radius argument and of the rect's pixel coordinates, same units
K_EVAL = 2.0       # rounding of a pixel's own evaluation of `power` (shader.cpp:256), x 2^-23 x the sum of |terms|
K_DET = 2.0        # the analytic member of the ensemble: common relative error of the conic, x 2^-23 x the determinant's cancellation
K_MEAN = 0.5       # floor of a pixel mean's uncertainty, x 2^-23 x (|mean| + S / 2): about an ulp of the coordinate
K_FLOOR = 0.5      # floor of the measured colour spread, in ulp
SENS_FACTOR = 2.0  # an entry's threshold windows: its own uncertainty x this (the spread over five evaluations is a SAMPLE)
RSS_FACTOR = 3.0   # the continuous bound: root-sum-square of the entries' terms (independent splats) x this; the largest
                   # ratio observed over three variants x 2 M pixels of C3 is 1.86
SENS_FLOOR = 2e-6  # what any pixel may move by (accumulated last-bit noise of ~100 blended entries)
IMPACT_FLOOR = 1e-5  # a class bit is set when the decision's flip could move the pixel by more than this (a tenth of the bar)

VARIANTS = {  # name -> (contracted build, orc_set_numerics flags, libm expf in the blend)
    "contracted": (True, 0, False),
    "rcp_div": (False, NUM_RCP_DIV, False),
    "rsqrt": (False, NUM_RSQRT, False),
    "fast_math": (True, NUM_RCP_DIV | NUM_RSQRT, False),
    "reassociated": (False, NUM_REASSOC, False),
    # not in the ensemble that measures the per-splat uncertainties: the independent checks of the bound -- another exp, the
    # third grouping of every three- and four-term sum, and everything at once in two combinations
    "libm_expf": (False, 0, True),
    "reassociated_other_grouping": (False, NUM_REASSOC2, False),
    "fast_math_reassociated_libm": (True, NUM_RCP_DIV | NUM_RSQRT | NUM_REASSOC, True),
    "fast_math_other_grouping_libm": (True, NUM_RCP_DIV | NUM_RSQRT | NUM_REASSOC2, True),
}

_cache = {}


def oracle(precision="f32", contracted=False):
    key = (precision, contracted)
    if key not in _cache:
        _cache[key] = Oracle(precision, contracted=contracted)
    return _cache[key]


def _staged(o, scene, cam, scale_modifier, sh_deg, want_lists=True):
    """the frame's per-splat stages through the stage-level entry points (the same C functions orc_render chains)"""
    W, H = cam.width, cam.height
    color = o.sh_process(np.asarray(cam.position, o.dtype), scene["pos"], scene["sh"], deg=sh_deg)
    ndc, depth, cov = o.project(scene["pos"], scene["scale"], scene["rotq"], cam, scale_modifier=scale_modifier)
    pix, conic, tiles, radii = o.allocate_tiles(W, H, depth, ndc, cov)
    out = {"color": color, "ndc": ndc, "depth": depth, "cov": cov, "pix": pix, "conic": conic, "tiles": tiles,
           "radii": radii}
    if want_lists:
        offsets = o.inclusive_sum(tiles)
        keys, vals = o.copy_with_keys(W, H, pix, offsets, radii, depth)
        keys, vals = o.sort_pairs(keys, vals)
        G = ((W + 15) // 16) * ((H + 15) // 16)
        out.update(ranges=o.get_ranges(keys, G), point_list=vals, num_rendered=int(vals.shape[0]))
    return out


def _radius_arg(cov):
    """3 sqrt(max(lambda1, lambda2)) of gs_tile_splatter/shader.cpp:139-148 in the array's own precision (numpy's
    elementwise arithmetic rounds every operation: the oracle's expressions, term by term)"""
    t = cov.dtype.type
    cx, cy, cz = cov[:, 0] + t(0.3), cov[:, 1], cov[:, 2] + t(0.3)
    det = cx * cz - cy * cy
    mid = t(0.5) * (cx + cz)
    s = np.sqrt(np.maximum(t(0.1), mid * mid - det))
    return t(3.0) * np.sqrt(np.maximum(mid + s, mid - s))


ENSEMBLE = ("f64", "contracted", "rcp_div", "rsqrt", "fast_math", "reassociated")


def _variant_oracle(name):
    if name == "f64":
        return oracle("f64"), 0
    contracted, flags, _ = VARIANTS[name]
    return oracle("f32", contracted), flags


def uncertainties(scene, cam32, scale_modifier=1.0, sh_deg=3, st32=None, k_depth=K_DEPTH, k_rect=K_RECT,
                  ensemble=ENSEMBLE):
    """Per-splat uncertainty of what the sort and the blend read.  MEASURED, not assumed: the spread of the splat's own
    record (depth, pixel mean, conic, colour, radius argument) over the evaluations this checker can make of the same
    expressions -- the f64 twin and the numerics variants -- floored at half an ulp, plus worst-case rounding windows for
    the quantities that feed a discontinuity (depth order, rect).  A well-conditioned splat spreads by 1e-7 relative, a
    needle seen end-on (`cov2d` cancelling to 1e-3 of its terms) by 1e-4.
    -> dict(depth_tol[P], drec[P,V,5] (signed distance of every evaluation's pixel mean and conic from the parity
    oracle's), dmean_rect[P,2], dcolor[P], r_lo[P], r_hi[P], counts...)"""
    o32 = oracle("f32")
    W, H = cam32.width, cam32.height
    a = st32 if st32 is not None else _staged(o32, scene, cam32, scale_modifier, sh_deg, want_lists=False)
    vis = a["radii"] > 0
    P = vis.shape[0]
    f = lambda x: x.astype(np.float64)
    p32, c32, d32, col32, arg32 = f(a["pix"]), f(a["conic"]), f(a["depth"]), f(a["color"]), f(_radius_arg(a["cov"]))
    sp_depth, sp_arg, sp_col = np.zeros(P), np.zeros(P), np.zeros(P)
    sp_pix, sp_conic = np.zeros((P, 2)), np.zeros((P, 3))
    drec = np.zeros((P, len(ensemble) + 1, 5), np.float32)
    both = vis.copy()
    radii_differ = {}
    for k, name in enumerate(ensemble):
        o, flags = _variant_oracle(name)
        o.set_numerics(flags)
        try:
            b = _staged(o, scene, o.convert_camera(cam32), scale_modifier, sh_deg, want_lists=False)
        finally:
            o.set_numerics(0)
        ok = vis & (b["radii"] > 0)
        both &= ok
        z = lambda x: np.where(ok.reshape((-1,) + (1,) * (x.ndim - 1)), x, 0.0)
        drec[:, k, 0:2] = z(b["pix"] - p32)
        drec[:, k, 2:5] = z(b["conic"] - c32)
        sp_depth = np.maximum(sp_depth, z(np.abs(d32 - b["depth"])))
        sp_pix = np.maximum(sp_pix, z(np.abs(p32 - b["pix"])))
        sp_conic = np.maximum(sp_conic, z(np.abs(c32 - b["conic"])))
        sp_col = np.maximum(sp_col, np.abs(col32 - b["color"]).max(axis=1))
        sp_arg = np.maximum(sp_arg, z(np.abs(arg32 - _radius_arg(b["cov"]))))
        radii_differ[name] = int((ok & (a["radii"] != b["radii"])).sum())
    # ... and one ANALYTIC member: conic = (c, -b, a) / (a c - b^2) divides by a difference that cancels -- kappa = (a c + b^2) /
    # |a c - b^2| is 5 for a round footprint and 1e3-1e5 for a needle or a screen-filling splat -- so ANY f32 evaluation carries
    # a common relative error of about kappa ulp in all three components, whether or not one of the ensemble's six happens
    # to show it for this splat (tools/numerics_soak.py: two of 3 600 frames had a splat whose samples agreed to 1e-6 while a
    # seventh evaluation moved it by 3e-4 = 0.9 kappa eps).  A common factor moves `power` by that fraction of itself.
    cov = f(a["cov"])
    ca, cb, cc = cov[:, 0] + 0.3, cov[:, 1], cov[:, 2] + 0.3
    with np.errstate(invalid="ignore", divide="ignore"):
        kappa = np.where(vis, (ca * cc + cb * cb) / np.maximum(np.abs(ca * cc - cb * cb), 1e-300), 0.0)
    kappa = np.where(np.isfinite(kappa), kappa, 0.0)
    drec[:, len(ensemble), 2:5] = (c32 * (K_DET * EPS32 * kappa)[:, None]).astype(np.float32) * vis[:, None]
    # depth = front . p + tz (camera.h:38-51 as a matrix row): rounding scales with the TERMS, not with the result
    pos = f(scene["pos"])
    front = np.asarray(cam32.front, np.float64)
    tz = -float(np.dot(np.asarray(cam32.position, np.float64), front))
    terms = np.abs(pos * front).sum(axis=1) + abs(tz)
    depth_tol = np.maximum(k_depth * EPS32 * terms, sp_depth)
    dcolor = np.maximum(K_FLOOR * EPS32, sp_col)
    # the rect's window of the pixel mean: a worst-case bound (the edges are discontinuities)
    S = np.array([W, H], np.float64)
    dmean_rect = np.maximum(k_rect * EPS32 * (np.abs(p32) + S), sp_pix)
    dmean_rect[~vis] = 0.0
    # radius: ceil(arg) flips where arg sits within its window of an integer
    win = np.maximum(k_rect * EPS32 * np.abs(arg32), sp_arg)
    r = a["radii"].astype(np.int64)
    with np.errstate(invalid="ignore"):
        lo = np.where(np.isfinite(arg32), np.ceil(arg32 - win), r)
        hi = np.where(np.isfinite(arg32), np.ceil(arg32 + win), r)
    r_lo = np.clip(np.where(vis, np.minimum(r, lo), 0), 0, 2**31 - 1).astype(np.int64)
    r_hi = np.clip(np.where(vis, np.maximum(r, hi), 0), 0, 2**31 - 1).astype(np.int64)
    # the near cull (gs_projector/shader.cpp:121): a splat within its depth uncertainty of z = 0.2 may vanish or appear
    # (the projector leaves a culled splat's depth unwritten, so the test uses the f64 value of the same expression)
    near = np.abs((pos * front).sum(axis=1) + tz - 0.2) <= depth_tol
    near_in = near & vis
    r_lo[near_in] = 0
    ill = vis & (sp_conic.max(axis=1) > 1e-5 * np.abs(c32).max(axis=1))
    return {"depth_tol": depth_tol, "drec": drec, "dmean_rect": dmean_rect, "dcolor": dcolor,
            "r_lo": r_lo.astype(np.int32), "r_hi": r_hi.astype(np.int32), "visible": int(vis.sum()),
            "radius_window_splats": int((vis & (r_lo != r_hi)).sum()),
            "near_cull_window_visible": int(near_in.sum()), "near_cull_window_culled": int((near & ~vis).sum()),
            "ill_conditioned_splats": int(ill.sum()), "ensemble_radii_differ": radii_differ}


def classify(scene, cam32, bg=(0.0, 0.0, 0.0), scale_modifier=1.0, sh_deg=3, ambig_eps=AMBIG_EPS, **k):
    """The parity oracle's frame with the classes of its rounding-sensitive pixels.
    -> dict(img, num_rendered, radii, cls[H,W] (CLS_* bits), sens[H,W], counts{...})"""
    o32 = oracle("f32")
    W, H = cam32.width, cam32.height
    st = _staged(o32, scene, cam32, scale_modifier, sh_deg)
    u = uncertainties(scene, cam32, scale_modifier, sh_deg, st32=st, **k)
    if st["num_rendered"] == 0:  # nothing is drawn: the image is left untouched (gs_tile_splatter/impl.cpp:109), like orc_render
        z = np.zeros((H, W), np.float32)
        counts = {k: v for k, v in u.items() if isinstance(v, (int, dict))}
        counts.update(rect_uncertain_splats=0, pixels=W * H, threshold_pixels=0, depth_pixels=0, rect_pixels=0, flagged_pixels=0,
                      pixels_that_may_move_over_1e_4=0, of_them_by_conditioning_alone=0)
        return {"img": np.zeros((3, H, W), np.float32), "num_rendered": 0, "radii": st["radii"], "final_T": z.copy(),
                "n_contrib": np.zeros((H, W), np.uint32), "cls": np.zeros((H, W), np.uint8), "sens": z.copy(), "rss": z.copy(),
                "flip": z.copy(), "bound": np.full((H, W), SENS_FLOOR), "counts": counts}
    img, final_T, n_contrib, cls, sens, rss, flip = o32.render_forward_ex(
        W, H, bg, st["ranges"], st["point_list"], st["pix"], st["conic"], scene["opacity"], st["color"], ambig_eps,
        depth=st["depth"], depth_tol=u["depth_tol"], drec=u["drec"], dcolor=u["dcolor"],
        eval_eps=K_EVAL * EPS32, mean_eps=K_MEAN * EPS32, window_factor=SENS_FACTOR, impact_floor=IMPACT_FLOOR)
    n_rect = o32.mark_rect_uncertain(W, H, st["pix"], st["conic"], scene["opacity"], u["r_lo"], u["r_hi"], u["dmean_rect"],
                                     ambig_eps, IMPACT_FLOOR, cls, flip)
    bound = RSS_FACTOR * rss.astype(np.float64) + flip.astype(np.float64) + SENS_FLOOR
    counts = {k: v for k, v in u.items() if isinstance(v, (int, dict))}
    counts["rect_uncertain_splats"] = n_rect
    n = W * H
    counts.update(pixels=n, threshold_pixels=int(((cls & CLS_THRESHOLD) != 0).sum()),
                  depth_pixels=int(((cls & CLS_DEPTH) != 0).sum()), rect_pixels=int(((cls & CLS_RECT) != 0).sum()),
                  flagged_pixels=int((cls != 0).sum()),
                  pixels_that_may_move_over_1e_4=int((bound > 1e-4).sum()),
                  of_them_by_conditioning_alone=int((RSS_FACTOR * rss.astype(np.float64) + SENS_FLOOR > 1e-4).sum()))
    return {"img": img, "num_rendered": st["num_rendered"], "radii": st["radii"], "final_T": final_T,
            "n_contrib": n_contrib, "cls": cls, "sens": sens, "rss": rss, "flip": flip, "bound": bound, "counts": counts}


def render_variant(name, scene, cam32, bg=(0.0, 0.0, 0.0), scale_modifier=1.0, sh_deg=3):
    """the frame under one numerics variant (orc_render, one C call) -> Oracle.render's dict"""
    contracted, flags, libm = VARIANTS[name]
    o = oracle("f32", contracted)
    o.set_numerics(flags)
    o.set_blend_exp(libm)
    try:
        return o.render(scene, o.convert_camera(cam32), bg=bg, scale_modifier=scale_modifier, sh_deg=sh_deg)
    finally:
        o.set_numerics(0)
        o.set_blend_exp(False)


def explain(img, other, cl, bar=1e-4):
    """Hold |img - other| against the per-pixel bound of `cl` (classify's result for the frame `img` -- the parity oracle's
    = the HIP frame's, bit for bit): bound = RSS_FACTOR x rss (continuous: per-splat uncertainties, first order) + flip
    (the sum of what each ambiguous decision could move the pixel by) + SENS_FLOOR.  A pixel is EXPLAINED when its
    difference is within its bound; the pixels beyond `bar` are attributed to the class bits they carry."""
    with np.errstate(invalid="ignore"):
        diff = np.abs(img.astype(np.float64) - other.astype(np.float64)).max(axis=0)
    diff = np.where(np.isnan(img).any(axis=0) & np.isnan(other).any(axis=0), 0.0, diff)
    cls = cl["cls"]
    over = diff > bar
    unexplained = diff > cl["bound"]
    flagged = cls != 0
    out = {"pixels_over_1e-4": int(over.sum()), "pixels_over_1e-3": int((diff > 1e-3).sum()),
           "max_abs_diff": float(diff.max()),
           "over_threshold": int((over & ((cls & CLS_THRESHOLD) != 0)).sum()),
           "over_depth_order": int((over & ((cls & CLS_DEPTH) != 0)).sum()),
           "over_rect_radius": int((over & ((cls & CLS_RECT) != 0)).sum()),
           "over_conditioning_only": int((over & ~flagged & ~unexplained).sum()),
           "over_unexplained": int((over & unexplained).sum()),
           "unexplained_pixels": int(unexplained.sum()),
           "max_unexplained_excess": float((diff - cl["bound"])[unexplained].max()) if unexplained.any() else 0.0,
           "max_unflagged": float(diff[~flagged].max()) if (~flagged).any() else 0.0,
           "worst_ratio_diff_to_bound": float((diff / cl["bound"]).max())}
    out["all_explained"] = out["unexplained_pixels"] == 0
    return out


def report(scene, cam32, bg=(0.0, 0.0, 0.0), scale_modifier=1.0, sh_deg=3, variants=tuple(VARIANTS), img=None):
    """classify + every variant: the `parity.vs_contracted`-style block of bench.py and the -m gpu tests.  `img`: the frame to
    hold against the variants (the HIP frame); default the parity oracle's own."""
    cl = classify(scene, cam32, bg, scale_modifier, sh_deg)
    base = cl["img"] if img is None else img
    out = {"classes": cl["counts"], "variants": {}}
    for name in variants:
        v = render_variant(name, scene, cam32, bg, scale_modifier, sh_deg)
        e = explain(base, v["img"], cl)
        e["num_rendered"] = v["num_rendered"]
        e["num_rendered_base"] = cl["num_rendered"]
        e["radii_differ"] = int((v["radii"] != cl["radii"]).sum())
        out["variants"][name] = e
    return out, cl
