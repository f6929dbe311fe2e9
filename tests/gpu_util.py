"""Helpers shared by the `-m gpu` parity tests (torch is only the owner of device memory here)."""
import numpy as np
import torch

DEV = "cuda:0"


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV).contiguous()


def upload_scene(scene):
    return {k: dev(scene[k]) for k in ("pos", "scale", "rotq", "sh", "opacity")}


def assert_image_parity(gpu_img, orc, **_legacy):
    """The HIP frame must equal the oracle's BIT FOR BIT -- every pixel, no tolerance, no exempt pixels.

    BASELINE's bar is 1e-4 per-pixel L-inf; until round 3 this helper held every pixel to it except those whose
    oracle evaluation came within 1e-5 of a hard threshold (alpha < 1/255, T < 1e-4), because v_exp_f32 and libm's
    expf differ by an ulp or two and an ulp flips a threshold.  The blend's exp is now one defined sequence of binary32
    operations on both sides (gs_math.hpp::blend_exp = oracle/lcgs_oracle.c::orc_blend_exp), every other operation
    already was, so the comparison is exact equality (NaN pixels, where a test feeds non-finite inputs, must be NaN
    on both sides).  Returns (max |diff|, differing pixels) = (0.0, 0) for callers that report them."""
    ref = orc["img"]
    assert gpu_img.shape == ref.shape and gpu_img.dtype == ref.dtype == np.float32
    same = (gpu_img.view(np.uint32) == ref.view(np.uint32)) | ((gpu_img == ref))  # +0 / -0 compare equal
    both_nan = np.isnan(gpu_img) & np.isnan(ref)
    ok = same | both_nan
    if not ok.all():
        bad = ~ok.all(axis=0)
        with np.errstate(invalid="ignore"):
            diff = np.abs(gpu_img.astype(np.float64) - ref.astype(np.float64)).max(axis=0)
        amb = orc.get("ambig")
        n_amb = int((bad & amb.astype(bool)).sum()) if amb is not None else -1
        ys, xs = np.nonzero(bad)
        raise AssertionError(f"{int(bad.sum())} of {bad.size} pixels differ from the oracle (max |diff| "
                             f"{np.nanmax(diff[bad]):.3e}; {n_amb} of them threshold-ambiguous); first at "
                             f"(x={xs[0]}, y={ys[0]}): gpu {gpu_img[:, ys[0], xs[0]]} oracle {ref[:, ys[0], xs[0]]}")
    print(f"[parity] {ref.shape[2]}x{ref.shape[1]}: bit-identical to the oracle")
    return 0.0, 0
