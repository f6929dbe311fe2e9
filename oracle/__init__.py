"""CPU oracle for the lcgs hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  The product package ``luisacomputegaussiansplatting_amd`` never does.

``Oracle("f32")`` is the parity oracle (and the reported CPU baseline); ``Oracle("f64")`` is the
same source compiled with ``real = double`` for finite-difference gradient checks.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = ("liblcgs_oracle_f32.so", "liblcgs_oracle_f64.so", "liblcgs_oracle_f32_contract.so")
# numerics variants (comparison runs only; lcgs_oracle.h) and the classes of rounding-sensitive pixels
NUM_RCP_DIV, NUM_RSQRT, NUM_REASSOC, NUM_REASSOC2 = 1, 2, 4, 8
CLS_THRESHOLD, CLS_DEPTH, CLS_RECT = 1, 2, 4


def build(force: bool = False) -> None:
    """Compile the C restatement (and oracle/_ref when /root/reference is present)."""
    need = force or not all(os.path.exists(os.path.join(_HERE, f)) for f in _LIBS)
    if not need:
        srcs = [os.path.join(_HERE, f) for f in ("lcgs_oracle.c", "lcgs_oracle_bwd.c", "lcgs_oracle.h")]
        newest = max(os.path.getmtime(s) for s in srcs)
        oldest = min(os.path.getmtime(os.path.join(_HERE, f)) for f in _LIBS)
        need = newest > oldest
    if need:
        subprocess.check_call(["make", "-C", _HERE, "all"], stdout=subprocess.DEVNULL)


def _ptr(a, ctype):
    if a is None:
        return None
    return a.ctypes.data_as(C.POINTER(ctype))


class Oracle:
    def __init__(self, precision: str = "f32", contracted: bool = False):
        """contracted=True loads the numerics VARIANT built with -ffp-contract=fast (f32 only; comparison runs)."""
        assert precision in ("f32", "f64") and not (contracted and precision != "f32")
        build()
        self.precision = precision
        self.contracted = contracted
        self.dtype = np.float32 if precision == "f32" else np.float64
        self.creal = C.c_float if precision == "f32" else C.c_double
        self.lib = C.CDLL(os.path.join(_HERE, f"liblcgs_oracle_{precision}{'_contract' if contracted else ''}.so"))
        assert self.lib.orc_sizeof_real() == np.dtype(self.dtype).itemsize
        assert bool(self.lib.orc_build_contracted()) == contracted
        self.lib.orc_mark_rect_uncertain.restype = C.c_int64
        creal = self.creal

        class Camera(C.Structure):
            _fields_ = [
                ("position", creal * 3),
                ("front", creal * 3),
                ("up", creal * 3),
                ("right", creal * 3),
                ("fov", creal),
                ("aspect_ratio", creal),
                ("width", C.c_int),
                ("height", C.c_int),
            ]

        self.Camera = Camera
        self.lib.orc_render.restype = C.c_int64
        self.lib.orc_tile_splatter_forward.restype = C.c_int64
        self.lib.orc_get_threads.restype = C.c_int
        if hasattr(self.lib, "orc_render_backward_full"):
            self.lib.orc_render_backward_full.restype = C.c_int64

    # ------------------------------------------------------------------ helpers
    def arr(self, a, shape=None):
        a = np.ascontiguousarray(np.asarray(a, dtype=self.dtype))
        if shape is not None:
            a = a.reshape(shape)
        return a

    def rp(self, a):
        return _ptr(a, self.creal)

    def convert_camera(self, cam):
        """the same camera (e.g. another Oracle's) in this library's struct: the host-side camera is common to every
        numerics variant, and the f64 twin gets the f32 camera's values widened"""
        return self.camera_from_dict(self.camera_to_dict(cam))

    def set_threads(self, n: int) -> None:
        self.lib.orc_set_threads(C.c_int(n))

    def set_smooth(self, on: bool) -> None:
        self.lib.orc_set_smooth(C.c_int(1 if on else 0))

    def set_blend_exp(self, use_libm: bool) -> None:
        """False (default): the build-defined blend exp the HIP kernels repeat bit for bit; True: libm's expf"""
        self.lib.orc_set_blend_exp(C.c_int(1 if use_libm else 0))

    def set_numerics(self, flags: int) -> None:
        """NUM_RCP_DIV | NUM_RSQRT: device-side divisions / square roots the way a fast-math JIT may form them; 0 = the
        parity oracle.  Comparison runs only."""
        self.lib.orc_set_numerics(C.c_int(flags))

    def blend_exp(self, x):
        """orc_blend_exp over an array of binary32 values (f32 build only)"""
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty_like(x)
        self.lib.orc_blend_exp_array(C.c_int64(x.size), _ptr(x, C.c_float), _ptr(out, C.c_float))
        return out

    def set_lod_min_radius(self, px: int) -> None:
        """the product's opt-in lcgs_set_lod rule restated for orc_render (0 = off, the reference's behaviour)"""
        self.lib.orc_set_lod_min_radius(C.c_int(px))

    def get_threads(self) -> int:
        return int(self.lib.orc_get_threads())

    # ------------------------------------------------------------------ camera
    def lookat(self, pos, target, world_up, width=None, height=None, fov=None):
        cam = self.Camera()
        p, t, u = (self.arr(v) for v in (pos, target, world_up))
        self.lib.orc_get_lookat_cam(self.rp(p), self.rp(t), self.rp(u), C.byref(cam))
        if width is not None:
            # app/main.cpp:204-207
            cam.width, cam.height = int(width), int(height)
            cam.aspect_ratio = self.dtype(np.float32(width) / np.float32(height))
        if fov is not None:
            cam.fov = fov
        return cam

    def camera_from_dict(self, d):
        cam = self.Camera()
        for k in ("position", "front", "up", "right"):
            for i in range(3):
                getattr(cam, k)[i] = float(d[k][i])
        cam.fov = float(d["fov"])
        cam.aspect_ratio = float(d["aspect_ratio"])
        cam.width = int(d["width"])
        cam.height = int(d["height"])
        return cam

    @staticmethod
    def camera_to_dict(cam):
        return {
            "position": [float(x) for x in cam.position],
            "front": [float(x) for x in cam.front],
            "up": [float(x) for x in cam.up],
            "right": [float(x) for x in cam.right],
            "fov": float(cam.fov),
            "aspect_ratio": float(cam.aspect_ratio),
            "width": int(cam.width),
            "height": int(cam.height),
        }

    def _mat(self, fn, *args):
        m = np.zeros(16, dtype=self.dtype)
        fn(*args, self.rp(m))
        return m.reshape(4, 4).T.copy()  # column-major storage -> math (row, col) indexing

    def local_to_world(self, cam):
        return self._mat(self.lib.orc_local_to_world_matrix, C.byref(cam))

    def world_to_local(self, cam):
        return self._mat(self.lib.orc_world_to_local_matrix, C.byref(cam))

    def projection(self, tanfovx, tanfovy, znear=0.1, zfar=100.0):
        r = self.creal
        return self._mat(self.lib.orc_projection_matrix, r(tanfovx), r(tanfovy), r(znear), r(zfar))

    # ------------------------------------------------------------------ stages
    def sh_eval_dir(self, deg, dirs, shs):
        dirs = self.arr(dirs, (-1, 3))
        shs = self.arr(shs, (dirs.shape[0], -1))
        out = np.zeros((dirs.shape[0], 3), dtype=self.dtype)
        for i in range(dirs.shape[0]):
            self.lib.orc_sh_eval_dir(C.c_int(deg), self.rp(dirs[i]), self.rp(shs[i]), self.rp(out[i]))
        return out

    def sh_process(self, campos, xyz, sh, deg=3, want_raw=False):
        xyz = self.arr(xyz, (-1, 3))
        P = xyz.shape[0]
        sh = self.arr(sh, (P, -1))
        campos = self.arr(campos)
        color = np.zeros((P, 3), dtype=self.dtype)
        raw = np.zeros((P, 3), dtype=self.dtype) if want_raw else None
        self.lib.orc_sh_process(C.c_int(P), C.c_int(3), C.c_int(deg), self.rp(campos), self.rp(xyz), self.rp(sh),
                                self.rp(color), self.rp(raw))
        return (color, raw) if want_raw else color

    def project(self, pos, scale, rotq, cam, scale_modifier=1.0, use_focal=True, init=None):
        pos = self.arr(pos, (-1, 3))
        P = pos.shape[0]
        scale = self.arr(scale, (P, 3))
        rotq = self.arr(rotq, (P, 4))
        if init is None:
            means_2d = np.zeros((P, 2), dtype=self.dtype)
            depth = np.zeros(P, dtype=self.dtype)
            covs_2d = np.zeros((P, 3), dtype=self.dtype)
        else:
            means_2d, depth, covs_2d = (self.arr(x).copy() for x in init)
        self.lib.orc_project_gs(C.c_int(P), self.rp(pos), self.rp(scale), self.rp(rotq), self.creal(scale_modifier),
                                self.rp(means_2d), self.rp(depth), self.rp(covs_2d), C.byref(cam),
                                C.c_int(1 if use_focal else 0))
        return means_2d, depth, covs_2d

    def allocate_tiles(self, width, height, depth, means_2d, covs_2d, use_focal=True):
        depth = self.arr(depth)
        P = depth.shape[0]
        means = self.arr(means_2d, (P, 2)).copy()
        covs = self.arr(covs_2d, (P, 3)).copy()
        tiles = np.zeros(P, dtype=np.uint32)
        radii = np.zeros(P, dtype=np.int32)
        self.lib.orc_allocate_tiles(C.c_int(P), C.c_int(width), C.c_int(height), self.rp(depth), self.rp(means),
                                    self.rp(covs), _ptr(tiles, C.c_uint32), _ptr(radii, C.c_int32),
                                    C.c_int(1 if use_focal else 0))
        return means, covs, tiles, radii

    def inclusive_sum(self, x):
        x = np.ascontiguousarray(x, dtype=np.uint32)
        out = np.zeros_like(x)
        self.lib.orc_inclusive_sum(C.c_int(x.shape[0]), _ptr(x, C.c_uint32), _ptr(out, C.c_uint32))
        return out

    def copy_with_keys(self, width, height, means_pix, offsets, radii, depth):
        depth = self.arr(depth)
        P = depth.shape[0]
        means = self.arr(means_pix, (P, 2))
        offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
        radii = np.ascontiguousarray(radii, dtype=np.int32)
        L = int(offsets[-1]) if P else 0
        keys = np.zeros(L, dtype=np.uint64)
        vals = np.zeros(L, dtype=np.uint32)
        self.lib.orc_copy_with_keys(C.c_int(P), C.c_int(width), C.c_int(height), self.rp(means),
                                    _ptr(offsets, C.c_uint32), _ptr(radii, C.c_int32), self.rp(depth),
                                    _ptr(keys, C.c_uint64), _ptr(vals, C.c_uint32))
        return keys, vals

    def sort_pairs(self, keys, vals):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        vals = np.ascontiguousarray(vals, dtype=np.uint32)
        ko = np.zeros_like(keys)
        vo = np.zeros_like(vals)
        self.lib.orc_sort_pairs(C.c_int64(keys.shape[0]), _ptr(keys, C.c_uint64), _ptr(vals, C.c_uint32),
                                _ptr(ko, C.c_uint64), _ptr(vo, C.c_uint32))
        return ko, vo

    def get_ranges(self, keys, n_tiles):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        ranges = np.zeros((n_tiles, 2), dtype=np.uint32)
        self.lib.orc_get_ranges(C.c_int64(keys.shape[0]), _ptr(keys, C.c_uint64), _ptr(ranges, C.c_uint32))
        return ranges

    def render_forward(self, width, height, bg, ranges, point_list, means_pix, conic, opacity, color,
                       ambig_eps=0.0):
        ranges = np.ascontiguousarray(ranges, dtype=np.uint32)
        point_list = np.ascontiguousarray(point_list, dtype=np.uint32)
        means = self.arr(means_pix)
        conic = self.arr(conic)
        opacity = self.arr(opacity)
        color = self.arr(color)
        bg = self.arr(bg)
        img = np.zeros((3, height, width), dtype=self.dtype)
        final_T = np.zeros((height, width), dtype=self.dtype)
        n_contrib = np.zeros((height, width), dtype=np.uint32)
        ambig = np.zeros((height, width), dtype=np.uint8)
        self.lib.orc_render_forward(C.c_int(width), C.c_int(height), self.rp(bg), _ptr(ranges, C.c_uint32),
                                    _ptr(point_list, C.c_uint32), self.rp(means), self.rp(conic), self.rp(opacity),
                                    self.rp(color), self.rp(img), self.rp(final_T), _ptr(n_contrib, C.c_uint32),
                                    _ptr(ambig, C.c_uint8), self.creal(ambig_eps))
        return img, final_T, n_contrib, ambig

    def render_forward_ex(self, width, height, bg, ranges, point_list, means_pix, conic, opacity, color, ambig_eps,
                          depth=None, depth_tol=None, drec=None, dcolor=None, eval_eps=0.0, mean_eps=0.0, window_factor=0.0,
                          impact_floor=0.0):
        """orc_render_forward_ex: the image plus, per pixel, the classes of rounding-sensitive decisions (cls) and the
        first-order bound of what the per-splat uncertainties can do to it (sens)."""
        ranges = np.ascontiguousarray(ranges, dtype=np.uint32)
        point_list = np.ascontiguousarray(point_list, dtype=np.uint32)
        a = lambda x: None if x is None else self.arr(x)
        means, conic, opacity, color, bg = (self.arr(x) for x in (means_pix, conic, opacity, color, bg))
        depth, depth_tol, drec, dcolor = a(depth), a(depth_tol), a(drec), a(dcolor)
        n_var = 0 if drec is None else drec.shape[1]
        assert drec is None or drec.shape == (opacity.shape[0], n_var, 5)
        img = np.zeros((3, height, width), dtype=self.dtype)
        final_T = np.zeros((height, width), dtype=self.dtype)
        n_contrib = np.zeros((height, width), dtype=np.uint32)
        cls = np.zeros((height, width), dtype=np.uint8)
        sens = np.zeros((height, width), dtype=self.dtype)
        flip = np.zeros((height, width), dtype=self.dtype)
        rss = np.zeros((height, width), dtype=self.dtype)
        self.lib.orc_render_forward_ex(C.c_int(width), C.c_int(height), self.rp(bg), _ptr(ranges, C.c_uint32),
                                       _ptr(point_list, C.c_uint32), self.rp(means), self.rp(conic), self.rp(opacity),
                                       self.rp(color), self.rp(img), self.rp(final_T), _ptr(n_contrib, C.c_uint32),
                                       _ptr(cls, C.c_uint8), self.creal(ambig_eps), self.rp(depth), self.rp(depth_tol),
                                       C.c_int(n_var), self.rp(drec), self.rp(dcolor), self.creal(eval_eps), self.creal(mean_eps),
                                       self.creal(window_factor),
                                       self.creal(impact_floor), self.rp(sens), self.rp(rss), self.rp(flip))
        return img, final_T, n_contrib, cls, sens, rss, flip

    def mark_rect_uncertain(self, width, height, means_pix, conic, opacity, r_lo, r_hi, dmean, eps, impact_floor, cls,
                            flip):
        """orc_mark_rect_uncertain: ORs CLS_RECT into cls, adds the reach to flip (both in place); returns the number of
        splats with an uncertain tile"""
        means, conic, opacity, dmean = (self.arr(x) for x in (means_pix, conic, opacity, dmean))
        r_lo = np.ascontiguousarray(r_lo, dtype=np.int32)
        r_hi = np.ascontiguousarray(r_hi, dtype=np.int32)
        assert cls.dtype == np.uint8 and cls.flags.c_contiguous and cls.shape == (height, width)
        assert flip.dtype == self.dtype and flip.flags.c_contiguous and flip.shape == (height, width)
        return int(self.lib.orc_mark_rect_uncertain(C.c_int(opacity.shape[0]), C.c_int(width), C.c_int(height),
                                                    self.rp(means), self.rp(conic), self.rp(opacity),
                                                    _ptr(r_lo, C.c_int32), _ptr(r_hi, C.c_int32), self.rp(dmean),
                                                    self.creal(eps), self.creal(impact_floor), _ptr(cls, C.c_uint8),
                                                    self.rp(flip)))

    def render(self, scene, cam, bg=(0.0, 0.0, 0.0), scale_modifier=1.0, sh_deg=3, ambig_eps=0.0):
        """Whole forward pipeline (app/main.cpp:266-308).  scene = dict(pos, scale, rotq, sh, opacity)."""
        pos = self.arr(scene["pos"], (-1, 3))
        P = pos.shape[0]
        scale = self.arr(scene["scale"], (P, 3))
        rotq = self.arr(scene["rotq"], (P, 4))
        sh = self.arr(scene["sh"], (P, -1))
        opacity = self.arr(scene["opacity"], (P,))
        bg = self.arr(bg)
        W, H = cam.width, cam.height
        img = np.zeros((3, H, W), dtype=self.dtype)
        radii = np.zeros(P, dtype=np.int32)
        final_T = np.zeros((H, W), dtype=self.dtype)
        n_contrib = np.zeros((H, W), dtype=np.uint32)
        ambig = np.zeros((H, W), dtype=np.uint8)
        L = self.lib.orc_render(C.c_int(P), C.c_int(sh_deg), self.rp(pos), self.rp(scale), self.rp(rotq), self.rp(sh),
                                self.rp(opacity), C.byref(cam), self.rp(bg), self.creal(scale_modifier), self.rp(img),
                                _ptr(radii, C.c_int32), self.rp(final_T), _ptr(n_contrib, C.c_uint32),
                                _ptr(ambig, C.c_uint8), self.creal(ambig_eps))
        return {"img": img, "radii": radii, "final_T": final_T, "n_contrib": n_contrib, "ambig": ambig,
                "num_rendered": int(L)}

    def image_to_rgb8(self, img_chw):
        img = self.arr(img_chw)
        _, H, W = img.shape
        out = np.zeros((H, W, 3), dtype=np.uint8)
        self.lib.orc_image_to_rgb8(C.c_int(W), C.c_int(H), self.rp(img), _ptr(out, C.c_uint8))
        return out

    # ------------------------------------------------------------------ backward
    def preprocess_backward(self, scene, cam, radii, dL_dmean2d, dL_dconic, dL_dcolor, scale_modifier=1.0, sh_deg=3):
        pos = self.arr(scene["pos"], (-1, 3))
        P = pos.shape[0]
        scale = self.arr(scene["scale"], (P, 3))
        rotq = self.arr(scene["rotq"], (P, 4))
        sh = self.arr(scene["sh"], (P, -1))
        radii = np.ascontiguousarray(radii, dtype=np.int32)
        gm, gc, gcol = self.arr(dL_dmean2d, (P, 2)), self.arr(dL_dconic, (P, 3)), self.arr(dL_dcolor, (P, 3))
        g = {"pos": np.zeros((P, 3), self.dtype), "scale": np.zeros((P, 3), self.dtype),
             "rotq": np.zeros((P, 4), self.dtype), "sh": np.zeros_like(sh)}
        self.lib.orc_preprocess_backward(C.c_int(P), C.c_int(sh_deg), self.rp(pos), self.rp(scale), self.rp(rotq),
                                         self.rp(sh), C.byref(cam), self.creal(scale_modifier), _ptr(radii, C.c_int32),
                                         self.rp(gm), self.rp(gc), self.rp(gcol), self.rp(g["pos"]), self.rp(g["scale"]),
                                         self.rp(g["rotq"]), self.rp(g["sh"]))
        return g

    def render_backward(self, width, height, bg, ranges, point_list, means_pix, conic, opacity, color, final_T, n_contrib,
                        dL_dimg):
        """render half of the backward on 2-D arrays (orc_render_backward): -> dL/d{pixel mean, conic, opacity, colour}"""
        n = self.arr(opacity).shape[0]
        rng = np.ascontiguousarray(ranges, dtype=np.uint32)
        pl = np.ascontiguousarray(point_list, dtype=np.uint32)
        nc = np.ascontiguousarray(n_contrib, dtype=np.uint32)
        mp, cn, op, col, fT, dL, bga = (self.arr(x) for x in (means_pix, conic, opacity, color, final_T, dL_dimg, bg))
        gm, gc, go, gcol = (np.zeros(s, self.dtype) for s in ((n, 2), (n, 3), (n,), (n, 3)))
        self.lib.orc_render_backward(C.c_int(width), C.c_int(height), self.rp(bga), _ptr(rng, C.c_uint32),
                                     _ptr(pl, C.c_uint32), self.rp(mp), self.rp(cn), self.rp(op), self.rp(col), self.rp(fT),
                                     _ptr(nc, C.c_uint32), self.rp(dL), self.rp(gm), self.rp(gc), self.rp(go), self.rp(gcol))
        return gm, gc, go, gcol

    def render_backward_full(self, scene, cam, dL_dimg, bg=(0.0, 0.0, 0.0), scale_modifier=1.0, sh_deg=3):
        pos = self.arr(scene["pos"], (-1, 3))
        P = pos.shape[0]
        scale = self.arr(scene["scale"], (P, 3))
        rotq = self.arr(scene["rotq"], (P, 4))
        sh = self.arr(scene["sh"], (P, -1))
        opacity = self.arr(scene["opacity"], (P,))
        bg = self.arr(bg)
        W, H = cam.width, cam.height
        dL = self.arr(dL_dimg, (3, H, W))
        img = np.zeros((3, H, W), dtype=self.dtype)
        g = {
            "pos": np.zeros((P, 3), dtype=self.dtype),
            "scale": np.zeros((P, 3), dtype=self.dtype),
            "rotq": np.zeros((P, 4), dtype=self.dtype),
            "sh": np.zeros_like(sh),
            "opacity": np.zeros(P, dtype=self.dtype),
        }
        L = self.lib.orc_render_backward_full(
            C.c_int(P), C.c_int(sh_deg), self.rp(pos), self.rp(scale), self.rp(rotq), self.rp(sh), self.rp(opacity),
            C.byref(cam), self.rp(bg), self.creal(scale_modifier), self.rp(dL), self.rp(img),
            self.rp(g["pos"]), self.rp(g["scale"]), self.rp(g["rotq"]), self.rp(g["sh"]), self.rp(g["opacity"]))
        g["img"] = img
        g["num_rendered"] = int(L)
        return g


class Ref:
    """oracle/_ref/liblcgs_ref.so -- pieces of the real reference that build from their own sources.
    Exists only in the authoring container (never on the GPU box)."""

    def __init__(self):
        path = os.path.join(_HERE, "_ref", "liblcgs_ref.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = C.CDLL(path)
        self.lib.ref_ply_vertex_count.restype = C.c_int64

    @staticmethod
    def available() -> bool:
        return os.path.exists(os.path.join(_HERE, "_ref", "liblcgs_ref.so"))

    def sh_eval_dir(self, deg, dirs, shs):
        dirs = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
        shs = np.ascontiguousarray(shs, dtype=np.float32).reshape(dirs.shape[0], -1)
        full = np.zeros((dirs.shape[0], 48), dtype=np.float32)
        full[:, : shs.shape[1]] = shs
        out = np.zeros((dirs.shape[0], 3), dtype=np.float32)
        for i in range(dirs.shape[0]):
            self.lib.ref_sh_eval_dir(C.c_int(deg), _ptr(dirs[i], C.c_float), _ptr(full[i], C.c_float),
                                     _ptr(out[i], C.c_float))
        return out

    def sh_backward_coeffs(self, deg, dirs, dL_dcolor):
        dirs = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
        g = np.ascontiguousarray(dL_dcolor, dtype=np.float32).reshape(-1, 3)
        out = np.zeros((dirs.shape[0], 16, 3), dtype=np.float32)
        for i in range(dirs.shape[0]):
            self.lib.ref_sh_backward_coeffs(C.c_int(deg), _ptr(dirs[i], C.c_float), _ptr(g[i], C.c_float),
                                            _ptr(out[i], C.c_float))
        return out

    def ply_columns(self, path, names):
        n = int(self.lib.ref_ply_vertex_count(path.encode()))
        if n < 0:
            raise RuntimeError("happly failed to open " + path)
        cols = {}
        for name in names:
            buf = np.zeros(n, dtype=np.float32)
            rc = self.lib.ref_ply_read_column(path.encode(), name.encode(), _ptr(buf, C.c_float), C.c_int64(n))
            if rc != 0:
                raise RuntimeError(f"happly failed on property {name}: rc={rc}")
            cols[name] = buf
        return n, cols
