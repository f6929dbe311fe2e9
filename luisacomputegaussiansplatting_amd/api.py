"""ctypes binding of liblcgs_hip.so + the Python mirror of the reference's operator classes.

Reference interfaces mirrored here (paths relative to the reference repository):
  * ``Camera`` / ``get_lookat_cam`` ...............  lcgs/include/lcgs/util/camera.h:15-82
  * ``SHProcessor.process`` .......................  lcgs/include/lcgs/sh_preprocessor.h:30-37
  * ``GSProjector.forward`` + its two proxies .....  lcgs/include/lcgs/gs_projector.h:16-43
  * ``GSTileSplatter.forward`` + its three proxies   lcgs/include/lcgs/gs_tile_splatter.h:28-35, proxy.h:43-71
  * ``read_gs_ply`` ...............................  app/gaussians.cpp:75-171
  * ``Renderer`` (the per-frame loop body) ........  app/main.cpp:266-308

Device buffers are torch CUDA(=HIP) tensors; only their ``data_ptr()`` crosses the C ABI.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "liblcgs_hip.so"
_lib = None


class LcgsError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"lcgs status {status}: {message}")
        self.status = status


def library_path() -> str:
    return os.path.join(_PKG_DIR, _LIB_NAME)


def build_library(force: bool = False) -> str:
    """Compile every HIP source for gfx950 into liblcgs_hip.so (in-tree).  Works without a GPU."""
    args = ["make", "-C", os.path.join(_PKG_DIR, "csrc"), "-j8"]
    if force:
        subprocess.check_call(args + ["clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return library_path()


class _SelftestReport(C.Structure):
    """lcgs_comm_selftest_report"""

    _fields_ = [("world_size", C.c_int), ("rank", C.c_int), ("allreduce_ok", C.c_int), ("allreduce_ms", C.c_double),
                ("p2p_ok", C.c_int), ("p2p_ms", C.c_double), ("owner_step_ok", C.c_int), ("owner_step_ms", C.c_double),
                ("owner_max_grad_err", C.c_double), ("timed_out", C.c_int), ("message", C.c_char * 256)]


class Camera(C.Structure):
    """struct Camera, lcgs/include/lcgs/util/camera.h:15-25"""

    _fields_ = [
        ("position", C.c_float * 3),
        ("front", C.c_float * 3),
        ("up", C.c_float * 3),
        ("right", C.c_float * 3),
        ("fov", C.c_float),
        ("aspect_ratio", C.c_float),
        ("width", C.c_int),
        ("height", C.c_int),
    ]

    def to_dict(self):
        return {
            "position": list(self.position), "front": list(self.front), "up": list(self.up),
            "right": list(self.right), "fov": float(self.fov), "aspect_ratio": float(self.aspect_ratio),
            "width": int(self.width), "height": int(self.height),
        }

    @staticmethod
    def from_dict(d) -> "Camera":
        cam = Camera()
        for k in ("position", "front", "up", "right"):
            for i in range(3):
                getattr(cam, k)[i] = float(d[k][i])
        cam.fov = float(d["fov"])
        cam.aspect_ratio = float(d["aspect_ratio"])
        cam.width = int(d["width"])
        cam.height = int(d["height"])
        return cam


class _TileAccel(C.Structure):
    _fields_ = [
        ("tiles_touched", C.c_void_p), ("point_offsets", C.c_void_p), ("point_list_keys_unsorted", C.c_void_p),
        ("point_list_unsorted", C.c_void_p), ("point_list_keys", C.c_void_p), ("point_list", C.c_void_p),
        ("ranges", C.c_void_p), ("capacity", C.c_int64),
    ]


class _TileInput(C.Structure):
    _fields_ = [
        ("num_gaussians", C.c_int), ("bg_color", C.c_float * 3), ("means_2d", C.c_void_p),
        ("depth_features", C.c_void_p), ("conic", C.c_void_p), ("color_features", C.c_void_p),
        ("opacity_features", C.c_void_p),
    ]


class _TileOutput(C.Structure):
    _fields_ = [
        ("height", C.c_int), ("width", C.c_int), ("target_img", C.c_void_p), ("radii", C.c_void_p),
        ("final_T", C.c_void_p), ("n_contrib", C.c_void_p),
    ]


class _StageTimes(C.Structure):
    _fields_ = [("count", C.c_int), ("name", C.c_char_p * 16), ("ms", C.c_float * 16)]


class _FrameStats(C.Structure):
    _fields_ = [("num_gaussians", C.c_int64), ("num_visible", C.c_int64), ("num_rendered", C.c_int64),
                ("num_pairs", C.c_int64), ("num_tiles", C.c_int64), ("equal_depth_unresolved", C.c_int64),
                ("list_shift", C.c_int64)]


class _Grads(C.Structure):
    _fields_ = [("d_dL_dpos", C.c_void_p), ("d_dL_dscale", C.c_void_p), ("d_dL_drotq", C.c_void_p),
                ("d_dL_dsh", C.c_void_p), ("d_dL_dopacity", C.c_void_p)]


class _Params(C.Structure):
    _fields_ = [("pos", C.c_void_p), ("scale", C.c_void_p), ("rotq", C.c_void_p), ("sh", C.c_void_p),
                ("opacity", C.c_void_p)]


class _AdamConfig(C.Structure):
    _fields_ = [("lr_pos", C.c_float), ("lr_sh_dc", C.c_float), ("lr_sh_rest", C.c_float), ("lr_opacity", C.c_float),
                ("lr_scale", C.c_float), ("lr_rot", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float),
                ("eps", C.c_float), ("step", C.c_int), ("visible_only", C.c_int)]


class _SceneHost(C.Structure):
    _fields_ = [("num_gaussians", C.c_int), ("sh_degree", C.c_int), ("pos", C.POINTER(C.c_float)),
                ("feature", C.POINTER(C.c_float)), ("opacity", C.POINTER(C.c_float)),
                ("scale", C.POINTER(C.c_float)), ("rotq", C.POINTER(C.c_float))]


# every symbol include/lcgs_hip.h declares (checked by tests/test_abi.py)
EXPORTED_SYMBOLS = [
    "lcgs_version", "lcgs_last_error", "lcgs_create", "lcgs_destroy", "lcgs_set_stream", "lcgs_synchronize",
    "lcgs_get_lookat_cam", "lcgs_local_to_world_matrix", "lcgs_world_to_local_matrix", "lcgs_projection_matrix",
    "lcgs_sh_process", "lcgs_project_forward", "lcgs_tile_splat_forward", "lcgs_set_stage_mode", "lcgs_stage_flush",
    "lcgs_inclusive_sum_u32",
    "lcgs_sort_pairs_u64_u32", "lcgs_scene_bind", "lcgs_scene_upload", "lcgs_render_forward",
    "lcgs_set_profiling", "lcgs_get_stage_times", "lcgs_get_frame_stats", "lcgs_debug_last_lists", "lcgs_debug_last_state", "lcgs_set_list_policy", "lcgs_debug_blend_exp",
    "lcgs_render_backward", "lcgs_render_backward_adam", "lcgs_fit_views", "lcgs_render_backward_accumulate", "lcgs_render_backward_compact", "lcgs_visible_rows", "lcgs_ply_read", "lcgs_ply_write_raw", "lcgs_scene_host_free", "lcgs_synth_scene",
    "lcgs_image_to_rgb8", "lcgs_image_to_rgb8_device", "lcgs_write_png", "lcgs_l2_loss_backward",
    "lcgs_scene_load_ply", "lcgs_scene_pointers", "lcgs_scene_download", "lcgs_scene_reorder_spatial", "lcgs_adam_step",
    "lcgs_render_forward_batch", "lcgs_scene_use_half_sh", "lcgs_scene_modified", "lcgs_debug_verify_derived",
    "lcgs_comm_owner_rows", "lcgs_owner_step_forward", "lcgs_owner_step_backward", "lcgs_owner_step_set_async",
    "lcgs_owner_step_finish", "lcgs_comm_selftest", "lcgs_loopback_group_create",
    "lcgs_loopback_group_destroy", "lcgs_comm_create_loopback",
    "lcgs_set_ingest_order", "lcgs_scene_permutation", "lcgs_set_lod", "lcgs_comm_unique_id", "lcgs_comm_create", "lcgs_comm_destroy", "lcgs_comm_info", "lcgs_comm_set_transport", "lcgs_comm_shard_rows",
    "lcgs_grads_allreduce", "lcgs_adam_step_sharded", "lcgs_comm_track_touched_rows", "lcgs_comm_get_stats",
    "lcgs_adam_step_sparse", "lcgs_sparse_touched_rows", "lcgs_sparse_message_words", "lcgs_sparse_pack",
    "lcgs_sparse_accumulate", "lcgs_scene_declare_static", "lcgs_owner_project", "lcgs_owner_project_views", "lcgs_owner_counts", "lcgs_owner_render", "lcgs_owner_render_backward", "lcgs_owner_backward",
]


def load_library():
    """dlopen liblcgs_hip.so.  Raises (never falls back) if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: torch ships its own libamdhip64, and a process that initialises two copies loses
    # the device in the second.  When torch is installed it is imported first, so that liblcgs_hip.so (linked against
    # the same soname) binds to the copy torch already loaded.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = library_path()
    if not os.path.exists(path):
        raise LcgsError(-1, f"{path} is missing: run luisacomputegaussiansplatting_amd.build_library() "
                            f"(or __graft_entry__.build()); there is no CPU fallback")
    lib = C.CDLL(path)
    lib.lcgs_version.restype = C.c_char_p
    lib.lcgs_last_error.restype = C.c_char_p
    for name in EXPORTED_SYMBOLS:
        fn = getattr(lib, name)
        if name not in ("lcgs_version", "lcgs_last_error", "lcgs_get_lookat_cam", "lcgs_local_to_world_matrix",
                        "lcgs_world_to_local_matrix", "lcgs_projection_matrix", "lcgs_scene_host_free",
                        "lcgs_image_to_rgb8", "lcgs_comm_shard_rows", "lcgs_sparse_message_words", "lcgs_comm_owner_rows"):
            fn.restype = C.c_int
    lib.lcgs_get_lookat_cam.restype = None
    lib.lcgs_local_to_world_matrix.restype = None
    lib.lcgs_world_to_local_matrix.restype = None
    lib.lcgs_projection_matrix.restype = None
    lib.lcgs_scene_host_free.restype = None
    lib.lcgs_image_to_rgb8.restype = None
    lib.lcgs_comm_shard_rows.restype = None
    lib.lcgs_comm_owner_rows.restype = None
    lib.lcgs_sparse_message_words.restype = C.c_int64
    lib.lcgs_sparse_message_words.argtypes = [C.c_int64, C.c_int]
    lib.lcgs_comm_shard_rows.argtypes = [C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.lcgs_projection_matrix.argtypes = [C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float)]
    _lib = lib
    return lib


def _check(status: int):
    if status != 0:
        raise LcgsError(status, load_library().lcgs_last_error().decode(errors="replace"))


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def _ptr(t) -> C.c_void_p:
    """device (or host) pointer of a torch tensor / numpy array / int / None"""
    if t is None:
        return C.c_void_p(0)
    if isinstance(t, int):
        return C.c_void_p(t)
    if hasattr(t, "data_ptr"):
        assert t.is_contiguous(), "buffers crossing the C ABI must be contiguous"
        return C.c_void_p(t.data_ptr())
    if isinstance(t, np.ndarray):
        assert t.flags["C_CONTIGUOUS"]
        return C.c_void_p(t.ctypes.data)
    raise TypeError(type(t))


# ---------------------------------------------------------------------------------------------- camera
def get_lookat_cam(pos, target, world_up, width: Optional[int] = None, height: Optional[int] = None,
                   fov: Optional[float] = None) -> Camera:
    """get_lookat_cam (camera.h:74-82); width/height additionally apply app/main.cpp:204-207."""
    cam = Camera()
    load_library().lcgs_get_lookat_cam(_f3(pos), _f3(target), _f3(world_up), C.byref(cam))
    if width is not None:
        cam.width, cam.height = int(width), int(height)
        cam.aspect_ratio = float(np.float32(width) / np.float32(height))
    if fov is not None:
        cam.fov = float(fov)
    return cam


def _mat(fn, *args) -> np.ndarray:
    m = (C.c_float * 16)()
    fn(*args, m)
    return np.array(m, dtype=np.float32).reshape(4, 4).T.copy()  # (row, col) indexing


def local_to_world_matrix(cam: Camera) -> np.ndarray:
    return _mat(load_library().lcgs_local_to_world_matrix, C.byref(cam))


def world_to_local_matrix(cam: Camera) -> np.ndarray:
    return _mat(load_library().lcgs_world_to_local_matrix, C.byref(cam))


def projection_matrix(tanfovx: float, tanfovy: float, znear: float = 0.1, zfar: float = 100.0) -> np.ndarray:
    return _mat(load_library().lcgs_projection_matrix, C.c_float(tanfovx), C.c_float(tanfovy), C.c_float(znear),
                C.c_float(zfar))


# ---------------------------------------------------------------------------------------------- context
class Context:
    """One (GPU, stream) pair: Context::create_device + create_stream of app/main.cpp:162-163."""

    def __init__(self, device_id: int = 0, stream: Optional[int] = None):
        lib = load_library()
        if stream is None:
            try:
                import torch

                if torch.cuda.is_available():
                    stream = torch.cuda.current_stream(device_id).cuda_stream
            except ImportError:
                stream = 0
        self._h = C.c_void_p(0)
        _check(lib.lcgs_create(C.c_int(device_id), C.c_void_p(stream or 0), C.byref(self._h)))
        self.device_id = device_id

    def close(self):
        if self._h:
            load_library().lcgs_destroy(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _check(load_library().lcgs_synchronize(self._h))

    def set_stream(self, stream: int):
        _check(load_library().lcgs_set_stream(self._h, C.c_void_p(stream)))

    def set_stage_mode(self, mode: str):
        """lcgs_set_stage_mode: "exact" (default: every operator runs at once and leaves the reference's buffers behind) or
        "deferred" (process / forward are recorded, a matching splatter call renders the fused frame instead)"""
        _check(load_library().lcgs_set_stage_mode(self._h, C.c_int({"exact": 0, "deferred": 1}[mode])))

    def stage_flush(self):
        _check(load_library().lcgs_stage_flush(self._h))

    def blend_exp(self, d_x, d_out, n: int):
        """diagnostics: the compositing loop's exp (gs_math.hpp::blend_exp) over n device values"""
        _check(load_library().lcgs_debug_blend_exp(self._h, _ptr(d_x), _ptr(d_out), C.c_int64(n)))

    # lcpp primitives
    def inclusive_sum(self, d_in, d_out, n: int):
        _check(load_library().lcgs_inclusive_sum_u32(self._h, _ptr(d_in), _ptr(d_out), C.c_int64(n)))

    def sort_pairs(self, keys_in, keys_out, vals_in, vals_out, n: int, begin_bit: int = 0, end_bit: int = 64):
        _check(load_library().lcgs_sort_pairs_u64_u32(self._h, _ptr(keys_in), _ptr(keys_out), _ptr(vals_in),
                                                      _ptr(vals_out), C.c_int64(n), C.c_int(begin_bit),
                                                      C.c_int(end_bit)))


_default_ctx: Optional[Context] = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


# ---------------------------------------------------------------------------------------------- proxies
@dataclass
class GPUPointsProxy:  # sh_preprocessor.h:16-20
    N: int = 0
    stride: int = 3
    pos: object = None


@dataclass
class GSProjectorInputProxy:  # gs_projector.h:16-22
    num_gaussians: int
    pos: object
    scale: object
    rotq: object
    scale_modifier: float = 1.0


@dataclass
class GSProjectorOutputProxy:  # gs_projector.h:24-28
    means_2d: object
    covs_2d: object
    depth: object


@dataclass
class GSTileSplatterInputProxy:  # proxy.h:43-54
    num_gaussians: int
    bg_color: tuple
    means_2d: object
    depth_features: object
    conic: object
    color_features: object
    opacity_features: object


@dataclass
class GSTileSplatterAccelProxy:  # proxy.h:56-64
    tiles_touched: object
    point_offsets: object
    point_list_keys_unsorted: object
    point_list_unsorted: object
    point_list_keys: object
    point_list: object
    ranges: object


@dataclass
class GSSplatForwardOutputProxy:  # proxy.h:66-71
    height: int
    width: int
    target_img: object
    radii: object
    final_T: object = None
    n_contrib: object = None


# ---------------------------------------------------------------------------------------------- operators
class SHProcessor:
    """lcgs::SHProcessor (sh_preprocessor.h:22-57)."""

    def __init__(self):
        self.ctx: Optional[Context] = None

    def create(self, ctx: Optional[Context] = None):
        self.ctx = ctx or default_context()

    def process(self, proxy: GPUPointsProxy, camera: Camera, sh, color, channel: int = 3, level: int = 3):
        if self.ctx is None:
            raise LcgsError(-1, "SHProcessor.create() was not called")
        _check(load_library().lcgs_sh_process(self.ctx._h, C.c_int(proxy.N), _ptr(proxy.pos), C.byref(camera),
                                              _ptr(sh), _ptr(color), C.c_int(level), C.c_int(channel)))


class GSProjector:
    """lcgs::GSProjector (gs_projector.h:30-87)."""

    def __init__(self):
        self.ctx: Optional[Context] = None

    def create(self, ctx: Optional[Context] = None):
        self.ctx = ctx or default_context()

    def forward(self, input: GSProjectorInputProxy, output: GSProjectorOutputProxy, cam: Camera,
                use_focal: bool = True):
        if self.ctx is None:
            raise LcgsError(-1, "GSProjector.create() was not called")
        _check(load_library().lcgs_project_forward(
            self.ctx._h, C.c_int(input.num_gaussians), _ptr(input.pos), _ptr(input.scale), _ptr(input.rotq),
            C.c_float(input.scale_modifier), _ptr(output.means_2d), _ptr(output.covs_2d), _ptr(output.depth),
            C.byref(cam), C.c_int(1 if use_focal else 0)))


class GSTileSplatter:
    """lcgs::GSTileSplatter (gs_tile_splatter.h:19-106)."""

    m_blocks = (16, 16)  # module.h:17

    def __init__(self):
        self.ctx: Optional[Context] = None
        self.num_rendered = 0  # gs_tile_splatter.h:23

    def create(self, ctx: Optional[Context] = None):
        self.ctx = ctx or default_context()

    def forward(self, accel: GSTileSplatterAccelProxy, input: GSTileSplatterInputProxy,
                output: GSSplatForwardOutputProxy, use_focal: bool = True) -> int:
        if self.ctx is None:
            raise LcgsError(-1, "GSTileSplatter.create() was not called")
        cap = int(accel.point_list.numel()) if hasattr(accel.point_list, "numel") else int(accel.point_list.size)
        a = _TileAccel(_ptr(accel.tiles_touched), _ptr(accel.point_offsets), _ptr(accel.point_list_keys_unsorted),
                       _ptr(accel.point_list_unsorted), _ptr(accel.point_list_keys), _ptr(accel.point_list),
                       _ptr(accel.ranges), cap)
        i = _TileInput(int(input.num_gaussians), _f3(input.bg_color), _ptr(input.means_2d),
                       _ptr(input.depth_features), _ptr(input.conic), _ptr(input.color_features),
                       _ptr(input.opacity_features))
        o = _TileOutput(int(output.height), int(output.width), _ptr(output.target_img), _ptr(output.radii),
                        _ptr(output.final_T), _ptr(output.n_contrib))
        n = C.c_int(0)
        _check(load_library().lcgs_tile_splat_forward(self.ctx._h, C.byref(a), C.byref(i), C.byref(o),
                                                      C.c_int(1 if use_focal else 0), C.byref(n)))
        self.num_rendered = n.value
        return n.value


class Renderer:
    """The fused per-frame path: what app/main.cpp:266-308 does (SHProcessor.process + GSProjector.forward +
    GSTileSplatter.forward) as one stream submission."""

    def __init__(self, ctx: Optional[Context] = None):
        self.ctx = ctx or default_context()
        self._keep = []  # keeps bound tensors alive
        self.P = 0
        self.sh_degree = 3
        # bumped by everything that replaces the state lcgs_render_backward works from (scene binding, a new frame):
        # render_autograd's backward compares it with the value its own forward left behind
        self._generation = 0

    def bind_scene(self, pos, scale, rotq, sh, opacity, sh_degree: int = 3):
        P = int(pos.shape[0])
        self._keep = [pos, scale, rotq, sh, opacity]
        self.P, self.sh_degree = P, sh_degree
        self._generation += 1
        _check(load_library().lcgs_scene_bind(self.ctx._h, C.c_int(P), C.c_int(sh_degree), _ptr(pos), _ptr(scale),
                                              _ptr(rotq), _ptr(sh), _ptr(opacity)))

    def declare_static(self, pos=None, scale=None, rotq=None):
        """lcgs_scene_declare_static: caller-owned arrays that do not change between frames get the cull pass's 16-byte
        {position, extent bound} rows (what a context-owned scene has by itself).  No arguments: withdraw."""
        if pos is None:
            _check(load_library().lcgs_scene_declare_static(self.ctx._h, C.c_int(0), None, None, None))
            return
        self._static = [pos, scale, rotq]
        _check(load_library().lcgs_scene_declare_static(self.ctx._h, C.c_int(int(pos.shape[0])), _ptr(pos), _ptr(scale),
                                                        _ptr(rotq)))

    def reorder_scene_spatial(self):
        """lcgs_scene_reorder_spatial: the context re-orders its scene along a Morton curve and renders from its own
        copy from now on.  Returns the permutation (device int32 tensor): new splat r = old splat perm[r]."""
        import torch

        perm = torch.empty(self.P, dtype=torch.int32, device=f"cuda:{self.ctx.device_id}")
        _check(load_library().lcgs_scene_reorder_spatial(self.ctx._h, _ptr(perm)))
        self._generation += 1
        self._keep = None  # the caller's arrays are no longer read
        return perm

    def upload_scene(self, scene: dict, sh_degree: int = 3, order: Optional[str] = None):
        """lcgs_scene_upload: host arrays -> device copies owned by the context, kept in spatial order by default
        (order="file" keeps the given order); see permutation() / scene_tensors()."""
        if order is not None:
            _check(load_library().lcgs_set_ingest_order(self.ctx._h, C.c_int({"file": 0, "spatial": 1}[order])))
        arrs = [np.ascontiguousarray(scene[k], dtype=np.float32) for k in ("pos", "scale", "rotq", "sh", "opacity")]
        P = int(arrs[0].reshape(-1, 3).shape[0])
        self.P, self.sh_degree = P, sh_degree
        self._generation += 1
        _check(load_library().lcgs_scene_upload(self.ctx._h, C.c_int(P), C.c_int(sh_degree), *[_ptr(a) for a in arrs]))

    def load_ply(self, path: str, order: Optional[str] = None) -> int:
        """read_gs_ply + upload with the de-interleave / activations on the device (lcgs_scene_load_ply).  order:
        "spatial" (the library's default: the context keeps its scene along a Morton curve) or "file"."""
        if order is not None:
            _check(load_library().lcgs_set_ingest_order(self.ctx._h, C.c_int({"file": 0, "spatial": 1}[order])))
        n = C.c_int(0)
        _check(load_library().lcgs_scene_load_ply(self.ctx._h, path.encode(), C.byref(n)))
        self.P, self.sh_degree, self._keep = n.value, 3, []
        self._generation += 1
        return n.value

    def _device_view(self, ptr: int, shape, typestr: str):
        """a torch tensor ALIASING context-owned device memory (no copy; valid while the context keeps that array)"""
        import torch

        class _View:
            __cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}

        return torch.as_tensor(_View(), device=f"cuda:{self.ctx.device_id}")

    def scene_tensors(self) -> dict:
        """lcgs_scene_pointers as torch tensors aliasing the bound arrays (the context's own after load_ply / a re-order).
        The C ABI hands these out as const: the context keeps data derived from them (include/lcgs_hip.h).  torch has no
        read-only tensors, so the views are writable -- after writing them in place by anything but the library's own
        optimiser steps, call scene_modified(), or frames cull from stale rows (verify_derived() is the test-mode guard)."""
        ptrs = [C.c_void_p() for _ in range(5)]
        n, deg = C.c_int(0), C.c_int(0)
        _check(load_library().lcgs_scene_pointers(self.ctx._h, C.byref(n), C.byref(deg), *[C.byref(p) for p in ptrs]))
        P, feat = n.value, (deg.value + 1) ** 2 * 3
        shapes = {"pos": (P, 3), "scale": (P, 3), "rotq": (P, 4), "sh": (P, feat), "opacity": (P,)}
        return {k: self._device_view(p.value, shapes[k], "<f4") for k, p in zip(("pos", "scale", "rotq", "sh", "opacity"), ptrs)}

    def scene_modified(self):
        """lcgs_scene_modified: the bound arrays were written behind the library's back; derived data is dropped / rebuilt."""
        _check(load_library().lcgs_scene_modified(self.ctx._h))

    def verify_derived(self) -> int:
        """lcgs_debug_verify_derived: derived rows in use that no longer match the bound arrays (0 = consistent)."""
        n = C.c_int64(0)
        _check(load_library().lcgs_debug_verify_derived(self.ctx._h, C.byref(n)))
        return int(n.value)

    def permutation(self):
        """lcgs_scene_permutation: int32 device tensor, [r] = file index of splat r; None while in file / caller order."""
        p = C.c_void_p()
        _check(load_library().lcgs_scene_permutation(self.ctx._h, C.byref(p)))
        return self._device_view(p.value, (self.P,), "<i4") if p.value else None

    def use_half_sh(self, enable: bool = True):
        """lcgs_scene_use_half_sh: opt-in f16 copy of the SH coefficients for the fused forward (outside the 1e-4 bar)."""
        _check(load_library().lcgs_scene_use_half_sh(self.ctx._h, C.c_int(1 if enable else 0)))

    def set_lod(self, min_radius_px: int):
        """lcgs_set_lod: opt-in footprint cull (0 = off): splats whose reference radius is below min_radius_px pixels are
        dropped from the fused frame.  Changes the image; never the default."""
        _check(load_library().lcgs_set_lod(self.ctx._h, C.c_int(int(min_radius_px))))

    def download_scene(self) -> dict:
        """Host copies of the bound scene (same keys and shapes as read_gs_ply)."""
        P, feat = self.P, (self.sh_degree + 1) ** 2 * 3
        out = {"pos": np.zeros((P, 3), np.float32), "scale": np.zeros((P, 3), np.float32),
               "rotq": np.zeros((P, 4), np.float32), "sh": np.zeros((P, feat), np.float32),
               "opacity": np.zeros((P,), np.float32)}
        _check(load_library().lcgs_scene_download(self.ctx._h, *[_ptr(out[k]) for k in
                                                                  ("pos", "scale", "rotq", "sh", "opacity")]))
        return out

    def forward(self, cam: Camera, img, bg=(0.0, 0.0, 0.0), scale_modifier: float = 1.0, radii=None,
                keep_state: bool = False, sync: bool = True) -> Optional[int]:
        n = C.c_int(0)
        self._generation += 1
        _check(load_library().lcgs_render_forward(self.ctx._h, C.byref(cam), _f3(bg), C.c_float(scale_modifier),
                                                  _ptr(img), _ptr(radii), C.c_int(1 if keep_state else 0),
                                                  C.byref(n) if sync else None))
        return n.value if sync else None

    def forward_batch(self, cams, imgs, bg=(0.0, 0.0, 0.0), scale_modifier: float = 1.0):
        """lcgs_render_forward_batch: one image per camera, two frames in flight; enqueues only."""
        n = len(cams)
        assert n == len(imgs)
        cam_arr = (Camera * n)(*cams)
        ptrs = (C.c_void_p * n)(*[_ptr(t).value for t in imgs])
        self._generation += 1
        _check(load_library().lcgs_render_forward_batch(self.ctx._h, C.c_int(n), cam_arr, _f3(bg),
                                                        C.c_float(scale_modifier), ptrs))

    def backward(self, dL_dimg, dpos, dscale, drotq, dsh, dopacity, compact: bool = False, accumulate: bool = False):
        """lcgs_render_backward; compact=True: lcgs_render_backward_compact (row r = the frame's r-th on-screen
        splat, see visible_rows; only those rows are written); accumulate=True: lcgs_render_backward_accumulate (dense
        rows added to what the arrays hold: a further view of a multi-view batch)."""
        if compact and accumulate:
            raise ValueError("compact rows belong to one frame: they cannot be accumulated over views")
        g = _Grads(_ptr(dpos), _ptr(dscale), _ptr(drotq), _ptr(dsh), _ptr(dopacity))
        lib = load_library()
        fn = lib.lcgs_render_backward_compact if compact else (lib.lcgs_render_backward_accumulate if accumulate
                                                               else lib.lcgs_render_backward)
        _check(fn(self.ctx._h, _ptr(dL_dimg), C.byref(g)))

    def fit_views(self, cams, targets, dpos, dscale, drotq, dsh, dopacity, losses, bg=(0.0, 0.0, 0.0),
                  scale_modifier: float = 1.0):
        """lcgs_fit_views: the views of one optimiser step -- forward, L2 loss against targets[j], backward -- with the
        dense gradients summed into the five arrays and losses[j] (device tensor of len(cams) floats) = view j's loss;
        consecutive views overlap (forward beside the previous backward)."""
        n = len(cams)
        if len(targets) != n:
            raise ValueError("one target image per camera")
        cam_arr = (Camera * n)(*cams)
        ptrs = (C.c_void_p * n)(*[_ptr(t).value for t in targets])
        g = _Grads(_ptr(dpos), _ptr(dscale), _ptr(drotq), _ptr(dsh), _ptr(dopacity))
        _check(load_library().lcgs_fit_views(self.ctx._h, C.c_int(n), cam_arr, _f3(bg), C.c_float(scale_modifier), ptrs,
                                             C.byref(g), _ptr(losses)))
        self._generation += 1

    # ---- splat ownership: the frame in two halves
    OWNER_RECORD_FLOATS, OWNER_GRAD_FLOATS = 12, 12

    def owner_project(self, slot: int, cam: "Camera", row_first: int, row_count: int, keep_state: bool = True,
                      scale_modifier: float = 1.0):
        """lcgs_owner_project: the per-splat half of `cam`'s frame on the rows [row_first, row_first + row_count) of the bound
        scene -> (global row indices int32 [n], packed records float32 [n, 12]) of the rows that reach the screen."""
        import torch

        dev = torch.device("cuda", self.ctx.device_id)
        out_rows = torch.empty(max(row_count, 1), dtype=torch.int32, device=dev)
        out_recs = torch.empty(max(row_count, 1), self.OWNER_RECORD_FLOATS, dtype=torch.float32, device=dev)
        n = C.c_int(0)
        _check(load_library().lcgs_owner_project(self.ctx._h, C.c_int(slot), C.byref(cam), C.c_float(scale_modifier),
                                                 C.c_int(row_first), C.c_int(row_count), C.c_int(1 if keep_state else 0),
                                                 _ptr(out_rows), _ptr(out_recs), C.byref(n)))
        out_rows, out_recs = out_rows[:n.value], out_recs[:n.value]
        self._generation += 1
        return out_rows, out_recs

    def owner_project_all(self, cams, row_first: int, row_count: int, keep_state: bool = True, scale_modifier: float = 1.0):
        """The same for every view of a step -- view v into slot v -- with ONE synchronisation: lcgs_owner_project_views (N
        pipelines side by side), then lcgs_owner_counts.  -> [(rows, records)] per view."""
        import torch

        dev = torch.device("cuda", self.ctx.device_id)
        lib, N = load_library(), len(cams)
        outs = [(torch.empty(max(row_count, 1), dtype=torch.int32, device=dev),
                 torch.empty(max(row_count, 1), self.OWNER_RECORD_FLOATS, dtype=torch.float32, device=dev)) for _ in range(N)]
        # lcgs_owner_project_views: the N pipelines side by side on the context's lanes, joined on its stream
        rows_p = (C.c_void_p * N)(*[o[0].data_ptr() for o in outs])
        recs_p = (C.c_void_p * N)(*[o[1].data_ptr() for o in outs])
        _check(lib.lcgs_owner_project_views(self.ctx._h, C.c_int(0), C.c_int(N), (Camera * N)(*cams), C.c_float(scale_modifier),
                                            C.c_int(row_first), C.c_int(row_count), C.c_int(1 if keep_state else 0), rows_p, recs_p))
        counts = (C.c_int * len(outs))()
        _check(lib.lcgs_owner_counts(self.ctx._h, C.c_int(0), C.c_int(len(outs)), counts))
        self._generation += 1
        return [(r[:counts[v]], q[:counts[v]]) for v, (r, q) in enumerate(outs)]

    def owner_render(self, cam: "Camera", rows, recs, img, bg=(0.0, 0.0, 0.0), keep_state: bool = True):
        """lcgs_owner_render: the rest of the frame from received records (ascending global rows)"""
        _check(load_library().lcgs_owner_render(self.ctx._h, C.byref(cam), _f3(bg), C.c_int(int(rows.shape[0])), _ptr(rows),
                                                _ptr(recs), _ptr(img), C.c_int(1 if keep_state else 0)))
        self._generation += 1

    def owner_render_backward(self, dL_dimg, grads2d):
        """lcgs_owner_render_backward: 2-D gradients (float32 [n, 12]) of the rows the last owner_render drew"""
        _check(load_library().lcgs_owner_render_backward(self.ctx._h, _ptr(dL_dimg), _ptr(grads2d)))

    def owner_backward(self, slot: int, grads2d, dpos, dscale, drotq, dsh, dopacity, accumulate: bool):
        """lcgs_owner_backward: the slot's rows' 2-D gradients -> parameter gradients at their rows of the full-size arrays"""
        g = _Grads(_ptr(dpos), _ptr(dscale), _ptr(drotq), _ptr(dsh), _ptr(dopacity))
        _check(load_library().lcgs_owner_backward(self.ctx._h, C.c_int(slot), _ptr(grads2d), C.byref(g),
                                                  C.c_int(1 if accumulate else 0)))

    def l2_loss_backward(self, img, target, dL_dimg, loss):
        """lcgs_l2_loss_backward: loss[0] = mean((img - target)^2), dL_dimg = 2 (img - target) / numel (device tensors)"""
        _, H, W = img.shape
        _check(load_library().lcgs_l2_loss_backward(self.ctx._h, C.c_int(W), C.c_int(H), _ptr(img), _ptr(target),
                                                    _ptr(dL_dimg), _ptr(loss)))

    def visible_rows(self):
        """lcgs_visible_rows of the last forward frame: (splat index of every compact row, row count) as a device
        int32 tensor copy (synchronises the context)."""
        import torch

        rows, count = C.c_void_p(), C.c_void_p()
        _check(load_library().lcgs_visible_rows(self.ctx._h, C.byref(rows), C.byref(count)))
        self.ctx.synchronize()
        n = self.frame_stats()["num_visible"]
        dev = f"cuda:{self.ctx.device_id}"
        if n == 0:
            return torch.empty(0, dtype=torch.int32, device=dev)

        class _View:  # the context-owned device array, seen through the CUDA array interface; cloned before returning
            __cuda_array_interface__ = {"shape": (n,), "typestr": "<i4", "data": (int(rows.value), False), "version": 2}

        return torch.as_tensor(_View(), device=dev).clone()

    def set_profiling(self, enabled: bool):
        _check(load_library().lcgs_set_profiling(self.ctx._h, C.c_int(1 if enabled else 0)))

    def stage_times(self) -> dict:
        t = _StageTimes()
        _check(load_library().lcgs_get_stage_times(self.ctx._h, C.byref(t)))
        return {t.name[i].decode(): float(t.ms[i]) for i in range(t.count)}

    def frame_stats(self) -> dict:
        s = _FrameStats()
        _check(load_library().lcgs_get_frame_stats(self.ctx._h, C.byref(s)))
        return {k: int(getattr(s, k)) for k, _ in _FrameStats._fields_}

    def last_lists(self, d_list, d_ranges):
        """lcgs_debug_last_lists: the last fused frame's sorted lists (original splat indices) and ranges.  Per TILE after a frame
        with keep_state=True; a frame without it lists its pairs per block of 2 x 2 tiles (the first ceil(gx / 2) * ceil(gy / 2)
        ranges, row-major; the rest zero).  Either argument may be None."""
        _check(load_library().lcgs_debug_last_lists(self.ctx._h, _ptr(d_list), _ptr(d_ranges)))

    def set_list_policy(self, policy: str):
        """lcgs_set_list_policy: "tile" (the reference's per-tile lists), "block" (per 2 x 2 tiles), "auto" (default)"""
        _check(load_library().lcgs_set_list_policy(self.ctx._h, C.c_int({"tile": 0, "block": 1, "auto": 2}[policy])))

    def last_state(self, d_final_T, d_n_contrib):
        """lcgs_debug_last_state: per pixel, what the last keep_state frame kept for its backward (final transmittance; 1-based
        list position of the last contributor).  Either argument may be None."""
        _check(load_library().lcgs_debug_last_state(self.ctx._h, _ptr(d_final_T), _ptr(d_n_contrib)))

    def adam_step(self, grads: dict, raw: dict, m: dict, v: dict, activated: dict, step: int, lr: dict,
                  betas=(0.9, 0.999), eps: float = 1e-15, visible_only: bool = False, sh_degree: int = 3,
                  compact_grads: bool = False):
        """lcgs_adam_step: gradients w.r.t. the activated values -> Adam on the raw parameters -> refreshed activated
        arrays.  Every dict has the keys pos / scale / rotq / sh / opacity (device tensors); lr has pos, sh_dc,
        sh_rest, opacity, scale, rot.  compact_grads (with visible_only): `grads` hold backward(compact=True) rows."""
        if compact_grads and not visible_only:
            raise ValueError("compact gradient rows only exist for the on-screen splats: visible_only must be set")
        keys = ("pos", "scale", "rotq", "sh", "opacity")
        P = int(raw["pos"].shape[0])
        cfg = _AdamConfig(lr["pos"], lr["sh_dc"], lr["sh_rest"], lr["opacity"], lr["scale"], lr["rot"], betas[0], betas[1],
                          eps, int(step), (2 if compact_grads else 1) if visible_only else 0)
        g = _Grads(*[_ptr(grads[k]) for k in keys])
        packs = [_Params(*[_ptr(d[k]) for k in keys]) for d in (raw, m, v, activated)]
        _check(load_library().lcgs_adam_step(self.ctx._h, C.c_int(P), C.c_int(sh_degree), C.byref(cfg), C.byref(g),
                                             *[C.byref(p) for p in packs]))


    def backward_adam(self, dL_dimg, raw: dict, m: dict, v: dict, activated: dict, step: int, lr: dict,
                      betas=(0.9, 0.999), eps: float = 1e-15, sh_degree: int = 3):
        """lcgs_render_backward_adam: the backward of the last keep_state frame with the on-screen-only Adam update applied
        where the per-splat gradients are formed -- no gradient arrays (= backward(compact=True) + adam_step(visible_only,
        compact_grads), bit for bit)."""
        keys = ("pos", "scale", "rotq", "sh", "opacity")
        P = int(raw["pos"].shape[0])
        cfg = _AdamConfig(lr["pos"], lr["sh_dc"], lr["sh_rest"], lr["opacity"], lr["scale"], lr["rot"], betas[0], betas[1],
                          eps, int(step), 2)
        packs = [_Params(*[_ptr(d[k]) for k in keys]) for d in (raw, m, v, activated)]
        _check(load_library().lcgs_render_backward_adam(self.ctx._h, _ptr(dL_dimg), C.c_int(P), C.c_int(sh_degree),
                                                        C.byref(cfg), *[C.byref(p) for p in packs]))


def _adam_config(lr: dict, betas, eps: float, step: int, visible_only: int) -> _AdamConfig:
    return _AdamConfig(lr["pos"], lr["sh_dc"], lr["sh_rest"], lr["opacity"], lr["scale"], lr["rot"], betas[0], betas[1], eps,
                       int(step), int(visible_only))


_KEYS = ("pos", "scale", "rotq", "sh", "opacity")


def shard_rows(num_gaussians: int, world_size: int, rank: int):
    """lcgs_comm_shard_rows: (first, count) of the rows rank `rank` owns in the sharded optimiser step; the last
    num_gaussians mod world_size rows (the tail) are kept by every rank.  Pure host arithmetic (no GPU needed)."""
    first, count = C.c_int64(0), C.c_int64(0)
    load_library().lcgs_comm_shard_rows(C.c_int64(num_gaussians), C.c_int(world_size), C.c_int(rank), C.byref(first),
                                        C.byref(count))
    return first.value, count.value


def owner_rows(num_gaussians: int, world_size: int, rank: int):
    """lcgs_comm_owner_rows: (first, count) of the rows a rank OWNS in the ownership step (the tail with the last rank)"""
    first, count = C.c_int64(0), C.c_int64(0)
    load_library().lcgs_comm_owner_rows(C.c_int64(num_gaussians), C.c_int(world_size), C.c_int(rank), C.byref(first), C.byref(count))
    return int(first.value), int(count.value)


class LoopbackGroup:
    """lcgs_loopback_group: the rendezvous of N in-process communicators (N contexts on ONE device, one host thread each) --
    the ownership step's C code path with N > 1 participants on a single GPU.  Tests and rehearsals only."""

    def __init__(self, world_size: int):
        self.world_size = world_size
        self._h = C.c_void_p(0)
        _check(load_library().lcgs_loopback_group_create(C.c_int(world_size), C.byref(self._h)))

    def close(self):
        if self._h:
            _check(load_library().lcgs_loopback_group_destroy(self._h))
            self._h = C.c_void_p(0)


class Comm:
    """lcgs_comm: the RCCL communicator of a view-parallel job, attached to one context.

    `exchange(payload: bytes | None) -> bytes` carries the 128-byte rendezvous token from rank 0 (which passes it in)
    to every other rank (which pass None): any broadcast the host has (torch.distributed, a pipe, a file)."""

    def __init__(self, ctx: Context, rank: int, world_size: int, exchange=None, loopback: "LoopbackGroup" = None):
        lib = load_library()
        if loopback is not None:  # an in-process communicator: N contexts on one device, one host thread each
            self.ctx, self.rank, self.world_size = ctx, rank, loopback.world_size
            self._h = C.c_void_p(0)
            _check(lib.lcgs_comm_create_loopback(ctx._h, loopback._h, C.c_int(rank), C.byref(self._h)))
            return
        token = (C.c_char * 128)()
        if rank == 0:
            _check(lib.lcgs_comm_unique_id(token))
        if world_size > 1 or exchange is not None:
            if exchange is None:
                raise ValueError("world_size > 1 needs an exchange function for the rendezvous token")
            got = exchange(bytes(token.raw) if rank == 0 else None)
            assert len(got) == 128
            C.memmove(token, got, 128)
        self.ctx, self.rank, self.world_size = ctx, rank, world_size
        self._h = C.c_void_p(0)
        _check(lib.lcgs_comm_create(ctx._h, token, C.c_int(rank), C.c_int(world_size), C.byref(self._h)))

    def close(self):
        if self._h:
            load_library().lcgs_comm_destroy(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def abandon(self):
        """forget the handle WITHOUT destroying the communicator: for one whose self-test left a phase stuck inside RCCL
        (lcgs_comm_destroy would wait for its stream, i.e. for ever) -- the process is expected to report and exit"""
        self._h = C.c_void_p(0)

    def info(self):
        """lcgs_comm_info: (rank, world_size) as the communicator itself -- i.e. RCCL -- sees them"""
        rk, ws = C.c_int(-1), C.c_int(-1)
        _check(load_library().lcgs_comm_info(self._h, C.byref(rk), C.byref(ws)))
        return int(rk.value), int(ws.value)

    def owner_step_forward(self, cams, img, bg=(0.0, 0.0, 0.0), scale_modifier: float = 1.0):
        """lcgs_owner_step_forward: the ownership step's first half with its transport -- this rank's rows projected for every
        view, records exchanged (RCCL send / recv, or the loopback), this rank's view rendered into img"""
        arr = (Camera * len(cams))(*cams)
        _check(load_library().lcgs_owner_step_forward(self.ctx._h, self._h, arr, _f3(bg), C.c_float(scale_modifier), _ptr(img)))

    def owner_step_backward(self, dL_dimg, grads: dict):
        """lcgs_owner_step_backward: the view's 2-D gradients back to the owners, parameter gradients at this rank's own rows"""
        g = _Grads(*[_ptr(grads[k]) for k in _KEYS])
        _check(load_library().lcgs_owner_step_backward(self.ctx._h, self._h, _ptr(dL_dimg), C.byref(g)))

    def selftest(self, timeout_s: float = 30.0, check: bool = True) -> dict:
        """lcgs_comm_selftest (a collective: every rank calls it): a 1 KB all-reduce, zero- and one-byte messages to every peer
        in one group, an ownership step on a 10 000-splat scene with and without read-back -- each phase against timeout_s.
        Returns the report as a dict (`ok`, per-phase `*_ok` / `*_ms`, `timed_out`, `message`); check=True raises on failure."""
        rep = _SelftestReport()
        status = load_library().lcgs_comm_selftest(self.ctx._h, self._h, C.c_double(timeout_s), C.byref(rep))
        out = {"ok": status == 0, "world_size": int(rep.world_size), "rank": int(rep.rank),
               "allreduce_ok": int(rep.allreduce_ok), "allreduce_ms": round(float(rep.allreduce_ms), 3),
               "p2p_ok": int(rep.p2p_ok), "p2p_ms": round(float(rep.p2p_ms), 3),
               "owner_step_ok": int(rep.owner_step_ok), "owner_step_ms": round(float(rep.owner_step_ms), 3),
               "owner_max_grad_err": float(rep.owner_max_grad_err), "timed_out": int(rep.timed_out),
               "message": rep.message.decode(errors="replace")}
        if check:
            _check(status)
        return out

    def owner_step_set_async(self, enable: bool = True):
        """lcgs_owner_step_set_async: steps size their messages from the previous step's counts and read nothing back; every
        such step must be closed with owner_step_finish()"""
        _check(load_library().lcgs_owner_step_set_async(self._h, C.c_int(1 if enable else 0)))

    def owner_step_finish(self) -> bool:
        """lcgs_owner_step_finish: True = some rank's step was short (a clipped message, truncated pairs): EVERY rank calls
        forward + backward again"""
        redo = C.c_int(0)
        _check(load_library().lcgs_owner_step_finish(self.ctx._h, self._h, C.byref(redo)))
        return bool(redo.value)

    def owner_step(self, cams, img, dL_dimg, grads: dict, bg=(0.0, 0.0, 0.0), scale_modifier: float = 1.0) -> int:
        """forward + backward + finish, repeated while the step was short; returns the number of repetitions (0 normally)"""
        for attempt in range(3):
            self.owner_step_forward(cams, img, bg, scale_modifier)
            self.owner_step_backward(dL_dimg, grads)
            if not self.owner_step_finish():
                return attempt
        raise LcgsError("the ownership step did not settle after two repetitions")

    def set_transport(self, transport: str):
        """lcgs_comm_set_transport: "f32" (default, exact) or "f16" (opt-in: half the bytes, ~sqrt(N) x 5e-4 relative)"""
        _check(load_library().lcgs_comm_set_transport(self._h, C.c_int({"f32": 0, "f16": 1}[transport])))

    def allreduce_grads(self, grads: dict, sh_degree: int = 3):
        """lcgs_grads_allreduce: in-place sum over the ranks of the five dense gradient arrays (chunked, overlapping the
        backward's tail); the context's stream waits for the result."""
        P = int(grads["pos"].shape[0])
        g = _Grads(*[_ptr(grads[k]) for k in _KEYS])
        _check(load_library().lcgs_grads_allreduce(self.ctx._h, self._h, C.c_int(P), C.c_int(sh_degree), C.byref(g)))

    def adam_step_sharded(self, grads: dict, raw: dict, m: dict, v: dict, activated: dict, step: int, lr: dict,
                          betas=(0.9, 0.999), eps: float = 1e-15, sh_degree: int = 3):
        """lcgs_adam_step_sharded: reduce-scatter -> Adam on the own rows (+ tail) -> all-gather of the activated arrays."""
        P = int(raw["pos"].shape[0])
        cfg = _adam_config(lr, betas, eps, step, 0)
        g = _Grads(*[_ptr(grads[k]) for k in _KEYS])
        packs = [_Params(*[_ptr(d[k]) for k in _KEYS]) for d in (raw, m, v, activated)]
        _check(load_library().lcgs_adam_step_sharded(self.ctx._h, self._h, C.c_int(P), C.c_int(sh_degree), C.byref(cfg),
                                                     C.byref(g), *[C.byref(p) for p in packs]))


    # ---- sparse gradient exchange (lcgs_hip.h "Sparse gradient exchange")
    def track_touched_rows(self, enable: bool = True):
        """lcgs_comm_track_touched_rows: dense backward passes on this context flag the rows their frame touched"""
        _check(load_library().lcgs_comm_track_touched_rows(self._h, C.c_int(1 if enable else 0)))

    def stats(self) -> dict:
        """lcgs_comm_get_stats: what the last collective call of this communicator moved (per GPU, from actual counts)"""
        st = _CommStats()
        _check(load_library().lcgs_comm_get_stats(self._h, C.byref(st)))
        return {"bytes_sent": int(st.bytes_sent), "bytes_received": int(st.bytes_received),
                "touched_rows": int(st.touched_rows), "collective_groups": int(st.collective_groups)}

    def adam_step_sparse(self, grads: dict, raw: dict, m: dict, v: dict, activated: dict, step: int, lr: dict,
                         betas=(0.9, 0.999), eps: float = 1e-15, sh_degree: int = 3):
        """lcgs_adam_step_sparse: touched rows -> their owners (send / recv) -> Adam on the own rows -> all-gather"""
        P = int(raw["pos"].shape[0])
        cfg = _adam_config(lr, betas, eps, step, 0)
        g = _Grads(*[_ptr(grads[k]) for k in _KEYS])
        packs = [_Params(*[_ptr(d[k]) for k in _KEYS]) for d in (raw, m, v, activated)]
        _check(load_library().lcgs_adam_step_sparse(self.ctx._h, self._h, C.c_int(P), C.c_int(sh_degree), C.byref(cfg),
                                                    C.byref(g), *[C.byref(p) for p in packs]))

    def sparse_touched_rows(self, num_gaussians: int, world_size: Optional[int] = None):
        """lcgs_sparse_touched_rows -> (device address of the ascending row list, owner_first[0 .. N + 1]) for an exchange
        over `world_size` ranks (default: the communicator's own); consumes the step's touched set and synchronises"""
        n = self.world_size if world_size is None else world_size
        out = _SparseRows()
        _check(load_library().lcgs_sparse_touched_rows(self.ctx._h, self._h, C.c_int(num_gaussians), C.c_int(n), C.byref(out)))
        return int(out.d_rows or 0), [int(out.owner_first[i]) for i in range(n + 2)]

    def sparse_pack(self, grads: dict, rows_addr: int, first: int, count: int, msg, sh_degree: int = 3):
        """lcgs_sparse_pack: rows [first, first + count) of the touched list -> one message (a float32 tensor of
        sparse_message_words(count) elements)"""
        g = _Grads(*[_ptr(grads[k]) for k in _KEYS])
        _check(load_library().lcgs_sparse_pack(self.ctx._h, C.c_int(sh_degree), C.byref(g), C.c_void_p(rows_addr + 4 * first),
                                               C.c_int64(count), _ptr(msg)))

    def sparse_accumulate(self, grads: dict, msg, count: int, sh_degree: int = 3, row_first: int = 0, row_count: int = -1):
        """lcgs_sparse_accumulate: add a received message's rows to the dense gradient rows; only rows in
        [row_first, row_first + row_count) are accepted (default: every row of the arrays)"""
        g = _Grads(*[_ptr(grads[k]) for k in _KEYS])
        if row_count < 0:
            row_first, row_count = 0, int(grads["pos"].shape[0])
        _check(load_library().lcgs_sparse_accumulate(self.ctx._h, C.c_int(sh_degree), C.byref(g), _ptr(msg), C.c_int64(count),
                                                     C.c_int64(row_first), C.c_int64(row_count)))


def sparse_message_words(count: int, sh_degree: int = 3) -> int:
    """4-byte words of a sparse-exchange message of `count` rows: indices + the five attribute blocks"""
    return count * (1 + 3 + 3 + 4 + (sh_degree + 1) ** 2 * 3 + 1)


class _SparseRows(C.Structure):
    _fields_ = [("d_rows", C.c_void_p), ("num_rows", C.c_int64), ("owner_first", C.c_int64 * 66)]


class _CommStats(C.Structure):
    _fields_ = [("bytes_sent", C.c_int64), ("bytes_received", C.c_int64), ("touched_rows", C.c_int64),
                ("collective_groups", C.c_int)]


def render_autograd(renderer: "Renderer", cam: Camera, pos, scale, rotq, sh, opacity, bg=(0.0, 0.0, 0.0),
                    scale_modifier: float = 1.0):
    """Differentiable frame for torch: `img = render_autograd(r, cam, pos, scale, rotq, sh, opacity)` (activated
    parameters, CHW float image); `img.backward(...)` / `torch.autograd.grad` run lcgs_render_backward.  torch is the
    owner of the tensors and of the autograd graph only; both directions are the HIP kernels.

    lcgs_render_backward differentiates the LAST keep_state frame of the context.  Several render_autograd frames of one
    renderer may be alive at once (a multi-view loss), or the renderer may have been used for something else between a
    frame's forward and its backward: the backward notices (the renderer's generation counter has moved), binds its own
    saved scene again and re-renders its own view with keep_state before differentiating -- the cost of one extra
    forward instead of gradients silently computed from another view's lists."""
    import torch

    class _Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, pos, scale, rotq, sh, opacity):
            args = [t.detach().contiguous() for t in (pos, scale, rotq, sh, opacity)]
            renderer.bind_scene(*args)
            img = torch.empty(3, cam.height, cam.width, device=args[0].device, dtype=torch.float32)
            n = renderer.forward(cam, img, bg=bg, scale_modifier=scale_modifier, keep_state=True, sync=True)
            if n == 0:
                img[:] = torch.tensor(bg, device=img.device).view(3, 1, 1)  # nothing drawn: the image is the background
            ctx.shapes = [t.shape for t in (pos, scale, rotq, sh, opacity)]
            ctx.empty = n == 0
            ctx.generation = renderer._generation
            ctx.save_for_backward(*args)
            return img

        @staticmethod
        def backward(ctx, dL_dimg):
            args = ctx.saved_tensors
            grads = [torch.zeros_like(t) for t in args]
            if not ctx.empty:
                if renderer._generation != ctx.generation:  # the renderer's frame state is no longer this frame's
                    renderer.bind_scene(*args)
                    scratch = torch.empty(3, cam.height, cam.width, device=args[0].device, dtype=torch.float32)
                    renderer.forward(cam, scratch, bg=bg, scale_modifier=scale_modifier, keep_state=True, sync=True)
                    ctx.generation = renderer._generation
                renderer.backward(dL_dimg.contiguous(), *grads)
            return tuple(g.view(s) for g, s in zip(grads, ctx.shapes))

    return _Fn.apply(pos, scale, rotq, sh, opacity)


# ---------------------------------------------------------------------------------------------- host io
def read_gs_ply(path: str) -> dict:
    """read_gs_ply (app/gaussians.cpp:75-171): activated, repacked host arrays."""
    lib = load_library()
    s = _SceneHost()
    _check(lib.lcgs_ply_read(path.encode(), C.byref(s)))
    try:
        P = s.num_gaussians

        def grab(p, n, shape):
            if P == 0:
                return np.zeros(shape, dtype=np.float32)
            return np.ctypeslib.as_array(p, shape=(n,)).reshape(shape).copy()

        return {
            "pos": grab(s.pos, P * 3, (P, 3)), "sh": grab(s.feature, P * 48, (P, 48)),
            "opacity": grab(s.opacity, P, (P,)), "scale": grab(s.scale, P * 3, (P, 3)),
            "rotq": grab(s.rotq, P * 4, (P, 4)), "sh_degree": int(s.sh_degree),
        }
    finally:
        lib.lcgs_scene_host_free(C.byref(s))


def write_ply_raw(path: str, pos, f_dc, f_rest, opacity_logit, log_scale, rot):
    a = [np.ascontiguousarray(x, dtype=np.float32) for x in (pos, f_dc, f_rest, opacity_logit, log_scale, rot)]
    P = int(a[0].reshape(-1, 3).shape[0])
    _check(load_library().lcgs_ply_write_raw(path.encode(), C.c_int(P), *[_ptr(x) for x in a]))


def synth_scene(kind: int, seed: int, count: int, first: int = 0) -> dict:
    """Deterministic synthetic stand-in scene: kind 0 object-like, 1 unbounded-like."""
    out = {
        "pos": np.empty((count, 3), np.float32), "sh": np.empty((count, 48), np.float32),
        "opacity": np.empty((count,), np.float32), "scale": np.empty((count, 3), np.float32),
        "rotq": np.empty((count, 4), np.float32),
    }
    _check(load_library().lcgs_synth_scene(C.c_int(kind), C.c_uint64(seed), C.c_int64(first), C.c_int64(count),
                                           _ptr(out["pos"]), _ptr(out["sh"]), _ptr(out["opacity"]),
                                           _ptr(out["scale"]), _ptr(out["rotq"])))
    return out


def image_to_rgb8(img_chw: np.ndarray) -> np.ndarray:
    """app/main.cpp:323-335"""
    img = np.ascontiguousarray(img_chw, dtype=np.float32)
    _, H, W = img.shape
    out = np.empty((H, W, 3), np.uint8)
    load_library().lcgs_image_to_rgb8(C.c_int(W), C.c_int(H), _ptr(img), _ptr(out))
    return out


def write_png(path: str, rgb: np.ndarray):
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    H, W, _ = rgb.shape
    _check(load_library().lcgs_write_png(path.encode(), C.c_int(W), C.c_int(H), _ptr(rgb)))
