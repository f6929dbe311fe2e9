"""luisacomputegaussiansplatting_amd -- MI355X-native (gfx950) 3D Gaussian Splatting hot path.

The product is ``liblcgs_hip.so`` (C ABI in ``include/lcgs_hip.h``, hand-written HIP kernels in
``csrc/kernels``).  This package is the thin Python mirror of the reference's operator API
(``SHProcessor`` / ``GSProjector`` / ``GSTileSplatter`` + proxy structs, lcgs/include/lcgs/*.h) on top of
that C ABI.  torch is used only as the owner of device memory and streams.

There is no CPU fallback: every operator raises ``LcgsError`` if the HIP library is missing or no
gfx950 device is present.
"""
from . import api  # noqa: F401
from .api import (  # noqa: F401
    Camera,
    Comm,
    Context,
    GSProjector,
    GSProjectorInputProxy,
    GSProjectorOutputProxy,
    GSSplatForwardOutputProxy,
    GSTileSplatter,
    GSTileSplatterAccelProxy,
    GSTileSplatterInputProxy,
    GPUPointsProxy,
    LcgsError,
    Renderer,
    SHProcessor,
    build_library,
    get_lookat_cam,
    image_to_rgb8,
    library_path,
    load_library,
    local_to_world_matrix,
    projection_matrix,
    read_gs_ply,
    render_autograd,
    shard_rows,
    synth_scene,
    world_to_local_matrix,
    write_png,
    write_ply_raw,
)

__version__ = "0.1"
