/*
 * lcgs_hip.h -- C ABI of liblcgs_hip.so: the MI355X-native (gfx950) implementation of the
 * LuisaComputeGaussianSplatting hot path (SH colour -> 3D->2D projection -> tile keys -> sort ->
 * per-tile alpha compositing, plus the matching backward).
 *
 * Every entry point names the reference interface it replaces (paths relative to the reference
 * repository).  The reference's operator API is three exported C++ classes taking POD "proxy"
 * structs of non-owning device buffer views (lcgs/include/lcgs/proxy.h:21-73); here the same
 * contract is expressed as plain pointers + sizes:
 *   - all `d_` pointers are DEVICE pointers owned by the caller, laid out exactly as the
 *     reference's flat float buffers (AoS-packed: xyz xyz..., rxyz rxyz..., SH (P,16,3));
 *   - all structs are POD and passed by pointer, copied before the call returns;
 *   - every function returns an lcgs_status; no exceptions cross the boundary
 *     (the reference is `noexcept` + abort);
 *   - one lcgs_context per (GPU, stream); a context is not thread-safe, like the reference's
 *     operator objects (mutable num_rendered / temp buffers, gs_tile_splatter.h:23,50-55).
 */
#ifndef LCGS_HIP_H
#define LCGS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(_WIN32)
#define LCGS_API __declspec(dllexport)
#else
#define LCGS_API __attribute__((visibility("default")))
#endif

typedef enum lcgs_status {
    LCGS_OK                = 0,
    LCGS_ERR_INVALID_ARG   = 1,
    LCGS_ERR_HIP           = 2, /* a HIP runtime call failed; see lcgs_last_error() */
    LCGS_ERR_NO_DEVICE     = 3,
    LCGS_ERR_OUT_OF_MEMORY = 4,
    LCGS_ERR_CAPACITY      = 5, /* caller-provided pair buffers too small (the reference does not check, app/main.cpp:245) */
    LCGS_ERR_IO            = 6,
    LCGS_ERR_FORMAT        = 7,
    LCGS_ERR_STATE         = 8  /* e.g. backward without a preceding forward */
} lcgs_status;

typedef struct lcgs_context lcgs_context;

/* lcgs/include/lcgs/util/camera.h:15-25 (struct Camera) */
typedef struct lcgs_camera {
    float position[3];
    float front[3];
    float up[3];
    float right[3];
    float fov;          /* vertical field of view, degrees (default 60) */
    float aspect_ratio; /* width / height */
    int   width;
    int   height;
} lcgs_camera;

/* ------------------------------------------------------------------------------------------
 * Library / context
 * ------------------------------------------------------------------------------------------ */
LCGS_API const char* lcgs_version(void);
/* Thread-local message of the last failing call on this thread. */
LCGS_API const char* lcgs_last_error(void);

/* Replaces Context::create_device + Device::create_stream (app/main.cpp:162-163) and the three
 * `create(Device&)` calls (sh_preprocessor.cpp:16, gs_projector/impl.cpp:14, gs_tile_splatter/impl.cpp:25).
 * `stream` is a hipStream_t (NULL = the device's null stream); kernels are precompiled for gfx950,
 * nothing is JIT-compiled here. */
LCGS_API lcgs_status lcgs_create(int device_id, void* stream, lcgs_context** out_ctx);
LCGS_API lcgs_status lcgs_destroy(lcgs_context* ctx);
LCGS_API lcgs_status lcgs_set_stream(lcgs_context* ctx, void* stream);
/* Stream::synchronize (app/main.cpp:223,315) */
LCGS_API lcgs_status lcgs_synchronize(lcgs_context* ctx);

/* ------------------------------------------------------------------------------------------
 * Host camera helpers -- lcgs/include/lcgs/util/camera.h
 * Matrices are column-major float[16], m[c*4+r], like luisa::float4x4.
 * ------------------------------------------------------------------------------------------ */
/* get_lookat_cam, camera.h:74-82 (fov/aspect/width/height get the struct defaults 60/1/512/512) */
LCGS_API void lcgs_get_lookat_cam(const float pos[3], const float target[3], const float world_up[3],
                                  lcgs_camera* out_cam);
/* local_to_world_matrix, camera.h:27-36 */
LCGS_API void lcgs_local_to_world_matrix(const lcgs_camera* cam, float m[16]);
/* world_to_local_matrix, camera.h:38-51 */
LCGS_API void lcgs_world_to_local_matrix(const lcgs_camera* cam, float m[16]);
/* projection_matrix, camera.h:54-72 (reference defaults znear=0.1, zfar=100) */
LCGS_API void lcgs_projection_matrix(float tanfovx, float tanfovy, float znear, float zfar, float m[16]);

/* ------------------------------------------------------------------------------------------
 * Stage-level operators: one call per reference class method, on raw device pointers with the
 * reference's buffer layouts, so each stage can be swapped in (and parity-tested) alone.
 * ------------------------------------------------------------------------------------------ */

/* SHProcessor::process (lcgs/include/lcgs/sh_preprocessor.h:30-37, lcgs/src/sh_preprocessor.cpp:169-188).
 * d_pos[3P], d_sh[P*(level+1)^2*3] -> d_color[3P] = clamp(SH(dir) + 0.5, 0, 1). */
LCGS_API lcgs_status lcgs_sh_process(lcgs_context* ctx, int num_points, const float* d_pos,
                                     const lcgs_camera* camera, const float* d_sh, float* d_color,
                                     int level, int channel);

/* GSProjector::forward (lcgs/include/lcgs/gs_projector.h:37-43, lcgs/src/gs_projector/impl.cpp:26-93).
 * Input proxy {num, pos[3P], scale[3P], rotq[4P] (r,x,y,z), scale_modifier} (gs_projector.h:16-22),
 * output proxy {means_2d[2P] NDC, covs_2d[3P], depth[P]} (gs_projector.h:24-28).
 * Splats with view-space z < 0.2 are left unwritten, as in the reference (gs_projector/shader.cpp:121). */
LCGS_API lcgs_status lcgs_project_forward(lcgs_context* ctx, int num_gaussians, const float* d_pos,
                                          const float* d_scale, const float* d_rotq, float scale_modifier,
                                          float* d_means_2d, float* d_covs_2d, float* d_depth,
                                          const lcgs_camera* camera, int use_focal);

/* GSTileSplatterAccelProxy (lcgs/include/lcgs/proxy.h:56-64) + the pair capacity the app fixes at
 * 20M (app/main.cpp:245). */
typedef struct lcgs_tile_accel {
    uint32_t* tiles_touched;            /* P */
    uint32_t* point_offsets;            /* P */
    uint64_t* point_list_keys_unsorted; /* capacity */
    uint32_t* point_list_unsorted;      /* capacity */
    uint64_t* point_list_keys;          /* capacity */
    uint32_t* point_list;               /* capacity */
    uint32_t* ranges;                   /* 2 * ceil(W/16) * ceil(H/16) */
    int64_t   capacity;                 /* number of (tile,splat) pairs the four pair buffers hold */
} lcgs_tile_accel;

/* GSTileSplatterInputProxy (proxy.h:43-54).  means_2d and conic are read AND overwritten in place
 * (NDC -> pixel, cov -> conic), exactly like the reference (gs_tile_splatter/shader.cpp:160-161). */
typedef struct lcgs_tile_input {
    int          num_gaussians;
    float        bg_color[3];
    float*       means_2d;         /* 2P */
    const float* depth_features;   /* P  */
    float*       conic;            /* 3P */
    const float* color_features;   /* 3P */
    const float* opacity_features; /* P  */
} lcgs_tile_input;

/* GSSplatForwardOutputProxy (proxy.h:66-71).  target_img is written planar CHW
 * (gs_tile_splatter/shader.cpp:279-286).  final_T / n_contrib are optional extras (NULL to skip):
 * the state a backward pass needs, which the reference computes and drops (shader.cpp:219-220). */
typedef struct lcgs_tile_output {
    int       height;
    int       width;
    float*    target_img; /* 3*H*W */
    int32_t*  radii;      /* P */
    float*    final_T;    /* H*W, optional */
    uint32_t* n_contrib;  /* H*W, optional */
} lcgs_tile_output;

/* GSTileSplatter::forward (lcgs/include/lcgs/gs_tile_splatter.h:28-35, lcgs/src/gs_tile_splatter/impl.cpp:63-180).
 * *num_rendered receives the reference's return value; 0 means nothing was drawn and the image is
 * left untouched (impl.cpp:109). */
LCGS_API lcgs_status lcgs_tile_splat_forward(lcgs_context* ctx, const lcgs_tile_accel* accel,
                                             const lcgs_tile_input* input, const lcgs_tile_output* output,
                                             int use_focal, int* num_rendered);

/* How the three operators above execute (default LCGS_STAGES_EXACT: each call runs at once and leaves every buffer of
 * the reference behind, bit for bit).  LCGS_STAGES_DEFERRED is for a caller that drives them the way app/main.cpp:266-308
 * does -- process, forward, forward, back to back, reading only target_img / radii / num_rendered afterwards:
 * lcgs_sh_process and lcgs_project_forward only RECORD their arguments; an lcgs_tile_splat_forward whose input proxy is
 * exactly their outputs (same arrays, same camera, use_focal, no final_T / n_contrib request) renders the library's fused
 * frame from the 3-D arrays instead -- the same image, radii and num_rendered (bit for bit; tests/test_gpu_stages.py), with
 * one difference inherited from the fused frame: a splat whose covariance is NaN is invisible (exact mode reproduces the
 * reference's zero-filled pairs for it).  The intermediate buffers (color, means_2d, covs_2d, depth and
 * the accel proxy) are then NOT written.  A splat call that does not match, lcgs_stage_flush, lcgs_synchronize or a switch
 * back to exact mode run whatever was recorded, so the mode never changes a result, only when it is produced. */
typedef enum lcgs_stage_mode { LCGS_STAGES_EXACT = 0, LCGS_STAGES_DEFERRED = 1 } lcgs_stage_mode;
LCGS_API lcgs_status lcgs_set_stage_mode(lcgs_context* ctx, int mode);
LCGS_API lcgs_status lcgs_stage_flush(lcgs_context* ctx);

/* The two external parallel primitives the splatter borrows (lcpp, absent from the reference tree):
 * DeviceScan<>::InclusiveSum (call site gs_tile_splatter/impl.cpp:104) and
 * DeviceRadixSort<>::SortPairs<ulong,uint> (call site impl.cpp:135-143).  Temp storage is owned by
 * the context (the reference keeps it in the splatter, gs_tile_splatter.h:50-55). */
LCGS_API lcgs_status lcgs_inclusive_sum_u32(lcgs_context* ctx, const uint32_t* d_in, uint32_t* d_out, int64_t n);
LCGS_API lcgs_status lcgs_sort_pairs_u64_u32(lcgs_context* ctx, const uint64_t* d_keys_in, uint64_t* d_keys_out,
                                             const uint32_t* d_vals_in, uint32_t* d_vals_out, int64_t n,
                                             int begin_bit, int end_bit);

/* ------------------------------------------------------------------------------------------
 * Fused path: what app/main.cpp:266-308 does per frame (process + forward + forward), as one
 * stream submission with no host round trip.  Numerically identical to the three stage calls.
 * ------------------------------------------------------------------------------------------ */

/* Bind caller-owned device arrays (the five buffers of app/main.cpp:180-186 after the upload at
 * :216-223).  Activations are already applied (app/gaussians.cpp:140-168). */
LCGS_API lcgs_status lcgs_scene_bind(lcgs_context* ctx, int num_gaussians, int sh_degree, const float* d_pos,
                                     const float* d_scale, const float* d_rotq, const float* d_sh,
                                     const float* d_opacity);
/* Same from host arrays: allocates device copies owned by the context (app/main.cpp:180-186,216-223). */
LCGS_API lcgs_status lcgs_scene_upload(lcgs_context* ctx, int num_gaussians, int sh_degree, const float* h_pos,
                                       const float* h_scale, const float* h_rotq, const float* h_sh,
                                       const float* h_opacity);

/* read_gs_ply (app/gaussians.cpp:75-171) + the upload of app/main.cpp:180-186,216-223 in one call, with the
 * de-interleave and the activations done on the device: the binary vertex records are copied to
 * the GPU as they lie in the file and become the five activated arrays there.  Equal to lcgs_ply_read +
 * lcgs_scene_upload except for exp() (device libm vs host libm, <= 2 ulp in scale and opacity).  ascii files and
 * files with non-float columns take the host path.  Synchronises the context's stream. */
LCGS_API lcgs_status lcgs_scene_load_ply(lcgs_context* ctx, const char* path, int* num_gaussians);
/* Splat order of a scene the CONTEXT owns (lcgs_scene_load_ply, lcgs_scene_upload).  Default LCGS_ORDER_SPATIAL: both
 * finish with lcgs_scene_reorder_spatial -- same images (the blend order is by depth; see that call for the one caveat on exactly
 * equal depths), the splats of a view in runs of consecutive rows (what it buys: DESIGN.md 9).  Per-splat outputs (radii, gradients) then follow the new order;
 * lcgs_scene_permutation maps rows back to file indices.  LCGS_ORDER_FILE keeps the file's order, like the reference
 * (app/gaussians.cpp:93-168).  Arrays bound with lcgs_scene_bind always keep the caller's order. */
typedef enum lcgs_splat_order { LCGS_ORDER_FILE = 0, LCGS_ORDER_SPATIAL = 1 } lcgs_splat_order;
LCGS_API lcgs_status lcgs_set_ingest_order(lcgs_context* ctx, int order);
/* *d_perm: context-owned device array, (*d_perm)[r] = file index of splat r; NULL while the scene is in file / caller order.
 * Valid until the next call that binds, loads or re-orders a scene. */
LCGS_API lcgs_status lcgs_scene_permutation(lcgs_context* ctx, const uint32_t** d_perm);
/* Device pointers of the bound scene (any of the outputs may be NULL).  The arrays of a scene the CONTEXT owns are read-only
 * through these pointers: the context keeps data derived from them (the permutation; 16-byte {position, extent bound} rows
 * that let the frame's cull pass read 16 instead of 40 bytes per splat).  The library's own writers (lcgs_adam_step & co.
 * with these arrays as `activated`) drop the derived rows by themselves; after changing the arrays any other way, bind them
 * again (lcgs_scene_bind with the same pointers), which rebuilds the rows.  Caller-owned arrays (lcgs_scene_bind of anything
 * else) carry no derived data and may change between frames freely. */
LCGS_API lcgs_status lcgs_scene_pointers(lcgs_context* ctx, int* num_gaussians, int* sh_degree, const float** d_pos,
                                         const float** d_scale, const float** d_rotq, const float** d_sh,
                                         const float** d_opacity);
/* "I changed the bound arrays behind the library's back" (a write through a const-cast of the pointers above, a tensor that
 * aliases them, another process): everything any live context of the process derived from them is dropped -- the cull rows
 * (rebuilt at once for a context-owned scene, on ctx's stream), the f16 coefficient copy, the kept state of the last frame.
 * Order it behind the writes.  Not needed after the library's own writers (lcgs_adam_step & co., the sharded steps'
 * all-gather), which do the same by themselves, whichever context they are issued through. */
LCGS_API lcgs_status lcgs_scene_modified(lcgs_context* ctx);
/* Diagnostics: how many of the derived rows in use for the bound arrays are NOT what the arrays say now (0 = consistent, or
 * nothing derived in use).  A test-mode guard against a forgotten lcgs_scene_modified.  Synchronises. */
LCGS_API lcgs_status lcgs_debug_verify_derived(lcgs_context* ctx, int64_t* stale_rows);

/* Caller-owned arrays that do not change from frame to frame (a trained scene being viewed: the reference's own use,
 * app/main.cpp:180-223 uploads once) can have the same derived rows as a context-owned scene: the three arrays are declared
 * static, the context builds the 16-byte {position, extent bound} rows once (on its stream), and every fused frame whose
 * position / scale / rotation arrays are EXACTLY these pointers (lcgs_render_forward after lcgs_scene_bind) and that asks for
 * no radii culls from them (a frame that returns the reference's radii array -- the stage operators in deferred mode do --
 * projects every splat anyway; measured effect: DESIGN.md 3).  The declaration lasts until it is repeated (same
 * pointers: "the contents changed"), withdrawn (num_gaussians = 0 or d_pos = NULL), or the library itself writes the arrays
 * (lcgs_adam_step & co.), or the context builds rows for a scene of its own (lcgs_scene_upload / lcgs_scene_load_ply /
 * lcgs_scene_reorder_spatial: one set of rows per context).  Changing the arrays behind a standing declaration gives wrong
 * frames. */
LCGS_API lcgs_status lcgs_scene_declare_static(lcgs_context* ctx, int num_gaussians, const float* d_pos, const float* d_scale,
                                               const float* d_rotq);

/* Opt-in reduced-precision SH for the fused forward.  enable != 0 converts the bound degree-3
 * coefficients to an f16 copy owned by the context (on the device; call again after the coefficients change, and
 * after every lcgs_scene_bind); the per-frame colour pass then reads 96 instead of 192 bytes per on-screen splat.
 * f16 keeps 11 significant bits, so this path is outside the 1e-4 image bar by construction (observed: ~1e-3);
 * the backward and the stage-level operators keep reading the f32 coefficients.  enable == 0 returns to f32. */
LCGS_API lcgs_status lcgs_scene_use_half_sh(lcgs_context* ctx, int enable);

/* Opt-in footprint (level-of-detail) cull for the fused frame (the reference only names LOD on its
 * roadmap, doc/roadmap.md:8).  min_radius_px > 0: a splat whose reference radius -- ceil(3 sqrt(lambda_max)) in pixels,
 * gs_tile_splatter/shader.cpp:145-148 -- is below it is treated as touching no tile (radius 0, not counted in
 * num_rendered, no gradient).  The low-pass filter and the max(0.1, .) under the root make 3 the smallest radius a splat
 * can have (3 sqrt(0.3 + sqrt(0.1)) = 2.35), so 4 drops exactly the splats at that floor.  This changes the image (a quality / speed trade): it is outside the 1e-4
 * parity bar against the unculled frame by construction, never the default, and the stage-level operators ignore it.
 * 0 switches it off. */
LCGS_API lcgs_status lcgs_set_lod(lcgs_context* ctx, int min_radius_px);

/* Copies the bound scene to host arrays sized like lcgs_scene_upload's inputs (NULL outputs are skipped). */
LCGS_API lcgs_status lcgs_scene_download(lcgs_context* ctx, float* h_pos, float* h_scale, float* h_rotq, float* h_sh,
                                         float* h_opacity);

/* One frame.  d_img: 3*H*W floats, CHW.  d_radii: P ints or NULL.  If num_rendered is non-NULL the call
 * synchronises the stream and stores the reference's num_rendered (sum of tiles touched); if NULL the
 * call only enqueues work.  keep_state != 0 keeps what lcgs_render_backward needs. */
LCGS_API lcgs_status lcgs_render_forward(lcgs_context* ctx, const lcgs_camera* camera, const float bg_color[3],
                                         float scale_modifier, float* d_img, int32_t* d_radii,
                                         int keep_state, int* num_rendered);

/* A batch of views of the bound scene (no counterpart in the reference, whose app renders one
 * hard-coded camera, app/main.cpp:191-207).  d_imgs[i]: 3*H_i*W_i floats, CHW, for cameras[i].  Enqueues only: the
 * work is ordered after what is already on the context's stream, and the stream waits for the whole batch; views
 * alternate between two internal workspaces so that two frames are in flight.  Per-view counters are not returned
 * (lcgs_get_frame_stats describes the last view rendered by the context itself). */
LCGS_API lcgs_status lcgs_render_forward_batch(lcgs_context* ctx, int num_views, const lcgs_camera* cameras,
                                               const float bg_color[3], float scale_modifier, float* const* d_imgs);

/* Per-stage device time of the last lcgs_render_forward / lcgs_render_backward in milliseconds
 * (hipEvent pairs on the context's stream).  Enable with lcgs_set_profiling(ctx, 1). */
#define LCGS_MAX_STAGES 16
typedef struct lcgs_stage_times {
    int         count;
    const char* name[LCGS_MAX_STAGES];
    float       ms[LCGS_MAX_STAGES];
} lcgs_stage_times;
LCGS_API lcgs_status lcgs_set_profiling(lcgs_context* ctx, int enabled);
LCGS_API lcgs_status lcgs_get_stage_times(lcgs_context* ctx, lcgs_stage_times* out);

/* Counters of the last frame (valid after a synchronising call): P, visible splats (radius>0 and
 * >=1 tile), reference num_rendered, pairs actually sorted, tiles. */
typedef struct lcgs_frame_stats {
    int64_t num_gaussians;
    int64_t num_visible;
    int64_t num_rendered;
    int64_t num_pairs;
    int64_t num_tiles;
    int64_t equal_depth_unresolved; /* always 0 since round 3: runs of EXACTLY equal depths of any length are blended in file
                                     * order in a re-ordered scene too (runs beyond 4096 members are sorted through global
                                     * scratch; see lcgs_scene_reorder_spatial).  Kept for ABI stability. */
    int64_t list_shift; /* granularity of the last frame's pair lists (num_pairs, lcgs_debug_last_lists): 0 per 16 x 16 tile,
                         * 1 per block of 2 x 2 tiles (frames without backward state of a context whose frames exceed ~3 M
                         * per-tile pairs); round 6 */
} lcgs_frame_stats;
LCGS_API lcgs_status lcgs_get_frame_stats(lcgs_context* ctx, lcgs_frame_stats* out);

/* Diagnostics: the sorted per-tile lists of the last fused frame expressed in ORIGINAL splat indices
 * (what the reference's accel.point_list holds after GSTileSplatter::forward, proxy.h:62) and the tile
 * ranges (proxy.h:63).  d_list: num_pairs entries; d_ranges: 2 * tiles.  Either may be NULL.  Synchronises.
 * At the granularity the frame used (lcgs_set_list_policy below, lcgs_frame_stats.list_shift): per tile, or per block of
 * 2 x 2 tiles -- then the first ceil(grid_x / 2) * ceil(grid_y / 2) ranges are the blocks', row-major, and the rest are zero. */
LCGS_API lcgs_status lcgs_debug_last_lists(lcgs_context* ctx, uint32_t* d_list, uint32_t* d_ranges);
/* The granularity of the fused frame's sorted pair lists.  LCGS_LISTS_PER_TILE: one list per 16 x 16 tile, the reference's
 * (gs_tile_splatter/impl.cpp:135-149).  LCGS_LISTS_PER_BLOCK: one per block of 2 x 2 tiles -- fewer pairs to duplicate and
 * partition; every tile's workgroup walks its block's list and takes the entries whose pruned rect covers it, the same
 * per-pixel sequence and the same image bit for bit; frames with keep_state != 0 stay per tile (the backward walks per-tile
 * lists).  LCGS_LISTS_AUTO (default): per block while the last synchronised frame had >= 3 M per-tile pairs
 * and >= 2.2 tiles per on-screen splat, per tile otherwise (and for a context's first frame).  lcgs_frame_stats.list_shift
 * says what the last frame used; lcgs_debug_last_lists returns the lists at that granularity. */
#define LCGS_LISTS_PER_TILE 0
#define LCGS_LISTS_PER_BLOCK 1
#define LCGS_LISTS_AUTO 2
LCGS_API lcgs_status lcgs_set_list_policy(lcgs_context* ctx, int policy);
/* Diagnostics: what the last frame with keep_state != 0 kept per pixel -- the final transmittance and the 1-based position,
 * within the pixel's tile list, of its last contributor (the values the reference computes and drops,
 * gs_tile_splatter/shader.cpp:219-220,252,273).  width * height entries each; either may be NULL.  Synchronises. */
LCGS_API lcgs_status lcgs_debug_last_state(lcgs_context* ctx, float* d_final_T, uint32_t* d_n_contrib);

/* Diagnostics: the compositing loop's exp (`exp(power)`, gs_tile_splatter/shader.cpp:258) evaluated on the device for n
 * values.  The reference's exp is whatever LuisaCompute's JIT maps it to (unpinned); this library defines it as a fixed
 * sequence of binary32 operations (csrc/kernels/gs_math.hpp::blend_exp, <= 2.73 ulp on [-6, 0]) so that a CPU restatement
 * can reproduce the frame bit for bit.  Domain -86 <= x <= 0; outside it the result is unspecified.  Synchronises. */
LCGS_API lcgs_status lcgs_debug_blend_exp(lcgs_context* ctx, const float* d_x, float* d_out, int64_t n);

/* Backward of the last lcgs_render_forward(keep_state=1) -- no counterpart in the reference
 * (README.md:70); specified in DESIGN.md.  All outputs are device pointers, overwritten:
 * dL/dpos[3P], dL/dscale[3P] (activated scale), dL/drotq[4P] (r,x,y,z, as stored),
 * dL/dsh[P*(deg+1)^2*3], dL/dopacity[P] (activated). */
typedef struct lcgs_grads {
    float* d_dL_dpos;
    float* d_dL_dscale;
    float* d_dL_drotq;
    float* d_dL_dsh;
    float* d_dL_dopacity;
} lcgs_grads;
LCGS_API lcgs_status lcgs_render_backward(lcgs_context* ctx, const float* d_dL_dimg, const lcgs_grads* grads);
/* The same dense gradients ADDED to what the arrays hold (no zero-fill): the second and later views of a multi-view batch
 * whose first view went through lcgs_render_backward.  One optimiser step (and, on several GPUs, one gradient collective)
 * per batch instead of per view: B views per GPU amortise the gradient all-reduce B times. */
LCGS_API lcgs_status lcgs_render_backward_accumulate(lcgs_context* ctx, const float* d_dL_dimg, const lcgs_grads* grads);
/* The same gradients as compact rows: row r of every output belongs to the r-th on-screen splat of that forward
 * frame (ascending splat index; lcgs_visible_rows names the splats).  Only those rows are written -- consecutive
 * rows, no zero-fill of the other P - V (what the dense variant's scattered stores cost: DESIGN.md 5).  Buffers need one row per on-screen splat (lcgs_frame_stats.num_visible; P rows always
 * suffice).  For single-GPU steps: lcgs_adam_step(visible_only = 2) consumes this layout directly; gradients that
 * are to be summed over views (RCCL all-reduce) need the dense variant. */
LCGS_API lcgs_status lcgs_render_backward_compact(lcgs_context* ctx, const float* d_dL_dimg, const lcgs_grads* grads);
/* Row list of the last forward frame: (*d_rows)[r] = splat index of compact row r, *d_count = address of the
 * device-side row count.  Context-owned device memory, valid until the context's next forward frame. */
LCGS_API lcgs_status lcgs_visible_rows(lcgs_context* ctx, const uint32_t** d_rows, const uint32_t** d_count);

/* Spatial order (ingest option; no counterpart in the reference, whose buffers stay in file order).
 * Re-orders the context's scene along a Morton curve of the positions: afterwards the context renders from its own
 * re-ordered copy of the five arrays (a scene bound with lcgs_scene_bind is copied, the caller's arrays are left
 * alone and no longer read).  Splat r of the new order is splat d_perm[r] of the old one (d_perm: P entries, device,
 * may be NULL); every per-splat output of later calls (radii, gradients, lcgs_visible_rows) follows the new order,
 * lcgs_scene_pointers returns the new arrays.  Images are unchanged: the blend order is by depth, and splats of exactly
 * equal depth are still blended in ascending FILE index like the reference (a pass behind the depth sort restores that
 * order inside every run of equal depth keys, of any length).  Why: the splats of a view then sit in long runs
 * of consecutive rows instead of being scattered over every DRAM page (figures: DESIGN.md 9).  Synchronises the context's
 * stream. */
LCGS_API lcgs_status lcgs_scene_reorder_spatial(lcgs_context* ctx, uint32_t* d_perm);

/* Optimiser step (the reference only names training on its roadmap, doc/roadmap.md:4).
 * The scene is parameterised as in 3DGS training: raw.pos / raw.sh are the values themselves,
 * scale = exp(raw.scale), opacity = sigmoid(raw.opacity), rotq = raw.rotq / |raw.rotq|.  One call maps the
 * gradients w.r.t. the ACTIVATED values (lcgs_render_backward's outputs, possibly summed over views) to the raw
 * parameters, applies Adam (torch.optim.Adam semantics, per-attribute learning rates, `step` counts from 1) to
 * raw / m / v in place and rewrites the activated arrays (`activated`; pos and sh may alias raw.pos / raw.sh, or be
 * separate buffers -- the arrays bound to the renderer -- which are then rewritten too).
 * All pointers are device arrays laid out like the scene (3P, 3P, 4P, P*(deg+1)^2*3, P).
 * visible_only != 0: only the splats that reached the screen in the last lcgs_render_forward of this context are
 * touched ("sparse Adam").  Enqueues on the context's stream. */
typedef struct lcgs_params {
    float* pos;
    float* scale;
    float* rotq;
    float* sh;
    float* opacity;
} lcgs_params;
typedef struct lcgs_adam_config {
    float lr_pos, lr_sh_dc, lr_sh_rest, lr_opacity, lr_scale, lr_rot;
    float beta1, beta2, eps;
    int   step;         /* 1, 2, ... (bias correction) */
    int   visible_only; /* 0: every splat (dense Adam); 1: on-screen splats only, gradients laid out per splat;
                         * 2: on-screen splats only, gradients in lcgs_render_backward_compact's row layout */
} lcgs_adam_config;
LCGS_API lcgs_status lcgs_adam_step(lcgs_context* ctx, int num_gaussians, int sh_degree, const lcgs_adam_config* cfg,
                                    const lcgs_grads* grads, const lcgs_params* raw, const lcgs_params* m,
                                    const lcgs_params* v, const lcgs_params* activated);

/* Single-GPU training step without gradient arrays ("training without python binding", doc/roadmap.md:4, taken one step
 * further): the backward of the last lcgs_render_forward(keep_state = 1) frame with the on-screen-only Adam update
 * (lcgs_adam_step, visible_only semantics) applied in the kernel that forms the per-splat gradients.  Same result, bit for
 * bit, as lcgs_render_backward_compact + lcgs_adam_step(visible_only = 2); no gradient row is written or read back
 * (2 x 236 bytes per on-screen splat less).  cfg->visible_only is ignored (the step is on-screen-only by construction).
 * The fused kernel covers sh_degree 3 with 16-byte-aligned rotq / sh rows; anything else runs the two calls internally. */
LCGS_API lcgs_status lcgs_render_backward_adam(lcgs_context* ctx, const float* d_dL_dimg, int num_gaussians, int sh_degree,
                                               const lcgs_adam_config* cfg, const lcgs_params* raw, const lcgs_params* m,
                                               const lcgs_params* v, const lcgs_params* activated);

/* ------------------------------------------------------------------------------------------
 * Multi-GPU.  No counterpart in the reference, which drives one device (app/main.cpp:162-163).
 * One process per GPU; the scene is replicated; a batch of views is sharded one view per GPU (no data-path collective in
 * the forward); the backward's dense per-splat gradients are summed over the ranks by RCCL over xGMI.  RCCL is bound
 * at run time (the copy the process already carries, else the ROCm installation's); without it these calls fail with
 * LCGS_ERR_NO_DEVICE and nothing else in the library is affected.
 * ------------------------------------------------------------------------------------------ */
typedef struct lcgs_comm lcgs_comm;
/* The rendezvous token (an ncclUniqueId): made by ONE rank, carried to the others by whatever channel the host has
 * (a pipe, a file, torch.distributed's store -- lcgs-app and the Python driver show two). */
typedef struct lcgs_comm_id {
    char bytes[128];
} lcgs_comm_id;
LCGS_API lcgs_status lcgs_comm_unique_id(lcgs_comm_id* out);
/* Collective over all ranks (ncclCommInitRank).  The communicator is attached to `ctx` (one per context): from then on
 * lcgs_render_backward runs its per-splat pass as splat-range slices so that lcgs_grads_allreduce can overlap it. */
LCGS_API lcgs_status lcgs_comm_create(lcgs_context* ctx, const lcgs_comm_id* id, int rank, int world_size,
                                      lcgs_comm** out);
LCGS_API lcgs_status lcgs_comm_destroy(lcgs_comm* comm);
LCGS_API lcgs_status lcgs_comm_info(const lcgs_comm* comm, int* rank, int* world_size);
/* Transport of lcgs_grads_allreduce.  LCGS_TRANSPORT_F32 (default): exact f32 sums, chunked behind the backward.
 * LCGS_TRANSPORT_F16 (opt-in): every attribute is scaled by a power of two all ranks agree on (their largest magnitude,
 * max-reduced first, lands below 16384 / N), rounded to f16, summed as f16 and scaled back -- half the bytes on the wire
 * for about sqrt(N) x 5e-4 of relative error in the norm: at the 1e-3 gradient bar for a node of eight, outside it beyond,
 * hence never the default.  Zeros stay exact zeros.  The sharded step always moves f32. */
typedef enum lcgs_transport { LCGS_TRANSPORT_F32 = 0, LCGS_TRANSPORT_F16 = 1 } lcgs_transport;
LCGS_API lcgs_status lcgs_comm_set_transport(lcgs_comm* comm, int transport);
/* Row ownership of the sharded step: rank r owns rows [first, first + count) with count = floor(P / N); the last
 * P mod N rows ("the tail") are kept up to date by every rank. */
LCGS_API void lcgs_comm_shard_rows(int64_t num_gaussians, int world_size, int rank, int64_t* first, int64_t* count);

/* In-place sum over all ranks of the five dense gradient arrays (lcgs_render_backward's outputs).  Issued on the
 * communicator's own stream in splat-range chunks, each behind the event of the backward slice that produced its rows, so
 * the first chunks travel while the backward's tail is still computing; the context's stream then waits for the sums
 * (enqueue lcgs_adam_step right behind it).  1.45 GB per GPU for the 6.1 M-splat scene: 2 (N-1)/N of that crosses xGMI
 * per GPU.
 * Collective discipline: EVERY rank must call it once per step with the same num_gaussians / sh_degree / transport, whether
 * or not it ran a backward (a rank without a view passes zero-filled arrays); the number and sizes of the RCCL calls
 * issued depend on those shared values only.  Call it directly behind the backward: work the caller enqueues on the
 * context's stream between the two must not touch the gradient arrays (the first chunks start behind their slice of the
 * backward, only the last one behind the stream's tail). */
LCGS_API lcgs_status lcgs_grads_allreduce(lcgs_context* ctx, lcgs_comm* comm, int num_gaussians, int sh_degree,
                                          const lcgs_grads* grads);
/* The whole optimiser step at N > 1 without replicated optimiser work: reduce-scatter of the gradients -> lcgs_adam_step on
 * the rank's own rows (+ the tail) -> all-gather of the refreshed ACTIVATED arrays.  Same bytes on the wire as
 * lcgs_grads_allreduce + a dense lcgs_adam_step, exact same arithmetic per row, 1/N of the optimiser's HBM traffic per
 * GPU.  raw / m / v are authoritative for the own rows (and the tail) only; `activated` is complete on every rank
 * afterwards.  Dense only (cfg->visible_only must be 0). */
LCGS_API lcgs_status lcgs_adam_step_sharded(lcgs_context* ctx, lcgs_comm* comm, int num_gaussians, int sh_degree,
                                            const lcgs_adam_config* cfg, const lcgs_grads* grads, const lcgs_params* raw,
                                            const lcgs_params* m, const lcgs_params* v, const lcgs_params* activated);

/* ---- Sparse gradient exchange (no counterpart in the reference: single device, app/main.cpp:162-163) -------------------
 * A view touches only its on-screen splats (39 % of the 6.1 M-splat stand-in), so in the REDUCE half of a step a rank needs
 * to hand a row's owner only the rows its views touched: (N-1)/N x touched x (236 + 4) bytes out per GPU instead of
 * (N-1)/N x P x 236.  Dense rows stay the layout of every array; only the wire format is sparse.
 *
 * lcgs_comm_track_touched_rows(comm, 1): from now on every dense lcgs_render_backward on the communicator's context flags
 * its frame's on-screen rows (lcgs_render_backward starts a new set, lcgs_render_backward_accumulate adds to it).  ALL
 * ranks must enable it before the step's first backward; a rank that runs no backward in a step has an empty set (and must
 * hold zeros in its gradient arrays).
 *
 * lcgs_adam_step_sparse: the whole optimiser step, same contract and same result as lcgs_adam_step_sharded up to the order
 * of the f32 sums: touched rows -> one message per owner (ncclSend / ncclRecv, sizes agreed through one small all-gather:
 * the step's only host synchronisation) -> the owner adds the messages to its rows in rank order -> lcgs_adam_step on the
 * own rows (+ the < N tail rows, which are all-reduced densely) -> all-gather of the refreshed ACTIVATED arrays.
 * cfg->visible_only must be 0 (dense Adam semantics: rows nobody touched still decay their moments).
 *
 * The three stages are also exported one by one, for a host that brings its own transport (multi_gpu.TorchCollective
 * runs them over a torch.distributed process group): lcgs_sparse_touched_rows (synchronises; the set is consumed),
 * lcgs_sparse_pack, lcgs_sparse_accumulate.  A message of `count` rows is count x (1 + 11 + (deg+1)^2 x 3) 4-byte words:
 * [row indices, ascending][pos rows][scale rows][rotq rows][sh rows][opacity rows]. */
#define LCGS_MAX_RANKS 64
typedef struct lcgs_sparse_rows {
    const uint32_t* d_rows;   /* device: the touched rows, ascending (owned by the communicator, valid until the next call) */
    int64_t         num_rows;
    /* rows [owner_first[o], owner_first[o + 1]) of d_rows lie in rank o's shard (o < N); [owner_first[N], num_rows) is the tail */
    int64_t owner_first[LCGS_MAX_RANKS + 2];
} lcgs_sparse_rows;
/* What the last collective call of this communicator moved over the wire (per GPU; computed from the actual counts). */
typedef struct lcgs_comm_stats {
    int64_t bytes_sent, bytes_received; /* by this rank, all collectives of the call */
    int64_t touched_rows;               /* sparse step: rows this rank's views touched (0 for the dense calls) */
    int     collective_groups;          /* RCCL groups / calls issued: identical on every rank by construction */
} lcgs_comm_stats;
LCGS_API lcgs_status lcgs_comm_track_touched_rows(lcgs_comm* comm, int enable);
LCGS_API lcgs_status lcgs_comm_get_stats(const lcgs_comm* comm, lcgs_comm_stats* out);
LCGS_API lcgs_status lcgs_adam_step_sparse(lcgs_context* ctx, lcgs_comm* comm, int num_gaussians, int sh_degree,
                                           const lcgs_adam_config* cfg, const lcgs_grads* grads, const lcgs_params* raw,
                                           const lcgs_params* m, const lcgs_params* v, const lcgs_params* activated);
/* world_size: the N of the exchange the rows are sharded for (the communicator's own for lcgs_adam_step_sparse; a host with
 * its own transport passes its own -- the communicator then only serves as the context's row tracker) */
LCGS_API lcgs_status lcgs_sparse_touched_rows(lcgs_context* ctx, lcgs_comm* comm, int num_gaussians, int world_size,
                                              lcgs_sparse_rows* out);
LCGS_API int64_t     lcgs_sparse_message_words(int64_t count, int sh_degree);
LCGS_API lcgs_status lcgs_sparse_pack(lcgs_context* ctx, int sh_degree, const lcgs_grads* grads, const uint32_t* d_rows,
                                      int64_t count, float* d_msg);
/* Adds the message's rows to the dense gradient rows.  The row indices come out of the (received) message: only rows in
 * [row_first, row_first + row_count) -- the caller's shard, or [0, num_gaussians) -- are accepted, any other index is
 * dropped without a write (a short, corrupt or mismatched-P message cannot reach memory outside the arrays). */
LCGS_API lcgs_status lcgs_sparse_accumulate(lcgs_context* ctx, int sh_degree, const lcgs_grads* grads, const float* d_msg,
                                            int64_t count, int64_t row_first, int64_t row_count);

/* ------------------------------------------------------------------------------------------
 * Splat ownership: the frame in two halves.  No reference counterpart (one device, app/main.cpp:162-163).
 * A GPU that OWNS the rows [row_first, row_first + row_count) of the scene runs the per-splat half of a view's frame on them
 * (lcgs_owner_project: cull, compaction, SH colour) and gets the packed records of the rows that reach that view's screen --
 * LCGS_OWNER_RECORD_FLOATS 4-byte words a row: pixel mean (2), conic (3), opacity, rgb (3), depth, pruned tile rect (2 x u32) --
 * with their GLOBAL row indices, ascending.  The GPU that RENDERS the view concatenates what the owners sent in owner order
 * (= ascending rows) and runs the rest of the frame on it (lcgs_owner_render: depth sort with the reference's order of equal
 * depths, duplication, tile partition, compositing): the same image, bit for bit, as lcgs_render_forward of the whole scene.
 * lcgs_owner_render_backward returns the 2-D gradients of those rows (LCGS_OWNER_GRAD_FLOATS words a row: mean (2), conic
 * (3), opacity, rgb (3), 3 unused) and lcgs_owner_backward -- on the owner, per view slot -- turns its rows' share into
 * parameter gradients at rows row_first ... of the full-size arrays (the first view of a step zero-fills the range, the
 * others add).  These four calls leave the transport between the two halves to the caller (multi_gpu.TorchCollective.owner_step:
 * send / recv over a process group); lcgs_owner_step_forward / _backward below are the same step with RCCL as the wire.  slot: the view's index inside the step, < LCGS_MAX_OWNER_VIEWS; its buffers live until re-used.
 * ------------------------------------------------------------------------------------------ */
#define LCGS_MAX_OWNER_VIEWS 16
#define LCGS_OWNER_RECORD_FLOATS 12
#define LCGS_OWNER_GRAD_FLOATS 12
LCGS_API lcgs_status lcgs_owner_project(lcgs_context* ctx, int slot, const lcgs_camera* camera, float scale_modifier,
                                        int row_first, int row_count, int keep_state, uint32_t* d_rows /* [row_count] */,
                                        float* d_records /* [row_count x 12], 16-byte aligned */,
                                        int* num_rows); /* rows written; synchronises.  NULL: the call only enqueues --
                                                         * the count stays on the device until lcgs_owner_counts */
/* The row counts of slots [first_slot, first_slot + num_slots) with ONE synchronisation: an owner projects its rows for every
 * view of a step (N asynchronous lcgs_owner_project calls, num_rows = NULL) and reads the N message sizes at once. */
LCGS_API lcgs_status lcgs_owner_counts(lcgs_context* ctx, int first_slot, int num_slots, int* num_rows /* [num_slots] */);
LCGS_API lcgs_status lcgs_owner_render(lcgs_context* ctx, const lcgs_camera* camera, const float bg_color[3], int num_rows,
                                       const uint32_t* d_rows, const float* d_records, float* d_img, int keep_state);
LCGS_API lcgs_status lcgs_owner_render_backward(lcgs_context* ctx, const float* d_dL_dimg, float* d_grads2d);
LCGS_API lcgs_status lcgs_owner_backward(lcgs_context* ctx, int slot, const float* d_grads2d, const lcgs_grads* grads,
                                         int accumulate);

/* The ownership step WITH its transport (round 5): rank r of the communicator owns the rows lcgs_comm_owner_rows(P, N, r) of
 * the scene bound to ctx (equal contiguous shards, the P mod N tail with the last rank) and renders view r of cameras[N].
 *   lcgs_owner_step_forward   projects the own rows for all N views (asynchronous), agrees the message sizes through one
 *                             small all-gather + read-back (the step's one host synchronisation besides the view's own
 *                             pair-buffer check), sends every view's rank its rows ([u32 row] + 48-byte record each,
 *                             ncclSend / ncclRecv in one group on the communicator's stream), receives every owner's rows of
 *                             its own view in owner order (= ascending rows) and renders the view into d_img -- the image
 *                             lcgs_render_forward of the whole scene gives, bit for bit.
 *   lcgs_owner_step_backward  differentiates the view, returns every owner the 48-byte 2-D gradient rows of its rows,
 *                             receives those of the own rows for all N views and maps them to parameter gradients at the
 *                             own rows of the full-size arrays in `grads` (view 0 overwrites, the others add; rows outside
 *                             the own range are not touched).  The caller applies lcgs_adam_step to its own rows.
 * Every rank calls both, once per step, in this order.  lcgs_comm_get_stats reports the bytes of the whole step after the
 * backward call ((4 + 48) bytes a row out as an owner, 48 back as a view's renderer, + the count table).
 * An in-process stand-in for RCCL exists for tests and single-GPU rehearsals: lcgs_loopback_group_create(N) +
 * lcgs_comm_create_loopback(ctx_r, group, r) give N communicators for N contexts ON ONE DEVICE, driven by one host thread
 * each (N <= LCGS_MAX_OWNER_VIEWS).  Every collective call of this header (lcgs_grads_allreduce with the f32 transport,
 * lcgs_adam_step_sharded, lcgs_adam_step_sparse, the two calls above) then runs the same code with device-to-device copies
 * and rank-ordered sums as the wire (RCCL refuses a second rank on a device).  Not a production transport. */
LCGS_API void        lcgs_comm_owner_rows(int64_t num_gaussians, int world_size, int rank, int64_t* first, int64_t* count);
LCGS_API lcgs_status lcgs_owner_step_forward(lcgs_context* ctx, lcgs_comm* comm, const lcgs_camera* cameras /* [world_size] */,
                                             const float bg_color[3], float scale_modifier, float* d_img);
LCGS_API lcgs_status lcgs_owner_step_backward(lcgs_context* ctx, lcgs_comm* comm, const float* d_dL_dimg, const lcgs_grads* grads);
/* The step WITHOUT a host read-back (round 6; opt-in per communicator, every rank alike).  Once a step's count table is known,
 * the next step sizes its messages from it -- n + n / 4 + 1024 rows, clipped to the owner's range: every rank derives the
 * same sizes from the same all-gathered table -- the true counts stay on the device (the view's frame is built from padded
 * per-owner segments), the frame's pair-buffer check is not read back either, and no call between lcgs_owner_step_forward
 * and the end of lcgs_owner_step_backward waits for the device.  A message that was clipped or a frame whose pairs were
 * truncated raises a flag that is max-reduced over the ranks behind the forward half; lcgs_owner_step_finish -- REQUIRED
 * after the backward call of such a step, before the gradients are used -- waits for that flag only (the device is in the
 * backward by then), adopts the step's table for the next one and reports *redo = 1 on EVERY rank if ANY rank's step was
 * short: forward and backward are then called again (that repetition reads its sizes back and grows what was too small).
 * The first step of a communicator, and any step whose padded segments would not fit the workspace the scene sizes, read
 * back as before (finish then reports 0 at once).  Bytes on the wire: <= 1.25 x the exact step's + 1024 rows a message. */
LCGS_API lcgs_status lcgs_owner_step_set_async(lcgs_comm* comm, int enable);
LCGS_API lcgs_status lcgs_owner_step_finish(lcgs_context* ctx, lcgs_comm* comm, int* redo);
/* What a communicator says about ITSELF before anything is timed (round 6): a collective -- every rank calls it -- of three
 * phases, each watched against timeout_s from the calling thread (a phase that never returns is reported in `timed_out`, the
 * call returns LCGS_ERR_STATE and the process should exit: the worker thread is left behind):
 *   1. a 1 KB all-reduce (rank r contributes r + 1 in 256 floats; every element must come back as N (N + 1) / 2);
 *   2. one group of point-to-point messages: to every peer (to itself at N = 1) a zero-byte and a one-byte message, and back;
 *   3. an ownership step on a 10 000-splat scene the call generates, with and without read-back: the rank's image must be its
 *      fused frame of that scene bit for bit, its own rows' gradients the sum of the N views' ordinary backward passes (1e-4).
 *      (N > LCGS_MAX_OWNER_VIEWS: not run, owner_step_ok = -1.)
 * The context's scene binding is put back afterwards (lcgs_scene_bind of what was bound: frame state and the opt-in f16
 * coefficient copy are reset).  Returns LCGS_OK only when every phase ran and was right. */
typedef struct lcgs_comm_selftest_report {
    int    world_size, rank;
    int    allreduce_ok;
    double allreduce_ms;
    int    p2p_ok;
    double p2p_ms;
    int    owner_step_ok; /* 1 right, 0 wrong, -1 not run */
    double owner_step_ms;
    double owner_max_grad_err; /* worst relative L2 error of an attribute over the own rows, over the three steps */
    int    timed_out;          /* 0, or the phase (1..3) that did not finish within timeout_s */
    char   message[256];
} lcgs_comm_selftest_report;
LCGS_API lcgs_status lcgs_comm_selftest(lcgs_context* ctx, lcgs_comm* comm, double timeout_s, lcgs_comm_selftest_report* out);
typedef struct lcgs_loopback_group lcgs_loopback_group;
LCGS_API lcgs_status lcgs_loopback_group_create(int world_size, lcgs_loopback_group** out);
LCGS_API lcgs_status lcgs_loopback_group_destroy(lcgs_loopback_group* group); /* after its communicators */
LCGS_API lcgs_status lcgs_comm_create_loopback(lcgs_context* ctx, lcgs_loopback_group* group, int rank, lcgs_comm** out);

/* ------------------------------------------------------------------------------------------
 * Scene ingest / image egress (host side of `render(ply, camera) -> image`)
 * ------------------------------------------------------------------------------------------ */

/* GaussiansData (app/gaussians.h:15-35) after read_gs_ply (app/gaussians.cpp:75-171): activated,
 * repacked host arrays.  Free with lcgs_scene_host_free. */
typedef struct lcgs_scene_host {
    int    num_gaussians;
    int    sh_degree; /* 3 */
    float* pos;       /* 3P */
    float* feature;   /* P*16*3, [j*48 + k*3 + c] */
    float* opacity;   /* P, sigmoid applied */
    float* scale;     /* 3P, exp applied */
    float* rotq;      /* 4P, (r,x,y,z) normalised */
} lcgs_scene_host;

/* read_gs_ply (app/gaussians.cpp:75-171): binary-little-endian or ascii PLY, properties looked up by
 * name (x y z f_dc_0..2 f_rest_0..44 opacity scale_0..2 rot_0..3), other properties ignored. */
LCGS_API lcgs_status lcgs_ply_read(const char* path, lcgs_scene_host* out);
/* Inverse (raw, un-activated values in, INRIA property order with nx ny nz = 0) -- used to write the
 * synthetic stand-in scenes; no counterpart in the reference. */
LCGS_API lcgs_status lcgs_ply_write_raw(const char* path, int num_gaussians, const float* pos,
                                        const float* f_dc /*3P*/, const float* f_rest /*45P, channel-major*/,
                                        const float* opacity_logit, const float* log_scale, const float* rot);
LCGS_API void lcgs_scene_host_free(lcgs_scene_host* scene);

/* Deterministic synthetic stand-ins for the four BASELINE scenes: kind 0 = "synth_object"
 * (lego/chair-like), kind 1 = "synth_unbounded" (bicycle/garden-like).  Counter-based RNG, so any
 * sub-range [first, first+count) can be generated independently.  Outputs are ACTIVATED arrays in the
 * layout of lcgs_scene_host (caller-allocated, count*{3,48,1,3,4} floats). */
LCGS_API lcgs_status lcgs_synth_scene(int kind, uint64_t seed, int64_t first, int64_t count, float* pos,
                                      float* feature, float* opacity, float* scale, float* rotq);

/* app/main.cpp:323-335: CHW float -> HWC uint8, vertical flip, truncating *255.  Host buffers. */
LCGS_API void lcgs_image_to_rgb8(int width, int height, const float* h_img_chw, uint8_t* h_rgb);
/* Same on the device (d_img CHW float -> d_rgb HWC uint8), enqueued on the context's stream. */
LCGS_API lcgs_status lcgs_image_to_rgb8_device(lcgs_context* ctx, int width, int height, const float* d_img_chw,
                                               uint8_t* d_rgb);
/* The simplest photometric loss for the training step (the reference only names training on its roadmap,
 * doc/roadmap.md:4): *d_loss = mean((img - target)^2) over the 3*H*W samples, d_dL_dimg = its gradient 2 (img - target) /
 * (3 H W) -- what lcgs_render_backward takes.  All device pointers; enqueued on the context's stream. */
LCGS_API lcgs_status lcgs_l2_loss_backward(lcgs_context* ctx, int width, int height, const float* d_img_chw,
                                           const float* d_target_chw, float* d_dL_dimg, float* d_loss);
/* The views of ONE optimiser step (a multi-view batch on this GPU): for every view lcgs_render_forward(keep_state) ->
 * lcgs_l2_loss_backward against d_targets[j] -> lcgs_render_backward, the dense gradients summed into `grads` (the first
 * view overwrites, the others add) and d_losses[j] = view j's loss (device, num_views floats).  Same results as those
 * calls one view after the other (gradient sums up to float addition order), but the views alternate between the context
 * and its sibling (the one lcgs_render_forward_batch uses), so that a view's forward runs beside the previous view's
 * backward.  The last view runs on `ctx`: with a communicator attached its
 * preprocess-backward is sliced and lcgs_grads_allreduce overlaps it as usual.  Ordered after prior work on the context's
 * stream; the stream waits for the whole batch.  Afterwards the context holds the LAST view's frame state. */
LCGS_API lcgs_status lcgs_fit_views(lcgs_context* ctx, int num_views, const lcgs_camera* cameras, const float bg_color[3],
                                    float scale_modifier, const float* const* d_targets, const lcgs_grads* grads,
                                    float* d_losses);
/* stbi_write_png(name, w, h, 3, data, 0) (app/main.cpp:339): 8-bit RGB PNG (stored deflate blocks). */
LCGS_API lcgs_status lcgs_write_png(const char* path, int width, int height, const uint8_t* h_rgb);

#ifdef __cplusplus
}
#endif
#endif /* LCGS_HIP_H */
