/*
 * lcgs_hip.h -- C ABI of liblcgs_hip.so: the MI355X-native (gfx950) implementation of the LuisaComputeGaussianSplatting hot
 * path (SH colour -> 3D->2D projection -> tile keys -> sort -> per-tile alpha compositing) plus the matching backward.
 * Every entry point names the reference interface it replaces (paths relative to the reference repository).
 *
 * Contract, for every function unless it says otherwise:
 *   - `d_` pointers are DEVICE pointers owned by the caller, laid out as the reference's flat float buffers
 *     (lcgs/include/lcgs/proxy.h:21-73: pos xyz.., rotq rxyz.., SH [P][16][3]); `h_` pointers are host memory;
 *   - structs are POD, passed by pointer, copied before the call returns;
 *   - the return value is an lcgs_status (no exception crosses the boundary); lcgs_last_error() has the message;
 *   - work is ENQUEUED on the context's stream; "synchronises" is said where a call waits for the device;
 *   - one lcgs_context per (GPU, stream), used by one host thread at a time (like the reference's operator objects,
 *     gs_tile_splatter.h:23,50-55); different contexts may be driven from different threads.
 * Long-form notes (why, measured effects, history) per declaration: docs/API_NOTES.md.
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
    LCGS_ERR_HIP           = 2, /* a HIP / RCCL call failed; see lcgs_last_error() */
    LCGS_ERR_NO_DEVICE     = 3, /* no gfx950 device (there is no CPU fallback), or RCCL absent for lcgs_comm_* */
    LCGS_ERR_OUT_OF_MEMORY = 4,
    LCGS_ERR_CAPACITY      = 5, /* pair buffers too small (the reference does not check, app/main.cpp:245) */
    LCGS_ERR_IO            = 6,
    LCGS_ERR_FORMAT        = 7,
    LCGS_ERR_STATE         = 8  /* e.g. backward without a preceding keep_state forward */
} lcgs_status;

typedef struct lcgs_context lcgs_context;

/* struct Camera, lcgs/include/lcgs/util/camera.h:15-25 */
typedef struct lcgs_camera {
    float position[3], front[3], up[3], right[3];
    float fov;          /* vertical, degrees (default 60) */
    float aspect_ratio; /* width / height */
    int   width, height;
} lcgs_camera;

/* ---- library / context ------------------------------------------------------------------------------------------ */
LCGS_API const char* lcgs_version(void);
LCGS_API const char* lcgs_last_error(void); /* thread-local message of the last failing call */
/* Context::create_device + Device::create_stream (app/main.cpp:162-163) and the three create(Device&) calls.
 * stream: a hipStream_t (NULL = the null stream).  Kernels are precompiled for gfx950. */
LCGS_API lcgs_status lcgs_create(int device_id, void* stream, lcgs_context** out_ctx);
LCGS_API lcgs_status lcgs_destroy(lcgs_context* ctx);
LCGS_API lcgs_status lcgs_set_stream(lcgs_context* ctx, void* stream);
LCGS_API lcgs_status lcgs_synchronize(lcgs_context* ctx); /* Stream::synchronize (app/main.cpp:223,315) */

/* ---- host camera helpers: get_lookat_cam, local_to_world, world_to_local, projection matrix (util/camera.h:74-82,
 * 27-36, 38-51, 54-72); matrices column-major float[16], m[c*4+r] ------------------------------------------------ */
LCGS_API void lcgs_get_lookat_cam(const float pos[3], const float target[3], const float world_up[3], lcgs_camera* out_cam);
LCGS_API void lcgs_local_to_world_matrix(const lcgs_camera* cam, float m[16]);
LCGS_API void lcgs_world_to_local_matrix(const lcgs_camera* cam, float m[16]);
LCGS_API void lcgs_projection_matrix(float tanfovx, float tanfovy, float znear, float zfar, float m[16]);

/* ---- stage-level operators: one call per reference class method, the reference's buffers bit for bit ----------- */
/* SHProcessor::process (sh_preprocessor.h:30-37, sh_preprocessor.cpp:169-188): d_color[3P] = clamp(SH(dir) + 0.5, 0, 1) */
LCGS_API lcgs_status lcgs_sh_process(lcgs_context* ctx, int num_points, const float* d_pos, const lcgs_camera* camera,
                                     const float* d_sh, float* d_color, int level, int channel);
/* GSProjector::forward (gs_projector.h:37-43, gs_projector/impl.cpp:26-93): means_2d[2P] NDC, covs_2d[3P], depth[P];
 * splats with view z < 0.2 are left unwritten (gs_projector/shader.cpp:121). */
LCGS_API lcgs_status lcgs_project_forward(lcgs_context* ctx, int num_gaussians, const float* d_pos, const float* d_scale,
                                          const float* d_rotq, float scale_modifier, float* d_means_2d, float* d_covs_2d,
                                          float* d_depth, const lcgs_camera* camera, int use_focal);
/* GSTileSplatterAccelProxy (proxy.h:56-64) + the pair capacity the app fixes at 20 M (app/main.cpp:245) */
typedef struct lcgs_tile_accel {
    uint32_t* tiles_touched;            /* P */
    uint32_t* point_offsets;            /* P */
    uint64_t* point_list_keys_unsorted; /* capacity */
    uint32_t* point_list_unsorted;      /* capacity */
    uint64_t* point_list_keys;          /* capacity */
    uint32_t* point_list;               /* capacity */
    uint32_t* ranges;                   /* 2 * ceil(W/16) * ceil(H/16) */
    int64_t   capacity;                 /* (tile, splat) pairs the four pair buffers hold */
} lcgs_tile_accel;
/* GSTileSplatterInputProxy (proxy.h:43-54); means_2d / conic are overwritten in place (NDC -> pixel, cov -> conic) */
typedef struct lcgs_tile_input {
    int          num_gaussians;
    float        bg_color[3];
    float*       means_2d;         /* 2P */
    const float* depth_features;   /* P  */
    float*       conic;            /* 3P */
    const float* color_features;   /* 3P */
    const float* opacity_features; /* P  */
} lcgs_tile_input;
/* GSSplatForwardOutputProxy (proxy.h:66-71); target_img planar CHW; final_T / n_contrib optional (NULL to skip) */
typedef struct lcgs_tile_output {
    int       height, width;
    float*    target_img; /* 3*H*W */
    int32_t*  radii;      /* P */
    float*    final_T;    /* H*W */
    uint32_t* n_contrib;  /* H*W */
} lcgs_tile_output;
/* GSTileSplatter::forward (gs_tile_splatter.h:28-35, impl.cpp:63-180).  *num_rendered = the reference's return value
 * (0: nothing drawn, image untouched, impl.cpp:109).  Synchronises once, like impl.cpp:106-107. */
LCGS_API lcgs_status lcgs_tile_splat_forward(lcgs_context* ctx, const lcgs_tile_accel* accel, const lcgs_tile_input* input,
                                             const lcgs_tile_output* output, int use_focal, int* num_rendered);
/* LCGS_STAGES_EXACT (default): each call runs at once.  LCGS_STAGES_DEFERRED: lcgs_sh_process / lcgs_project_forward only
 * record; a splat call on exactly their outputs renders the fused frame into target_img / radii (same bits, intermediates
 * not written); anything else (a non-matching call, lcgs_stage_flush, lcgs_synchronize, a mode switch) runs what was recorded. */
typedef enum lcgs_stage_mode { LCGS_STAGES_EXACT = 0, LCGS_STAGES_DEFERRED = 1 } lcgs_stage_mode;
LCGS_API lcgs_status lcgs_set_stage_mode(lcgs_context* ctx, int mode);
LCGS_API lcgs_status lcgs_stage_flush(lcgs_context* ctx);
/* lcpp DeviceScan::InclusiveSum (call site impl.cpp:104) and DeviceRadixSort::SortPairs<ulong,uint> (impl.cpp:135-143):
 * u32 wrap-around sum; stable ascending sort on bits [begin_bit, end_bit).  Temp storage is the context's. */
LCGS_API lcgs_status lcgs_inclusive_sum_u32(lcgs_context* ctx, const uint32_t* d_in, uint32_t* d_out, int64_t n);
LCGS_API lcgs_status lcgs_sort_pairs_u64_u32(lcgs_context* ctx, const uint64_t* d_keys_in, uint64_t* d_keys_out,
                                             const uint32_t* d_vals_in, uint32_t* d_vals_out, int64_t n, int begin_bit, int end_bit);

/* ---- the scene a context renders ------------------------------------------------------------------------------- */
/* Caller-owned activated arrays (the five buffers of app/main.cpp:180-186 after :216-223); rotq 16-byte aligned. */
LCGS_API lcgs_status lcgs_scene_bind(lcgs_context* ctx, int num_gaussians, int sh_degree, const float* d_pos, const float* d_scale,
                                     const float* d_rotq, const float* d_sh, const float* d_opacity);
/* Host arrays -> device copies OWNED by the context.  Synchronises. */
LCGS_API lcgs_status lcgs_scene_upload(lcgs_context* ctx, int num_gaussians, int sh_degree, const float* h_pos, const float* h_scale,
                                       const float* h_rotq, const float* h_sh, const float* h_opacity);
/* read_gs_ply (app/gaussians.cpp:75-171) + upload, de-interleave and activations on the device (exp within 2 ulp of the
 * host reader's; ascii / non-float files take the host path).  Context-owned.  Synchronises. */
LCGS_API lcgs_status lcgs_scene_load_ply(lcgs_context* ctx, const char* path, int* num_gaussians);
/* Order of a context-owned scene: LCGS_ORDER_SPATIAL (default; upload / load_ply finish with lcgs_scene_reorder_spatial,
 * per-splat outputs follow the new order) or LCGS_ORDER_FILE.  Bound arrays always keep the caller's order. */
typedef enum lcgs_splat_order { LCGS_ORDER_FILE = 0, LCGS_ORDER_SPATIAL = 1 } lcgs_splat_order;
LCGS_API lcgs_status lcgs_set_ingest_order(lcgs_context* ctx, int order);
/* Morton re-order of the context's scene (a bound scene is copied first).  d_perm[P] (device, nullable): new row r = old
 * row d_perm[r].  Same images bit for bit (equal depths still blend in file order).  Synchronises. */
LCGS_API lcgs_status lcgs_scene_reorder_spatial(lcgs_context* ctx, uint32_t* d_perm);
/* *d_perm: context-owned, (*d_perm)[r] = file index of row r; NULL in file / caller order.  Valid until the next bind / load. */
LCGS_API lcgs_status lcgs_scene_permutation(lcgs_context* ctx, const uint32_t** d_perm);
/* Device pointers of the bound scene (outputs nullable).  A context-owned scene is READ-ONLY through them: the context keeps
 * rows derived from it; after writing it any other way than through this library, call lcgs_scene_modified. */
LCGS_API lcgs_status lcgs_scene_pointers(lcgs_context* ctx, int* num_gaussians, int* sh_degree, const float** d_pos,
                                         const float** d_scale, const float** d_rotq, const float** d_sh, const float** d_opacity);
/* "The bound arrays were written behind the library's back": derived rows, the f16 coefficient copy and kept frame state
 * are dropped (rows rebuilt at once for a context-owned scene); other contexts on the same arrays are told and act at their
 * next frame.  Order it behind the writes. */
LCGS_API lcgs_status lcgs_scene_modified(lcgs_context* ctx);
/* Diagnostics: derived rows in use that no longer match the bound arrays (0 = consistent).  Synchronises. */
LCGS_API lcgs_status lcgs_debug_verify_derived(lcgs_context* ctx, int64_t* stale_rows);
/* Caller-owned arrays that do not change between frames may have the derived 16-byte cull rows too: declared for exactly
 * these three pointers, until repeated ("contents changed"), withdrawn (num_gaussians 0 / d_pos NULL) or written by the library. */
LCGS_API lcgs_status lcgs_scene_declare_static(lcgs_context* ctx, int num_gaussians, const float* d_pos, const float* d_scale,
                                               const float* d_rotq);
/* Opt-in f16 copy of the degree-3 SH coefficients for the fused forward's colour pass: image moves by ~1e-3, outside the
 * 1e-4 bar by construction.  Call again after the coefficients change / after lcgs_scene_bind.  0 = back to f32. */
LCGS_API lcgs_status lcgs_scene_use_half_sh(lcgs_context* ctx, int enable);
/* Opt-in footprint cull: splats whose reference radius (pixels, gs_tile_splatter/shader.cpp:145-148) is below
 * min_radius_px touch no tile.  Changes the image; never the default; 0 = off.  Stage-level operators ignore it. */
LCGS_API lcgs_status lcgs_set_lod(lcgs_context* ctx, int min_radius_px);
/* Bound scene -> host arrays sized like lcgs_scene_upload's inputs (NULL outputs skipped).  Synchronises. */
LCGS_API lcgs_status lcgs_scene_download(lcgs_context* ctx, float* h_pos, float* h_scale, float* h_rotq, float* h_sh, float* h_opacity);

/* ---- the fused frame: app/main.cpp:266-308 (process + forward + forward) as one stream submission -------------- */
/* d_img: 3*H*W floats CHW; d_radii: P ints or NULL.  num_rendered non-NULL: synchronises and stores the reference's
 * num_rendered; NULL: only enqueues.  keep_state != 0 keeps what lcgs_render_backward needs. */
LCGS_API lcgs_status lcgs_render_forward(lcgs_context* ctx, const lcgs_camera* camera, const float bg_color[3], float scale_modifier,
                                         float* d_img, int32_t* d_radii, int keep_state, int* num_rendered);
/* A batch of views (two frames in flight on sibling workspaces); d_imgs[i]: 3*H_i*W_i floats for cameras[i].  Enqueues;
 * the context's stream waits for the whole batch. */
LCGS_API lcgs_status lcgs_render_forward_batch(lcgs_context* ctx, int num_views, const lcgs_camera* cameras, const float bg_color[3],
                                               float scale_modifier, float* const* d_imgs);
/* Per-stage device time (ms, hipEvent pairs) of the last forward / backward, while profiling is on (frames then run in order). */
#define LCGS_MAX_STAGES 16
typedef struct lcgs_stage_times {
    int         count;
    const char* name[LCGS_MAX_STAGES];
    float       ms[LCGS_MAX_STAGES];
} lcgs_stage_times;
LCGS_API lcgs_status lcgs_set_profiling(lcgs_context* ctx, int enabled);
LCGS_API lcgs_status lcgs_get_stage_times(lcgs_context* ctx, lcgs_stage_times* out);
/* Counters of the last synchronised frame. */
typedef struct lcgs_frame_stats {
    int64_t num_gaussians;
    int64_t num_visible;            /* radius > 0 and >= 1 tile */
    int64_t num_rendered;           /* the reference's: sum of tiles of the unpruned rects */
    int64_t num_pairs;              /* pairs actually sorted (pruned; at list_shift's granularity) */
    int64_t num_tiles;
    int64_t equal_depth_unresolved; /* always 0 (kept for layout stability) */
    int64_t list_shift;             /* 0: lists per 16 x 16 tile; 1: per block of 2 x 2 tiles */
} lcgs_frame_stats;
LCGS_API lcgs_status lcgs_get_frame_stats(lcgs_context* ctx, lcgs_frame_stats* out);
/* Granularity of the fused frame's sorted pair lists: per tile like the reference (impl.cpp:135-149), per block of 2 x 2
 * tiles (same image bit for bit; fewer pairs through duplication / partition), or AUTO (default): per block for frames
 * without backward state while the last synchronised frame had >= 3 M per-tile pairs and >= 2.2 tiles per on-screen splat. */
#define LCGS_LISTS_PER_TILE 0
#define LCGS_LISTS_PER_BLOCK 1
#define LCGS_LISTS_AUTO 2
LCGS_API lcgs_status lcgs_set_list_policy(lcgs_context* ctx, int policy);
/* Diagnostics (all synchronise).  last_lists: the sorted lists in ORIGINAL splat indices (accel.point_list, proxy.h:62) and
 * ranges (proxy.h:63) at the last frame's granularity (per block: the first ceil(gx/2)*ceil(gy/2) ranges, rest zero).
 * last_state: per pixel final T and 1-based list position of the last contributor of the last keep_state frame
 * (shader.cpp:219-220,252,273).  blend_exp: the compositing loop's defined exp (<= 2.73 ulp on [-6, 0]) for n values in [-86, 0]. */
LCGS_API lcgs_status lcgs_debug_last_lists(lcgs_context* ctx, uint32_t* d_list, uint32_t* d_ranges);
LCGS_API lcgs_status lcgs_debug_last_state(lcgs_context* ctx, float* d_final_T, uint32_t* d_n_contrib);
LCGS_API lcgs_status lcgs_debug_blend_exp(lcgs_context* ctx, const float* d_x, float* d_out, int64_t n);

/* ---- backward of the last lcgs_render_forward(keep_state = 1) (no reference counterpart, README.md:70) ---- */
/* Outputs w.r.t. the ACTIVATED inputs: dL/dpos[3P], dL/dscale[3P], dL/drotq[4P] (as stored; 16-byte aligned),
 * dL/dsh[P*(deg+1)^2*3], dL/dopacity[P]. */
typedef struct lcgs_grads {
    float *d_dL_dpos, *d_dL_dscale, *d_dL_drotq, *d_dL_dsh, *d_dL_dopacity;
} lcgs_grads;
/* dense rows, overwritten (exact zeros off screen) / ADDED to what the arrays hold (further views of a batch) */
LCGS_API lcgs_status lcgs_render_backward(lcgs_context* ctx, const float* d_dL_dimg, const lcgs_grads* grads);
LCGS_API lcgs_status lcgs_render_backward_accumulate(lcgs_context* ctx, const float* d_dL_dimg, const lcgs_grads* grads);
/* Compact rows: row r = the frame's r-th on-screen splat (ascending index; lcgs_visible_rows); only those rows are written. */
LCGS_API lcgs_status lcgs_render_backward_compact(lcgs_context* ctx, const float* d_dL_dimg, const lcgs_grads* grads);
/* (*d_rows)[r] = splat of compact row r, *d_count = device address of the row count.  Valid until the next forward. */
LCGS_API lcgs_status lcgs_visible_rows(lcgs_context* ctx, const uint32_t** d_rows, const uint32_t** d_count);

/* ---- optimiser step (doc/roadmap.md:4 names training; 3DGS parameterisation: scale = exp, opacity = sigmoid, rotq normalised) ---- */
typedef struct lcgs_params {
    float *pos, *scale, *rotq, *sh, *opacity; /* device arrays laid out like the scene */
} lcgs_params;
typedef struct lcgs_adam_config {
    float lr_pos, lr_sh_dc, lr_sh_rest, lr_opacity, lr_scale, lr_rot;
    float beta1, beta2, eps;
    int   step;         /* 1, 2, ... (bias correction) */
    int   visible_only; /* 0 every splat; 1 on-screen splats, per-splat gradient rows; 2 on-screen, compact rows */
} lcgs_adam_config;
/* Gradients w.r.t. activated values -> Adam (torch.optim.Adam semantics) on raw / m / v in place -> `activated` rewritten
 * (pos / sh may alias raw).  visible_only refers to the context's last forward frame. */
LCGS_API lcgs_status lcgs_adam_step(lcgs_context* ctx, int num_gaussians, int sh_degree, const lcgs_adam_config* cfg, const lcgs_grads* grads,
                                    const lcgs_params* raw, const lcgs_params* m, const lcgs_params* v, const lcgs_params* activated);
/* lcgs_render_backward_compact + lcgs_adam_step(visible_only = 2) in one pass, no gradient arrays (same bits). */
LCGS_API lcgs_status lcgs_render_backward_adam(lcgs_context* ctx, const float* d_dL_dimg, int num_gaussians, int sh_degree,
                                               const lcgs_adam_config* cfg, const lcgs_params* raw, const lcgs_params* m,
                                               const lcgs_params* v, const lcgs_params* activated);
/* *d_loss = mean((img - target)^2), d_dL_dimg = its gradient (what lcgs_render_backward takes).  Device pointers. */
LCGS_API lcgs_status lcgs_l2_loss_backward(lcgs_context* ctx, int width, int height, const float* d_img_chw, const float* d_target_chw,
                                           float* d_dL_dimg, float* d_loss);
/* The views of ONE optimiser step: forward(keep_state) -> L2 loss vs d_targets[j] -> backward per view, dense gradients
 * summed into `grads`, d_losses[j] (device); views alternate between the context and its sibling.  The context's stream
 * waits for the batch; the context holds the last view's frame state. */
LCGS_API lcgs_status lcgs_fit_views(lcgs_context* ctx, int num_views, const lcgs_camera* cameras, const float bg_color[3],
                                    float scale_modifier, const float* const* d_targets, const lcgs_grads* grads, float* d_losses);

/* ---- multi-GPU (no reference counterpart: one device, app/main.cpp:162-163) ------------------------ */
/* One process per GPU, scene replicated, one view per GPU; RCCL is bound at run time (absent: LCGS_ERR_NO_DEVICE). */
typedef struct lcgs_comm lcgs_comm;
typedef struct lcgs_comm_id {
    char bytes[128]; /* an ncclUniqueId: made by ONE rank, carried to the others by the host */
} lcgs_comm_id;
LCGS_API lcgs_status lcgs_comm_unique_id(lcgs_comm_id* out);
/* Collective (ncclCommInitRank).  Attached to ctx (one per context): its dense backward then runs in slices. */
LCGS_API lcgs_status lcgs_comm_create(lcgs_context* ctx, const lcgs_comm_id* id, int rank, int world_size, lcgs_comm** out);
LCGS_API lcgs_status lcgs_comm_destroy(lcgs_comm* comm);
LCGS_API lcgs_status lcgs_comm_info(const lcgs_comm* comm, int* rank, int* world_size);
/* LCGS_TRANSPORT_F32 (default, exact) / LCGS_TRANSPORT_F16 (opt-in: half the bytes, ~sqrt(N) x 5e-4 relative) for lcgs_grads_allreduce */
typedef enum lcgs_transport { LCGS_TRANSPORT_F32 = 0, LCGS_TRANSPORT_F16 = 1 } lcgs_transport;
LCGS_API lcgs_status lcgs_comm_set_transport(lcgs_comm* comm, int transport);
/* Sharded / sparse steps: rank r owns rows [first, first + floor(P / N)); the P mod N tail rows are everybody's. */
LCGS_API void lcgs_comm_shard_rows(int64_t num_gaussians, int world_size, int rank, int64_t* first, int64_t* count);
/* In-place f32 sum over the ranks of the five dense gradient arrays, chunked behind the backward's slices on the
 * communicator's stream; the context's stream waits for it.  EVERY rank calls it once per step with the same
 * num_gaussians / sh_degree / transport (a rank without a view passes zeros), directly behind its backward. */
LCGS_API lcgs_status lcgs_grads_allreduce(lcgs_context* ctx, lcgs_comm* comm, int num_gaussians, int sh_degree, const lcgs_grads* grads);
/* Reduce-scatter -> lcgs_adam_step on the own rows (+ tail) -> all-gather of the ACTIVATED arrays.  Dense (visible_only 0). */
LCGS_API lcgs_status lcgs_adam_step_sharded(lcgs_context* ctx, lcgs_comm* comm, int num_gaussians, int sh_degree,
                                            const lcgs_adam_config* cfg, const lcgs_grads* grads, const lcgs_params* raw,
                                            const lcgs_params* m, const lcgs_params* v, const lcgs_params* activated);
/* Sparse reduce half: only rows the step's views touched travel ((1 + 59) words a row).  track_touched_rows(1) on ALL ranks
 * before the step's first backward; lcgs_adam_step_sparse = the sharded step with that reduce half (one host read-back for
 * the message sizes).  The three stages are exported for a host with its own transport. */
#define LCGS_MAX_RANKS 64
typedef struct lcgs_sparse_rows {
    const uint32_t* d_rows; /* touched rows, ascending (communicator-owned, valid until the next call) */
    int64_t         num_rows;
    int64_t         owner_first[LCGS_MAX_RANKS + 2]; /* rows [owner_first[o], owner_first[o+1]) lie in rank o's shard; [N].. = tail */
} lcgs_sparse_rows;
typedef struct lcgs_comm_stats { /* what the last collective call (or ownership step) moved, per GPU */
    int64_t bytes_sent, bytes_received;
    int64_t touched_rows;      /* sparse step: rows touched; ownership step: rows on this rank's screen */
    int     collective_groups; /* RCCL groups / calls issued: identical on every rank */
} lcgs_comm_stats;
LCGS_API lcgs_status lcgs_comm_track_touched_rows(lcgs_comm* comm, int enable);
LCGS_API lcgs_status lcgs_comm_get_stats(const lcgs_comm* comm, lcgs_comm_stats* out);
LCGS_API lcgs_status lcgs_adam_step_sparse(lcgs_context* ctx, lcgs_comm* comm, int num_gaussians, int sh_degree,
                                           const lcgs_adam_config* cfg, const lcgs_grads* grads, const lcgs_params* raw,
                                           const lcgs_params* m, const lcgs_params* v, const lcgs_params* activated);
/* (touched_rows synchronises and consumes the set; accumulate drops indices outside [row_first, row_first + row_count) unwritten) */
LCGS_API lcgs_status lcgs_sparse_touched_rows(lcgs_context* ctx, lcgs_comm* comm, int num_gaussians, int world_size, lcgs_sparse_rows* out);
LCGS_API int64_t     lcgs_sparse_message_words(int64_t count, int sh_degree);
LCGS_API lcgs_status lcgs_sparse_pack(lcgs_context* ctx, int sh_degree, const lcgs_grads* grads, const uint32_t* d_rows, int64_t count,
                                      float* d_msg);
LCGS_API lcgs_status lcgs_sparse_accumulate(lcgs_context* ctx, int sh_degree, const lcgs_grads* grads, const float* d_msg, int64_t count,
                                            int64_t row_first, int64_t row_count);

/* ---- splat ownership: the frame in two halves ---------------------------------------------------- */
/* An OWNER of rows [row_first, row_first + row_count) runs the per-splat half of a view's frame on them and gets the packed
 * records (12 words: pixel mean 2, conic 3, opacity, rgb 3, depth, pruned rect 2 x u32) of the rows on that view's screen with
 * their global row indices, ascending; the view's RENDERER runs the rest on the rows of all owners concatenated in owner
 * order (same image as lcgs_render_forward, bit for bit) and returns 12-word 2-D gradient rows; the owner maps them to
 * parameter gradients at its rows.  slot < LCGS_MAX_OWNER_VIEWS: the view's index inside the step.
 * project: d_rows[row_count], d_records[row_count x 12] (16-byte aligned); num_rows = rows written (synchronises), NULL:
 * enqueue only and read all slots' counts with ONE synchronisation through lcgs_owner_counts.
 * backward: accumulate 0 clears the range first, 1 adds the rows. */
#define LCGS_MAX_OWNER_VIEWS 16
#define LCGS_OWNER_RECORD_FLOATS 12
#define LCGS_OWNER_GRAD_FLOATS 12
LCGS_API lcgs_status lcgs_owner_project(lcgs_context* ctx, int slot, const lcgs_camera* camera, float scale_modifier, int row_first,
                                        int row_count, int keep_state, uint32_t* d_rows, float* d_records, int* num_rows);
/* The N views of a step over ONE row range in one call (view k -> slot first_slot + k; d_rows / d_records: host arrays of N
 * device pointers): N independent pipelines run side by side and are joined on the context's stream; enqueue only. */
LCGS_API lcgs_status lcgs_owner_project_views(lcgs_context* ctx, int first_slot, int num_views, const lcgs_camera* cameras,
                                              float scale_modifier, int row_first, int row_count, int keep_state,
                                              uint32_t* const* d_rows, float* const* d_records);
LCGS_API lcgs_status lcgs_owner_counts(lcgs_context* ctx, int first_slot, int num_slots, int* num_rows);
LCGS_API lcgs_status lcgs_owner_render(lcgs_context* ctx, const lcgs_camera* camera, const float bg_color[3], int num_rows,
                                       const uint32_t* d_rows, const float* d_records, float* d_img, int keep_state);
LCGS_API lcgs_status lcgs_owner_render_backward(lcgs_context* ctx, const float* d_dL_dimg, float* d_grads2d);
LCGS_API lcgs_status lcgs_owner_backward(lcgs_context* ctx, int slot, const float* d_grads2d, const lcgs_grads* grads, int accumulate);
/* The step WITH its transport (RCCL send / recv on the communicator's stream): rank r owns lcgs_comm_owner_rows(P, N, r)
 * (equal shards, tail with the last rank) and renders cameras[r] (cameras: world_size entries).  Every rank calls forward
 * then backward once per step; gradients land at the own rows of `grads` (other rows untouched); lcgs_comm_get_stats has
 * the step's bytes. */
LCGS_API void        lcgs_comm_owner_rows(int64_t num_gaussians, int world_size, int rank, int64_t* first, int64_t* count);
LCGS_API lcgs_status lcgs_owner_step_forward(lcgs_context* ctx, lcgs_comm* comm, const lcgs_camera* cameras, const float bg_color[3],
                                             float scale_modifier, float* d_img);
LCGS_API lcgs_status lcgs_owner_step_backward(lcgs_context* ctx, lcgs_comm* comm, const float* d_dL_dimg, const lcgs_grads* grads);
/* Opt-in (every rank alike): from the second step on NOTHING is read back -- messages are sized from the previous step's
 * all-gathered counts (n + n/4 + 1024 rows), counts and the pair-buffer verdict stay on the device.  Such a step MUST be
 * closed with lcgs_owner_step_finish before its gradients are used: it waits for the forward half's max-reduced verdict
 * only; *redo = 1 on EVERY rank if ANY rank's message was clipped / frame truncated: call forward + backward again. */
LCGS_API lcgs_status lcgs_owner_step_set_async(lcgs_comm* comm, int enable);
LCGS_API lcgs_status lcgs_owner_step_finish(lcgs_context* ctx, lcgs_comm* comm, int* redo);
/* A collective self-test: 1 KB all-reduce; a zero- and a one-byte message to every peer in one group; an ownership step on a
 * 10 000-splat scratch scene (image = the fused frame bit for bit, gradients 1e-4).  Each phase against timeout_s; a phase
 * that never returns is reported (timed_out) and the process should exit.  The context's scene binding is put back. */
typedef struct lcgs_comm_selftest_report {
    int    world_size, rank;
    int    allreduce_ok;
    double allreduce_ms;
    int    p2p_ok;
    double p2p_ms;
    int    owner_step_ok; /* 1 right, 0 wrong, -1 not run (N > LCGS_MAX_OWNER_VIEWS) */
    double owner_step_ms;
    double owner_max_grad_err;
    int    timed_out; /* 0, or the phase (1..3) that did not finish */
    char   message[256];
} lcgs_comm_selftest_report;
LCGS_API lcgs_status lcgs_comm_selftest(lcgs_context* ctx, lcgs_comm* comm, double timeout_s, lcgs_comm_selftest_report* out);
/* In-process stand-in for RCCL (tests, single-GPU rehearsals): N communicators for N contexts ON ONE DEVICE, one host
 * thread each; every collective of this header runs the same code over device copies.  Not a production transport. */
typedef struct lcgs_loopback_group lcgs_loopback_group;
LCGS_API lcgs_status lcgs_loopback_group_create(int world_size, lcgs_loopback_group** out);
LCGS_API lcgs_status lcgs_loopback_group_destroy(lcgs_loopback_group* group); /* after its communicators */
LCGS_API lcgs_status lcgs_comm_create_loopback(lcgs_context* ctx, lcgs_loopback_group* group, int rank, lcgs_comm** out);

/* ---- scene ingest / image egress (host side of render(ply, camera) -> image) ------------------------------------ */
/* GaussiansData (app/gaussians.h:15-35) after read_gs_ply: activated host arrays.  Free with lcgs_scene_host_free. */
typedef struct lcgs_scene_host {
    int    num_gaussians;
    int    sh_degree; /* 3 */
    float* pos;       /* 3P */
    float* feature;   /* P*16*3, [j*48 + k*3 + c] */
    float* opacity;   /* P, sigmoid applied */
    float* scale;     /* 3P, exp applied */
    float* rotq;      /* 4P, (r,x,y,z) normalised */
} lcgs_scene_host;
/* read_gs_ply (app/gaussians.cpp:75-171): binary-LE or ascii PLY, properties by name, others ignored. */
LCGS_API lcgs_status lcgs_ply_read(const char* path, lcgs_scene_host* out);
/* Raw (un-activated) values, INRIA property order, nx ny nz = 0 (f_dc 3P, f_rest 45P channel-major): writes the stand-ins. */
LCGS_API lcgs_status lcgs_ply_write_raw(const char* path, int num_gaussians, const float* pos, const float* f_dc, const float* f_rest,
                                        const float* opacity_logit, const float* log_scale, const float* rot);
LCGS_API void lcgs_scene_host_free(lcgs_scene_host* scene);
/* Deterministic stand-ins for the BASELINE scenes: kind 0 object-like, 1 unbounded-like; counter-based RNG
 * (any sub-range independently); ACTIVATED arrays, caller-allocated (count * {3, 48, 1, 3, 4} floats). */
LCGS_API lcgs_status lcgs_synth_scene(int kind, uint64_t seed, int64_t first, int64_t count, float* pos, float* feature, float* opacity,
                                      float* scale, float* rotq);
/* app/main.cpp:323-335: CHW float -> HWC uint8, vertical flip, truncating *255 (host / device variants). */
LCGS_API void        lcgs_image_to_rgb8(int width, int height, const float* h_img_chw, uint8_t* h_rgb);
LCGS_API lcgs_status lcgs_image_to_rgb8_device(lcgs_context* ctx, int width, int height, const float* d_img_chw, uint8_t* d_rgb);
/* stbi_write_png(name, w, h, 3, data, 0) (app/main.cpp:339): 8-bit RGB PNG. */
LCGS_API lcgs_status lcgs_write_png(const char* path, int width, int height, const uint8_t* h_rgb);

#ifdef __cplusplus
}
#endif
#endif /* LCGS_HIP_H */
