// context.hpp -- the (opaque to callers) lcgs_context, shared by the translation units behind the C ABI
// (abi_*.cpp: frames, scene, optimiser -- see abi_internal.hpp; host/comm.cpp: the RCCL gradient collectives).
#pragma once

#include <atomic>

#include "common.hpp"
#include "kernels/launch.hpp"

struct lcgs_comm;

namespace lcgs
{
constexpr int kMaxEvents    = LCGS_MAX_STAGES + 1;
constexpr int kMaxGradSlices = 16; // splat-range slices of the dense gradient rows (chunked all-reduce)
// (host/comm.cpp) a context that goes away before its communicator: the communicator forgets it
void comm_forget_context(lcgs_comm* comm);
// (host/comm.cpp) sparse exchange: a dense backward flags the rows of the frame it has just differentiated
lcgs_status comm_mark_touched(lcgs_comm* comm, const uint32_t* vis_index, const uint32_t* d_counts, int64_t P, int64_t hint_V,
                              bool accumulate, hipStream_t stream);
namespace abi
{
// (abi_scene.cpp) position / scale / rotation rows inside these arrays are being written: every live context drops the rows
// it derived from them (the cull pass's {position, extent bound} rows)
void scene_arrays_written(lcgs_context* ctx, const float* pos, const float* scale, const float* rotq);
} // namespace abi
} // namespace lcgs

using lcgs::CamParams;
using lcgs::DeviceBuffer;

struct lcgs_context {
    int         device = 0;
    hipStream_t stream = nullptr;

    // scene (lcgs_scene_bind / lcgs_scene_upload)
    int          P = 0, sh_deg = 3;
    const float *pos = nullptr, *scale = nullptr, *rotq = nullptr, *sh = nullptr, *opacity = nullptr;
    DeviceBuffer owned[5];
    // Splat order of a context-owned scene: lcgs_scene_load_ply re-orders along a Morton curve unless told otherwise
    // (lcgs_set_ingest_order); scene_perm[r] = file index of splat r while perm_valid (lcgs_scene_permutation)
    int          lod_min_radius = 0; // lcgs_set_lod: opt-in footprint cull of the fused frame (0 = off)
    int          ingest_order = 1; // LCGS_ORDER_SPATIAL
    DeviceBuffer scene_perm;
    bool         perm_valid = false;      // the BOUND arrays are owned[] in the order scene_perm describes
    bool         perm_for_owned = false;  // scene_perm describes the current contents of owned[] (bound or not)
    DeviceBuffer sh_half;            // opt-in f16 copy of sh for the fused forward's colour pass (lcgs_scene_use_half_sh)
    bool         use_half_sh = false;
    // {position, extent bound} rows of a scene the context OWNS (abi_scene.cpp refresh_cull_bound): the cull pass's phase 1
    // reads these 16 bytes instead of 40 bytes of position + scale + rotation.  cull_bound is what frames use -- the
    // context's own buffer, a sibling's borrowed pointer, or NULL (caller-bound arrays, or owned arrays the library has
    // written since: lcgs_adam_step & co. drop it; binding the owned arrays again rebuilds it)
    DeviceBuffer          cull_bound_buf;
    DeviceBuffer          verify_ws; // lcgs_debug_verify_derived: one counter
    const float4*         cull_bound = nullptr;
    // ... valid for exactly these arrays (the context's own, or caller arrays declared static: lcgs_scene_declare_static);
    // a frame uses the rows only when its position / scale / rotation arrays are these (cull_rows())
    struct {
        const float *pos = nullptr, *scale = nullptr, *rotq = nullptr;
        int          P = 0;
    } cull_key;
    // THREADING: a context (with its batch sibling) is used by one host thread at a time; the only cross-thread traffic is
    // this word.  Another thread's writer (an optimiser step through ITS context on arrays this one renders, or its
    // lcgs_scene_modified) never touches this context's fields: it compares against the copy of the keys this context
    // PUBLISHED under the registry mutex (abi_scene.cpp) and posts bits here; the owning thread honours them at its next
    // use -- kRowsStale at once (cull_rows() below; cleared when the rows are rebuilt), kSceneModified in
    // frame_state_valid() / note_foreign_writes().
    static constexpr uint32_t kRowsStale = 1u, kSceneModified = 2u;
    std::atomic<uint32_t>     foreign_writes{ 0 };
    const float4* cull_rows() const
    {
        if (foreign_writes.load(std::memory_order_acquire) & kRowsStale) return nullptr;
        return (cull_bound && pos == cull_key.pos && scale == cull_key.scale && rotq == cull_key.rotq && P == cull_key.P)
                   ? cull_bound
                   : nullptr;
    }
    // another thread declared the bound arrays modified: the f16 coefficient copy and the last frame's kept state are stale
    void note_foreign_writes()
    {
        if (foreign_writes.load(std::memory_order_acquire) & kSceneModified) {
            foreign_writes.fetch_and(~kSceneModified, std::memory_order_acq_rel);
            use_half_sh = false;
            last.valid  = false;
        }
    }
    bool frame_state_valid()
    {
        note_foreign_writes();
        return last.valid;
    }

    // workspace of the fused frame
    DeviceBuffer cull_slab, chunk_info, chunk_base; // the cull pass's per-chunk output (fused_forward.hip k_cull_compact)
    DeviceBuffer recs, sortk[2], sortv[2], vis_index, rects, rects_sorted, pairk[2], pairv[2], zero_ws[3], counts, sort_ws,
        expand_ws, final_T, n_contrib, list_idx, grads2d, strip_masks, shjac, tie_ws, fused_grads, bwd_counter, keep_list,
        keep_ranges; // (the last two: per-tile lists a keep-state frame on per-block lists writes for its backward)
    bool         last_has_jac = false; // the last keep_state frame stored the colour Jacobian (degree 3)
    // zero_ws holds what a frame needs zeroed: the tile ranges.  Three
    // copies rotate: while frame N runs, the auxiliary stream clears the copy of frame N + 2.  Two frames ahead, not
    // one, so that no wait is needed when a frame starts: the fill issued during frame N - 1 sits on the auxiliary
    // stream in front of frame N's record builder, whose completion frame N's renderer waited for -- and frame N + 1
    // starts behind that renderer.
    size_t    zero_scan_bytes = 0, zero_bytes = 0;
    int       zero_cur        = 0;
    bool      zero_ready[3]   = { false, false, false };
    uint32_t* ranges          = nullptr; // tile ranges of the last frame (inside zero_ws[...])
    // device-resident per-call parameters + the captured frame graph (replayed while its key is unchanged)
    DeviceBuffer   frame_params;
    hipGraphExec_t graph_exec = nullptr;
    struct GraphKey {
        const void *pos = nullptr, *scale = nullptr, *rotq = nullptr, *sh = nullptr, *opacity = nullptr, *sh_half = nullptr,
                   *img = nullptr, *radii = nullptr, *cull_bound = nullptr;
        int         P = -1, sh_deg = -1, width = 0, height = 0, keep_state = -1, list_shift = -1;
        int64_t     hint_V = -1, hint_L = -1;
        uint32_t    capacity = 0;
        hipStream_t stream = nullptr;
        bool        operator==(const GraphKey& o) const
        {
            return pos == o.pos && scale == o.scale && rotq == o.rotq && sh == o.sh && opacity == o.opacity &&
                   sh_half == o.sh_half && img == o.img && radii == o.radii && cull_bound == o.cull_bound && P == o.P && sh_deg == o.sh_deg && width == o.width &&
                   height == o.height && keep_state == o.keep_state && list_shift == o.list_shift && hint_V == o.hint_V && hint_L == o.hint_L &&
                   capacity == o.capacity && stream == o.stream;
        }
    } graph_key;
    // Longest-list-first tile schedule: a scheduling hint, so a frame uses the order derived from the PREVIOUS
    // frame's list lengths (computed on the auxiliary stream while that frame rendered); only the first frame of a
    // resolution computes its own order in line.
    DeviceBuffer tile_order[2];
    int          order_cur = 0;  // tile_order[order_cur] is the newest complete order ...
    uint32_t     order_G   = 0;  // ... valid for this many tiles (0: none yet)
    uint32_t*    last_tile_order = nullptr; // the order the last frame rendered with (reused by the backward)
    // Persistent renderers (render.hip / backward.hip PERSIST): a bounded grid of `k x CUs` workgroups that pull tiles from a
    // counter, instead of one workgroup per tile.  Used while SEVERAL frames are in flight (camera batches, lcgs_fit_views):
    // the cap leaves wave slots, registers and LDS free on every CU, so the other frame's short sort-chain kernels start at
    // once instead of queueing behind thousands of pending tile workgroups.  0 = one workgroup per tile (in-order frames).
    int       num_cus = 0;
    int       persist_in_flight = 0;   // k while several frames are in flight (lcgs_render_forward_batch / lcgs_fit_views)
    int       persist_forced = -1;     // tuning hook LCGS_RENDER_WGS_PER_CU: k for EVERY frame when >= 0
    int       persist_bwd_in_flight = 0, persist_bwd_forced = -1; // the same for the render-backward (LCGS_BWD_WGS_PER_CU)
    bool      frames_in_flight = false; // set around the calls of a batch
    uint32_t* work_counters = nullptr; // [0] forward renderer, [1] render-backward: inside the frame's zeroed block
    // CU-partitioned streams (tuning hook LCGS_CHAIN_CUS=K, measured in round 4): the sort chain on a stream masked to K
    // CUs, record builder + renderer on the complement
    hipStream_t chain_stream = nullptr, render_stream = nullptr;
    hipEvent_t  ev_begin = nullptr, ev_chain = nullptr;
    bool use_graph = false; // opt-in (LCGS_GRAPH=1): measured no gain on MI355X, the short kernels are GPU-latency-bound
    // second stream: work that is independent of the sort chain (record building; gradient zero-fill) overlaps it
    hipStream_t aux_stream = nullptr;
    hipEvent_t  ev_fork = nullptr, ev_join = nullptr, ev_ranges = nullptr, ev_aux_done = nullptr, ev_render = nullptr,
                ev_counts = nullptr;
    bool        aux_pending = false, counts_pending = false;
    // keep_state frames clear the 2-D gradient rows on the auxiliary stream (beside the renderer) so that the backward
    // does not start with a 40 us zero-fill; consumed by the first backward of that frame
    hipEvent_t  ev_g2d_zero = nullptr;
    bool        g2d_zeroed  = false;
    // lcgs_render_forward_batch: a sibling context (own workspace, own streams) that renders every other view, so
    // that one view's latency-bound sort chain overlaps the other's bandwidth- and VALU-bound kernels
    lcgs_context* twin         = nullptr;
    hipStream_t   twin_stream  = nullptr; // owned
    hipEvent_t    ev_batch_fork = nullptr, ev_batch_join = nullptr;
    // lcgs_fit_views: the view's image and loss gradient (per context: two views are in flight), the order of the
    // backward passes across the two contexts
    DeviceBuffer fit_img, fit_dL;
    hipEvent_t   ev_fit_bwd = nullptr;
    // launch-size hints from the last synchronised frame (live counts stay on the device; larger counts are
    // still handled correctly by chunk striding)
    int64_t hint_V = 0, hint_L = 0, hint_Lb = 0; // (hint_L: frames with per-tile lists; hint_Lb: per 2 x 2-tile block)
    // workspace of the stage-level path / primitives
    DeviceBuffer st_keys_tmp, st_vals_tmp, st_sort_temp, st_scan_temp, st_scalar;
    DeviceBuffer st_flags, st_u32[8], st_keys_exp, st_vals_exp; // the splatter's sort-before-duplicate (lcgs_tile_splat_forward)
    DeviceBuffer st_win, st_win2, st_offs; // ... and the output-balanced pair copies' window table (one word per 1024 pairs) and offsets
    uint32_t     pair_capacity = 0;
    uint32_t*    h_counts      = nullptr; // pinned, 8 x u32
    uint32_t*    h_stage       = nullptr; // pinned, 4 x u32: the stage-level splatter's one read-back (num_rendered & co.)
    uint32_t*    h_stage_dev   = nullptr; // the same words as the device sees them (posted by the compaction's scan launch)

    // state of the last forward (for backward and stats)
    struct {
        bool      valid = false;
        bool      has_state = false;
        CamParams cp;
        float     bg[3];
        float     scale_modifier;
        int       list_buf = 0; // pairv[list_buf] holds the sorted per-tile lists (dense ids)
    } last;
    lcgs_frame_stats stats{};

    // Chunked gradient all-reduce (host/comm.cpp).  With a communicator attached, the dense backward splits its
    // preprocess pass into `grad_slices` splat-range slices and records an event behind each; lcgs_grads_allreduce
    // starts reducing a slice's rows when its event fires, while the later slices are still being computed.
    lcgs_comm*   comm = nullptr;        // attached by lcgs_comm_create (not owned)
    int          grad_slices = 1;       // > 1: slice the dense preprocess-backward
    DeviceBuffer slice_bounds;          // (grad_slices + 1) x u32, dense-id boundaries of the slices (device)
    hipEvent_t   ev_slice[lcgs::kMaxGradSlices]{};
    int          slices_recorded = 0;   // slices of the last dense backward whose events are valid (0: not sliced)
    const void*  slices_of = nullptr;   // the dL_dpos array those events belong to

    // Stage-level operators in DEFERRED mode (lcgs_set_stage_mode): SHProcessor::process and GSProjector::forward record
    // their arguments instead of running; a GSTileSplatter::forward whose inputs are exactly their outputs then renders
    // the fused frame from the 3-D arrays (same image, radii, num_rendered); anything else runs the recorded calls first.
    int stage_mode = 0; // LCGS_STAGES_EXACT
    uint32_t stage_serial = 0;   // lcgs_tile_splat_forward's per-frame mark of the "unwritten pair slots" word
    bool stage_side_copy = true; // the unsorted pair buffers' copy beside the depth sort (A/B hook LCGS_STAGE_SIDE_COPY=0)
    // Frames that keep no backward state may list their pairs per block of 2 x 2 tiles (CamParams::list_shift).  It pays from
    // ~3 M per-tile pairs up (-2 % at 0.2-0.9 M, 0 at 2.5 M, +2.5 % at 4-8 M, +6 % at 10 M, +15 % at 17 M: profiles/
    // r05_coarse_lists_ab.txt), so the default decides per context from the last synchronised frame's pair count, with hysteresis;
    // LCGS_COARSE_LISTS=0 / 1 force it off / on (A/B and test hook).
    int  coarse_mode     = 2;     // 0 never, 1 always, 2 by the pair count
    bool coarse_on       = false; // (mode 2) the current decision
    bool coarse_keep     = false; // frames that keep backward state follow the decision too (render.hip COMPACT): built and
                                  // measured in round 6, -1.1 % on forward+backward (REJECTED.md) -- off; LCGS_COARSE_KEEP=1: A/B hook
    double coarse_yield  = 0.6;   // pruned per-tile pairs / reference num_rendered, from this context's last per-tile frame
    bool bwd_use_masks   = true; // the render-backward walks the forward's kept strip bits (test hook LCGS_BWD_USE_MASKS=0: it repeats the strip tests)
    bool stage_mailbox   = true; // the splatter's scalars posted to pinned memory and polled (A/B hook LCGS_STAGE_MAILBOX=0: copy + sync)
    int stage_sort = 0; // lcgs_tile_splat_forward's sort route: 0 = by frame size, 1 = literal six passes, 2 = sort-before-duplicate
                        // (LCGS_STAGE_SORT=literal|splats, read once when the context is created)
    struct {
        bool         pending = false;
        int          num = 0, level = 3;
        const float *pos = nullptr, *sh = nullptr;
        float*       color = nullptr;
        lcgs_camera  cam{};
    } def_sh;
    struct {
        bool         pending = false;
        int          num = 0, use_focal = 1;
        const float *pos = nullptr, *scale = nullptr, *rotq = nullptr;
        float        scale_modifier = 1.0f;
        float *      means = nullptr, *covs = nullptr, *depth = nullptr;
        lcgs_camera  cam{};
    } def_proj;

    // Splat ownership (abi_owner.cpp): per view of a step, what this context computed as an OWNER of a row range
    // (kept for that view's backward), and -- as the RENDERER of a view -- the received records the last frame was drawn from
    struct OwnerSlot {
        bool         valid = false, has_jac = false;
        CamParams    cp;
        float        scale_modifier = 1.0f;
        int          row_first = 0, row_count = 0, num = 0;
        DeviceBuffer vis, shjac, counts;
    } owner[LCGS_MAX_OWNER_VIEWS];
    uint32_t*                h_owner_counts = nullptr; // pinned, LCGS_MAX_OWNER_VIEWS words (lcgs_owner_counts)
    // lcgs_owner_project_views: the N views of a step are N INDEPENDENT short pipelines over the rank's P / N rows (five launches
    // of a few hundred workgroups each); they run side by side on kOwnerLanes streams at most (two by default), each lane with its own scratch of
    // the cull / first-sort-pass stages (sized by the row range), joined on the context's stream at the end of the call
    static constexpr int kOwnerLanes = 4;
    struct OwnerLane {
        DeviceBuffer slab, chunk_info, chunk_base, sortk[2], sortv[2], rects, sort_ws;
        hipStream_t  stream = nullptr;
        hipEvent_t   done   = nullptr;
    } owner_lane[kOwnerLanes];
    hipEvent_t               ev_owner_fork = nullptr;
    const lcgs::SplatRecord* owner_recs = nullptr;
    int                      owner_rows = 0;

    // per-stage timing
    bool             profiling = false;
    hipEvent_t       events[lcgs::kMaxEvents]{};
    bool             events_created = false;
    int              n_marks        = 0;
    const char*      mark_names[lcgs::kMaxEvents]{};
    lcgs_stage_times times{};
};

