// abi_owner.cpp -- the C ABI, part 7: the frame in two halves, for the splat-ownership step of DESIGN.md 7b.  A GPU that
// OWNS a range of rows runs the per-splat half of a view's frame on them (cull, compaction, SH colour: the packed 48-byte
// records of the rows that reach that view's screen) and, later, the per-splat half of its backward; the GPU that RENDERS
// the view runs everything from the depth sort on -- on records it received from the owners -- and the render-backward, and
// hands the 2-D gradients back.  The kernels are the fused frame's own (abi_frame.cpp / abi_backward.cpp): only the seam
// between the two halves is new.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "abi_internal.hpp"
#include "kernels/tie_order.hpp"

using namespace lcgs;
using namespace lcgs::abi;

namespace
{
struct RowRange {
    const float *pos, *scale, *rotq, *sh, *opacity;
};
RowRange shard_of(lcgs_context* ctx, int row_first)
{
    const size_t feat = (size_t)(ctx->sh_deg + 1) * (ctx->sh_deg + 1) * 3;
    return { ctx->pos + 3 * (size_t)row_first, ctx->scale + 3 * (size_t)row_first, ctx->rotq + 4 * (size_t)row_first,
             ctx->sh + feat * (size_t)row_first, ctx->opacity + (size_t)row_first };
}

// The per-splat half of one view's frame on the slot's row range: the fused frame's first stages -- cull, compaction (the depth
// sort's first pass only: the renderer of the view sorts what it receives), colour + packed records -- on the context's own
// workspace and stream, or (lane != NULL) on a lane's scratch and stream beside other views of the same step.
lcgs_status project_rows(lcgs_context* ctx, int slot, const CamParams& cp, float scale_modifier, bool keep_state, uint32_t* d_rows,
                         float* d_records, lcgs_context::OwnerLane* lane)
{
    auto&        s = ctx->owner[slot];
    const int    row_first = s.row_first, row_count = s.row_count;
    const size_t n = (size_t)row_count;
    LCGS_TRY(s.vis.ensure(n * 4 + 16));
    LCGS_TRY(s.counts.ensure(64));
    if (keep_state) LCGS_TRY(s.shjac.ensure(n * 48));
    hipStream_t    st = lane ? lane->stream : ctx->stream;
    const RowRange r  = shard_of(ctx, row_first);
    uint32_t*      dc = s.counts.as<uint32_t>();
    uint4*         slab       = lane ? lane->slab.as<uint4>() : ctx->cull_slab.as<uint4>();
    uint2*         chunk_info = lane ? lane->chunk_info.as<uint2>() : ctx->chunk_info.as<uint2>();
    uint32_t*      chunk_base = lane ? lane->chunk_base.as<uint32_t>() : ctx->chunk_base.as<uint32_t>();
    uint32_t*      k0 = lane ? lane->sortk[0].as<uint32_t>() : ctx->sortk[0].as<uint32_t>();
    uint32_t*      k1 = lane ? lane->sortk[1].as<uint32_t>() : ctx->sortk[1].as<uint32_t>();
    uint32_t*      v0 = lane ? lane->sortv[0].as<uint32_t>() : ctx->sortv[0].as<uint32_t>();
    uint32_t*      v1 = lane ? lane->sortv[1].as<uint32_t>() : ctx->sortv[1].as<uint32_t>();
    uint2*         rects   = lane ? lane->rects.as<uint2>() : ctx->rects.as<uint2>();
    void*          sort_ws = lane ? lane->sort_ws.ptr : ctx->sort_ws.ptr;
    const DepthSortFirstPass dfirst = depth_sort_first_pass(row_count, sort_ws);
    launch_cull_compact(row_count, cp, scale_modifier, nullptr, r.pos, r.scale, r.rotq, r.opacity, nullptr, slab, chunk_info, dfirst,
                        st, ctx->cull_rows() ? ctx->cull_rows() + row_first : nullptr); // (the scene's bound rows, if it has them)
    launch_depth_sort_from_chunks(row_count, row_count, slab, chunk_info, chunk_base, k0, k1, v0, v1, s.vis.as<uint32_t>(), rects, dc,
                                  sort_ws, st, nullptr, nullptr, /*first_pass_only=*/true);
    launch_build_records(row_count, ctx->sh_deg, cp, scale_modifier, nullptr, r.pos, r.scale, r.rotq, r.sh, r.opacity,
                         s.vis.as<uint32_t>(), dc, reinterpret_cast<SplatRecord*>(d_records), st, nullptr,
                         keep_state ? s.shjac.as<float4>() : nullptr);
    s.has_jac = keep_state && build_records_writes_jacobian(ctx->sh_deg, r.sh, false);
    launch_rows_global(s.vis.as<uint32_t>(), dc, (uint32_t)row_first, d_rows, row_count, st);
    LCGS_HIP_CHECK(hipGetLastError());
    s.valid          = true;
    s.cp             = cp;
    s.scale_modifier = scale_modifier;
    s.num            = -1; // (on the device until read back)
    ctx->last.valid  = false; // (the context's own frame state was overwritten)
    return LCGS_OK;
}
} // namespace

extern "C" {

lcgs_status lcgs_owner_project(lcgs_context* ctx, int slot, const lcgs_camera* camera, float scale_modifier, int row_first,
                               int row_count, int keep_state, uint32_t* d_rows, float* d_records, int* num_rows)
{
    LCGS_REQUIRE(ctx != nullptr, "ctx is NULL");
    LCGS_REQUIRE(row_count == 0 || (d_rows && d_records), "NULL output buffer");
    LCGS_REQUIRE((reinterpret_cast<uintptr_t>(d_records) & 15) == 0, "records must be 16-byte aligned");
    LCGS_REQUIRE(slot >= 0 && slot < LCGS_MAX_OWNER_VIEWS, "slot out of range");
    LCGS_HIP_CHECK(hipSetDevice(ctx->device));
    LCGS_TRY(check_camera(camera));
    if (num_rows) *num_rows = 0;
    LCGS_REQUIRE(ctx->pos != nullptr && ctx->P > 0, "no scene bound");
    LCGS_REQUIRE(row_first >= 0 && row_count >= 0 && (int64_t)row_first + row_count <= ctx->P, "row range outside the scene");
    auto& s = ctx->owner[slot];
    s.valid = false;
    s.row_first = row_first;
    s.row_count = row_count;
    s.num       = 0;
    if (row_count == 0) return LCGS_OK;
    const CamParams cp = make_cam_params(*camera);
    LCGS_TRY(ensure_fused_workspace(ctx, cp, keep_state != 0));
    if (ctx->aux_pending) { // a pipelined frame of this context may still be using the workspace through the auxiliary stream
        LCGS_HIP_CHECK(hipStreamWaitEvent(ctx->stream, ctx->ev_aux_done, 0));
        ctx->aux_pending = false;
    }
    LCGS_TRY(project_rows(ctx, slot, cp, scale_modifier, keep_state != 0, d_rows, d_records, nullptr));
    if (!num_rows) return LCGS_OK; // asynchronous: lcgs_owner_counts reads the count (with the other views' counts)
    return lcgs_owner_counts(ctx, slot, 1, num_rows);
}

// The N views of a step in one call: view k -> slot first_slot + k.  N independent short pipelines over the same row range;
// they run side by side on the context's lanes (context.hpp OwnerLane) and are joined on the context's stream before the call
// returns, so work the caller enqueues there afterwards sees every output.  Counts stay on the device (lcgs_owner_counts).
lcgs_status lcgs_owner_project_views(lcgs_context* ctx, int first_slot, int num_views, const lcgs_camera* cameras,
                                     float scale_modifier, int row_first, int row_count, int keep_state, uint32_t* const* d_rows,
                                     float* const* d_records)
{
    LCGS_REQUIRE(ctx && cameras && d_rows && d_records, "NULL argument");
    LCGS_REQUIRE(first_slot >= 0 && num_views >= 1 && first_slot + num_views <= LCGS_MAX_OWNER_VIEWS, "slots out of range");
    LCGS_HIP_CHECK(hipSetDevice(ctx->device));
    LCGS_REQUIRE(ctx->pos != nullptr && ctx->P > 0, "no scene bound");
    LCGS_REQUIRE(row_first >= 0 && row_count >= 0 && (int64_t)row_first + row_count <= ctx->P, "row range outside the scene");
    // TWO lanes by default: eight views of 766 K rows each, back to back without read-back, take 0.484 ms on one lane (the
    // context's stream), 0.368 on two, 0.413 on four -- the host's forty launches and the event traffic bound it from there
    // (tools/gpu/owner_rank_compute.py; A/B hook LCGS_OWNER_LANES=1..4)
    static const int max_lanes = [] {
        const char* e = getenv("LCGS_OWNER_LANES");
        const int   v = e ? atoi(e) : 2;
        return std::max(1, std::min(v, (int)lcgs_context::kOwnerLanes));
    }();
    const int lanes = std::min(max_lanes, num_views);
    for (int k = 0; k < num_views; ++k) {
        LCGS_TRY(check_camera(&cameras[k]));
        LCGS_REQUIRE(row_count == 0 || (d_rows[k] && d_records[k]), "NULL output buffer");
        LCGS_REQUIRE((reinterpret_cast<uintptr_t>(d_records[k]) & 15) == 0, "records must be 16-byte aligned");
        auto& s     = ctx->owner[first_slot + k];
        s.valid     = false;
        s.row_first = row_first;
        s.row_count = row_count;
        s.num       = 0;
    }
    if (row_count == 0) return LCGS_OK;
    LCGS_TRY(ensure_fused_workspace(ctx, make_cam_params(cameras[0]), keep_state != 0));
    if (ctx->aux_pending) {
        LCGS_HIP_CHECK(hipStreamWaitEvent(ctx->stream, ctx->ev_aux_done, 0));
        ctx->aux_pending = false;
    }
    if (lanes <= 1) { // one view (or the hook): the context's own workspace and stream
        for (int k = 0; k < num_views; ++k)
            LCGS_TRY(project_rows(ctx, first_slot + k, make_cam_params(cameras[k]), scale_modifier, keep_state != 0, d_rows[k],
                                  d_records[k], nullptr));
        return LCGS_OK;
    }
    // lanes: scratch sized by the row range, a stream and an event each; every lane starts behind what the context's stream
    // holds now (the caller's writes to the scene, an optimiser step) and the context's stream continues behind every lane
    const size_t n = (size_t)row_count, chunks = (size_t)cull_chunk_count(row_count);
    if (!ctx->ev_owner_fork) LCGS_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_owner_fork, hipEventDisableTiming));
    LCGS_HIP_CHECK(hipEventRecord(ctx->ev_owner_fork, ctx->stream));
    for (int l = 0; l < lanes; ++l) {
        auto& L = ctx->owner_lane[l];
        if (!L.stream) LCGS_HIP_CHECK(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
        if (!L.done) LCGS_HIP_CHECK(hipEventCreateWithFlags(&L.done, hipEventDisableTiming));
        LCGS_TRY(L.slab.ensure(chunks * 2048 * 16));
        LCGS_TRY(L.chunk_info.ensure(chunks * 8));
        LCGS_TRY(L.chunk_base.ensure(chunks * 4));
        for (int i = 0; i < 2; ++i) {
            LCGS_TRY(L.sortk[i].ensure(n * 4));
            LCGS_TRY(L.sortv[i].ensure(n * 4));
        }
        LCGS_TRY(L.rects.ensure(n * 8));
        LCGS_TRY(L.sort_ws.ensure(pair_sort_ws_bytes((int64_t)n)));
        LCGS_HIP_CHECK(hipStreamWaitEvent(L.stream, ctx->ev_owner_fork, 0));
    }
    lcgs_status st = LCGS_OK;
    for (int k = 0; k < num_views && st == LCGS_OK; ++k)
        st = project_rows(ctx, first_slot + k, make_cam_params(cameras[k]), scale_modifier, keep_state != 0, d_rows[k], d_records[k],
                          &ctx->owner_lane[k % lanes]);
    for (int l = 0; l < lanes; ++l) { // (joined whatever happened: nothing may be left running beside the context's stream)
        auto& L = ctx->owner_lane[l];
        if (hipEventRecord(L.done, L.stream) != hipSuccess || hipStreamWaitEvent(ctx->stream, L.done, 0) != hipSuccess) {
            (void)hipStreamSynchronize(L.stream);
            if (st == LCGS_OK) st = LCGS_ERR_HIP;
        }
    }
    return st;
}

lcgs_status lcgs_owner_counts(lcgs_context* ctx, int first_slot, int num_slots, int* num_rows)
{
    LCGS_REQUIRE(ctx && num_rows, "NULL argument");
    LCGS_REQUIRE(first_slot >= 0 && num_slots >= 0 && first_slot + num_slots <= LCGS_MAX_OWNER_VIEWS, "slots out of range");
    LCGS_HIP_CHECK(hipSetDevice(ctx->device));
    if (!ctx->h_owner_counts)
        LCGS_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_owner_counts), LCGS_MAX_OWNER_VIEWS * 4, hipHostMallocDefault));
    bool pending = false;
    for (int k = 0; k < num_slots; ++k) {
        auto& s = ctx->owner[first_slot + k];
        ctx->h_owner_counts[first_slot + k] = 0;
        if (s.valid && s.row_count > 0 && s.num < 0) {
            LCGS_HIP_CHECK(hipMemcpyAsync(ctx->h_owner_counts + first_slot + k, s.counts.ptr, 4, hipMemcpyDeviceToHost, ctx->stream));
            pending = true;
        }
    }
    if (pending) LCGS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < num_slots; ++k) {
        auto& s = ctx->owner[first_slot + k];
        if (s.valid && s.row_count > 0 && s.num < 0) s.num = (int)ctx->h_owner_counts[first_slot + k];
        num_rows[k] = (s.valid && s.row_count > 0) ? s.num : 0;
    }
    return LCGS_OK;
}

lcgs_status lcgs_owner_render(lcgs_context* ctx, const lcgs_camera* camera, const float bg_color[3], int num_rows,
                              const uint32_t* d_rows, const float* d_records, float* d_img, int keep_state)
{
    return lcgs::abi::owner_render_frame(ctx, camera, bg_color, num_rows, d_rows, d_records, d_img, keep_state, nullptr);
}

} // extern "C"

// lcgs_owner_render; with `af` the variant of the ownership step that reads nothing back (comm.cpp): the rows sit in padded
// per-owner segments whose true counts are on the device, num_rows is the segments' total CAPACITY (every launch is sized by
// it, every count is read on the device), there is one attempt and no synchronisation -- the pair-buffer verdict is OR-ed
// into *af->overflow and the frame's counters are copied to the pinned block for whoever waits on the step later.
lcgs_status lcgs::abi::owner_render_frame(lcgs_context* ctx, const lcgs_camera* camera, const float bg_color[3], int num_rows,
                                          const uint32_t* d_rows, const float* d_records, float* d_img, int keep_state,
                                          const OwnerAsyncFrame* af)
{
    LCGS_REQUIRE(ctx && bg_color && d_img, "NULL argument");
    LCGS_HIP_CHECK(hipSetDevice(ctx->device));
    LCGS_TRY(check_camera(camera));
    LCGS_REQUIRE(num_rows >= 0 && num_rows <= ctx->P, "num_rows out of range (the context's scene sizes the workspace)");
    ctx->last.valid = false;
    ctx->owner_recs = nullptr;
    if (num_rows == 0) return LCGS_OK; // nothing on screen: image untouched, like gs_tile_splatter/impl.cpp:109
    LCGS_REQUIRE(d_rows && d_records, "NULL device pointer");
    LCGS_REQUIRE((reinterpret_cast<uintptr_t>(d_records) & 15) == 0, "records must be 16-byte aligned");
    const CamParams    cp   = make_cam_params(*camera);
    hipStream_t        st   = ctx->stream;
    const SplatRecord* recs = reinterpret_cast<const SplatRecord*>(d_records);
    const uint32_t     G    = cp.grid_x * cp.grid_y;
    for (int attempt = 0; attempt < 4; ++attempt) {
        LCGS_TRY(ensure_fused_workspace(ctx, cp, keep_state != 0));
        uint32_t* dc = ctx->counts.as<uint32_t>();
        if (ctx->aux_pending) { // a pipelined frame of this context may still be using the auxiliary stream's buffers
            LCGS_HIP_CHECK(hipStreamWaitEvent(st, ctx->ev_aux_done, 0));
            ctx->aux_pending = false;
        }
        for (bool& z : ctx->zero_ready) z = false;
        LCGS_HIP_CHECK(hipMemsetAsync(ctx->zero_ws[0].ptr, 0, ctx->zero_bytes, st));
        ctx->ranges        = reinterpret_cast<uint32_t*>(ctx->zero_ws[0].as<char>() + ctx->zero_scan_bytes);
        ctx->work_counters = reinterpret_cast<uint32_t*>(ctx->zero_ws[0].as<char>() + ctx->zero_bytes - 256);
        // equal depths blend in ascending FILE index (tie_order.hpp): the received rows are ascending rows of the scene as
        // the contexts hold it -- its file order, unless the scene was re-ordered, in which case the tags restore it
        TieOrder   tie;
        uint32_t   id_mask = 0xFFFFFFFFu;
        const bool tied    = ctx->perm_valid;
        if (tied) {
            tie.d_counts   = dc;
            tie.vis_index  = ctx->vis_index.as<uint32_t>();
            tie.perm       = ctx->scene_perm.as<uint32_t>();
            tie.id_bits    = (uint32_t)std::max(1, ceil_log2_u32((uint32_t)ctx->P));
            tie.tag_shift  = 2u * tie.id_bits > 32u ? 2u * tie.id_bits - 32u : 0u;
            id_mask        = (1u << tie.id_bits) - 1u;
            tie.scratch_k1 = ctx->tie_ws.as<uint32_t>();
        }
        if (af)
            launch_unpack_records_seg(af->segs, af->table, af->view, recs, d_rows, tied ? tie.perm : nullptr, tie.id_bits,
                                      tie.tag_shift, ctx->sortk[0].as<uint32_t>(), ctx->sortv[0].as<uint32_t>(),
                                      ctx->rects.as<uint2>(), ctx->vis_index.as<uint32_t>(), dc, af->overflow, (uint32_t)ctx->P,
                                      cp.grid_x, cp.grid_y, st);
        else
            launch_unpack_records(num_rows, recs, d_rows, tied ? tie.perm : nullptr, tie.id_bits, tie.tag_shift,
                                  ctx->sortk[0].as<uint32_t>(), ctx->sortv[0].as<uint32_t>(), ctx->rects.as<uint2>(),
                                  ctx->vis_index.as<uint32_t>(), dc, (uint32_t)ctx->P, cp.grid_x, cp.grid_y, st);
        const int w = launch_pair_sort_u32(ctx->sortk[0].as<uint32_t>(), ctx->sortk[1].as<uint32_t>(),
                                           ctx->sortv[0].as<uint32_t>(), ctx->sortv[1].as<uint32_t>(), dc, ctx->P, num_rows, 0, 32,
                                           ctx->sort_ws.ptr, st);
        if (tied)
            launch_fix_equal_depth_order(ctx->sortk[w].as<uint32_t>(), ctx->sortv[w].as<uint32_t>(),
                                         ctx->sortk[w ^ 1].as<uint32_t>(), ctx->sortv[w ^ 1].as<uint32_t>(), ctx->P, num_rows, tie,
                                         st);
        const uint32_t*         order     = ctx->sortv[w].as<uint32_t>();
        const int               tile_bits = std::max(1, ceil_log2_u32(G));
        const int64_t           hint_L    = ctx->hint_L > 0 ? ctx->hint_L : ctx->pair_capacity;
        const PairSortFirstPass first     = pair_sort_first_pass(ctx->pair_capacity, hint_L, 0, tile_bits, ctx->sort_ws.ptr);
        const bool counted = launch_expand(ctx->P, num_rows, hint_L, dc, cp.grid_x, order, ctx->rects.as<uint2>(),
                                           ctx->rects_sorted.as<uint2>(), ctx->pairk[0].as<uint32_t>(),
                                           ctx->pairv[0].as<uint32_t>(), ctx->pair_capacity, ctx->expand_ws.as<uint32_t>(), st,
                                           &first, id_mask);
        const int where2 = launch_pair_sort_u32(ctx->pairk[0].as<uint32_t>(), ctx->pairk[1].as<uint32_t>(),
                                                ctx->pairv[0].as<uint32_t>(), ctx->pairv[1].as<uint32_t>(), dc + 2,
                                                ctx->pair_capacity, hint_L, 0, tile_bits, ctx->sort_ws.ptr, st, counted);
        launch_get_ranges_u32(hint_L, ctx->pair_capacity, dc, ctx->pairk[where2].as<uint32_t>(), ctx->ranges, nullptr, st, nullptr);
        uint32_t* order_now = ctx->tile_order[0].as<uint32_t>();
        launch_tile_order(ctx->ranges, G, order_now, st, cp.grid_x, 0u);
        ctx->order_G = 0; // (the pipelined frames' schedule buffers were used out of turn)
        launch_render_forward_rec(cp, bg_color, ctx->ranges, ctx->pairv[where2].as<uint32_t>(), recs, d_img,
                                  keep_state ? ctx->final_T.as<float>() : nullptr,
                                  keep_state ? ctx->n_contrib.as<uint32_t>() : nullptr, dc, nullptr, order_now, st,
                                  keep_state ? ctx->strip_masks.as<uint8_t>() : nullptr, nullptr);
        LCGS_HIP_CHECK(hipGetLastError());
        if (af) { // no read-back: the verdict travels with the step's flag, the counters to the pinned block
            launch_owner_pair_verdict(dc, af->overflow, st);
            LCGS_HIP_CHECK(hipMemcpyAsync(ctx->h_counts, dc, 40, hipMemcpyDeviceToHost, st));
            ctx->counts_pending      = false;
            ctx->last.valid          = true;
            ctx->last.has_state      = keep_state != 0;
            ctx->last.cp             = cp;
            ctx->last.scale_modifier = 1.0f;
            ctx->last.list_buf       = where2;
            memcpy(ctx->last.bg, bg_color, sizeof(float) * 3);
            ctx->last_tile_order = order_now;
            ctx->owner_recs      = recs;
            ctx->owner_rows      = num_rows;
            ctx->g2d_zeroed      = false;
            return LCGS_OK;
        }
        LCGS_HIP_CHECK(hipMemcpyAsync(ctx->h_counts, dc, 40, hipMemcpyDeviceToHost, st));
        LCGS_HIP_CHECK(hipStreamSynchronize(st));
        ctx->counts_pending = false;
        if ((int64_t)ctx->h_counts[4] > ctx->hint_L || (int64_t)ctx->h_counts[4] * 2 < ctx->hint_L)
            ctx->hint_L = (int64_t)ctx->h_counts[4] + ctx->h_counts[4] / 4 + 4096;
        if (ctx->h_counts[3] != 0) { // the pair workspace was too small for this view: grow, redo
            const uint64_t want = (uint64_t)ctx->h_counts[7] + ctx->h_counts[7] / 4;
            LCGS_REQUIRE(want <= 0x7FFFFFFFull, "num_rendered exceeds 2^31 pairs");
            ctx->pair_capacity = std::max(ctx->pair_capacity, (uint32_t)want);
            ctx->h_counts[3] = ctx->h_counts[6] = ctx->h_counts[7] = 0;
            LCGS_HIP_CHECK(hipMemsetAsync(dc + 6, 0, 8, st));
            continue;
        }
        if (ctx->h_counts[6] != 0) { // (sticky record of this very frame's demand: nothing left to report)
            ctx->h_counts[6] = ctx->h_counts[7] = 0;
            LCGS_HIP_CHECK(hipMemsetAsync(dc + 6, 0, 8, st));
        }
        ctx->last.valid          = true;
        ctx->last.has_state      = keep_state != 0;
        ctx->last.cp             = cp;
        ctx->last.scale_modifier = 1.0f;
        ctx->last.list_buf       = where2;
        memcpy(ctx->last.bg, bg_color, sizeof(float) * 3);
        ctx->last_tile_order = order_now;
        ctx->owner_recs      = recs;
        ctx->owner_rows      = num_rows;
        ctx->g2d_zeroed      = false;
        return LCGS_OK;
    }
    set_last_error("pair buffer growth did not converge");
    return LCGS_ERR_CAPACITY;
}

// what a step that read nothing back learns later from the pinned counters (comm.cpp lcgs_owner_step_finish): the launch
// hints follow the frame, a frame that overflowed its pair buffers grows them for the redo
void lcgs::abi::owner_frame_settle(lcgs_context* ctx)
{
    if (!ctx->h_counts) return;
    if ((int64_t)ctx->h_counts[4] > ctx->hint_L || (int64_t)ctx->h_counts[4] * 2 < ctx->hint_L)
        ctx->hint_L = (int64_t)ctx->h_counts[4] + ctx->h_counts[4] / 4 + 4096;
    if (ctx->h_counts[3] != 0 || ctx->h_counts[6] != 0) {
        const uint64_t want = (uint64_t)ctx->h_counts[7] + ctx->h_counts[7] / 4;
        if (want <= 0x7FFFFFFFull) ctx->pair_capacity = std::max(ctx->pair_capacity, (uint32_t)want);
        ctx->h_counts[3] = ctx->h_counts[6] = ctx->h_counts[7] = 0;
        (void)hipMemsetAsync(ctx->counts.as<uint32_t>() + 6, 0, 8, ctx->stream); // (the sticky record: reported here)
    }
}

extern "C" {

lcgs_status lcgs_owner_render_backward(lcgs_context* ctx, const float* d_dL_dimg, float* d_grads2d)
{
    return lcgs::abi::owner_render_backward_into(ctx, d_dL_dimg, d_grads2d, nullptr);
}

} // extern "C"

// lcgs_owner_render_backward.  The 2-D gradient rows are formed where the caller wants them (one 48-byte row per received
// row, by position: no staging copy); `fill`: dense gradient arrays (the caller's own rows) cleared as a side job of the
// VALU-bound kernel, like the fused backward's (abi_backward.cpp), instead of 0.3 ms of memsets behind it.
lcgs_status lcgs::abi::owner_render_backward_into(lcgs_context* ctx, const float* d_dL_dimg, float* d_grads2d, const DenseFill* fill)
{
    LCGS_REQUIRE(ctx && d_dL_dimg && d_grads2d, "NULL argument");
    LCGS_REQUIRE((reinterpret_cast<uintptr_t>(d_grads2d) & 15) == 0, "d_grads2d must be 16-byte aligned (rows are float4 x 3)");
    LCGS_HIP_CHECK(hipSetDevice(ctx->device));
    if (!ctx->frame_state_valid() || !ctx->last.has_state || !ctx->owner_recs) {
        set_last_error("lcgs_owner_render_backward needs a preceding lcgs_owner_render(..., keep_state = 1)");
        return LCGS_ERR_STATE;
    }
    hipStream_t st = ctx->stream;
    // (rows are addressed by POSITION; a frame from padded segments has positions beyond its row count: all of them are cleared)
    LCGS_HIP_CHECK(hipMemsetAsync(d_grads2d, 0, (size_t)ctx->owner_rows * LCGS_OWNER_GRAD_FLOATS * 4, st));
    LCGS_HIP_CHECK(hipMemsetAsync(ctx->bwd_counter.ptr, 0, 4, st)); // the persistent render-backward's tile counter
    launch_render_backward(ctx->last.cp, ctx->last.bg, ctx->ranges, ctx->pairv[ctx->last.list_buf].as<uint32_t>(), ctx->owner_recs,
                           ctx->final_T.as<float>(), ctx->n_contrib.as<uint32_t>(), d_dL_dimg, d_grads2d, ctx->last_tile_order, st,
                           ctx->strip_masks.as<uint8_t>(), ctx->counts.as<uint32_t>(), nullptr, 0, fill);
    LCGS_HIP_CHECK(hipGetLastError());
    return LCGS_OK;
}

extern "C" {

lcgs_status lcgs_owner_backward(lcgs_context* ctx, int slot, const float* d_grads2d, const lcgs_grads* grads, int accumulate)
{
    return lcgs::abi::owner_backward_rows(ctx, slot, d_grads2d, grads, accumulate ? 1 : 0);
}

} // extern "C"

// lcgs_owner_backward.  mode 0: the range is cleared, the view's rows are written; 1: the view's rows are ADDED to what the
// range holds; 2: the range has been cleared already (the render-backward's side job, comm.cpp), the view's rows are written
// -- plain stores instead of a read-modify-write of 236 bytes a row.
lcgs_status lcgs::abi::owner_backward_rows(lcgs_context* ctx, int slot, const float* d_grads2d, const lcgs_grads* grads, int mode)
{
    LCGS_REQUIRE(ctx && grads, "NULL argument");
    LCGS_REQUIRE(slot >= 0 && slot < LCGS_MAX_OWNER_VIEWS, "slot out of range");
    LCGS_REQUIRE(grads->d_dL_dpos && grads->d_dL_dscale && grads->d_dL_drotq && grads->d_dL_dsh && grads->d_dL_dopacity,
                 "NULL gradient buffer");
    LCGS_REQUIRE((reinterpret_cast<uintptr_t>(grads->d_dL_drotq) & 15) == 0, "dL_drotq must be 16-byte aligned");
    LCGS_REQUIRE((reinterpret_cast<uintptr_t>(d_grads2d) & 15) == 0, "d_grads2d must be 16-byte aligned (rows are float4 x 3)");
    LCGS_HIP_CHECK(hipSetDevice(ctx->device));
    auto& s = ctx->owner[slot];
    if (s.row_count == 0) return LCGS_OK; // an owner of nothing
    if (!s.valid) {
        set_last_error("lcgs_owner_backward needs a preceding lcgs_owner_project into this slot");
        return LCGS_ERR_STATE;
    }
    hipStream_t  st   = ctx->stream;
    const size_t feat = (size_t)(ctx->sh_deg + 1) * (ctx->sh_deg + 1) * 3, f = (size_t)s.row_first, c = (size_t)s.row_count;
    float *gp = grads->d_dL_dpos + 3 * f, *gs = grads->d_dL_dscale + 3 * f, *gq = grads->d_dL_drotq + 4 * f,
          *gsh = grads->d_dL_dsh + feat * f, *go = grads->d_dL_dopacity + f;
    if (mode == 0) { // the first view of the step: every row of the range that is not on its screen is an exact zero
        LCGS_HIP_CHECK(hipMemsetAsync(gp, 0, c * 3 * 4, st));
        LCGS_HIP_CHECK(hipMemsetAsync(gs, 0, c * 3 * 4, st));
        LCGS_HIP_CHECK(hipMemsetAsync(gq, 0, c * 4 * 4, st));
        LCGS_HIP_CHECK(hipMemsetAsync(gsh, 0, c * feat * 4, st));
        LCGS_HIP_CHECK(hipMemsetAsync(go, 0, c * 4, st));
    }
    if (s.num < 0) {
        set_last_error("the slot's row count is still on the device: call lcgs_owner_counts after an asynchronous lcgs_owner_project");
        return LCGS_ERR_STATE;
    }
    if (s.num == 0) return LCGS_OK;
    LCGS_REQUIRE(d_grads2d != nullptr, "NULL 2-D gradients");
    const RowRange r = shard_of(ctx, s.row_first);
    launch_preprocess_backward(s.num, ctx->sh_deg, s.cp, s.scale_modifier, r.pos, r.scale, r.rotq, r.sh, s.vis.as<uint32_t>(),
                               s.counts.as<uint32_t>(), d_grads2d, gp, gs, gq, gsh, go, st,
                               s.has_jac ? s.shjac.as<float4>() : nullptr, /*compact=*/false, nullptr, 0, 1, /*accumulate=*/mode == 1);
    LCGS_HIP_CHECK(hipGetLastError());
    return LCGS_OK;
}


extern "C" {

} // extern "C"
