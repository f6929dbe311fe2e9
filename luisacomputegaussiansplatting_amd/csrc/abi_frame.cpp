// abi_frame.cpp -- the C ABI, part 4: the fused frame -- workspace, the one-submission enqueue, 
// lcgs_render_forward, camera batches and the sibling context they alternate with.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <string>
#include <vector>

#include "abi_internal.hpp"
#include "kernels/tie_order.hpp"

using namespace lcgs;
using namespace lcgs::abi;

namespace lcgs
{
namespace abi
{
lcgs_status ensure_fused_workspace(lcgs_context* ctx, const CamParams& cp, bool keep_state)
{
    const size_t P = (size_t)ctx->P;
    LCGS_TRY(ctx->recs.ensure(P * sizeof(SplatRecord)));
    for (int i = 0; i < 2; ++i) {
        LCGS_TRY(ctx->sortk[i].ensure(P * 4));
        LCGS_TRY(ctx->sortv[i].ensure(P * 4));
    }
    {
        const size_t chunks = (size_t)cull_chunk_count((int)P);
        LCGS_TRY(ctx->cull_slab.ensure(chunks * 2048 * 16));
        LCGS_TRY(ctx->chunk_info.ensure(chunks * 8));
        LCGS_TRY(ctx->chunk_base.ensure(chunks * 4));
    }
    LCGS_TRY(ctx->vis_index.ensure(P * 4));
    LCGS_TRY(ctx->rects.ensure(P * 8));
    LCGS_TRY(ctx->rects_sorted.ensure(P * 8));
    // scratch for runs of more than 4096 equal depths of a re-ordered scene (kernels/tie_order.hpp).  Allocated HERE, with
    // every other buffer of the frame: enqueue_forward may run inside a stream capture (LCGS_GRAPH=1), where hipMalloc fails
    if (ctx->perm_valid) LCGS_TRY(ctx->tie_ws.ensure(P * 4));
    if (ctx->pair_capacity == 0) {
        // generous default: 288 GB of HBM makes over-provisioning the pair buffers free
        uint64_t cap       = std::max<uint64_t>((uint64_t)4 * P, (uint64_t)1 << 22);
        ctx->pair_capacity = (uint32_t)std::min<uint64_t>(cap, 0x7FFFFFFFull);
    }
    for (int i = 0; i < 2; ++i) {
        LCGS_TRY(ctx->pairk[i].ensure((size_t)ctx->pair_capacity * 4));
        LCGS_TRY(ctx->pairv[i].ensure((size_t)ctx->pair_capacity * 4));
    }
    const size_t G = (size_t)cp.grid_x * cp.grid_y;
    auto         al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    // (b0: the chained scan's state block, gone with the scan; b1: the tile ranges + the persistent renderers' tile counters)
    const size_t b0 = 0, b1 = al(G * 2 * 4) + 256;
    for (int i = 0; i < 3; ++i) {
        const void* before = ctx->zero_ws[i].ptr;
        LCGS_TRY(ctx->zero_ws[i].ensure(b0 + b1));
        if (ctx->zero_ws[i].ptr != before || ctx->zero_bytes != b0 + b1) ctx->zero_ready[i] = false;
    }
    for (int i = 0; i < 2; ++i) {
        const void* before = ctx->tile_order[i].ptr;
        LCGS_TRY(ctx->tile_order[i].ensure(G * 8)); // (G slots + the sorted blocks of a per-block schedule: render.hip k_tile_order)
        if (ctx->tile_order[i].ptr != before) ctx->order_G = 0;
    }
    ctx->zero_scan_bytes = b0;
    ctx->zero_bytes      = b0 + b1;
    {
        // the counter block starts zeroed: words [6] / [7] (overflow since the last read-back) are only ever added to
        const void* before = ctx->counts.ptr;
        LCGS_TRY(ctx->counts.ensure(64));
        if (ctx->counts.ptr != before) LCGS_HIP_CHECK(hipMemsetAsync(ctx->counts.ptr, 0, 64, ctx->stream));
    }
    LCGS_TRY(ctx->sort_ws.ensure(pair_sort_ws_bytes(std::max<int64_t>((int64_t)P, (int64_t)ctx->pair_capacity))));
    LCGS_TRY(ctx->expand_ws.ensure(expand_ws_bytes((int)P)));
    if (keep_state) {
        LCGS_TRY(ctx->final_T.ensure((size_t)cp.width * cp.height * 4));
        LCGS_TRY(ctx->n_contrib.ensure((size_t)cp.width * cp.height * 4));
        // (per-block lists: every tile owns a segment that could hold its whole block's list -- 4 x the pairs, render.hip COMPACT)
        const size_t list_slots = (size_t)ctx->pair_capacity * (cp.list_shift ? 4u : 1u);
        LCGS_TRY(ctx->strip_masks.ensure(list_slots));
        if (cp.list_shift) {
            LCGS_TRY(ctx->keep_list.ensure(list_slots * 4));
            LCGS_TRY(ctx->keep_ranges.ensure(G * 8));
        }
        LCGS_TRY(ctx->grads2d.ensure(grads2d_bytes((int64_t)P)));
        LCGS_TRY(ctx->shjac.ensure(P * 48));
        if (!ctx->bwd_counter.ptr) { // (allocated once; starts at zero whatever runs first)
            LCGS_TRY(ctx->bwd_counter.ensure(256));
            LCGS_HIP_CHECK(hipMemsetAsync(ctx->bwd_counter.ptr, 0, 256, ctx->stream));
        }
    }
    if (!ctx->h_counts) LCGS_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_counts), 64, hipHostMallocDefault));
    return LCGS_OK;
}

} // namespace abi
} // namespace lcgs

namespace
{
// enqueue one fused forward frame (no synchronisation)
// d_fp: when non-NULL, camera / bg / scale_modifier are read from device memory by the kernels (graph replay)
// in_capture: the frame is being recorded into a hipGraph (fixed pointers): no per-frame buffer alternation
lcgs_status enqueue_forward(lcgs_context* ctx, const CamParams& cp, const float bg[3], float scale_modifier,
                            float* d_img, int32_t* d_radii, bool keep_state, const FrameParams* d_fp,
                            bool in_capture = false)
{
    const hipStream_t vis  = ctx->stream; // the stream whose order the caller sees
    uint32_t*    d_counts = ctx->counts.as<uint32_t>();
    const int    P        = ctx->P;
    SplatRecord* recs     = ctx->recs.as<SplatRecord>();
    const uint32_t G      = cp.grid_x * cp.grid_y;
    ctx->n_marks          = 0;
    LCGS_TRY(mark(ctx, "begin"));

    // With per-stage profiling on, everything runs in order on the main stream so that stage times stay
    // attributable; otherwise independent work moves to the auxiliary stream (see below).
    const bool overlap  = !ctx->profiling;
    const bool deferred = overlap && !in_capture;
    // CU-partitioned streams (tuning hook): the sort chain on `st` = the chain stream, record builder + renderer on the
    // render stream; the caller's stream only orders the frame (waits for what came before, is waited on by the end)
    const bool  part = deferred && ctx->chain_stream != nullptr;
    hipStream_t st   = part ? ctx->chain_stream : vis;
    hipStream_t rst  = part ? ctx->render_stream : vis;
    if (part) {
        LCGS_HIP_CHECK(hipEventRecord(ctx->ev_begin, vis));
        LCGS_HIP_CHECK(hipStreamWaitEvent(st, ctx->ev_begin, 0));
    }
    // Zeroed per frame: the tile ranges (the reference zero-fills ranges too,
    // gs_tile_splatter/impl.cpp:147).  Normally the auxiliary stream cleared this frame's copy during the last frame.
    const int zb = deferred ? ctx->zero_cur : 0;
    if (!deferred && ctx->aux_pending && !in_capture) {
        // leaving the pipelined mode (profiling switched on): the auxiliary stream may still be filling a copy or
        // writing a tile schedule this in-order frame is about to use
        LCGS_HIP_CHECK(hipStreamWaitEvent(st, ctx->ev_aux_done, 0));
        ctx->aux_pending = false;
        for (bool& r : ctx->zero_ready) r = false;
    }
    if (!(deferred && ctx->zero_ready[zb])) LCGS_HIP_CHECK(hipMemsetAsync(ctx->zero_ws[zb].ptr, 0, ctx->zero_bytes, st));
    ctx->zero_ready[zb]  = false;
    ctx->ranges          = reinterpret_cast<uint32_t*>(ctx->zero_ws[zb].as<char>() + ctx->zero_scan_bytes);
    ctx->work_counters   = reinterpret_cast<uint32_t*>(ctx->zero_ws[zb].as<char>() + ctx->zero_bytes - 256);
    const DepthSortFirstPass dfirst = depth_sort_first_pass(P, ctx->sort_ws.ptr);
    launch_cull_compact(P, cp, scale_modifier, d_fp, ctx->pos, ctx->scale, ctx->rotq, ctx->opacity, d_radii,
                        ctx->cull_slab.as<uint4>(), ctx->chunk_info.as<uint2>(), dfirst, st, ctx->cull_rows());
    LCGS_TRY(mark(ctx, "cull_compact"));
    const int64_t hint_V = ctx->hint_V > 0 ? ctx->hint_V : P;
    // (per-tile and per-block frames of one context keep a launch-size hint each: their pair counts differ by up to 3 x, and a
    // context that alternates between them -- a viewer beside a trainer -- would size every forward-only frame for the other's)
    const int64_t hint_own = cp.list_shift ? (ctx->hint_Lb > 0 ? ctx->hint_Lb : ctx->hint_L) : ctx->hint_L;
    const int64_t hint_L   = hint_own > 0 ? hint_own : ctx->pair_capacity;
    // survivors by depth bits (the low 32 bits of the reference key), sorted before duplication.  The first pass reads
    // the cull pass's chunk slabs, hands out the dense ids and writes vis_index / rects; its completion is the fork
    // point of the record builder.
    // a scene the context re-ordered: equal depths must still blend in ascending FILE index, as in the reference
    // (kernels/tie_order.hpp); the sorted values then carry a file-index tag above the dense id's id_bits
    TieOrder tie;
    uint32_t id_mask = 0xFFFFFFFFu;
    if (ctx->perm_valid) {
        tie.d_counts  = d_counts;
        tie.vis_index = ctx->vis_index.as<uint32_t>();
        tie.perm      = ctx->scene_perm.as<uint32_t>();
        tie.id_bits   = (uint32_t)std::max(1, ceil_log2_u32((uint32_t)P));
        tie.tag_shift = 2u * tie.id_bits > 32u ? 2u * tie.id_bits - 32u : 0u;
        id_mask       = (1u << tie.id_bits) - 1u;
        tie.scratch_k1 = ctx->tie_ws.as<uint32_t>(); // (sized by ensure_fused_workspace: no allocation in here)
    }
    launch_depth_sort_from_chunks(P, hint_V, ctx->cull_slab.as<uint4>(), ctx->chunk_info.as<uint2>(),
                                  ctx->chunk_base.as<uint32_t>(), ctx->sortk[0].as<uint32_t>(), ctx->sortk[1].as<uint32_t>(),
                                  ctx->sortv[0].as<uint32_t>(), ctx->sortv[1].as<uint32_t>(), ctx->vis_index.as<uint32_t>(),
                                  ctx->rects.as<uint2>(), d_counts, ctx->sort_ws.ptr, st,
                                  (overlap && !in_capture) ? ctx->ev_fork : nullptr, ctx->perm_valid ? &tie : nullptr);
    const uint32_t* order = ctx->sortv[0].as<uint32_t>();
    LCGS_TRY(mark(ctx, "depth_sort"));
    // Record building (SH fetch + colour: bandwidth-bound) is independent of the rest of the sort chain (latency-bound
    // short kernels): fork it onto the auxiliary stream so the two overlap; the renderer joins.
    hipStream_t rec_stream = part ? rst : (overlap ? ctx->aux_stream : st);
    if (overlap) {
        // (in a capture the fork is recorded here, after the whole depth sort; otherwise the first pass's scatter
        //  dispatch carries it)
        if (in_capture) LCGS_HIP_CHECK(hipEventRecord(ctx->ev_fork, st));
        LCGS_HIP_CHECK(hipStreamWaitEvent(rec_stream, ctx->ev_fork, 0));
    }
    launch_build_records((int)std::min<int64_t>(P, hint_V), ctx->sh_deg, cp, scale_modifier, d_fp, ctx->pos, ctx->scale,
                         ctx->rotq, ctx->sh, ctx->opacity, ctx->vis_index.as<uint32_t>(), d_counts, recs, rec_stream,
                         ctx->use_half_sh ? ctx->sh_half.as<uint16_t>() : nullptr,
                         keep_state ? ctx->shjac.as<float4>() : nullptr);
    ctx->last_has_jac = keep_state && build_records_writes_jacobian(ctx->sh_deg, ctx->sh, ctx->use_half_sh);
    if (overlap && !part) LCGS_HIP_CHECK(hipEventRecord(ctx->ev_join, ctx->aux_stream));
    // keep_state frames: the 2-D gradient rows the backward adds to are cleared by the RENDERER as a side job (render.hip):
    // no launch on the auxiliary stream, no cross-stream wait in front of the render-backward
    ctx->g2d_zeroed = false;
    const bool g2d_in_render = deferred && keep_state;
    LCGS_TRY(mark(ctx, "build_records"));

    // stable partition by tile id: only ceil(log2 G) key bits are live.  The kernel that writes the pairs also leaves
    // the partition's first per-chunk digit counts in the sort workspace (the depth sort is done with it by then).
    const int tile_bits = std::max(1, ceil_log2_u32(list_grid_x(cp) * list_grid_y(cp)));
    const PairSortFirstPass first = pair_sort_first_pass(ctx->pair_capacity, hint_L, 0, tile_bits, ctx->sort_ws.ptr);
    const bool counted =
        launch_expand(P, hint_V, hint_L, d_counts, list_grid_x(cp), order, ctx->rects.as<uint2>(), ctx->rects_sorted.as<uint2>(),
                      ctx->pairk[0].as<uint32_t>(), ctx->pairv[0].as<uint32_t>(), ctx->pair_capacity,
                      ctx->expand_ws.as<uint32_t>(), st, &first, id_mask);
    LCGS_TRY(mark(ctx, "expand"));

    const int where2 = launch_pair_sort_u32(ctx->pairk[0].as<uint32_t>(), ctx->pairk[1].as<uint32_t>(),
                                            ctx->pairv[0].as<uint32_t>(), ctx->pairv[1].as<uint32_t>(), d_counts + 2,
                                            ctx->pair_capacity, hint_L, 0, tile_bits, ctx->sort_ws.ptr, st,
                                            /*first_hist_done=*/counted);
    LCGS_TRY(mark(ctx, "tile_sort"));

    launch_get_ranges_u32(hint_L, ctx->pair_capacity, d_counts, ctx->pairk[where2].as<uint32_t>(), ctx->ranges,
                          nullptr, st, deferred ? ctx->ev_ranges : nullptr);
    // tile schedule: the newest complete order if it matches this grid, else computed here
    uint32_t* order_now = nullptr;
    if (deferred && ctx->order_G == G) {
        order_now = ctx->tile_order[ctx->order_cur].as<uint32_t>();
    } else {
        const int ob = deferred ? (ctx->order_cur ^ 1) : 0;
        order_now    = ctx->tile_order[ob].as<uint32_t>();
        launch_tile_order(ctx->ranges, G, order_now, st, cp.grid_x, cp.list_shift);
        if (deferred) {
            ctx->order_cur = ob;
            ctx->order_G   = G;
        }
    }
    LCGS_TRY(mark(ctx, "ranges"));
    if (deferred) {
        // behind the records on the auxiliary stream, beside the renderer: this frame's list lengths -> next
        // frame's schedule, and the next frame's zeroed copy
        const int ob = ctx->order_cur ^ 1, znext = (zb + 2) % 3; // the copy of the frame after the next
        LCGS_HIP_CHECK(hipStreamWaitEvent(ctx->aux_stream, ctx->ev_ranges, 0)); // (carried by the ranges dispatch)
        launch_tile_order(ctx->ranges, G, ctx->tile_order[ob].as<uint32_t>(), ctx->aux_stream, cp.grid_x, cp.list_shift);
        LCGS_HIP_CHECK(hipMemsetAsync(ctx->zero_ws[znext].ptr, 0, ctx->zero_bytes, ctx->aux_stream));
        LCGS_HIP_CHECK(hipEventRecord(ctx->ev_aux_done, ctx->aux_stream));
        ctx->aux_pending       = true;
        ctx->zero_ready[znext] = true;
        ctx->zero_cur          = (zb + 1) % 3;
        ctx->order_cur         = ob; // written before the next frame's record builder runs: its renderer waits for that
    }

    if (part) { // the records are ahead of the renderer on its own stream; it waits for the chain
        LCGS_HIP_CHECK(hipEventRecord(ctx->ev_chain, st));
        LCGS_HIP_CHECK(hipStreamWaitEvent(rst, ctx->ev_chain, 0));
    } else if (overlap) {
        LCGS_HIP_CHECK(hipStreamWaitEvent(st, ctx->ev_join, 0)); // records are ready
    }
    // several frames in flight: a bounded, persistent grid (context.hpp) -- same image, free wave slots on every CU
    const int      k_persist = ctx->persist_forced >= 0 ? ctx->persist_forced : (ctx->frames_in_flight ? ctx->persist_in_flight : 0);
    const uint32_t persist_wgs = (deferred && k_persist > 0) ? (uint32_t)(k_persist * std::max(ctx->num_cus, 1)) : 0u;
    launch_render_forward_rec(cp, bg, ctx->ranges, ctx->pairv[where2].as<uint32_t>(), recs, d_img,
                              keep_state ? ctx->final_T.as<float>() : nullptr,
                              keep_state ? ctx->n_contrib.as<uint32_t>() : nullptr, d_counts, d_fp, order_now, part ? rst : st,
                              keep_state ? ctx->strip_masks.as<uint8_t>() : nullptr, deferred ? ctx->ev_render : nullptr,
                              ctx->work_counters, persist_wgs, g2d_in_render ? ctx->grads2d.as<float>() : nullptr,
                              g2d_in_render ? ctx->bwd_counter.as<uint32_t>() : nullptr,
                              keep_state && cp.list_shift ? ctx->keep_list.as<uint32_t>() : nullptr,
                              keep_state && cp.list_shift ? ctx->keep_ranges.as<uint32_t>() : nullptr);
    ctx->g2d_zeroed = g2d_in_render; // (consumed by the first backward of this frame; same stream: no event)
    ctx->last_tile_order = order_now;
    LCGS_TRY(mark(ctx, "render"));
    if (part) LCGS_HIP_CHECK(hipStreamWaitEvent(vis, ctx->ev_render, 0)); // the caller's stream sees the finished frame

    if (deferred) {
        // the counter read-back leaves through the auxiliary stream: the next frame does not queue behind it
        LCGS_HIP_CHECK(hipStreamWaitEvent(ctx->aux_stream, ctx->ev_render, 0)); // (carried by the render dispatch)
        LCGS_HIP_CHECK(hipMemcpyAsync(ctx->h_counts, d_counts, 40, hipMemcpyDeviceToHost, ctx->aux_stream));
        LCGS_HIP_CHECK(hipEventRecord(ctx->ev_counts, ctx->aux_stream));
        ctx->counts_pending = true;
    } else {
        LCGS_HIP_CHECK(hipMemcpyAsync(ctx->h_counts, d_counts, 40, hipMemcpyDeviceToHost, st));
    }
    ctx->last.valid          = true;
    ctx->last.has_state      = keep_state;
    ctx->last.cp             = cp;
    ctx->last.scale_modifier = scale_modifier;
    ctx->last.list_buf       = where2;
    memcpy(ctx->last.bg, bg, sizeof(float) * 3);
    return LCGS_OK;
}
} // namespace

extern "C" {

lcgs_status lcgs_render_forward(lcgs_context* ctx, const lcgs_camera* camera, const float bg_color[3],
                                float scale_modifier, float* d_img, int32_t* d_radii, int keep_state,
                                int* num_rendered)
{
    LCGS_REQUIRE(ctx != nullptr, "ctx is NULL");
    LCGS_HIP_CHECK(hipSetDevice(ctx->device)); // multi-GPU processes: every entry point selects its device
    ctx->note_foreign_writes();
    LCGS_TRY(check_camera(camera));
    LCGS_REQUIRE(bg_color != nullptr, "bg_color is NULL");
    LCGS_REQUIRE(d_img != nullptr, "d_img is NULL");
    if (num_rendered) *num_rendered = 0;
    if (ctx->P == 0) return LCGS_OK; // nothing to draw: image untouched, like gs_tile_splatter/impl.cpp:109
    LCGS_REQUIRE(ctx->pos != nullptr, "no scene bound (call lcgs_scene_bind / lcgs_scene_upload first)");
    CamParams cp      = make_cam_params(*camera);
    cp.lod_min_radius = ctx->lod_min_radius;
    // (frames that keep backward state stay per tile: letting them follow -- their renderer then writes per-tile lists for the
    // backward while it stages, render.hip COMPACT -- was built in round 6 and lost 1.1 %: REJECTED.md; LCGS_COARSE_KEEP=1 is
    // the A/B hook; the segments need 4 x the pair capacity in 32-bit positions)
    cp.list_shift     = ((!keep_state || (ctx->coarse_keep && ctx->pair_capacity < (1u << 30))) &&
                     (ctx->coarse_mode == 1 || (ctx->coarse_mode == 2 && ctx->coarse_on))) ? 1u : 0u;
    ctx->owner_recs   = nullptr; // (an ordinary frame: its backward is lcgs_render_backward again)
    uint32_t        earlier_truncated = 0; // asynchronous frames before this one that overflowed the pair workspace
    for (int attempt = 0; attempt < 4; ++attempt) {
        LCGS_TRY(ensure_fused_workspace(ctx, cp, keep_state != 0));
        if (ctx->use_graph && !ctx->profiling && ctx->stream != nullptr) { // the legacy NULL stream cannot be captured
            // refresh the device-resident parameters (one tiny eager launch), then replay the captured frame
            LCGS_TRY(ctx->frame_params.ensure(sizeof(FrameParams)));
            FrameParams fp;
            fp.cp = cp;
            memcpy(fp.bg, bg_color, sizeof(float) * 3);
            fp.scale_modifier = scale_modifier;
            launch_set_frame_params(fp, ctx->frame_params.as<FrameParams>(), ctx->stream);
            lcgs_context::GraphKey key;
            key.pos = ctx->pos; key.scale = ctx->scale; key.rotq = ctx->rotq; key.sh = ctx->sh; key.opacity = ctx->opacity;
            key.sh_half = ctx->use_half_sh ? ctx->sh_half.ptr : nullptr; // (selects the kernel and its coefficient rows)
            key.cull_bound = ctx->cull_rows(); // (selects the cull kernel)
            key.img = d_img; key.radii = d_radii; key.P = ctx->P; key.sh_deg = ctx->sh_deg;
            key.width = camera->width; key.height = camera->height; key.keep_state = keep_state != 0;
            key.list_shift = (int)cp.list_shift;
            key.hint_V = ctx->hint_V; key.hint_L = cp.list_shift ? (ctx->hint_Lb > 0 ? ctx->hint_Lb : ctx->hint_L) : ctx->hint_L; key.capacity = ctx->pair_capacity; key.stream = ctx->stream;
            if (!ctx->graph_exec || !(key == ctx->graph_key)) {
                if (ctx->graph_exec) {
                    LCGS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
                    (void)hipGraphExecDestroy(ctx->graph_exec);
                    ctx->graph_exec = nullptr;
                }
                hipGraph_t graph = nullptr;
                LCGS_HIP_CHECK(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
                lcgs_status cs = enqueue_forward(ctx, cp, bg_color, scale_modifier, d_img, d_radii, keep_state != 0,
                                                 ctx->frame_params.as<FrameParams>(), /*in_capture=*/true);
                hipError_t ce = hipStreamEndCapture(ctx->stream, &graph);
                if (cs != LCGS_OK) {
                    if (graph) (void)hipGraphDestroy(graph);
                    return cs;
                }
                LCGS_HIP_CHECK(ce);
                LCGS_HIP_CHECK(hipGraphInstantiate(&ctx->graph_exec, graph, nullptr, nullptr, 0));
                (void)hipGraphDestroy(graph);
                ctx->graph_key = key;
            }
            LCGS_HIP_CHECK(hipGraphLaunch(ctx->graph_exec, ctx->stream));
            ctx->last.valid          = true;
            ctx->last.has_state      = keep_state != 0;
            ctx->last.cp             = cp;
            ctx->last.scale_modifier = scale_modifier;
            memcpy(ctx->last.bg, bg_color, sizeof(float) * 3);
        } else {
            LCGS_TRY(enqueue_forward(ctx, cp, bg_color, scale_modifier, d_img, d_radii, keep_state != 0, nullptr));
        }
        if (!num_rendered && !ctx->profiling) return LCGS_OK; // fully asynchronous frame
        LCGS_TRY(sync_frame(ctx));
        LCGS_TRY(collect_marks(ctx));
        ctx->stats.num_gaussians = ctx->P;
        ctx->stats.num_visible   = ctx->h_counts[0];
        ctx->stats.num_rendered  = ctx->h_counts[1];
        ctx->stats.num_pairs     = ctx->h_counts[2];
        ctx->stats.num_tiles     = (int64_t)cp.grid_x * cp.grid_y;
        ctx->stats.equal_depth_unresolved = ctx->perm_valid ? ctx->h_counts[9] : 0;
        if (num_rendered) *num_rendered = (int)ctx->h_counts[1];
        if (ctx->h_counts[5] != 0) return check_frame_flags(ctx);
        // launch-size hints for the following asynchronous frames
        // (kept unless the live counts leave the [hint/2, hint] band, so a captured graph stays valid)
        if ((int64_t)ctx->h_counts[0] > ctx->hint_V || (int64_t)ctx->h_counts[0] * 2 < ctx->hint_V)
            ctx->hint_V = (int64_t)ctx->h_counts[0] + ctx->h_counts[0] / 4 + 4096;
        {
            int64_t& hl = cp.list_shift ? ctx->hint_Lb : ctx->hint_L;
            if ((int64_t)ctx->h_counts[4] > hl || (int64_t)ctx->h_counts[4] * 2 < hl) hl = (int64_t)ctx->h_counts[4] + ctx->h_counts[4] / 4 + 4096;
        }
        {   // per-block lists for the following frames without backward state?  (context.hpp coarse_mode)  The decision is
            // made in PER-TILE pairs whichever granularity this frame used: a per-tile frame has the count itself and notes
            // its share of the reference's num_rendered (the pruning's yield: a property of the scene, 0.58 on the bicycle
            // stand-in); a per-block frame multiplies its num_rendered -- which does not depend on the granularity -- by
            // that share.  (Until round 6 the per-block count was scaled by 5/3, a ratio that really varies from 0.45 to 0.7:
            // near the thresholds successive frames could flip the decision back and forth.)
            if (!cp.list_shift && ctx->h_counts[1] > 0)
                ctx->coarse_yield = std::min(1.0, std::max(0.05, (double)ctx->h_counts[4] / (double)ctx->h_counts[1]));
            // Round 6's workload sweep (profiles/r06_workload_sweep.json) added the second condition: what per-block lists save
            // is the duplication of footprints that span several tiles of a block, and a frame of SMALL footprints (scale
            // modifier 0.5: 1.6-1.9 tiles per on-screen splat) has little of it -- its per-block lists are 0.8 of the per-tile
            // ones and the 2.4 x stagings cost 4-10 % of the frame; from ~2.2 tiles per splat up they win.
            const int64_t per_tile = cp.list_shift ? (int64_t)((double)ctx->h_counts[1] * ctx->coarse_yield) : (int64_t)ctx->h_counts[4];
            const int64_t V_now    = (int64_t)ctx->h_counts[0];
            if (per_tile >= 3000000 && per_tile * 10 >= V_now * 22) ctx->coarse_on = true;
            else if (per_tile < 2400000 || per_tile * 10 < V_now * 19) ctx->coarse_on = false;
        }
        ctx->stats.list_shift = cp.list_shift;
        // overflow bookkeeping: [3] this frame, [6] / [7] every frame since the last read-back (sticky on the device)
        const bool     own    = ctx->h_counts[3] != 0;
        const uint32_t sticky = ctx->h_counts[6], sticky_want = ctx->h_counts[7];
        if (sticky) {
            ctx->h_counts[6] = ctx->h_counts[7] = 0;
            LCGS_HIP_CHECK(hipMemsetAsync(ctx->counts.as<uint32_t>() + 6, 0, 8, ctx->stream));
            uint64_t want = (uint64_t)sticky_want + sticky_want / 4;
            if (want > 0x7FFFFFFFull) {
                set_last_error("num_rendered exceeds 2^31 pairs");
                return LCGS_ERR_CAPACITY;
            }
            ctx->pair_capacity = std::max(ctx->pair_capacity, (uint32_t)want);
        }
        if (sticky > (own ? 1u : 0u)) earlier_truncated += sticky - (own ? 1u : 0u);
        if (own) continue; // pair buffers were too small for this view: grown above, redo the frame
        if (earlier_truncated) {
            char buf[256];
            snprintf(buf, sizeof(buf),
                     "%u earlier asynchronous frame(s) needed more (tile, splat) pairs than the workspace held; their "
                     "images are truncated (this frame is complete).  The workspace has been grown: render them again",
                     earlier_truncated);
            set_last_error(buf);
            return LCGS_ERR_CAPACITY;
        }
        return LCGS_OK;
    }
    set_last_error("pair buffer growth did not converge");
    return LCGS_ERR_CAPACITY;
}

// Camera batches.  Views are independent, and a single frame leaves the GPU half idle while its
// sort chain waits on memory round trips, so the batch alternates between this context and a sibling context with
// its own workspace and streams: two frames are in flight at any time (measured on the bicycle stand-in: 1250 vs
// 1110 frames/s; three or four in flight were slower).  Everything is ordered after prior work on the context's
// stream and the stream waits for the whole batch, so callers see ordinary stream semantics.
lcgs_status lcgs_render_forward_batch(lcgs_context* ctx, int num_views, const lcgs_camera* cameras,
                                      const float bg_color[3], float scale_modifier, float* const* d_imgs)
{
    LCGS_REQUIRE(ctx != nullptr, "ctx is NULL");
    LCGS_REQUIRE(num_views >= 0, "num_views is negative");
    if (num_views == 0) return LCGS_OK;
    LCGS_REQUIRE(cameras != nullptr && d_imgs != nullptr && bg_color != nullptr, "NULL argument");
    LCGS_HIP_CHECK(hipSetDevice(ctx->device));
    ctx->note_foreign_writes();
    const bool two = num_views > 1 && !ctx->profiling && !ctx->use_graph && ctx->P > 0;
    // frames in flight: 2 (measured best through round 3); LCGS_BATCH_IN_FLIGHT = 3 / 4 is a tuning hook (a chain of siblings)
    static const int want = [] {
        const char* e = getenv("LCGS_BATCH_IN_FLIGHT");
        return e ? std::min(std::max(atoi(e), 2), 4) : 2;
    }();
    lcgs_context* ring[4] = { ctx, nullptr, nullptr, nullptr };
    int           n_ring  = 1;
    if (two)
        for (; n_ring < std::min(want, num_views); ++n_ring) {
            LCGS_TRY(prepare_twin(ring[n_ring - 1]));
            ring[n_ring] = ring[n_ring - 1]->twin;
        }
    InFlight in_flight(ctx, two);
    for (int i = 0; i < num_views; ++i) {
        LCGS_REQUIRE(d_imgs[i] != nullptr, "NULL image pointer in the batch");
        LCGS_TRY(lcgs_render_forward(ring[i % n_ring], &cameras[i], bg_color, scale_modifier, d_imgs[i], nullptr, 0, nullptr));
    }
    for (int k = 0; k + 1 < n_ring; ++k) { // the caller's stream waits for every sibling's frames
        LCGS_HIP_CHECK(hipEventRecord(ring[k]->ev_batch_join, ring[k]->twin_stream));
        LCGS_HIP_CHECK(hipStreamWaitEvent(ctx->stream, ring[k]->ev_batch_join, 0));
    }
    return LCGS_OK;
}

} // extern "C"

namespace lcgs
{
namespace abi
{
// The sibling context of camera / view batches: created on first use, bound to the same scene, ordered after the work
// already on the context's stream.
lcgs_status prepare_twin(lcgs_context* ctx)
{
    if (!ctx->twin) {
        LCGS_HIP_CHECK(hipStreamCreateWithFlags(&ctx->twin_stream, hipStreamNonBlocking));
        LCGS_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_batch_fork, hipEventDisableTiming));
        LCGS_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_batch_join, hipEventDisableTiming));
        LCGS_TRY(lcgs_create(ctx->device, ctx->twin_stream, &ctx->twin));
    }
    {
        lcgs_context* t = ctx->twin;
        if (t->pos != ctx->pos || t->P != ctx->P || t->sh != ctx->sh || t->sh_deg != ctx->sh_deg)
            LCGS_TRY(lcgs_scene_bind(t, ctx->P, ctx->sh_deg, ctx->pos, ctx->scale, ctx->rotq, ctx->sh, ctx->opacity));
        t->use_half_sh    = false;
        t->lod_min_radius = ctx->lod_min_radius;
        t->coarse_mode    = ctx->coarse_mode;
        t->coarse_on      = ctx->coarse_on;
        t->coarse_yield   = ctx->coarse_yield;
        // (borrowed, like the permutation: built on ctx->stream before the fork below; rows another thread's writer has
        // declared stale -- context.hpp foreign_writes -- are not handed on)
        const bool rows_ok = ctx->cull_rows() != nullptr || !(ctx->foreign_writes.load(std::memory_order_acquire) & lcgs_context::kRowsStale);
        t->cull_bound      = rows_ok ? ctx->cull_bound : nullptr;
        t->cull_key        = ctx->cull_key;
        if (!rows_ok) t->cull_key = {};
        t->foreign_writes.fetch_and(~lcgs_context::kRowsStale, std::memory_order_acq_rel);
        registry_publish(t);
        // the sibling renders the same (possibly re-ordered) arrays: it borrows their permutation for the order of equal depths
        t->scene_perm.ptr   = ctx->scene_perm.ptr;
        t->scene_perm.bytes = 0;
        t->perm_valid       = ctx->perm_valid;
        if (ctx->use_half_sh) { // the sibling reads the same f16 copy (not owned: never grown or freed through it)
            t->sh_half.ptr   = ctx->sh_half.ptr;
            t->sh_half.bytes = 0;
            t->use_half_sh   = true;
        }
        // launch sizes and pair capacity learnt by the synchronised frames of this context serve the sibling too
        t->hint_V        = std::max(t->hint_V, ctx->hint_V);
        t->hint_L        = std::max(t->hint_L, ctx->hint_L);
        t->hint_Lb       = std::max(t->hint_Lb, ctx->hint_Lb);
        t->pair_capacity = std::max(t->pair_capacity, ctx->pair_capacity);
        LCGS_HIP_CHECK(hipEventRecord(ctx->ev_batch_fork, ctx->stream));
        LCGS_HIP_CHECK(hipStreamWaitEvent(ctx->twin_stream, ctx->ev_batch_fork, 0));
    }
    return LCGS_OK;
}
} // namespace abi
} // namespace lcgs
