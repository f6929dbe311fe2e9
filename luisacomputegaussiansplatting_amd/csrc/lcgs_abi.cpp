// lcgs_abi.cpp -- the C ABI of liblcgs_hip.so (include/lcgs_hip.h): context, workspace management and
// the host-side orchestration of the HIP kernels.  No compute happens on the host; if there is no GPU
// lcgs_create fails with LCGS_ERR_NO_DEVICE -- there is no CPU fallback.
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <thread>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <string>
#include <vector>

#include "common.hpp"
#include "context.hpp"
#include "kernels/launch.hpp"
#include "kernels/tie_order.hpp"

namespace lcgs
{

static thread_local std::string g_last_error;

void set_last_error(const std::string& msg) { g_last_error = msg; }

lcgs_status hip_fail(hipError_t e, const char* what, const char* file, int line)
{
    char buf[512];
    snprintf(buf, sizeof(buf), "HIP error %d (%s) at %s:%d in `%s`", (int)e, hipGetErrorString(e), file, line, what);
    g_last_error = buf;
    if (e == hipErrorOutOfMemory) return LCGS_ERR_OUT_OF_MEMORY;
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice) return LCGS_ERR_NO_DEVICE;
    return LCGS_ERR_HIP;
}

lcgs_status DeviceBuffer::ensure(size_t need)
{
    if (need <= bytes) return LCGS_OK;
    // geometric growth, like ensure_*_temp_buffer (lcgs/src/gs_tile_splatter/impl.cpp:38-41)
    size_t new_bytes = bytes == 0 ? need : std::max(need, bytes * 2);
    new_bytes        = (new_bytes + 255) & ~(size_t)255;
    if (ptr) {
        LCGS_HIP_CHECK(hipFree(ptr));
        ptr   = nullptr;
        bytes = 0;
    }
    LCGS_HIP_CHECK(hipMalloc(&ptr, new_bytes));
    bytes = new_bytes;
    // test hook: fresh workspace starts as garbage instead of whatever the allocator hands out (usually zeros), so
    // that a kernel reading what no kernel wrote shows up in the parity tests
    static const bool poison = getenv("LCGS_POISON") != nullptr;
    if (poison) {
        LCGS_HIP_CHECK(hipMemset(ptr, 0xA5, new_bytes)); // (legacy stream: not ordered against non-blocking streams,
        LCGS_HIP_CHECK(hipDeviceSynchronize());          //  so finish it before anybody writes real data)
    }
    return LCGS_OK;
}

void DeviceBuffer::release()
{
    if (ptr) (void)hipFree(ptr);
    ptr   = nullptr;
    bytes = 0;
}

} // namespace lcgs

using namespace lcgs;

namespace
{
inline int ceil_log2_u32(uint32_t v)
{
    int b = 0;
    while ((1ull << b) < v) ++b;
    return b;
}
} // namespace

namespace lcgs
{
hipStream_t context_stream(lcgs_context* ctx) { return ctx->stream; }
int         context_device(lcgs_context* ctx) { return ctx->device; }
} // namespace lcgs

namespace
{

lcgs_status mark(lcgs_context* ctx, const char* name)
{
    // debugging hook: LCGS_DEBUG_SYNC=1 waits for the device after every stage and names it (a faulting kernel
    // then aborts right after its stage's line instead of at some later synchronisation)
    static const bool debug_sync = getenv("LCGS_DEBUG_SYNC") != nullptr;
    if (debug_sync) {
        fprintf(stderr, "[lcgs %p] %s ...\n", (void*)ctx, name);
        (void)hipDeviceSynchronize();
        fprintf(stderr, "[lcgs %p] %s done\n", (void*)ctx, name);
    }
    if (!ctx->profiling) return LCGS_OK;
    if (!ctx->events_created) {
        for (int i = 0; i < kMaxEvents; ++i) LCGS_HIP_CHECK(hipEventCreate(&ctx->events[i]));
        ctx->events_created = true;
    }
    if (ctx->n_marks >= kMaxEvents) return LCGS_OK;
    ctx->mark_names[ctx->n_marks] = name;
    LCGS_HIP_CHECK(hipEventRecord(ctx->events[ctx->n_marks], ctx->stream));
    ctx->n_marks++;
    return LCGS_OK;
}

lcgs_status collect_marks(lcgs_context* ctx)
{
    ctx->times.count = 0;
    if (!ctx->profiling || ctx->n_marks < 2) return LCGS_OK;
    LCGS_HIP_CHECK(hipEventSynchronize(ctx->events[ctx->n_marks - 1]));
    for (int i = 1; i < ctx->n_marks; ++i) {
        float ms = 0;
        LCGS_HIP_CHECK(hipEventElapsedTime(&ms, ctx->events[i - 1], ctx->events[i]));
        ctx->times.name[i - 1] = ctx->mark_names[i];
        ctx->times.ms[i - 1]   = ms;
    }
    ctx->times.count = ctx->n_marks - 1;
    return LCGS_OK;
}

// waits for the context's stream and for the counter read-back of the last frame (which travels on the auxiliary
// stream, see enqueue_forward)
lcgs_status sync_frame(lcgs_context* ctx)
{
    LCGS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->counts_pending) {
        LCGS_HIP_CHECK(hipEventSynchronize(ctx->ev_counts));
        ctx->counts_pending = false;
    }
    return LCGS_OK;
}

// after a stream synchronisation: problems the last (possibly asynchronous) frame reported through its counters
lcgs_status check_frame_flags(lcgs_context* ctx)
{
    if (!ctx->last.valid || !ctx->h_counts) return LCGS_OK;
    if (ctx->h_counts[5] != 0) {
        set_last_error("a device-side wait timed out (bounded spin expired): the frame is invalid");
        return LCGS_ERR_HIP;
    }
    if (ctx->h_counts[6] != 0) {
        // The pair workspace was too small for one or more frames since the last check (the device keeps the count and
        // the largest demand in sticky words, so a truncated asynchronous frame is not forgotten when later frames
        // fit): those images are truncated.  Grow for the next frame and clear the record.
        uint64_t want = (uint64_t)ctx->h_counts[7] + ctx->h_counts[7] / 4;
        if (want > 0x7FFFFFFFull) want = 0x7FFFFFFFull;
        ctx->pair_capacity = std::max(ctx->pair_capacity, (uint32_t)want);
        char buf[320];
        snprintf(buf, sizeof(buf),
                 "%u asynchronous frame(s) needed more (tile, splat) pairs than the workspace held (up to %u); their "
                 "images are truncated.  The workspace has been grown: render those frames again",
                 ctx->h_counts[6], ctx->h_counts[7]);
        ctx->h_counts[3] = ctx->h_counts[6] = ctx->h_counts[7] = 0;
        LCGS_HIP_CHECK(hipMemsetAsync(ctx->counts.as<uint32_t>() + 6, 0, 8, ctx->stream));
        set_last_error(buf);
        return LCGS_ERR_CAPACITY;
    }
    return LCGS_OK;
}

// deferred stage mode: run a recorded SHProcessor::process / GSProjector::forward now (context.hpp def_sh / def_proj)
lcgs_status run_deferred_sh(lcgs_context* ctx)
{
    if (!ctx->def_sh.pending) return LCGS_OK;
    ctx->def_sh.pending = false;
    CamParams cp{};
    for (int i = 0; i < 3; ++i) cp.campos[i] = ctx->def_sh.cam.position[i];
    launch_sh_process(ctx->def_sh.num, ctx->def_sh.level, cp, ctx->def_sh.pos, ctx->def_sh.sh, ctx->def_sh.color, ctx->stream);
    LCGS_HIP_CHECK(hipGetLastError());
    return LCGS_OK;
}

lcgs_status run_deferred_proj(lcgs_context* ctx)
{
    if (!ctx->def_proj.pending) return LCGS_OK;
    auto& d   = ctx->def_proj;
    d.pending = false;
    launch_project(d.num, make_cam_params(d.cam), d.use_focal != 0, d.pos, d.scale, d.rotq, d.scale_modifier, d.means, d.depth,
                   d.covs, ctx->stream);
    LCGS_HIP_CHECK(hipGetLastError());
    return LCGS_OK;
}

lcgs_status check_camera(const lcgs_camera* cam)
{
    LCGS_REQUIRE(cam != nullptr, "camera is NULL");
    LCGS_REQUIRE(cam->width > 0 && cam->height > 0, "camera width/height must be positive");
    LCGS_REQUIRE(cam->width <= 65535 * 16 && cam->height <= 65535 * 16, "resolution too large");
    LCGS_REQUIRE(cam->fov > 0.0f && cam->fov < 180.0f, "camera fov must be in (0,180) degrees");
    LCGS_REQUIRE(cam->aspect_ratio > 0.0f, "camera aspect_ratio must be positive");
    return LCGS_OK;
}

#define LCGS_TRY(expr)                    \
    do {                                  \
        lcgs_status _s = (expr);          \
        if (_s != LCGS_OK) return _s;     \
    } while (0)

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
        LCGS_TRY(ctx->tile_order[i].ensure(G * 4));
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
        LCGS_TRY(ctx->strip_masks.ensure((size_t)ctx->pair_capacity));
        LCGS_TRY(ctx->grads2d.ensure(grads2d_bytes((int64_t)P)));
        LCGS_TRY(ctx->shjac.ensure(P * 48));
        LCGS_TRY(ctx->bwd_counter.ensure(256));
    }
    if (!ctx->h_counts) LCGS_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_counts), 64, hipHostMallocDefault));
    return LCGS_OK;
}

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
                        ctx->cull_slab.as<uint4>(), ctx->chunk_info.as<uint2>(), dfirst, st);
    LCGS_TRY(mark(ctx, "cull_compact"));
    const int64_t hint_V = ctx->hint_V > 0 ? ctx->hint_V : P;
    const int64_t hint_L = ctx->hint_L > 0 ? ctx->hint_L : ctx->pair_capacity;
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
    ctx->g2d_zeroed = false;
    if (deferred && keep_state) { // behind the records, beside the sort chain and the renderer
        launch_zero_grads2d(d_counts, ctx->grads2d.as<float>(), rec_stream, ctx->bwd_counter.as<uint32_t>());
        LCGS_HIP_CHECK(hipEventRecord(ctx->ev_g2d_zero, rec_stream));
        ctx->g2d_zeroed = true;
    }
    LCGS_TRY(mark(ctx, "build_records"));

    // stable partition by tile id: only ceil(log2 G) key bits are live.  The kernel that writes the pairs also leaves
    // the partition's first per-chunk digit counts in the sort workspace (the depth sort is done with it by then).
    const int tile_bits = std::max(1, ceil_log2_u32(cp.grid_x * cp.grid_y));
    const PairSortFirstPass first = pair_sort_first_pass(ctx->pair_capacity, hint_L, 0, tile_bits, ctx->sort_ws.ptr);
    const bool counted =
        launch_expand(P, hint_V, hint_L, d_counts, cp.grid_x, order, ctx->rects.as<uint2>(), ctx->rects_sorted.as<uint2>(),
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
        launch_tile_order(ctx->ranges, G, order_now, st);
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
        launch_tile_order(ctx->ranges, G, ctx->tile_order[ob].as<uint32_t>(), ctx->aux_stream);
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
                              ctx->work_counters, persist_wgs);
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

lcgs_status prepare_twin(lcgs_context* ctx); // (the sibling context of camera / view batches, defined with them)

// marks a context and its sibling as rendering several frames at once for the duration of a batch call
struct InFlight {
    lcgs_context* c;
    InFlight(lcgs_context* ctx, bool on) : c(on ? ctx : nullptr)
    {
        for (lcgs_context* t = c; t; t = t->twin) t->frames_in_flight = true;
    }
    ~InFlight()
    {
        for (lcgs_context* t = c; t; t = t->twin) t->frames_in_flight = false;
    }
};

} // namespace

extern "C" {

const char* lcgs_version(void) { return "lcgs-hip 0.1 (gfx950)"; }
const char* lcgs_last_error(void) { return g_last_error.c_str(); }

lcgs_status lcgs_create(int device_id, void* stream, lcgs_context** out_ctx)
{
    LCGS_REQUIRE(out_ctx != nullptr, "out_ctx is NULL");
    *out_ctx  = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        set_last_error(std::string("no HIP device available (hipGetDeviceCount: ") + hipGetErrorString(e) + ", count " +
                       std::to_string(count) + "): liblcgs_hip has no CPU path");
        return LCGS_ERR_NO_DEVICE;
    }
    LCGS_REQUIRE(device_id >= 0 && device_id < count, "device_id out of range");
    LCGS_HIP_CHECK(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    LCGS_HIP_CHECK(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_last_error(std::string("liblcgs_hip is built for gfx950 only; device reports ") + prop.gcnArchName);
        return LCGS_ERR_NO_DEVICE;
    }
    lcgs_context* ctx = new (std::nothrow) lcgs_context();
    if (!ctx) return LCGS_ERR_OUT_OF_MEMORY;
    ctx->device  = device_id;
    ctx->stream  = reinterpret_cast<hipStream_t>(stream);
    ctx->num_cus = prop.multiProcessorCount;
    // tuning hooks of the frames-in-flight machinery (context.hpp): workgroups per CU of the persistent renderers
    if (const char* e = getenv("LCGS_RENDER_WGS_PER_CU")) ctx->persist_forced = atoi(e);
    if (const char* e = getenv("LCGS_RENDER_WGS_IN_FLIGHT")) ctx->persist_in_flight = atoi(e);
    if (const char* e = getenv("LCGS_BWD_WGS_PER_CU")) ctx->persist_bwd_forced = atoi(e);
    if (const char* e = getenv("LCGS_BWD_WGS_IN_FLIGHT")) ctx->persist_bwd_in_flight = atoi(e);
    if (const char* e = getenv("LCGS_GRAPH")) ctx->use_graph = (e[0] == '1'); // tuning hook
    if (const char* e = getenv("LCGS_STAGE_SORT")) ctx->stage_sort = e[0] == 'l' ? 1 : (e[0] == 's' ? 2 : 0); // test hook
    // The auxiliary stream has the LOWEST dispatch priority: its bandwidth-bound workgroups fill the gaps the main
    // stream's short, latency-bound kernels leave instead of competing with them.
    hipError_t se;
    {
        int         lo = 0, hi = 0;
        const char* m  = getenv("LCGS_AUX_PRIORITY"); // tuning hook: "low" (default), "same"
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi); // lo = numerically greatest = lowest priority
        const int prio = (m && m[0] == 's') ? 0 : lo;
        se             = hipStreamCreateWithPriority(&ctx->aux_stream, hipStreamNonBlocking, prio);
        if (se != hipSuccess) se = hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking);
    }
    for (hipEvent_t* ev : { &ctx->ev_fork, &ctx->ev_join, &ctx->ev_ranges, &ctx->ev_aux_done, &ctx->ev_render, &ctx->ev_counts,
                            &ctx->ev_g2d_zero })
        if (se == hipSuccess) se = hipEventCreateWithFlags(ev, hipEventDisableTiming);
    if (const char* e = getenv("LCGS_CHAIN_CUS")) {
        // tuning hook: the sort chain on a stream masked to K CUs, record builder + renderer on the complement.  The K
        // units are spread evenly over the 8 XCDs whichever way the driver numbers the mask bits (XCD-interleaved or
        // XCD-major): XCD x gets the bits 32 x + ((x + j) % 8 + 8 (j % 4)), j < K / 8.
        const int K = atoi(e), n = ctx->num_cus;
        if (K >= 8 && K <= 64 && K % 8 == 0 && n == 256) {
            uint32_t chain[8] = {}, rest[8];
            for (int x = 0; x < 8; ++x)
                for (int j = 0; j < K / 8; ++j) chain[x] |= 1u << ((x + j) % 8 + 8 * (j % 4));
            for (int x = 0; x < 8; ++x) rest[x] = ~chain[x];
            if (se == hipSuccess) se = hipExtStreamCreateWithCUMask(&ctx->chain_stream, 8, chain);
            if (se == hipSuccess) se = hipExtStreamCreateWithCUMask(&ctx->render_stream, 8, rest);
            for (hipEvent_t* ev : { &ctx->ev_begin, &ctx->ev_chain })
                if (se == hipSuccess) se = hipEventCreateWithFlags(ev, hipEventDisableTiming);
        }
    }
    if (se != hipSuccess) {
        lcgs_status s = hip_fail(se, "aux stream / events", __FILE__, __LINE__);
        (void)lcgs_destroy(ctx); // releases whichever of the stream / events were created
        return s;
    }
    *out_ctx    = ctx;
    return LCGS_OK;
}

lcgs_status lcgs_destroy(lcgs_context* ctx)
{
    if (!ctx) return LCGS_OK;
    (void)hipSetDevice(ctx->device);
    (void)lcgs_stage_flush(ctx); // deferred stage calls still recorded: their outputs are the caller's buffers
    if (ctx->comm) {
        comm_forget_context(ctx->comm); // (drains the communicator's stream; the caller still owns and destroys it)
        ctx->comm = nullptr;
    }
    if (ctx->twin) {
        ctx->twin->sh_half.ptr    = nullptr; // borrowed from this context
        ctx->twin->scene_perm.ptr = nullptr; // likewise
        (void)lcgs_destroy(ctx->twin);
        ctx->twin = nullptr;
    }
    if (ctx->twin_stream) {
        (void)hipStreamSynchronize(ctx->twin_stream);
        (void)hipStreamDestroy(ctx->twin_stream);
    }
    for (hipEvent_t ev : { ctx->ev_batch_fork, ctx->ev_batch_join })
        if (ev) (void)hipEventDestroy(ev);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->aux_stream) (void)hipStreamSynchronize(ctx->aux_stream);
    for (hipStream_t s : { ctx->chain_stream, ctx->render_stream })
        if (s) {
            (void)hipStreamSynchronize(s);
            (void)hipStreamDestroy(s);
        }
    for (hipEvent_t ev : { ctx->ev_begin, ctx->ev_chain })
        if (ev) (void)hipEventDestroy(ev);
    DeviceBuffer* bufs[] = { &ctx->recs, &ctx->sortk[0], &ctx->sortk[1], &ctx->sortv[0], &ctx->sortv[1], &ctx->vis_index,
                             &ctx->rects, &ctx->rects_sorted, &ctx->cull_slab, &ctx->chunk_info, &ctx->chunk_base, &ctx->pairk[0], &ctx->pairk[1], &ctx->pairv[0],
                             &ctx->pairv[1], &ctx->zero_ws[0], &ctx->zero_ws[1], &ctx->zero_ws[2], &ctx->counts, &ctx->sort_ws,
                             &ctx->expand_ws, &ctx->final_T, &ctx->n_contrib, &ctx->list_idx, &ctx->grads2d, &ctx->tile_order[0], &ctx->tile_order[1], &ctx->st_keys_tmp,
                             &ctx->st_vals_tmp, &ctx->st_sort_temp, &ctx->st_scan_temp, &ctx->st_scalar, &ctx->sh_half, &ctx->strip_masks, &ctx->shjac, &ctx->tie_ws, &ctx->fused_grads, &ctx->bwd_counter, &ctx->st_flags, &ctx->st_keys_exp, &ctx->st_vals_exp,
                             &ctx->st_u32[0], &ctx->st_u32[1], &ctx->st_u32[2], &ctx->st_u32[3], &ctx->st_u32[4], &ctx->st_u32[5], &ctx->st_u32[6], &ctx->st_u32[7] };
    for (DeviceBuffer* b : bufs) b->release();
    for (auto& b : ctx->owned) b.release();
    if (ctx->graph_exec) (void)hipGraphExecDestroy(ctx->graph_exec);
    if (ctx->aux_stream) {
        (void)hipStreamSynchronize(ctx->aux_stream);
        (void)hipStreamDestroy(ctx->aux_stream);
    }
    for (hipEvent_t ev : { ctx->ev_fork, ctx->ev_join, ctx->ev_ranges, ctx->ev_aux_done, ctx->ev_render, ctx->ev_counts,
                           ctx->ev_g2d_zero })
        if (ev) (void)hipEventDestroy(ev);
    ctx->fit_img.release();
    ctx->fit_dL.release();
    if (ctx->ev_fit_bwd) (void)hipEventDestroy(ctx->ev_fit_bwd);
    ctx->frame_params.release();
    ctx->slice_bounds.release();
    ctx->scene_perm.release();
    for (hipEvent_t ev : ctx->ev_slice)
        if (ev) (void)hipEventDestroy(ev);
    if (ctx->h_counts) (void)hipHostFree(ctx->h_counts);
    if (ctx->events_created)
        for (auto& ev : ctx->events) (void)hipEventDestroy(ev);
    delete ctx;
    return LCGS_OK;
}

lcgs_status lcgs_set_stream(lcgs_context* ctx, void* stream)
{
    LCGS_REQUIRE(ctx != nullptr, "ctx is NULL");
    if (ctx->stream != reinterpret_cast<hipStream_t>(stream)) {
        // deferred stage calls were recorded against the OLD stream's order: they run there, before the switch
        LCGS_TRY(lcgs_stage_flush(ctx));
        // frames still in flight were ordered against the old stream: drain them before switching
        LCGS_HIP_CHECK(hipSetDevice(ctx->device));
        LCGS_TRY(sync_frame(ctx));
        for (lcgs_context* t = ctx->twin; t; t = t->twin) LCGS_TRY(sync_frame(t));
    }
    ctx->stream = reinterpret_cast<hipStream_t>(stream);
    return LCGS_OK;
}

lcgs_status lcgs_synchronize(lcgs_context* ctx)
{
    LCGS_REQUIRE(ctx != nullptr, "ctx is NULL");
    LCGS_TRY(lcgs_stage_flush(ctx)); // deferred stage mode: whatever was recorded is produced before the caller looks
    LCGS_TRY(sync_frame(ctx));
    lcgs_status twin_status = LCGS_OK;
    for (lcgs_context* t = ctx->twin; t; t = t->twin) { // (every workspace is checked, and grown, by one call)
        LCGS_TRY(sync_frame(t));
        const lcgs_status s = check_frame_flags(t);
        if (twin_status == LCGS_OK) twin_status = s;
    }
    const lcgs_status own_status = check_frame_flags(ctx);
    return own_status != LCGS_OK ? own_status : twin_status;
}

// ---------------------------------------------------------------------------------------------- stages

lcgs_status lcgs_sh_process(lcgs_context* ctx, int num_points, const float* d_pos, const lcgs_camera* camera,
                            const float* d_sh, float* d_color, int level, int channel)
{
    LCGS_REQUIRE(ctx != nullptr, "ctx is NULL");
    LCGS_HIP_CHECK(hipSetDevice(ctx->device)); // multi-GPU processes: every entry point selects its device
    LCGS_REQUIRE(num_points >= 0, "num_points < 0");
    LCGS_REQUIRE(camera != nullptr, "camera is NULL");
    LCGS_REQUIRE(level >= -1 && level <= 3, "SH level must be in [-1,3]");
    LCGS_REQUIRE(channel == 3, "only 3 colour channels are supported (as in the reference)");
    if (num_points == 0) return LCGS_OK;
    LCGS_REQUIRE(d_pos && d_sh && d_color, "NULL device pointer");
    if (ctx->stage_mode == LCGS_STAGES_DEFERRED) { // recorded; run by the splatter's fused frame, or by a flush
        if (ctx->def_sh.pending) LCGS_TRY(run_deferred_sh(ctx));
        ctx->def_sh.pending = true;
        ctx->def_sh.num = num_points; ctx->def_sh.level = level;
        ctx->def_sh.pos = d_pos; ctx->def_sh.sh = d_sh; ctx->def_sh.color = d_color;
        ctx->def_sh.cam = *camera;
        return LCGS_OK;
    }
    CamParams cp{};
    for (int i = 0; i < 3; ++i) cp.campos[i] = camera->position[i];
    launch_sh_process(num_points, level, cp, d_pos, d_sh, d_color, ctx->stream);
    LCGS_HIP_CHECK(hipGetLastError());
    return LCGS_OK;
}

lcgs_status lcgs_project_forward(lcgs_context* ctx, int num_gaussians, const float* d_pos, const float* d_scale,
                                 const float* d_rotq, float scale_modifier, float* d_means_2d, float* d_covs_2d,
                                 float* d_depth, const lcgs_camera* camera, int use_focal)
{
    LCGS_REQUIRE(ctx != nullptr, "ctx is NULL");
    LCGS_HIP_CHECK(hipSetDevice(ctx->device)); // multi-GPU processes: every entry point selects its device
    LCGS_REQUIRE(num_gaussians >= 0, "num_gaussians < 0");
    LCGS_TRY(check_camera(camera));
    if (num_gaussians == 0) return LCGS_OK;
    LCGS_REQUIRE(d_pos && d_scale && d_rotq && d_means_2d && d_covs_2d && d_depth, "NULL device pointer");
    if (ctx->stage_mode == LCGS_STAGES_DEFERRED) {
        if (ctx->def_proj.pending) LCGS_TRY(run_deferred_proj(ctx));
        auto& d = ctx->def_proj;
        d.pending = true;
        d.num = num_gaussians; d.use_focal = use_focal;
        d.pos = d_pos; d.scale = d_scale; d.rotq = d_rotq; d.scale_modifier = scale_modifier;
        d.means = d_means_2d; d.covs = d_covs_2d; d.depth = d_depth;
        d.cam = *camera;
        return LCGS_OK;
    }
    CamParams cp = make_cam_params(*camera);
    launch_project(num_gaussians, cp, use_focal != 0, d_pos, d_scale, d_rotq, scale_modifier, d_means_2d, d_depth,
                   d_covs_2d, ctx->stream);
    LCGS_HIP_CHECK(hipGetLastError());
    return LCGS_OK;
}

lcgs_status lcgs_set_stage_mode(lcgs_context* ctx, int mode)
{
    LCGS_REQUIRE(ctx != nullptr, "ctx is NULL");
    LCGS_REQUIRE(mode == LCGS_STAGES_EXACT || mode == LCGS_STAGES_DEFERRED, "unknown stage mode");
    LCGS_TRY(lcgs_stage_flush(ctx));
    ctx->stage_mode = mode;
    return LCGS_OK;
}

lcgs_status lcgs_stage_flush(lcgs_context* ctx)
{
    LCGS_REQUIRE(ctx != nullptr, "ctx is NULL");
    if (!ctx->def_sh.pending && !ctx->def_proj.pending) return LCGS_OK;
    LCGS_HIP_CHECK(hipSetDevice(ctx->device));
    LCGS_TRY(run_deferred_sh(ctx));
    return run_deferred_proj(ctx);
}

lcgs_status lcgs_inclusive_sum_u32(lcgs_context* ctx, const uint32_t* d_in, uint32_t* d_out, int64_t n)
{
    LCGS_REQUIRE(ctx != nullptr, "ctx is NULL");
    LCGS_HIP_CHECK(hipSetDevice(ctx->device)); // multi-GPU processes: every entry point selects its device
    LCGS_REQUIRE(n >= 0 && n < ((int64_t)1 << 31), "n out of range");
    if (n == 0) return LCGS_OK;
    LCGS_REQUIRE(d_in && d_out, "NULL device pointer");
    LCGS_TRY(ctx->st_scan_temp.ensure(scan_temp_bytes(n)));
    launch_inclusive_sum_u32(d_in, d_out, n, ctx->st_scan_temp.ptr, ctx->stream);
    LCGS_HIP_CHECK(hipGetLastError());
    return LCGS_OK;
}

lcgs_status lcgs_sort_pairs_u64_u32(lcgs_context* ctx, const uint64_t* d_keys_in, uint64_t* d_keys_out,
                                    const uint32_t* d_vals_in, uint32_t* d_vals_out, int64_t n, int begin_bit,
                                    int end_bit)
{
    LCGS_REQUIRE(ctx != nullptr, "ctx is NULL");
    LCGS_HIP_CHECK(hipSetDevice(ctx->device)); // multi-GPU processes: every entry point selects its device
    LCGS_REQUIRE(n >= 0 && n < ((int64_t)1 << 30), "n out of range");
    LCGS_REQUIRE(begin_bit >= 0 && end_bit <= 64 && begin_bit <= end_bit, "bad bit range");
    if (n == 0) return LCGS_OK;
    LCGS_REQUIRE(d_keys_in && d_keys_out && d_vals_in && d_vals_out, "NULL device pointer");
    LCGS_REQUIRE((const void*)d_keys_in != (const void*)d_keys_out && (const void*)d_vals_in != (const void*)d_vals_out,
                 "in-place sort is not supported");
    LCGS_TRY(ctx->st_keys_tmp.ensure((size_t)n * 8));
    LCGS_TRY(ctx->st_vals_tmp.ensure((size_t)n * 4));
    LCGS_TRY(ctx->st_sort_temp.ensure(pair_sort_ws_bytes(n)));
    launch_pair_sort_u64_preserve(d_keys_in, d_vals_in, d_keys_out, d_vals_out, ctx->st_keys_tmp.as<uint64_t>(),
                                  ctx->st_vals_tmp.as<uint32_t>(), n, begin_bit, end_bit, ctx->st_sort_temp.ptr,
                                  ctx->stream);
    LCGS_HIP_CHECK(hipGetLastError());
    return LCGS_OK;
}

// GSTileSplatter::forward, lcgs/src/gs_tile_splatter/impl.cpp:63-180 -- same stage order, same buffers.
lcgs_status lcgs_tile_splat_forward(lcgs_context* ctx, const lcgs_tile_accel* accel, const lcgs_tile_input* input,
                                    const lcgs_tile_output* output, int use_focal, int* num_rendered)
{
    LCGS_REQUIRE(ctx && accel && input && output, "NULL argument");
    LCGS_HIP_CHECK(hipSetDevice(ctx->device)); // multi-GPU processes: every entry point selects its device
    if (num_rendered) *num_rendered = 0;
    const int P = input->num_gaussians;
    LCGS_REQUIRE(P >= 0, "num_gaussians < 0");
    LCGS_REQUIRE(output->width > 0 && output->height > 0, "output size must be positive");
    LCGS_REQUIRE(accel->tiles_touched && accel->point_offsets && accel->point_list_keys_unsorted &&
                     accel->point_list_unsorted && accel->point_list_keys && accel->point_list && accel->ranges,
                 "NULL accel buffer");
    LCGS_REQUIRE(output->target_img && output->radii, "NULL output buffer");
    if (P == 0) return LCGS_OK; // num_rendered = 0: nothing drawn, image untouched (impl.cpp:109)
    LCGS_REQUIRE(input->means_2d && input->depth_features && input->conic && input->color_features &&
                     input->opacity_features,
                 "NULL input buffer");
    if (ctx->def_sh.pending || ctx->def_proj.pending) {
        // Deferred mode: are this call's inputs exactly what the two recorded calls would have produced?  Then the fused
        // frame gives the same image / radii / num_rendered from the 3-D arrays; else the recorded calls run now.
        const auto& a = ctx->def_sh;
        const auto& b = ctx->def_proj;
        const bool  same_cam = memcmp(a.cam.position, b.cam.position, sizeof(float) * 3) == 0;
        const bool  match =
            a.pending && b.pending && a.num == P && b.num == P && a.pos == b.pos && a.color == input->color_features &&
            b.means == input->means_2d && b.covs == input->conic && b.depth == input->depth_features && same_cam &&
            b.use_focal != 0 && use_focal != 0 && a.level >= 0 && a.level <= 3 && b.cam.width == output->width &&
            b.cam.height == output->height && !output->final_T && !output->n_contrib && ctx->lod_min_radius == 0 &&
            (reinterpret_cast<uintptr_t>(b.rotq) & 15) == 0 && P < (1 << 30);
        if (match) {
            // the recorded arrays stand in for the context's scene for this one frame
            struct Saved {
                int P, sh_deg; const float *pos, *scale, *rotq, *sh, *opacity; bool perm_valid, use_half_sh;
            } sv = { ctx->P, ctx->sh_deg, ctx->pos, ctx->scale, ctx->rotq, ctx->sh, ctx->opacity, ctx->perm_valid,
                     ctx->use_half_sh };
            ctx->P = P; ctx->sh_deg = a.level; ctx->pos = a.pos; ctx->scale = b.scale; ctx->rotq = b.rotq; ctx->sh = a.sh;
            ctx->opacity = input->opacity_features; ctx->perm_valid = false; ctx->use_half_sh = false;
            int         n  = 0;
            lcgs_status rs = lcgs_render_forward(ctx, &b.cam, input->bg_color, b.scale_modifier, output->target_img,
                                                 output->radii, 0, &n);
            ctx->P = sv.P; ctx->sh_deg = sv.sh_deg; ctx->pos = sv.pos; ctx->scale = sv.scale; ctx->rotq = sv.rotq;
            ctx->sh = sv.sh; ctx->opacity = sv.opacity; ctx->perm_valid = sv.perm_valid; ctx->use_half_sh = sv.use_half_sh;
            ctx->last.valid = false; // (the frame state belongs to the borrowed arrays)
            if (num_rendered) *num_rendered = n;
            // the recorded calls are consumed only by a frame that was rendered: after an error (capacity, HIP) they stay
            // pending, and a later flush / synchronise / splatter call still produces what they promised
            if (rs == LCGS_OK) ctx->def_sh.pending = ctx->def_proj.pending = false;
            return rs;
        }
        LCGS_TRY(run_deferred_sh(ctx));
        LCGS_TRY(run_deferred_proj(ctx));
    }
    hipStream_t st = ctx->stream;
    CamParams   cp{};
    cp.width  = (uint32_t)output->width;
    cp.height = (uint32_t)output->height;
    cp.grid_x = div_up(cp.width, kBlockX); // impl.cpp:76-79
    cp.grid_y = div_up(cp.height, kBlockY);

    LCGS_TRY(ctx->st_scalar.ensure(16));
    uint32_t* d_hole = ctx->st_scalar.as<uint32_t>();
    LCGS_HIP_CHECK(hipMemsetAsync(d_hole, 0, 4, st));
    launch_allocate_tiles(P, cp, use_focal != 0, input->depth_features, input->means_2d, input->conic,
                          accel->tiles_touched, output->radii, st, d_hole); // impl.cpp:87-99
    LCGS_TRY(ctx->st_scan_temp.ensure(scan_temp_bytes(P)));
    launch_inclusive_sum_u32(accel->tiles_touched, accel->point_offsets, P, ctx->st_scan_temp.ptr, st); // impl.cpp:104
    // (for the sort below: the splats that claim pair slots, ascending -- compacted before the one synchronisation)
    const size_t fbytes = sparse_flag_bytes(P);
    LCGS_TRY(ctx->st_flags.ensure(fbytes));
    LCGS_TRY(ctx->st_u32[0].ensure((size_t)sparse_flag_chunks(P) * 4 + 4));
    LCGS_TRY(ctx->st_u32[1].ensure((size_t)P * 4 + 4));
    uint8_t*  d_flags = ctx->st_flags.as<uint8_t>();
    uint32_t* d_vis   = ctx->st_u32[1].as<uint32_t>();
    uint32_t* d_nvis  = d_hole + 1;
    if (fbytes > (size_t)P) LCGS_HIP_CHECK(hipMemsetAsync(d_flags + P, 0, fbytes - (size_t)P, st)); // the padding
    launch_tile_flags(P, accel->tiles_touched, d_flags, st);
    launch_compact_flags(d_flags, P, ctx->st_u32[0].as<uint32_t>(), d_vis, d_nvis, st);
    int32_t  L    = 0;
    uint32_t hole = 0, n_vis = 0;
    LCGS_HIP_CHECK(hipMemcpyAsync(&L, accel->point_offsets + (P - 1), 4, hipMemcpyDeviceToHost, st)); // impl.cpp:106
    LCGS_HIP_CHECK(hipMemcpyAsync(&hole, d_hole, 4, hipMemcpyDeviceToHost, st));
    LCGS_HIP_CHECK(hipMemcpyAsync(&n_vis, d_nvis, 4, hipMemcpyDeviceToHost, st));
    LCGS_HIP_CHECK(hipStreamSynchronize(st));                                                        // impl.cpp:107
    if (num_rendered) *num_rendered = L;
    if (L <= 0) return LCGS_OK; // impl.cpp:109
    if ((int64_t)L > accel->capacity) {
        char buf[160];
        snprintf(buf, sizeof(buf), "num_rendered = %d exceeds the pair buffer capacity %lld", L, (long long)accel->capacity);
        set_last_error(buf);
        return LCGS_ERR_CAPACITY;
    }
    // The reference zero-fills both unsorted pair buffers every frame (impl.cpp:117-118) and then overwrites every slot --
    // except those of a splat whose covariance is NaN (radius 0, tiles > 0: copy_with_keys skips it, shader.cpp:41-42), which
    // keep the fill: key 0, splat 0.  The fill (12 bytes x num_rendered: 156 MB on the bicycle stand-in) is therefore issued
    // only in a frame that has such a splat; the buffers' contents are the reference's either way.
    if (hole) {
        LCGS_HIP_CHECK(hipMemsetAsync(accel->point_list_unsorted, 0, (size_t)L * 4, st));      // impl.cpp:117
        LCGS_HIP_CHECK(hipMemsetAsync(accel->point_list_keys_unsorted, 0, (size_t)L * 8, st)); // impl.cpp:118
    }
    launch_copy_with_keys(P, cp, input->means_2d, accel->point_offsets, output->radii, input->depth_features,
                          accel->point_list_keys_unsorted, accel->point_list_unsorted, st); // impl.cpp:120-130
    // impl.cpp:135-143 sorts all 64 key bits; only 32 + ceil(log2 G) of them can differ
    const int tile_bits = std::max(1, ceil_log2_u32(cp.grid_x * cp.grid_y));
    LCGS_TRY(ctx->st_keys_tmp.ensure((size_t)L * 8));
    LCGS_TRY(ctx->st_vals_tmp.ensure((size_t)L * 4));
    LCGS_TRY(ctx->st_sort_temp.ensure(pair_sort_ws_bytes(L)));
    // (small frames: the six passes over few pairs beat the longer chain of short launches -- 0.36 M pairs: 4270 vs 3930
    //  frames/s, 2.0 M: equal, 13 M: 610 vs 710; LCGS_STAGE_SORT=literal|splats forces either, a tuning / test hook)
    const bool literal = ctx->stage_sort ? ctx->stage_sort == 1 : L < (4 << 20); // (the hook is read once, at lcgs_create)
    if (hole || n_vis == 0 || literal) {
        // the reference's sort as it stands: all live key bits of the unsorted pairs (always for frames with zero-filled
        // slots, whose pairs exist nowhere but in those buffers)
        launch_pair_sort_u64_preserve(accel->point_list_keys_unsorted, accel->point_list_unsorted, accel->point_list_keys,
                                      accel->point_list, ctx->st_keys_tmp.as<uint64_t>(), ctx->st_vals_tmp.as<uint32_t>(),
                                      L, 0, 32 + tile_bits, ctx->st_sort_temp.ptr, st);
    } else {
        // The same sorted arrays by sort-before-duplicate: a stable LSD sort on (tile << 32 | depth bits) sorts
        // the low 32 bits first -- and all pairs of a splat share them.  So the n_vis splats that claim slots are sorted by
        // depth bits (stable: ascending index inside equal depths, the order their pairs have in the unsorted buffers), their
        // pairs are written out again in THAT order (k_copy_with_keys over the sorted sequence, same (y, x) order inside a
        // splat), and only the tile bits remain to be sorted over the num_rendered pairs: 4 passes over n_vis + 2 over L
        // instead of 6 over L, bit-identical keys / lists (tests/test_gpu_stages.py compares every entry).
        const int n = (int)n_vis;
        for (int i = 2; i < 8; ++i) LCGS_TRY(ctx->st_u32[i].ensure((size_t)n * 4 + 16));
        uint32_t *ka = ctx->st_u32[2].as<uint32_t>(), *kb = ctx->st_u32[3].as<uint32_t>(), *va = ctx->st_u32[4].as<uint32_t>(),
                 *vb = ctx->st_u32[5].as<uint32_t>(), *cnt = ctx->st_u32[6].as<uint32_t>(), *offs = ctx->st_u32[7].as<uint32_t>();
        launch_gather_depth_keys(n, d_vis, input->depth_features, ka, va, st);
        const int which = launch_pair_sort_u32(ka, kb, va, vb, d_nvis, n, n, 0, 32, ctx->st_sort_temp.ptr, st);
        const uint32_t* order = which ? vb : va;
        launch_gather_u32(n, order, accel->tiles_touched, cnt, st);
        LCGS_TRY(ctx->st_scan_temp.ensure(scan_temp_bytes(n)));
        launch_inclusive_sum_u32(cnt, offs, n, ctx->st_scan_temp.ptr, st);
        LCGS_TRY(ctx->st_keys_exp.ensure((size_t)L * 8));
        LCGS_TRY(ctx->st_vals_exp.ensure((size_t)L * 4));
        launch_copy_with_keys_ordered(n, cp, input->means_2d, offs, output->radii, input->depth_features, order,
                                      ctx->st_keys_exp.as<uint64_t>(), ctx->st_vals_exp.as<uint32_t>(), st);
        launch_pair_sort_u64_preserve(ctx->st_keys_exp.as<uint64_t>(), ctx->st_vals_exp.as<uint32_t>(), accel->point_list_keys,
                                      accel->point_list, ctx->st_keys_tmp.as<uint64_t>(), ctx->st_vals_tmp.as<uint32_t>(), L,
                                      32, 32 + tile_bits, ctx->st_sort_temp.ptr, st);
    }
    const size_t G = (size_t)cp.grid_x * cp.grid_y;
    LCGS_HIP_CHECK(hipMemsetAsync(accel->ranges, 0, G * 2 * 4, st)); // impl.cpp:147
    launch_get_ranges_u64(L, accel->point_list_keys, accel->ranges, st); // impl.cpp:150-156
    launch_render_forward_aos(cp, input->bg_color, accel->ranges, accel->point_list, input->means_2d, input->conic,
                              input->opacity_features, input->color_features, output->target_img, output->final_T,
                              output->n_contrib, st); // impl.cpp:159-174
    LCGS_HIP_CHECK(hipGetLastError());
    return LCGS_OK; // no sync, like impl.cpp:177
}

// ---------------------------------------------------------------------------------------------- fused

lcgs_status lcgs_scene_bind(lcgs_context* ctx, int num_gaussians, int sh_degree, const float* d_pos,
                            const float* d_scale, const float* d_rotq, const float* d_sh, const float* d_opacity)
{
    LCGS_REQUIRE(ctx != nullptr, "ctx is NULL");
    LCGS_REQUIRE(num_gaussians >= 0 && num_gaussians < (1 << 30), "num_gaussians out of range");
    LCGS_REQUIRE(sh_degree >= 0 && sh_degree <= 3, "sh_degree must be in [0,3]");
    if (num_gaussians > 0) LCGS_REQUIRE(d_pos && d_scale && d_rotq && d_sh && d_opacity, "NULL device pointer");
    LCGS_REQUIRE((reinterpret_cast<uintptr_t>(d_rotq) & 15) == 0, "rotq must be 16-byte aligned");
    ctx->P       = num_gaussians;
    ctx->sh_deg  = sh_degree;
    ctx->pos     = d_pos;
    ctx->scale   = d_scale;
    ctx->rotq    = d_rotq;
    ctx->sh      = d_sh;
    ctx->opacity = d_opacity;
    ctx->last.valid = false;
    ctx->use_half_sh = false; // a new scene: the f16 copy (if any) is stale
    // the caller's arrays, the caller's order -- unless these ARE the context's own re-ordered arrays (bound again after
    // something else was): their permutation, and with it the reference's order of equal depths, still applies
    ctx->perm_valid = ctx->perm_for_owned && num_gaussians > 0 && d_pos == ctx->owned[0].as<float>() &&
                      d_scale == ctx->owned[1].as<float>() && d_rotq == ctx->owned[2].as<float>() &&
                      d_sh == ctx->owned[3].as<float>() && d_opacity == ctx->owned[4].as<float>();
    return LCGS_OK;
}

lcgs_status lcgs_set_lod(lcgs_context* ctx, int min_radius_px)
{
    LCGS_REQUIRE(ctx != nullptr, "ctx is NULL");
    LCGS_REQUIRE(min_radius_px >= 0 && min_radius_px <= 4096, "min_radius_px out of range");
    ctx->lod_min_radius = min_radius_px;
    if (ctx->twin) ctx->twin->lod_min_radius = min_radius_px;
    return LCGS_OK;
}

lcgs_status lcgs_set_ingest_order(lcgs_context* ctx, int order)
{
    LCGS_REQUIRE(ctx != nullptr, "ctx is NULL");
    LCGS_REQUIRE(order == LCGS_ORDER_FILE || order == LCGS_ORDER_SPATIAL, "order must be LCGS_ORDER_FILE or LCGS_ORDER_SPATIAL");
    ctx->ingest_order = order;
    return LCGS_OK;
}

lcgs_status lcgs_scene_permutation(lcgs_context* ctx, const uint32_t** d_perm)
{
    LCGS_REQUIRE(ctx && d_perm, "NULL argument");
    *d_perm = ctx->perm_valid ? ctx->scene_perm.as<uint32_t>() : nullptr;
    return LCGS_OK;
}

lcgs_status lcgs_scene_upload(lcgs_context* ctx, int num_gaussians, int sh_degree, const float* h_pos,
                              const float* h_scale, const float* h_rotq, const float* h_sh, const float* h_opacity)
{
    LCGS_REQUIRE(ctx != nullptr, "ctx is NULL");
    LCGS_HIP_CHECK(hipSetDevice(ctx->device)); // multi-GPU processes: every entry point selects its device
    LCGS_REQUIRE(num_gaussians >= 0 && num_gaussians < (1 << 30), "num_gaussians out of range");
    LCGS_REQUIRE(sh_degree >= 0 && sh_degree <= 3, "sh_degree must be in [0,3]");
    if (num_gaussians > 0) LCGS_REQUIRE(h_pos && h_scale && h_rotq && h_sh && h_opacity, "NULL host pointer");
    const size_t P        = (size_t)num_gaussians;
    const size_t feat     = (size_t)(sh_degree + 1) * (sh_degree + 1) * 3;
    const size_t sizes[5] = { P * 3 * 4, P * 3 * 4, P * 4 * 4, P * feat * 4, P * 4 };
    const float* src[5]   = { h_pos, h_scale, h_rotq, h_sh, h_opacity };
    ctx->perm_for_owned = false; // owned[] is rewritten in the given order
    for (int i = 0; i < 5; ++i) {
        LCGS_TRY(ctx->owned[i].ensure(std::max<size_t>(sizes[i], 16)));
        if (sizes[i])
            LCGS_HIP_CHECK(hipMemcpyAsync(ctx->owned[i].ptr, src[i], sizes[i], hipMemcpyHostToDevice, ctx->stream));
    }
    LCGS_HIP_CHECK(hipStreamSynchronize(ctx->stream)); // app/main.cpp:223
    LCGS_TRY(lcgs_scene_bind(ctx, num_gaussians, sh_degree, ctx->owned[0].as<float>(), ctx->owned[1].as<float>(),
                             ctx->owned[2].as<float>(), ctx->owned[3].as<float>(), ctx->owned[4].as<float>()));
    // a scene the context owns is kept in spatial order unless the caller asked for the given one (lcgs_set_ingest_order)
    if (ctx->ingest_order == LCGS_ORDER_SPATIAL && num_gaussians > 0) LCGS_TRY(lcgs_scene_reorder_spatial(ctx, nullptr));
    return LCGS_OK;
}

lcgs_status lcgs_scene_reorder_spatial(lcgs_context* ctx, uint32_t* d_perm)
{
    LCGS_REQUIRE(ctx != nullptr, "ctx is NULL");
    LCGS_HIP_CHECK(hipSetDevice(ctx->device));
    if (ctx->P == 0) return LCGS_OK;
    LCGS_REQUIRE(ctx->pos != nullptr, "no scene bound (call lcgs_scene_bind / lcgs_scene_upload / lcgs_scene_load_ply first)");
    LCGS_TRY(sync_frame(ctx)); // frames in flight still read the old arrays
    for (lcgs_context* t = ctx->twin; t; t = t->twin) LCGS_TRY(sync_frame(t));
    hipStream_t   st = ctx->stream;
    const int64_t P  = ctx->P;
    // ---- the box: mean +- 4 sigma per axis (finite positions only), 1024 cells per axis
    DeviceBuffer partial;
    const int    nb = pos_moment_blocks();
    LCGS_TRY(partial.ensure((size_t)nb * 7 * sizeof(double)));
    launch_pos_moments(P, ctx->pos, partial.as<double>(), st);
    std::vector<double> h((size_t)nb * 7);
    hipError_t          e = hipMemcpyAsync(h.data(), partial.ptr, h.size() * sizeof(double), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    partial.release();
    LCGS_HIP_CHECK(e);
    double acc[7] = { 0, 0, 0, 0, 0, 0, 0 };
    for (int b = 0; b < nb; ++b)
        for (int k = 0; k < 7; ++k) acc[k] += h[(size_t)b * 7 + k];
    float lo[3] = { 0, 0, 0 }, cells[3] = { 1, 1, 1 };
    for (int a = 0; a < 3; ++a) {
        const double n = std::max(acc[6], 1.0), mean = acc[a] / n;
        const double sd = std::sqrt(std::max(acc[3 + a] / n - mean * mean, 0.0));
        const double half = std::max(4.0 * sd, 1e-6);
        lo[a]    = (float)(mean - half);
        cells[a] = (float)(1024.0 / (2.0 * half));
    }
    // ---- keys, stable sort, gather
    DeviceBuffer keys[2], vals[2], ws, fresh[5];
    auto         drop = [&]() {
        for (int i = 0; i < 2; ++i) {
            keys[i].release();
            vals[i].release();
        }
        ws.release();
    };
    lcgs_status s = LCGS_OK;
    for (int i = 0; i < 2 && s == LCGS_OK; ++i) {
        s = keys[i].ensure((size_t)P * 4);
        if (s == LCGS_OK) s = vals[i].ensure((size_t)P * 4);
    }
    if (s == LCGS_OK) s = ws.ensure(pair_sort_ws_bytes(P));
    const size_t feat    = (size_t)(ctx->sh_deg + 1) * (ctx->sh_deg + 1) * 3;
    const int    rowf[5] = { 3, 3, 4, (int)feat, 1 };
    for (int i = 0; i < 5 && s == LCGS_OK; ++i) s = fresh[i].ensure(std::max<size_t>((size_t)P * rowf[i] * 4, 16));
    if (s != LCGS_OK) {
        drop();
        for (DeviceBuffer& b : fresh) b.release();
        return s;
    }
    launch_morton_keys(P, ctx->pos, lo, cells, keys[0].as<uint32_t>(), vals[0].as<uint32_t>(), st);
    const int where = launch_pair_sort_u32(keys[0].as<uint32_t>(), keys[1].as<uint32_t>(), vals[0].as<uint32_t>(),
                                           vals[1].as<uint32_t>(), nullptr, P, P, 0, 30, ws.ptr, st);
    const uint32_t* perm   = vals[where].as<uint32_t>();
    const float*    src[5] = { ctx->pos, ctx->scale, ctx->rotq, ctx->sh, ctx->opacity };
    for (int i = 0; i < 5; ++i) launch_gather_rows(P, rowf[i], perm, src[i], fresh[i].as<float>(), st);
    e = hipGetLastError();
    DeviceBuffer kept_perm;
    if (e == hipSuccess && kept_perm.ensure((size_t)P * 4) != LCGS_OK) e = hipErrorOutOfMemory;
    if (e == hipSuccess) e = hipMemcpyAsync(kept_perm.ptr, perm, (size_t)P * 4, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess && d_perm) e = hipMemcpyAsync(d_perm, perm, (size_t)P * 4, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    drop();
    if (e != hipSuccess) {
        for (DeviceBuffer& b : fresh) b.release();
        kept_perm.release();
        LCGS_HIP_CHECK(e);
    }
    // ---- the context now owns (and renders from) the re-ordered copy
    const bool half = ctx->use_half_sh, had_perm = ctx->perm_valid;
    for (int i = 0; i < 5; ++i) {
        ctx->owned[i].release();
        ctx->owned[i] = fresh[i];
    }
    LCGS_TRY(lcgs_scene_bind(ctx, ctx->P, ctx->sh_deg, ctx->owned[0].as<float>(), ctx->owned[1].as<float>(),
                             ctx->owned[2].as<float>(), ctx->owned[3].as<float>(), ctx->owned[4].as<float>()));
    // the permutation stays with the context (composed with an earlier one: file index of every row)
    if (had_perm) {
        DeviceBuffer composed;
        lcgs_status  cs = composed.ensure((size_t)P * 4);
        if (cs == LCGS_OK) {
            launch_gather_rows(P, 1, kept_perm.as<uint32_t>(), ctx->scene_perm.as<float>(), composed.as<float>(), st);
            hipError_t ce = hipStreamSynchronize(st);
            kept_perm.release();
            if (ce != hipSuccess) {
                composed.release();
                LCGS_HIP_CHECK(ce);
            }
            ctx->scene_perm.release();
            ctx->scene_perm = composed;
        } else {
            kept_perm.release();
            return cs;
        }
    } else {
        ctx->scene_perm.release();
        ctx->scene_perm = kept_perm;
    }
    ctx->perm_valid     = true;
    ctx->perm_for_owned = true;
    if (half) LCGS_TRY(lcgs_scene_use_half_sh(ctx, 1)); // the f16 copy follows the new order
    return LCGS_OK;
}

lcgs_status lcgs_adam_step(lcgs_context* ctx, int num_gaussians, int sh_degree, const lcgs_adam_config* cfg,
                           const lcgs_grads* grads, const lcgs_params* raw, const lcgs_params* m, const lcgs_params* v,
                           const lcgs_params* activated)
{
    LCGS_REQUIRE(ctx && cfg && grads && raw && m && v && activated, "NULL argument");
    LCGS_REQUIRE(num_gaussians >= 0, "num_gaussians is negative");
    LCGS_REQUIRE(sh_degree >= 0 && sh_degree <= 3, "sh_degree must be in [0,3]");
    LCGS_REQUIRE(cfg->step >= 1, "step counts from 1");
    LCGS_REQUIRE(cfg->beta1 >= 0.0f && cfg->beta1 < 1.0f && cfg->beta2 >= 0.0f && cfg->beta2 < 1.0f, "betas must be in [0,1)");
    if (num_gaussians == 0) return LCGS_OK;
    const lcgs_params* packs[4] = { raw, m, v, activated };
    for (const lcgs_params* p : packs)
        LCGS_REQUIRE(p->pos && p->scale && p->rotq && p->sh && p->opacity, "NULL device pointer in a parameter pack");
    LCGS_REQUIRE(grads->d_dL_dpos && grads->d_dL_dscale && grads->d_dL_drotq && grads->d_dL_dsh && grads->d_dL_dopacity,
                 "NULL gradient pointer");
    LCGS_HIP_CHECK(hipSetDevice(ctx->device));
    const uint32_t* row_list = nullptr;
    const uint32_t* d_rows   = nullptr;
    int64_t         hint     = num_gaussians;
    if (cfg->visible_only) {
        LCGS_REQUIRE(ctx->last.valid && ctx->P == num_gaussians,
                     "visible_only needs a forward frame of this scene in this context");
        row_list = ctx->vis_index.as<uint32_t>();
        d_rows   = ctx->counts.as<uint32_t>(); // [0] = survivors of the last frame
        hint     = ctx->hint_V > 0 ? std::min<int64_t>(ctx->hint_V, num_gaussians) : num_gaussians;
    }
    auto pack = [](const lcgs_params* p) { return AdamArrays{ p->pos, p->scale, p->rotq, p->sh, p->opacity }; };
    const AdamArrays g = { grads->d_dL_dpos, grads->d_dL_dscale, grads->d_dL_drotq, grads->d_dL_dsh, grads->d_dL_dopacity };
    const AdamRates  lr = { cfg->lr_pos, cfg->lr_sh_dc, cfg->lr_sh_rest, cfg->lr_opacity, cfg->lr_scale, cfg->lr_rot };
    launch_adam_step(num_gaussians, (sh_degree + 1) * (sh_degree + 1) * 3, row_list, d_rows, hint, g, pack(raw), pack(m),
                     pack(v), pack(activated), lr, cfg->beta1, cfg->beta2, cfg->eps, cfg->step, ctx->stream,
                     /*grad_compact=*/cfg->visible_only == 2);
    LCGS_HIP_CHECK(hipGetLastError());
    return LCGS_OK;
}

lcgs_status lcgs_scene_pointers(lcgs_context* ctx, int* num_gaussians, int* sh_degree, const float** d_pos,
                                const float** d_scale, const float** d_rotq, const float** d_sh, const float** d_opacity)
{
    LCGS_REQUIRE(ctx != nullptr, "ctx is NULL");
    if (num_gaussians) *num_gaussians = ctx->P;
    if (sh_degree) *sh_degree = ctx->sh_deg;
    if (d_pos) *d_pos = ctx->pos;
    if (d_scale) *d_scale = ctx->scale;
    if (d_rotq) *d_rotq = ctx->rotq;
    if (d_sh) *d_sh = ctx->sh;
    if (d_opacity) *d_opacity = ctx->opacity;
    return LCGS_OK;
}

lcgs_status lcgs_scene_use_half_sh(lcgs_context* ctx, int enable)
{
    LCGS_REQUIRE(ctx != nullptr, "ctx is NULL");
    LCGS_HIP_CHECK(hipSetDevice(ctx->device));
    if (!enable) {
        ctx->use_half_sh = false;
        return LCGS_OK;
    }
    LCGS_REQUIRE(ctx->pos != nullptr || ctx->P == 0, "no scene bound");
    LCGS_REQUIRE(ctx->sh_deg == 3, "the f16 coefficient path exists for sh_degree 3 only");
    const int64_t n = (int64_t)ctx->P * 48;
    LCGS_TRY(ctx->sh_half.ensure(std::max<size_t>((size_t)n * 2, 16)));
    launch_sh_to_half(n, ctx->sh, ctx->sh_half.as<uint16_t>(), ctx->stream);
    LCGS_HIP_CHECK(hipGetLastError());
    ctx->use_half_sh = true;
    if (ctx->twin) ctx->twin->use_half_sh = false; // the sibling of a camera batch re-binds; see below
    return LCGS_OK;
}

lcgs_status lcgs_scene_download(lcgs_context* ctx, float* h_pos, float* h_scale, float* h_rotq, float* h_sh,
                                float* h_opacity)
{
    LCGS_REQUIRE(ctx != nullptr, "ctx is NULL");
    LCGS_HIP_CHECK(hipSetDevice(ctx->device)); // multi-GPU processes: every entry point selects its device
    const size_t P    = (size_t)ctx->P;
    const size_t feat = (size_t)(ctx->sh_deg + 1) * (ctx->sh_deg + 1) * 3;
    const size_t sizes[5] = { P * 3 * 4, P * 3 * 4, P * 4 * 4, P * feat * 4, P * 4 };
    const float* src[5]   = { ctx->pos, ctx->scale, ctx->rotq, ctx->sh, ctx->opacity };
    float*       dst[5]   = { h_pos, h_scale, h_rotq, h_sh, h_opacity };
    for (int i = 0; i < 5; ++i)
        if (dst[i] && sizes[i]) LCGS_HIP_CHECK(hipMemcpyAsync(dst[i], src[i], sizes[i], hipMemcpyDeviceToHost, ctx->stream));
    LCGS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return LCGS_OK;
}

// Scene ingest with the de-interleave and the activations on the device.  The vertex records go
// to the GPU exactly as they lie in the file -- mmap -> two pinned staging buffers filled by host threads ->
// async copies -- and k_ply_activate turns each chunk into the five activated arrays while the next chunk is in
// flight.  The host never touches a float: no 62 column vectors, no scalar activation loops
// (app/gaussians.cpp:93-168), no second 1.45 GB host copy.
lcgs_status lcgs_scene_load_ply(lcgs_context* ctx, const char* path, int* num_gaussians)
{
    LCGS_REQUIRE(ctx != nullptr && path != nullptr, "NULL argument");
    if (num_gaussians) *num_gaussians = 0;
    LCGS_HIP_CHECK(hipSetDevice(ctx->device));
    PlyProbe probe;
    LCGS_TRY(ply_probe(path, &probe));
    if (!probe.device_ok) { // ascii, or non-float columns: the general host parser
        lcgs_scene_host h;
        LCGS_TRY(lcgs_ply_read(path, &h));
        lcgs_status s = lcgs_scene_upload(ctx, h.num_gaussians, h.sh_degree, h.pos, h.scale, h.rotq, h.feature, h.opacity);
        if (num_gaussians) *num_gaussians = h.num_gaussians;
        lcgs_scene_host_free(&h);
        return s; // (lcgs_scene_upload applied the ingest order)
    }
    const int64_t N = probe.num_vertices;
    LCGS_REQUIRE(N < (1 << 30), "too many vertices");
    ctx->perm_for_owned = false; // owned[] is rewritten in the file's order
    const size_t sizes[5] = { (size_t)N * 3 * 4, (size_t)N * 3 * 4, (size_t)N * 4 * 4, (size_t)N * 48 * 4, (size_t)N * 4 };
    for (int i = 0; i < 5; ++i) LCGS_TRY(ctx->owned[i].ensure(std::max<size_t>(sizes[i], 16)));
    if (N > 0) {
        const int fd = open(path, O_RDONLY);
        if (fd < 0) {
            set_last_error(std::string("cannot open ") + path);
            return LCGS_ERR_IO;
        }
        const size_t map_bytes = probe.payload_offset + (size_t)N * probe.stride;
        void*        map       = mmap(nullptr, map_bytes, PROT_READ, MAP_PRIVATE, fd, 0);
        close(fd);
        if (map == MAP_FAILED) {
            set_last_error(std::string("cannot map ") + path);
            return LCGS_ERR_IO;
        }
        (void)madvise(map, map_bytes, MADV_SEQUENTIAL);
        const unsigned char* payload = static_cast<const unsigned char*>(map) + probe.payload_offset;
        const int64_t        chunk   = std::max<int64_t>(1, ((int64_t)64 << 20) / (int64_t)probe.stride); // records
        const size_t         cbytes  = (size_t)chunk * probe.stride;
        unsigned char*       pinned[2] = { nullptr, nullptr };
        hipEvent_t           done[2]   = { nullptr, nullptr };
        DeviceBuffer         d_raw[2];
        lcgs_status          st = LCGS_OK;
        auto                 cleanup = [&]() {
            for (int i = 0; i < 2; ++i) {
                if (pinned[i]) (void)hipHostFree(pinned[i]);
                if (done[i]) (void)hipEventDestroy(done[i]);
                d_raw[i].release();
            }
            munmap(map, map_bytes);
        };
        for (int i = 0; i < 2 && st == LCGS_OK; ++i) {
            if (hipHostMalloc(reinterpret_cast<void**>(&pinned[i]), cbytes, hipHostMallocDefault) != hipSuccess ||
                hipEventCreateWithFlags(&done[i], hipEventDisableTiming) != hipSuccess)
                st = LCGS_ERR_OUT_OF_MEMORY;
            else
                st = d_raw[i].ensure(cbytes);
        }
        PlyColumns cols;
        for (int w = 0; w < 59; ++w) cols.offset[w] = probe.column_offset[w];
        int b = 0;
        for (int64_t first = 0; first < N && st == LCGS_OK; first += chunk, b ^= 1) {
            const int64_t count = std::min<int64_t>(chunk, N - first);
            const size_t  bytes = (size_t)count * probe.stride;
            if (first >= 2 * chunk && hipEventSynchronize(done[b]) != hipSuccess) st = LCGS_ERR_HIP; // buffer b is free again
            if (st != LCGS_OK) break;
            // page cache -> pinned memory with a few threads (one memcpy stream tops out well below PCIe rate)
            const unsigned char* src = payload + (size_t)first * probe.stride;
            const int            nt  = 8;
            std::vector<std::thread> th;
            for (int t = 0; t < nt; ++t) {
                const size_t a = bytes * t / nt, e = bytes * (t + 1) / nt;
                th.emplace_back([=] { memcpy(pinned[b] + a, src + a, e - a); });
            }
            for (auto& x : th) x.join();
            if (hipMemcpyAsync(d_raw[b].ptr, pinned[b], bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) st = LCGS_ERR_HIP;
            launch_ply_activate(d_raw[b].as<unsigned char>(), first, count, (uint32_t)probe.stride, cols,
                                ctx->owned[0].as<float>(), ctx->owned[1].as<float>(), ctx->owned[2].as<float>(),
                                ctx->owned[3].as<float>(), ctx->owned[4].as<float>(), ctx->stream);
            if (hipEventRecord(done[b], ctx->stream) != hipSuccess) st = LCGS_ERR_HIP;
        }
        if (hipStreamSynchronize(ctx->stream) != hipSuccess && st == LCGS_OK) st = LCGS_ERR_HIP;
        cleanup();
        if (st != LCGS_OK) {
            if (st == LCGS_ERR_HIP) set_last_error("device ingest of the PLY payload failed");
            return st;
        }
    }
    if (num_gaussians) *num_gaussians = (int)N;
    LCGS_TRY(lcgs_scene_bind(ctx, (int)N, 3, ctx->owned[0].as<float>(), ctx->owned[1].as<float>(), ctx->owned[2].as<float>(),
                             ctx->owned[3].as<float>(), ctx->owned[4].as<float>()));
    // a scene the context owns is kept in spatial order unless the caller asked for the file's (lcgs_set_ingest_order)
    if (ctx->ingest_order == LCGS_ORDER_SPATIAL && N > 0) LCGS_TRY(lcgs_scene_reorder_spatial(ctx, nullptr));
    return LCGS_OK;
}

lcgs_status lcgs_render_forward(lcgs_context* ctx, const lcgs_camera* camera, const float bg_color[3],
                                float scale_modifier, float* d_img, int32_t* d_radii, int keep_state,
                                int* num_rendered)
{
    LCGS_REQUIRE(ctx != nullptr, "ctx is NULL");
    LCGS_HIP_CHECK(hipSetDevice(ctx->device)); // multi-GPU processes: every entry point selects its device
    LCGS_TRY(check_camera(camera));
    LCGS_REQUIRE(bg_color != nullptr, "bg_color is NULL");
    LCGS_REQUIRE(d_img != nullptr, "d_img is NULL");
    if (num_rendered) *num_rendered = 0;
    if (ctx->P == 0) return LCGS_OK; // nothing to draw: image untouched, like gs_tile_splatter/impl.cpp:109
    LCGS_REQUIRE(ctx->pos != nullptr, "no scene bound (call lcgs_scene_bind / lcgs_scene_upload first)");
    CamParams cp      = make_cam_params(*camera);
    cp.lod_min_radius = ctx->lod_min_radius;
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
            key.img = d_img; key.radii = d_radii; key.P = ctx->P; key.sh_deg = ctx->sh_deg;
            key.width = camera->width; key.height = camera->height; key.keep_state = keep_state != 0;
            key.hint_V = ctx->hint_V; key.hint_L = ctx->hint_L; key.capacity = ctx->pair_capacity; key.stream = ctx->stream;
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
        if ((int64_t)ctx->h_counts[4] > ctx->hint_L || (int64_t)ctx->h_counts[4] * 2 < ctx->hint_L)
            ctx->hint_L = (int64_t)ctx->h_counts[4] + ctx->h_counts[4] / 4 + 4096;
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

namespace
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
        t->pair_capacity = std::max(t->pair_capacity, ctx->pair_capacity);
        LCGS_HIP_CHECK(hipEventRecord(ctx->ev_batch_fork, ctx->stream));
        LCGS_HIP_CHECK(hipStreamWaitEvent(ctx->twin_stream, ctx->ev_batch_fork, 0));
    }
    return LCGS_OK;
}
} // namespace

extern "C" {

lcgs_status lcgs_fit_views(lcgs_context* ctx, int num_views, const lcgs_camera* cameras, const float bg_color[3],
                           float scale_modifier, const float* const* d_targets, const lcgs_grads* grads, float* d_losses)
{
    LCGS_REQUIRE(ctx != nullptr, "ctx is NULL");
    LCGS_REQUIRE(num_views >= 0, "num_views is negative");
    if (num_views == 0) return LCGS_OK;
    LCGS_REQUIRE(cameras && d_targets && grads && d_losses && bg_color, "NULL argument");
    LCGS_REQUIRE(ctx->P > 0 && ctx->pos != nullptr, "no scene bound");
    LCGS_HIP_CHECK(hipSetDevice(ctx->device));
    const bool two = num_views > 1 && !ctx->profiling && !ctx->use_graph;
    if (two) LCGS_TRY(prepare_twin(ctx));
    InFlight      in_flight(ctx, two);
    lcgs_context* prev = nullptr; // the context whose backward wrote `grads` last
    for (int j = 0; j < num_views; ++j) {
        // alternate, ending on `ctx`: the last backward is the one a gradient all-reduce overlaps (its slices)
        lcgs_context* c = (two && ((num_views - 1 - j) & 1)) ? ctx->twin : ctx;
        LCGS_REQUIRE(d_targets[j] != nullptr, "NULL target image in the batch");
        LCGS_TRY(check_camera(&cameras[j]));
        const size_t img_bytes = (size_t)cameras[j].width * cameras[j].height * 3 * sizeof(float);
        LCGS_TRY(c->fit_img.ensure(img_bytes));
        LCGS_TRY(c->fit_dL.ensure(img_bytes));
        if (!c->ev_fit_bwd) LCGS_HIP_CHECK(hipEventCreateWithFlags(&c->ev_fit_bwd, hipEventDisableTiming));
        LCGS_TRY(lcgs_render_forward(c, &cameras[j], bg_color, scale_modifier, c->fit_img.as<float>(), nullptr, 1, nullptr));
        LCGS_TRY(lcgs_l2_loss_backward(c, cameras[j].width, cameras[j].height, c->fit_img.as<float>(), d_targets[j],
                                       c->fit_dL.as<float>(), d_losses + j));
        // the gradient arrays are shared: this view's backward after the previous view's (on the other context)
        if (prev && prev != c) LCGS_HIP_CHECK(hipStreamWaitEvent(c->stream, prev->ev_fit_bwd, 0));
        LCGS_TRY(j == 0 ? lcgs_render_backward(c, c->fit_dL.as<float>(), grads)
                        : lcgs_render_backward_accumulate(c, c->fit_dL.as<float>(), grads));
        LCGS_HIP_CHECK(hipEventRecord(c->ev_fit_bwd, c->stream));
        prev = c;
    }
    if (two) { // (the last view ran on ctx; the one before it on the sibling, and ctx's backward already waited for it)
        LCGS_HIP_CHECK(hipEventRecord(ctx->ev_batch_join, ctx->twin_stream));
        LCGS_HIP_CHECK(hipStreamWaitEvent(ctx->stream, ctx->ev_batch_join, 0));
    }
    return LCGS_OK;
}

lcgs_status lcgs_set_profiling(lcgs_context* ctx, int enabled)
{
    LCGS_REQUIRE(ctx != nullptr, "ctx is NULL");
    ctx->profiling = enabled != 0;
    return LCGS_OK;
}

lcgs_status lcgs_get_stage_times(lcgs_context* ctx, lcgs_stage_times* out)
{
    LCGS_REQUIRE(ctx && out, "NULL argument");
    *out = ctx->times;
    return LCGS_OK;
}

lcgs_status lcgs_get_frame_stats(lcgs_context* ctx, lcgs_frame_stats* out)
{
    LCGS_REQUIRE(ctx && out, "NULL argument");
    LCGS_REQUIRE(ctx->last.valid, "no frame rendered yet");
    LCGS_TRY(sync_frame(ctx));
    ctx->stats.num_gaussians = ctx->P;
    ctx->stats.num_visible   = ctx->h_counts[0];
    ctx->stats.num_rendered  = ctx->h_counts[1];
    ctx->stats.num_pairs     = ctx->h_counts[2];
    ctx->stats.num_tiles     = (int64_t)ctx->last.cp.grid_x * ctx->last.cp.grid_y;
    ctx->stats.equal_depth_unresolved = ctx->perm_valid ? ctx->h_counts[9] : 0;
    *out                     = ctx->stats;
    return LCGS_OK;
}

// Debug/parity hook: the sorted per-tile lists of the last fused frame in ORIGINAL splat indices (what the
// reference's point_list holds) and the tile ranges.  d_list must hold num_pairs entries, d_ranges 2*G.
lcgs_status lcgs_debug_last_lists(lcgs_context* ctx, uint32_t* d_list, uint32_t* d_ranges)
{
    LCGS_REQUIRE(ctx && ctx->last.valid, "no frame rendered yet");
    LCGS_HIP_CHECK(hipSetDevice(ctx->device)); // multi-GPU processes: every entry point selects its device
    LCGS_TRY(sync_frame(ctx));
    const uint32_t L = ctx->h_counts[2];
    if (d_list && L)
        launch_map_to_index(L, ctx->counts.as<uint32_t>(), ctx->pairv[ctx->last.list_buf].as<uint32_t>(),
                            ctx->vis_index.as<uint32_t>(), d_list, ctx->stream);
    if (d_ranges)
        LCGS_HIP_CHECK(hipMemcpyAsync(d_ranges, ctx->ranges,
                                      (size_t)ctx->last.cp.grid_x * ctx->last.cp.grid_y * 8, hipMemcpyDeviceToDevice,
                                      ctx->stream));
    LCGS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return LCGS_OK;
}

// Debug/parity hook: the compositing loop's exp on its own (gs_math.hpp::blend_exp).
lcgs_status lcgs_debug_blend_exp(lcgs_context* ctx, const float* d_x, float* d_out, int64_t n)
{
    LCGS_REQUIRE(ctx && (n == 0 || (d_x && d_out)) && n >= 0, "null pointer / negative count");
    LCGS_HIP_CHECK(hipSetDevice(ctx->device));
    launch_blend_exp(d_x, d_out, n, ctx->stream);
    LCGS_HIP_CHECK(hipGetLastError());
    LCGS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return LCGS_OK;
}

namespace
{
// the optimiser folded into the per-splat pass (lcgs_render_backward_adam): no gradient arrays at all
struct FusedAdam {
    AdamArrays raw, m, v, act;
    AdamRates  lr;
    AdamStep   step;
};
lcgs_status render_backward(lcgs_context* ctx, const float* d_dL_dimg, const lcgs_grads* grads, bool compact,
                            bool accumulate = false, const FusedAdam* fused = nullptr);
}

lcgs_status lcgs_render_backward(lcgs_context* ctx, const float* d_dL_dimg, const lcgs_grads* grads)
{
    return render_backward(ctx, d_dL_dimg, grads, /*compact=*/false);
}

lcgs_status lcgs_render_backward_compact(lcgs_context* ctx, const float* d_dL_dimg, const lcgs_grads* grads)
{
    return render_backward(ctx, d_dL_dimg, grads, /*compact=*/true);
}

lcgs_status lcgs_render_backward_accumulate(lcgs_context* ctx, const float* d_dL_dimg, const lcgs_grads* grads)
{
    return render_backward(ctx, d_dL_dimg, grads, /*compact=*/false, /*accumulate=*/true);
}

lcgs_status lcgs_render_backward_adam(lcgs_context* ctx, const float* d_dL_dimg, int num_gaussians, int sh_degree,
                                      const lcgs_adam_config* cfg, const lcgs_params* raw, const lcgs_params* m,
                                      const lcgs_params* v, const lcgs_params* activated)
{
    LCGS_REQUIRE(ctx && d_dL_dimg && cfg && raw && m && v && activated, "NULL argument");
    LCGS_REQUIRE(num_gaussians == ctx->P && sh_degree == ctx->sh_deg, "num_gaussians / sh_degree must be the bound scene's");
    LCGS_REQUIRE(cfg->step >= 1, "step counts from 1");
    LCGS_REQUIRE(cfg->beta1 >= 0.0f && cfg->beta1 < 1.0f && cfg->beta2 >= 0.0f && cfg->beta2 < 1.0f, "betas must be in [0,1)");
    const lcgs_params* packs[4] = { raw, m, v, activated };
    for (const lcgs_params* p : packs)
        LCGS_REQUIRE(p->pos && p->scale && p->rotq && p->sh && p->opacity, "NULL device pointer in a parameter pack");
    auto aligned16 = [](const lcgs_params* p) {
        return ((reinterpret_cast<uintptr_t>(p->rotq) | reinterpret_cast<uintptr_t>(p->sh)) & 15) == 0;
    };
    const bool fusable = ctx->sh_deg == 3 && ctx->last.valid && ctx->last.has_state && ctx->last_has_jac && aligned16(raw) &&
                         aligned16(m) && aligned16(v) && aligned16(activated);
    if (!fusable) {
        // other SH degrees, frames without the kept colour Jacobian, unaligned rows: the same step as two calls on
        // context-owned compact gradient rows (identical result; the fused kernel exists for the degree-3 training case)
        LCGS_HIP_CHECK(hipSetDevice(ctx->device));
        const size_t feat = (size_t)(ctx->sh_deg + 1) * (ctx->sh_deg + 1) * 3;
        const size_t rows = (size_t)ctx->P;
        auto         al   = [](size_t x) { return (x + 3) & ~(size_t)3; }; // every array starts on a 16-byte boundary
        const size_t o_scale = al(rows * 3), o_rotq = al(o_scale + rows * 3), o_sh = al(o_rotq + rows * 4),
                     o_op = al(o_sh + rows * feat);
        LCGS_TRY(ctx->fused_grads.ensure((o_op + rows) * 4));
        float*     g  = ctx->fused_grads.as<float>();
        lcgs_grads gr = { g, g + o_scale, g + o_rotq, g + o_sh, g + o_op };
        LCGS_TRY(render_backward(ctx, d_dL_dimg, &gr, /*compact=*/true));
        lcgs_adam_config c2 = *cfg;
        c2.visible_only     = 2;
        return lcgs_adam_step(ctx, num_gaussians, sh_degree, &c2, &gr, raw, m, v, activated);
    }
    auto      pack = [](const lcgs_params* p) { return AdamArrays{ p->pos, p->scale, p->rotq, p->sh, p->opacity }; };
    FusedAdam fa   = { pack(raw), pack(m), pack(v), pack(activated),
                       { cfg->lr_pos, cfg->lr_sh_dc, cfg->lr_sh_rest, cfg->lr_opacity, cfg->lr_scale, cfg->lr_rot },
                       make_adam_step(cfg->beta1, cfg->beta2, cfg->eps, cfg->step) };
    lcgs_grads none{};
    return render_backward(ctx, d_dL_dimg, &none, /*compact=*/true, /*accumulate=*/false, &fa);
}

lcgs_status lcgs_visible_rows(lcgs_context* ctx, const uint32_t** d_rows, const uint32_t** d_count)
{
    LCGS_REQUIRE(ctx && d_rows && d_count, "NULL argument");
    LCGS_REQUIRE(ctx->last.valid, "no frame rendered yet");
    *d_rows  = ctx->vis_index.as<uint32_t>();
    *d_count = ctx->counts.as<uint32_t>(); // [0] = on-screen splats of the last frame
    return LCGS_OK;
}

} // extern "C"

namespace
{
lcgs_status render_backward(lcgs_context* ctx, const float* d_dL_dimg, const lcgs_grads* grads, bool compact,
                            bool accumulate, const FusedAdam* fused)
{
    LCGS_REQUIRE(ctx && d_dL_dimg && grads, "NULL argument");
    LCGS_HIP_CHECK(hipSetDevice(ctx->device)); // multi-GPU processes: every entry point selects its device
    LCGS_REQUIRE(fused || (grads->d_dL_dpos && grads->d_dL_dscale && grads->d_dL_drotq && grads->d_dL_dsh &&
                           grads->d_dL_dopacity),
                 "NULL gradient buffer");
    if (!ctx->last.valid || !ctx->last.has_state) {
        set_last_error("lcgs_render_backward needs a preceding lcgs_render_forward(..., keep_state = 1)");
        return LCGS_ERR_STATE;
    }
    LCGS_REQUIRE((reinterpret_cast<uintptr_t>(grads->d_dL_drotq) & 15) == 0, "dL_drotq must be 16-byte aligned");
    hipStream_t  st   = ctx->stream;
    const size_t P    = (size_t)ctx->P;
    const size_t feat = (size_t)(ctx->sh_deg + 1) * (ctx->sh_deg + 1) * 3;
    ctx->n_marks      = 0;
    LCGS_TRY(mark(ctx, "begin"));
    // dense per-splat gradients: splats that did not reach the screen get exact zeros.  The 236 B/splat zero-fill
    // is pure HBM writes and independent of the render-backward: it runs on the auxiliary stream beside it.
    // (Compact rows: every row that exists is written by the preprocess-backward, nothing to clear.)
    // accumulate: the arrays hold the sum of earlier views of the batch -- no fill, the rows are added to
    const bool  overlap = !ctx->profiling && !compact && !accumulate;
    hipStream_t zs      = overlap ? ctx->aux_stream : st;
    if (overlap) {
        LCGS_HIP_CHECK(hipEventRecord(ctx->ev_fork, st));
        LCGS_HIP_CHECK(hipStreamWaitEvent(ctx->aux_stream, ctx->ev_fork, 0));
    }
    // dense rows with a communicator attached: the preprocess pass runs as splat-range slices so that the gradient
    // all-reduce (lcgs_grads_allreduce) can start on the first rows while the later ones are still being computed
    const bool sliced = !compact && ctx->grad_slices > 1 && P >= 4096;
    if (sliced) {
        LCGS_TRY(ctx->slice_bounds.ensure((lcgs::kMaxGradSlices + 1) * sizeof(uint32_t)));
        for (int k = 0; k < ctx->grad_slices; ++k)
            if (!ctx->ev_slice[k]) LCGS_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_slice[k], hipEventDisableTiming));
        launch_slice_bounds(ctx->vis_index.as<uint32_t>(), ctx->counts.as<uint32_t>(), (int64_t)P, ctx->grad_slices,
                            ctx->slice_bounds.as<uint32_t>(), zs); // (before the fill: ev_join / stream order covers it)
    }
    if (!compact && !accumulate) {
        LCGS_HIP_CHECK(hipMemsetAsync(grads->d_dL_dpos, 0, P * 3 * 4, zs));
        LCGS_HIP_CHECK(hipMemsetAsync(grads->d_dL_dscale, 0, P * 3 * 4, zs));
        LCGS_HIP_CHECK(hipMemsetAsync(grads->d_dL_drotq, 0, P * 4 * 4, zs));
        LCGS_HIP_CHECK(hipMemsetAsync(grads->d_dL_dsh, 0, P * feat * 4, zs));
        LCGS_HIP_CHECK(hipMemsetAsync(grads->d_dL_dopacity, 0, P * 4, zs));
    }
    if (overlap) LCGS_HIP_CHECK(hipEventRecord(ctx->ev_join, ctx->aux_stream));
    if (ctx->g2d_zeroed && !ctx->profiling) { // cleared during the forward (first backward of this frame only)
        LCGS_HIP_CHECK(hipStreamWaitEvent(st, ctx->ev_g2d_zero, 0));
        ctx->g2d_zeroed = false;
    } else {
        LCGS_TRY(ctx->grads2d.ensure(grads2d_bytes((int64_t)P)));
        launch_zero_grads2d(ctx->counts.as<uint32_t>(), ctx->grads2d.as<float>(), st, ctx->bwd_counter.as<uint32_t>());
        ctx->g2d_zeroed = false;
    }
    LCGS_TRY(mark(ctx, "zero_grads"));
    // another view's forward in flight (lcgs_fit_views): a bounded persistent grid leaves its sort chain room on every CU
    const int      k_bwd = ctx->persist_bwd_forced >= 0 ? ctx->persist_bwd_forced
                                                        : (ctx->frames_in_flight ? ctx->persist_bwd_in_flight : 0);
    const uint32_t bwd_wgs = (!ctx->profiling && k_bwd > 0) ? (uint32_t)(k_bwd * std::max(ctx->num_cus, 1)) : 0u;
    launch_render_backward(ctx->last.cp, ctx->last.bg, ctx->ranges, ctx->pairv[ctx->last.list_buf].as<uint32_t>(),
                           ctx->recs.as<SplatRecord>(), ctx->final_T.as<float>(), ctx->n_contrib.as<uint32_t>(),
                           d_dL_dimg, ctx->grads2d.as<float>(), ctx->last_tile_order, st,
                           render_forward_writes_strip_masks() ? ctx->strip_masks.as<uint8_t>() : nullptr,
                           ctx->counts.as<uint32_t>(), ctx->bwd_counter.as<uint32_t>(), bwd_wgs);
    LCGS_TRY(mark(ctx, "render_backward"));
    if (overlap) LCGS_HIP_CHECK(hipStreamWaitEvent(st, ctx->ev_join, 0));
    const int slices = fused ? 0 : (sliced ? ctx->grad_slices : 1);
    if (fused) // (compact, unsliced: the update is applied where the gradients are formed; nothing is written out)
        launch_preprocess_backward_adam(ctx->hint_V > 0 ? ctx->hint_V : (int64_t)P, ctx->last.cp, ctx->last.scale_modifier,
                                        ctx->pos, ctx->scale, ctx->rotq, ctx->vis_index.as<uint32_t>(),
                                        ctx->counts.as<uint32_t>(), ctx->grads2d.as<float>(), ctx->shjac.as<float4>(),
                                        fused->raw, fused->m, fused->v, fused->act, fused->lr, fused->step, st);
    for (int k = 0; k < slices; ++k) {
        launch_preprocess_backward(ctx->hint_V > 0 ? ctx->hint_V : (int64_t)P, ctx->sh_deg, ctx->last.cp,
                                   ctx->last.scale_modifier, ctx->pos, ctx->scale, ctx->rotq, ctx->sh,
                                   ctx->vis_index.as<uint32_t>(), ctx->counts.as<uint32_t>(), ctx->grads2d.as<float>(),
                                   grads->d_dL_dpos, grads->d_dL_dscale, grads->d_dL_drotq, grads->d_dL_dsh,
                                   grads->d_dL_dopacity, st, ctx->last_has_jac ? ctx->shjac.as<float4>() : nullptr, compact,
                                   sliced ? ctx->slice_bounds.as<uint32_t>() : nullptr, k, slices, accumulate);
        if (sliced) LCGS_HIP_CHECK(hipEventRecord(ctx->ev_slice[k], st));
    }
    ctx->slices_recorded = sliced ? slices : 0;
    ctx->slices_of       = sliced ? grads->d_dL_dpos : nullptr;
    // sparse exchange (opt-in, lcgs_comm_track_touched_rows): the rows this frame wrote join the step's touched set
    if (!compact && !fused && ctx->comm)
        LCGS_TRY(lcgs::comm_mark_touched(ctx->comm, ctx->vis_index.as<uint32_t>(), ctx->counts.as<uint32_t>(), (int64_t)P,
                                         ctx->hint_V, accumulate, st));
    LCGS_TRY(mark(ctx, "preprocess_backward"));
    LCGS_HIP_CHECK(hipGetLastError());
    if (ctx->profiling) {
        LCGS_HIP_CHECK(hipStreamSynchronize(st));
        LCGS_TRY(collect_marks(ctx));
    }
    return LCGS_OK;
}
} // namespace
