// abi_core.cpp -- the C ABI of liblcgs_hip.so (include/lcgs_hip.h), part 1: errors, device buffers, the context and its
// streams.  No compute happens on the host; if there is no GPU lcgs_create fails with LCGS_ERR_NO_DEVICE -- there is no CPU
// fallback.  (The other parts: abi_internal.hpp.)
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <string>
#include <vector>

#include "abi_internal.hpp"

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
using namespace lcgs::abi;

namespace lcgs
{
hipStream_t context_stream(lcgs_context* ctx) { return ctx->stream; }
int         context_device(lcgs_context* ctx) { return ctx->device; }
} // namespace lcgs

namespace lcgs
{
namespace abi
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

lcgs_status check_camera(const lcgs_camera* cam)
{
    LCGS_REQUIRE(cam != nullptr, "camera is NULL");
    LCGS_REQUIRE(cam->width > 0 && cam->height > 0, "camera width/height must be positive");
    LCGS_REQUIRE(cam->width <= 65535 * 16 && cam->height <= 65535 * 16, "resolution too large");
    LCGS_REQUIRE(cam->fov > 0.0f && cam->fov < 180.0f, "camera fov must be in (0,180) degrees");
    LCGS_REQUIRE(cam->aspect_ratio > 0.0f, "camera aspect_ratio must be positive");
    return LCGS_OK;
}
} // namespace abi
} // namespace lcgs

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
    if (const char* e = getenv("LCGS_STAGE_SIDE_COPY")) ctx->stage_side_copy = e[0] != '0'; // A/B hook
    if (const char* e = getenv("LCGS_STAGE_MAILBOX")) ctx->stage_mailbox = e[0] != '0';      // A/B hook
    if (const char* e = getenv("LCGS_COARSE_LISTS")) ctx->coarse_mode = e[0] == '0' ? 0 : (e[0] == '1' ? 1 : 2); // A/B / test hook
    if (const char* e = getenv("LCGS_COARSE_KEEP")) ctx->coarse_keep = e[0] != '0';                               // A/B / test hook
    if (const char* e = getenv("LCGS_BWD_USE_MASKS")) ctx->bwd_use_masks = e[0] != '0';      // test hook: the launcher's mask-less form
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
    abi::registry_add(ctx);
    *out_ctx    = ctx;
    return LCGS_OK;
}

lcgs_status lcgs_destroy(lcgs_context* ctx)
{
    if (!ctx) return LCGS_OK;
    abi::registry_remove(ctx);
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
                             &ctx->st_vals_tmp, &ctx->st_sort_temp, &ctx->st_scan_temp, &ctx->st_scalar, &ctx->sh_half, &ctx->strip_masks, &ctx->keep_list, &ctx->keep_ranges, &ctx->shjac, &ctx->tie_ws, &ctx->fused_grads, &ctx->bwd_counter, &ctx->st_flags, &ctx->st_keys_exp, &ctx->st_vals_exp,
                             &ctx->st_u32[0], &ctx->st_u32[1], &ctx->st_u32[2], &ctx->st_u32[3], &ctx->st_u32[4], &ctx->st_u32[5], &ctx->st_u32[6], &ctx->st_u32[7],
                             &ctx->cull_bound_buf, &ctx->verify_ws, &ctx->st_win, &ctx->st_win2, &ctx->st_offs };
    for (DeviceBuffer* b : bufs) b->release();
    for (auto& s : ctx->owner)
        for (DeviceBuffer* b : { &s.vis, &s.shjac, &s.counts }) b->release();
    for (auto& l : ctx->owner_lane) {
        if (l.stream) {
            (void)hipStreamSynchronize(l.stream);
            (void)hipStreamDestroy(l.stream);
        }
        if (l.done) (void)hipEventDestroy(l.done);
        for (DeviceBuffer* b : { &l.slab, &l.chunk_info, &l.chunk_base, &l.sortk[0], &l.sortk[1], &l.sortv[0], &l.sortv[1], &l.rects,
                                 &l.sort_ws })
            b->release();
    }
    if (ctx->ev_owner_fork) (void)hipEventDestroy(ctx->ev_owner_fork);
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
    if (ctx->h_owner_counts) (void)hipHostFree(ctx->h_owner_counts);
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
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
    LCGS_REQUIRE(ctx->frame_state_valid(), "no frame rendered yet");
    LCGS_TRY(sync_frame(ctx));
    ctx->stats.num_gaussians = ctx->P;
    ctx->stats.num_visible   = ctx->h_counts[0];
    ctx->stats.num_rendered  = ctx->h_counts[1];
    ctx->stats.num_pairs     = ctx->h_counts[2];
    ctx->stats.num_tiles     = (int64_t)ctx->last.cp.grid_x * ctx->last.cp.grid_y;
    ctx->stats.equal_depth_unresolved = ctx->perm_valid ? ctx->h_counts[9] : 0;
    ctx->stats.list_shift             = ctx->last.cp.list_shift;
    *out                     = ctx->stats;
    return LCGS_OK;
}

// Debug/parity hook: the sorted per-tile lists of the last fused frame in ORIGINAL splat indices (what the
// reference's point_list holds) and the tile ranges.  d_list must hold num_pairs entries, d_ranges 2*G.
lcgs_status lcgs_debug_last_lists(lcgs_context* ctx, uint32_t* d_list, uint32_t* d_ranges)
{
    LCGS_REQUIRE(ctx && ctx->frame_state_valid(), "no frame rendered yet");
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

// The granularity of the fused frame's pair lists (context.hpp coarse_mode): per tile like the reference, per 2 x 2-tile block,
// or the library's decision from the last synchronised frame's counts (the default).
lcgs_status lcgs_set_list_policy(lcgs_context* ctx, int policy)
{
    LCGS_REQUIRE(ctx != nullptr, "ctx is NULL");
    LCGS_REQUIRE(policy == LCGS_LISTS_PER_TILE || policy == LCGS_LISTS_PER_BLOCK || policy == LCGS_LISTS_AUTO, "unknown list policy");
    for (lcgs_context* c = ctx; c; c = c->twin) c->coarse_mode = policy;
    return LCGS_OK;
}

// Debug/parity hook: what the last keep_state frame stored for its backward -- the values the reference computes and drops
// (gs_tile_splatter/shader.cpp:219-220,252,273): per pixel the final transmittance and the 1-based list position of the
// last contributor.
lcgs_status lcgs_debug_last_state(lcgs_context* ctx, float* d_final_T, uint32_t* d_n_contrib)
{
    LCGS_REQUIRE(ctx && ctx->frame_state_valid() && ctx->last.has_state, "no keep_state frame rendered yet");
    LCGS_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t n = (size_t)ctx->last.cp.width * ctx->last.cp.height;
    if (d_final_T) LCGS_HIP_CHECK(hipMemcpyAsync(d_final_T, ctx->final_T.ptr, n * 4, hipMemcpyDeviceToDevice, ctx->stream));
    if (d_n_contrib) LCGS_HIP_CHECK(hipMemcpyAsync(d_n_contrib, ctx->n_contrib.ptr, n * 4, hipMemcpyDeviceToDevice, ctx->stream));
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

} // extern "C"
