// abi_stages.cpp -- the C ABI, part 2: the reference's three operators at stage level (SHProcessor::process,
// GSProjector::forward, GSTileSplatter::forward: same stage order, same buffers), the two lcpp primitives they borrow, and
// the deferred stage mode.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <string>
#include <vector>

#include "abi_internal.hpp"
#include <atomic>
#include <chrono>
#include <thread>

using namespace lcgs;
using namespace lcgs::abi;

namespace lcgs
{
namespace abi
{
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
} // namespace abi
} // namespace lcgs

extern "C" {

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

    if (!ctx->st_scalar.ptr) { // (once: the hole word starts at zero and is only ever set to a frame's own mark)
        LCGS_TRY(ctx->st_scalar.ensure(16));
        LCGS_HIP_CHECK(hipMemsetAsync(ctx->st_scalar.ptr, 0, 16, st));
    }
    uint32_t*      d_hole    = ctx->st_scalar.as<uint32_t>();
    const uint32_t hole_mark = ++ctx->stage_serial ? ctx->stage_serial : ++ctx->stage_serial; // never 0
    // (for the sort below: the splats that claim pair slots, ascending -- flagged by the allocation pass itself, compacted
    //  before the one synchronisation)
    const size_t fbytes = sparse_flag_bytes(P);
    LCGS_TRY(ctx->st_flags.ensure(fbytes));
    LCGS_TRY(ctx->st_u32[0].ensure((size_t)sparse_flag_chunks(P) * 4 + 4));
    LCGS_TRY(ctx->st_u32[1].ensure((size_t)P * 4 + 4));
    uint8_t*  d_flags = ctx->st_flags.as<uint8_t>();
    uint32_t* d_vis   = ctx->st_u32[1].as<uint32_t>();
    uint32_t* d_nvis  = d_hole + 1;
    launch_allocate_tiles(P, cp, use_focal != 0, input->depth_features, input->means_2d, input->conic,
                          accel->tiles_touched, output->radii, st, d_hole, hole_mark, d_flags, fbytes); // impl.cpp:87-99
    LCGS_TRY(ctx->st_scan_temp.ensure(scan_temp_bytes(P)));
    launch_inclusive_sum_u32(accel->tiles_touched, accel->point_offsets, P, ctx->st_scan_temp.ptr, st); // impl.cpp:104
    // (the compaction's scan launch also carries num_rendered = point_offsets[P - 1] (impl.cpp:106) next to the two scalars)
    // the frame's one read-back (impl.cpp:106-107): num_rendered beside the two scalars of this implementation.  The
    // compaction's one-workgroup scan launch POSTS the three words and the frame's serial to pinned host memory and this thread
    // polls the serial: no copy launch, no stream synchronisation (whose wake-up alone costs tens of microseconds), and the
    // host's next launches overlap the scatter launch still running behind the scan.  (Round 4: one 12-byte copy + a
    // synchronisation; before that three 4-byte copies into pageable words, ~80 us of idle GPU per frame.)
    if (!ctx->h_stage) {
        // (coherent + mapped, asked for explicitly: the host polls this block while the stream that posts to it is still
        // running -- the hand-off must not depend on what HIP_HOST_COHERENT makes of a default allocation)
        LCGS_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_stage), 16, hipHostMallocCoherent | hipHostMallocMapped));
        memset(ctx->h_stage, 0, 16);
        if (hipHostGetDevicePointer(reinterpret_cast<void**>(&ctx->h_stage_dev), ctx->h_stage, 0) != hipSuccess) {
            (void)hipGetLastError();
            ctx->h_stage_dev = nullptr;
        }
    }
    uint32_t* box = ctx->stage_mailbox ? ctx->h_stage_dev : nullptr;
    launch_compact_flags(d_flags, P, ctx->st_u32[0].as<uint32_t>(), d_vis, d_nvis, st, accel->point_offsets + (P - 1), d_hole + 2,
                         d_hole, box, hole_mark);
    bool posted = false;
    if (box) {
        LCGS_HIP_CHECK(hipGetLastError());
        volatile uint32_t* hs = ctx->h_stage;
        // (bounded: a launch that failed never posts -- after ~2 s, or at the first error the stream reports, fall back to the
        //  synchronising read-back, which returns that error)
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t spins = 0;; ++spins) {
            if (hs[3] == hole_mark) {
                posted = true;
                break;
            }
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#else
            std::this_thread::yield();
#endif
            if ((spins & 0xFFFFu) == 0xFFFFu) {
                if (hipStreamQuery(st) != hipErrorNotReady) break; // finished (posted by now, re-checked below) or failed
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) break;
            }
        }
        if (!posted && hs[3] == hole_mark) posted = true;
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (!posted) {
        LCGS_HIP_CHECK(hipMemcpyAsync(ctx->h_stage, d_hole, 12, hipMemcpyDeviceToHost, st));
        LCGS_HIP_CHECK(hipStreamSynchronize(st));                                                    // impl.cpp:107
    }
    const uint32_t hole = ctx->h_stage[0] == hole_mark ? 1u : 0u, n_vis = ctx->h_stage[1];
    const int32_t  L    = (int32_t)ctx->h_stage[2];
    if (num_rendered) *num_rendered = L;
    if (L <= 0) return LCGS_OK; // impl.cpp:109
    // impl.cpp:147's zero-fill of the ranges, issued here: right behind the synchronisation the GPU waits for the host's first
    // launches anyway, and nothing before get_ranges touches the buffer
    LCGS_HIP_CHECK(hipMemsetAsync(accel->ranges, 0, (size_t)cp.grid_x * cp.grid_y * 2 * 4, st));
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
    // impl.cpp:135-143 sorts all 64 key bits; only 32 + ceil(log2 G) of them can differ
    const int tile_bits = std::max(1, ceil_log2_u32(cp.grid_x * cp.grid_y));
    LCGS_TRY(ctx->st_keys_tmp.ensure((size_t)L * 8));
    LCGS_TRY(ctx->st_vals_tmp.ensure((size_t)L * 4));
    LCGS_TRY(ctx->st_sort_temp.ensure(pair_sort_ws_bytes(L)));
    // (small frames: the six passes over few pairs beat the longer chain of short launches -- 0.36 M pairs: 4270 vs 3930
    //  frames/s, 2.0 M: equal, 13 M: 610 vs 710; LCGS_STAGE_SORT=literal|splats forces either, a tuning / test hook)
    const bool literal = ctx->stage_sort ? ctx->stage_sort == 1 : L < (4 << 20); // (the hook is read once, at lcgs_create)
    // The reference's unsorted pair buffers (impl.cpp:120-130).  On the sort-before-duplicate route nothing downstream reads
    // them -- they are outputs only -- so their copy runs on the auxiliary stream beside the depth sort's short, latency-bound
    // launches and is joined before the renderer (round 5: -60 us of the frame); every exit path joins (SideJoin).
    struct SideJoin {
        lcgs_context* c      = nullptr;
        bool          forked = false;
        ~SideJoin()
        {
            if (forked) (void)hipStreamWaitEvent(c->stream, c->ev_join, 0);
        }
    } side{ ctx };
    if (ctx->ev_fork) LCGS_HIP_CHECK(hipEventRecord(ctx->ev_fork, st)); // "the frame's counts are known": where the side copy may start
    const bool splat_route = !(hole || n_vis == 0 || literal);
    auto copy_unsorted = [&](hipStream_t cs) -> lcgs_status { // impl.cpp:120-130
        if (hole) { // (slots a NaN-covariance splat claims stay unwritten: the per-splat kernel skips them as shader.cpp:41-42 does)
            launch_copy_with_keys(P, cp, input->means_2d, accel->point_offsets, output->radii, input->depth_features,
                                  accel->point_list_keys_unsorted, accel->point_list_unsorted, cs);
            return LCGS_OK;
        }
        // the same pairs from workgroups that own consecutive output slots (coalesced stores): the sources are the n_vis
        // splats that claim slots, in index order, with their own inclusive offsets (= point_offsets at their rows)
        LCGS_TRY(ctx->st_offs.ensure((size_t)n_vis * 4 + 16));
        LCGS_TRY(ctx->st_win2.ensure(copy_with_keys_windows_bytes((uint32_t)L)));
        launch_gather_u32((int)n_vis, d_vis, accel->point_offsets, ctx->st_offs.as<uint32_t>(), cs);
        launch_copy_with_keys_balanced((int)n_vis, cp, input->means_2d, ctx->st_offs.as<uint32_t>(), output->radii,
                                       input->depth_features, d_vis, accel->point_list_keys_unsorted, accel->point_list_unsorted,
                                       (uint32_t)L, ctx->st_win2.as<uint32_t>(), cs);
        return LCGS_OK;
    };
    if (!splat_route) LCGS_TRY(copy_unsorted(st)); // (the literal sort reads them)
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
        // (enqueued behind the depth sort's launches: right after the frame's synchronisation the host's launch rate is what
        //  the GPU waits for, and the main chain goes first; the copy still has the whole chain to hide behind)
        if (ctx->stage_side_copy && ctx->aux_stream && ctx->ev_fork && ctx->ev_join) {
            LCGS_HIP_CHECK(hipStreamWaitEvent(ctx->aux_stream, ctx->ev_fork, 0));
            LCGS_TRY(copy_unsorted(ctx->aux_stream));
            LCGS_HIP_CHECK(hipEventRecord(ctx->ev_join, ctx->aux_stream));
            side.forked = true;
        } else {
            LCGS_TRY(copy_unsorted(st));
        }
        const uint32_t* order = which ? vb : va;
        launch_gather_u32(n, order, accel->tiles_touched, cnt, st);
        LCGS_TRY(ctx->st_scan_temp.ensure(scan_temp_bytes(n)));
        launch_inclusive_sum_u32(cnt, offs, n, ctx->st_scan_temp.ptr, st);
        LCGS_TRY(ctx->st_keys_exp.ensure((size_t)L * 8));
        LCGS_TRY(ctx->st_vals_exp.ensure((size_t)L * 4));
        LCGS_TRY(ctx->st_win.ensure(copy_with_keys_windows_bytes((uint32_t)L)));
        launch_copy_with_keys_balanced(n, cp, input->means_2d, offs, output->radii, input->depth_features, order,
                                       ctx->st_keys_exp.as<uint64_t>(), ctx->st_vals_exp.as<uint32_t>(), (uint32_t)L,
                                       ctx->st_win.as<uint32_t>(), st, /*the sorted depth keys:*/ which ? kb : ka);
        launch_pair_sort_u64_preserve(ctx->st_keys_exp.as<uint64_t>(), ctx->st_vals_exp.as<uint32_t>(), accel->point_list_keys,
                                      accel->point_list, ctx->st_keys_tmp.as<uint64_t>(), ctx->st_vals_tmp.as<uint32_t>(), L,
                                      32, 32 + tile_bits, ctx->st_sort_temp.ptr, st);
    }
    launch_get_ranges_u64(L, accel->point_list_keys, accel->ranges, st); // impl.cpp:150-156
    if (side.forked) { // the unsorted buffers are complete before anything the caller enqueues behind this call
        LCGS_HIP_CHECK(hipStreamWaitEvent(st, ctx->ev_join, 0));
        side.forked = false;
    }
    launch_render_forward_aos(cp, input->bg_color, accel->ranges, accel->point_list, input->means_2d, input->conic,
                              input->opacity_features, input->color_features, output->target_img, output->final_T,
                              output->n_contrib, st); // impl.cpp:159-174
    LCGS_HIP_CHECK(hipGetLastError());
    return LCGS_OK; // no sync, like impl.cpp:177
}

} // extern "C"
