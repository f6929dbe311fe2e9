// abi_backward.cpp -- the C ABI, part 5: the backward of the fused frame -- dense per-splat rows, compact
// rows, accumulation over views, and the variant with the optimiser folded into the per-splat pass.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <string>
#include <vector>

#include "abi_internal.hpp"

using namespace lcgs;
using namespace lcgs::abi;

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

extern "C" {

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
    const bool fusable = ctx->sh_deg == 3 && ctx->frame_state_valid() && ctx->last.has_state && ctx->last_has_jac && aligned16(raw) &&
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
    scene_arrays_written(ctx, activated->pos, activated->scale, activated->rotq); // (a context-owned scene trained in place)
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
    LCGS_REQUIRE(ctx->frame_state_valid(), "no frame rendered yet");
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
    if (!ctx->frame_state_valid() || !ctx->last.has_state) {
        set_last_error("lcgs_render_backward needs a preceding lcgs_render_forward(..., keep_state = 1)");
        return LCGS_ERR_STATE;
    }
    if (ctx->owner_recs) { // the last frame was lcgs_owner_render's: its records are not this context's own
        set_last_error("the last frame was drawn from received records (lcgs_owner_render): use lcgs_owner_render_backward");
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
    // Round 4: the fill is a SIDE JOB of the render-backward kernel (backward.hip, DenseFill) -- its workgroups clear their
    // share of the five arrays with fire-and-forget stores before they turn to their tile -- instead of 0.33 ms of memset
    // kernels on the auxiliary stream, a fork and a join: 1.741 -> 1.724 ms per step (most of what the fill costs the
    // VALU-bound kernel is the memory system's either way).
    // (LCGS_BWD_FILL=aux keeps the memsets: the A/B hook; per-stage profiling keeps them too, in order, as "zero_grads".)
    static const bool fill_on_aux = [] { const char* e = getenv("LCGS_BWD_FILL"); return e && e[0] == 'a'; }();
    const bool  dense_fill = !compact && !accumulate;
    const bool  fill_in_kernel = dense_fill && !ctx->profiling && !fill_on_aux && P * feat < ((size_t)1 << 32); // (u32 lengths)
    const bool  overlap = !ctx->profiling && dense_fill && !fill_in_kernel;
    hipStream_t zs      = overlap ? ctx->aux_stream : st;
    DenseFill   fill;
    if (fill_in_kernel) {
        fill.b0 = grads->d_dL_dpos; fill.b1 = grads->d_dL_dscale; fill.b2 = grads->d_dL_drotq; fill.b3 = grads->d_dL_dsh;
        fill.b4 = grads->d_dL_dopacity;
        const size_t n[5] = { P * 3, P * 3, P * 4, P * feat, P };
        for (int a = 0; a < 5; ++a) fill.n[a] = (uint32_t)n[a];
    }
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
    if (dense_fill && !fill_in_kernel) {
        LCGS_HIP_CHECK(hipMemsetAsync(grads->d_dL_dpos, 0, P * 3 * 4, zs));
        LCGS_HIP_CHECK(hipMemsetAsync(grads->d_dL_dscale, 0, P * 3 * 4, zs));
        LCGS_HIP_CHECK(hipMemsetAsync(grads->d_dL_drotq, 0, P * 4 * 4, zs));
        LCGS_HIP_CHECK(hipMemsetAsync(grads->d_dL_dsh, 0, P * feat * 4, zs));
        LCGS_HIP_CHECK(hipMemsetAsync(grads->d_dL_dopacity, 0, P * 4, zs));
    }
    if (overlap) LCGS_HIP_CHECK(hipEventRecord(ctx->ev_join, ctx->aux_stream));
    if (ctx->g2d_zeroed && !ctx->profiling) { // cleared by the forward's renderer (first backward of this frame only)
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
    // (a frame on per-block lists left every tile a list of its own for this walk: render.hip COMPACT)
    const bool      own_lists = ctx->last.cp.list_shift != 0u;
    const uint32_t* bw_ranges = own_lists ? ctx->keep_ranges.as<uint32_t>() : ctx->ranges;
    const uint32_t* bw_list   = own_lists ? ctx->keep_list.as<uint32_t>() : ctx->pairv[ctx->last.list_buf].as<uint32_t>();
    CamParams       bw_cp     = ctx->last.cp;
    bw_cp.list_shift          = 0u;
    launch_render_backward(bw_cp, ctx->last.bg, bw_ranges, bw_list,
                           ctx->recs.as<SplatRecord>(), ctx->final_T.as<float>(), ctx->n_contrib.as<uint32_t>(),
                           d_dL_dimg, ctx->grads2d.as<float>(), ctx->last_tile_order, st,
                           render_forward_writes_strip_masks() && ctx->bwd_use_masks ? ctx->strip_masks.as<uint8_t>() : nullptr,
                           ctx->counts.as<uint32_t>(), ctx->bwd_counter.as<uint32_t>(), bwd_wgs, fill_in_kernel ? &fill : nullptr);
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
