// abi_train.cpp -- the C ABI, part 6: the optimiser step (csrc/kernels/train.hip) and multi-view steps on one GPU
// (lcgs_fit_views: a view's forward beside the previous view's backward, on the context and its sibling).
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

extern "C" {

lcgs_status lcgs_adam_step(lcgs_context* ctx, int num_gaussians, int sh_degree, const lcgs_adam_config* cfg,
                           const lcgs_grads* grads, const lcgs_params* raw, const lcgs_params* m, const lcgs_params* v,
                           const lcgs_params* activated)
{
    LCGS_REQUIRE(ctx && cfg && grads && raw && m && v && activated, "NULL argument");
    LCGS_REQUIRE(num_gaussians >= 0, "num_gaussians is negative");
    LCGS_REQUIRE(sh_degree >= 0 && sh_degree <= 3, "sh_degree must be in [0,3]");
    LCGS_REQUIRE(cfg->step >= 1, "step counts from 1");
    LCGS_REQUIRE(cfg->beta1 >= 0.0f && cfg->beta1 < 1.0f && cfg->beta2 >= 0.0f && cfg->beta2 < 1.0f, "betas must be in [0,1)");
    // (before the empty-range return: a rank whose shard is empty still belongs to a step that rewrites the arrays)
    scene_arrays_written(ctx, activated->pos, activated->scale, activated->rotq); // (a context-owned scene trained in place)
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
        LCGS_REQUIRE(ctx->frame_state_valid() && ctx->P == num_gaussians,
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

} // extern "C"
