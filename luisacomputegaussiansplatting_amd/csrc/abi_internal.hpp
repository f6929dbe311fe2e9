// abi_internal.hpp -- what the translation units behind the C ABI share: the context (context.hpp), the small helpers of
// abi_core.cpp and the frame / sibling plumbing of abi_frame.cpp.  The C ABI itself is split by subject (round 4; one
// 1 700-line file until then), every exported symbol unchanged:
//   abi_core.cpp      errors, device buffers, context create / destroy, stream, synchronise, profiling, stats, debug hooks
//   abi_stages.cpp    the reference's three operators + the two lcpp primitives (stage level), deferred stage mode
//   abi_scene.cpp     scene bind / upload / download / PLY ingest / spatial re-order / f16 SH / LOD
//   abi_frame.cpp     the fused frame: workspace, enqueue, lcgs_render_forward, camera batches, the sibling context
//   abi_backward.cpp  lcgs_render_backward and its variants (compact rows, accumulate, fused Adam)
//   abi_train.cpp     lcgs_adam_step, lcgs_fit_views
#pragma once

#include "common.hpp"
#include "context.hpp"
#include "kernels/launch.hpp"

#define LCGS_TRY(expr)                    \
    do {                                  \
        lcgs_status _s = (expr);          \
        if (_s != LCGS_OK) return _s;     \
    } while (0)

namespace lcgs
{
namespace abi
{
inline int ceil_log2_u32(uint32_t v)
{
    int b = 0;
    while ((1ull << b) < v) ++b;
    return b;
}

// abi_core.cpp
lcgs_status mark(lcgs_context* ctx, const char* name);          // per-stage timing mark (+ LCGS_DEBUG_SYNC)
lcgs_status collect_marks(lcgs_context* ctx);
lcgs_status sync_frame(lcgs_context* ctx);                      // the context's stream + the last frame's counter read-back
lcgs_status check_frame_flags(lcgs_context* ctx);               // problems an asynchronous frame reported through its counters
lcgs_status check_camera(const lcgs_camera* cam);
// abi_stages.cpp: deferred stage mode -- run a recorded SHProcessor::process / GSProjector::forward now
lcgs_status run_deferred_sh(lcgs_context* ctx);
lcgs_status run_deferred_proj(lcgs_context* ctx);
// abi_scene.cpp: the cull pass's {position, extent bound} rows of a context-owned scene (context.hpp cull_bound)
lcgs_status refresh_cull_bound(lcgs_context* ctx);
lcgs_status build_cull_bound(lcgs_context* ctx, int P, const float* pos, const float* scale, const float* rotq);
// ... dropped when the library itself writes activated arrays that ARE the context's scene (optimiser steps): the frames
// fall back to reading position + scale + rotation until the arrays are bound again
// (EVERY live context of the process is looked at -- a second context that renders the same arrays keeps rows of its own)
void scene_arrays_written(lcgs_context* ctx, const float* pos, const float* scale, const float* rotq);
// abi_owner.cpp: lcgs_owner_render, and its variant for the ownership step that reads nothing back (comm.cpp)
struct OwnerAsyncFrame {
    OwnerSegs       segs;            // padded per-owner segments of the received rows / records
    const uint32_t* table = nullptr; // device: the all-gathered counts, [o * N + view] = owner o's rows on `view`'s screen
    uint32_t        view  = 0;
    uint32_t*       overflow = nullptr; // device word: bit 0 a segment was clipped, bit 1 the pair buffers were too small
};
lcgs_status owner_render_frame(lcgs_context* ctx, const lcgs_camera* camera, const float bg_color[3], int num_rows,
                               const uint32_t* d_rows, const float* d_records, float* d_img, int keep_state,
                               const OwnerAsyncFrame* af);
lcgs_status owner_render_backward_into(lcgs_context* ctx, const float* d_dL_dimg, float* d_grads2d, const DenseFill* fill);
lcgs_status owner_backward_rows(lcgs_context* ctx, int slot, const float* d_grads2d, const lcgs_grads* grads, int mode);
void owner_frame_settle(lcgs_context* ctx); // hints / pair capacity from the pinned counters of such a frame
// the process's live contexts (lcgs_create / lcgs_destroy), for scene_arrays_written
void registry_add(lcgs_context* ctx);
void registry_remove(lcgs_context* ctx);
// the owning thread publishes the arrays its derived rows were built from and the arrays it has bound (threading:
// context.hpp foreign_writes)
void registry_publish(lcgs_context* ctx);
// abi_frame.cpp
lcgs_status ensure_fused_workspace(lcgs_context* ctx, const CamParams& cp, bool keep_state);
lcgs_status prepare_twin(lcgs_context* ctx); // the sibling context of camera / view batches: created on first use, same scene

// marks a context and its siblings as rendering several frames at once for the duration of a batch call
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
} // namespace abi
} // namespace lcgs
