// abi_scene.cpp -- the C ABI, part 3: the scene a context renders -- bind / upload / download, PLY ingest with the
// de-interleave and the activations on the device, the spatial (Morton) order of context-owned scenes, the opt-in f16
// coefficients and footprint cull.
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <thread>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "abi_internal.hpp"

using namespace lcgs;
using namespace lcgs::abi;

namespace lcgs
{
namespace abi
{
// The bound arrays are the context's own (lcgs_scene_upload / lcgs_scene_load_ply / lcgs_scene_reorder_spatial, or those
// arrays bound again): (re)build the 16-byte {position, extent bound} rows the cull pass's phase 1 reads.  Caller-owned
// arrays get none -- the library cannot know when they change.  LCGS_CULL_BOUND=0 switches it off (A/B hook).
lcgs_status refresh_cull_bound(lcgs_context* ctx)
{
    const bool own = ctx->P > 0 && ctx->pos == ctx->owned[0].as<float>() && ctx->scale == ctx->owned[1].as<float>() &&
                     ctx->rotq == ctx->owned[2].as<float>();
    if (!own) { // (rows of other arrays -- declared static -- stay: a frame uses them only for THOSE arrays)
        registry_publish(ctx); // (the bound arrays changed)
        return LCGS_OK;
    }
    return build_cull_bound(ctx, ctx->P, ctx->pos, ctx->scale, ctx->rotq);
}

// rows for the given arrays (on ctx->stream, like the frames that read them); LCGS_CULL_BOUND=0 switches them off (A/B hook)
lcgs_status build_cull_bound(lcgs_context* ctx, int P, const float* pos, const float* scale, const float* rotq)
{
    static const bool enabled = [] {
        const char* e = getenv("LCGS_CULL_BOUND");
        return !(e && e[0] == '0');
    }();
    ctx->cull_bound = nullptr;
    ctx->cull_key   = {};
    // (whatever another thread posted about the OLD rows is settled: a write it makes from here on posts again)
    ctx->foreign_writes.fetch_and(~lcgs_context::kRowsStale, std::memory_order_acq_rel);
    lcgs_status s = LCGS_OK;
    if (enabled && P > 0 && pos && scale && rotq) {
        s = ctx->cull_bound_buf.ensure((size_t)P * sizeof(float4));
        if (s == LCGS_OK) {
            launch_cull_bound(P, pos, scale, rotq, ctx->cull_bound_buf.as<float4>(), ctx->stream);
            if (hipGetLastError() != hipSuccess) s = LCGS_ERR_HIP;
        }
        if (s == LCGS_OK) {
            ctx->cull_bound = ctx->cull_bound_buf.as<float4>();
            ctx->cull_key   = { pos, scale, rotq, P };
        }
    }
    registry_publish(ctx);
    return s;
}

namespace
{
// What a context PUBLISHES for other threads' writers (under g_registry_mutex): the arrays its derived rows were built from
// and the arrays it has bound.  Writers compare against these copies, never against the context's own fields.
struct Published {
    lcgs_context* ctx = nullptr;
    const float * key_pos = nullptr, *key_scale = nullptr, *key_rotq = nullptr; // cull_key
    int           key_P = 0;
    const float * arr[5] = { nullptr, nullptr, nullptr, nullptr, nullptr }; // bound pos / scale / rotq / sh / opacity
    size_t        arr_floats[5] = { 0, 0, 0, 0, 0 };
};
std::mutex             g_registry_mutex;
std::vector<Published> g_registry;

Published* find_published(lcgs_context* ctx) // (mutex held)
{
    for (Published& e : g_registry)
        if (e.ctx == ctx) return &e;
    return nullptr;
}
bool same_thread_family(const lcgs_context* a, const lcgs_context* b) // b is a (or a's batch sibling, or a is b's)
{
    if (a == b) return true;
    for (const lcgs_context* t = a->twin; t; t = t->twin)
        if (t == b) return true;
    for (const lcgs_context* t = b->twin; t; t = t->twin)
        if (t == a) return true;
    return false;
}
} // namespace
void registry_add(lcgs_context* ctx)
{
    std::lock_guard<std::mutex> lock(g_registry_mutex);
    Published e;
    e.ctx = ctx;
    g_registry.push_back(e);
}
void registry_remove(lcgs_context* ctx)
{
    std::lock_guard<std::mutex> lock(g_registry_mutex);
    g_registry.erase(std::remove_if(g_registry.begin(), g_registry.end(), [&](const Published& e) { return e.ctx == ctx; }),
                     g_registry.end());
}
// the owning thread, whenever its cull_key or its bound arrays change
void registry_publish(lcgs_context* ctx)
{
    std::lock_guard<std::mutex> lock(g_registry_mutex);
    Published* e = find_published(ctx);
    if (!e) return;
    e->key_pos   = ctx->cull_bound ? ctx->cull_key.pos : nullptr;
    e->key_scale = ctx->cull_bound ? ctx->cull_key.scale : nullptr;
    e->key_rotq  = ctx->cull_bound ? ctx->cull_key.rotq : nullptr;
    e->key_P     = ctx->cull_bound ? ctx->cull_key.P : 0;
    const size_t feat = (size_t)(ctx->sh_deg + 1) * (ctx->sh_deg + 1) * 3;
    const float* arr[5]    = { ctx->pos, ctx->scale, ctx->rotq, ctx->sh, ctx->opacity };
    const size_t floats[5] = { 3, 3, 4, feat, 1 };
    for (int i = 0; i < 5; ++i) {
        e->arr[i]        = arr[i];
        e->arr_floats[i] = floats[i] * (size_t)std::max(ctx->P, 0);
    }
}

// The library is about to write (or has let someone write) position / scale / rotation rows inside the given arrays: every
// context whose derived rows were built from arrays that overlap them drops the rows -- this context, its batch siblings AND
// any other context of the process that renders the same arrays (an optimiser step issued through context A on arrays
// context B owns).  Frames fall back to reading the arrays themselves until the rows are rebuilt (bind / lcgs_scene_modified).
// Contexts of OTHER host threads are never touched: the comparison runs on what they published, and they get a bit posted
// (lcgs_context::foreign_writes) that their own thread honours at its next use of the rows.
void scene_arrays_written(lcgs_context* ctx, const float* pos, const float* scale, const float* rotq)
{
    std::lock_guard<std::mutex> lock(g_registry_mutex);
    for (Published& e : g_registry) {
        if (e.key_P <= 0) continue;
        auto inside = [&](const float* p, const float* base, size_t floats) {
            return p != nullptr && base != nullptr && p >= base && p < base + floats * (size_t)e.key_P;
        };
        if (!(inside(pos, e.key_pos, 3) || inside(scale, e.key_scale, 3) || inside(rotq, e.key_rotq, 4))) continue;
        e.key_pos = e.key_scale = e.key_rotq = nullptr;
        e.key_P                               = 0;
        if (ctx && same_thread_family(ctx, e.ctx)) { // the caller's own: dropped in place
            e.ctx->cull_bound = nullptr;
            e.ctx->cull_key   = {};
        } else {
            e.ctx->foreign_writes.fetch_or(lcgs_context::kRowsStale, std::memory_order_acq_rel);
        }
    }
}

// lcgs_scene_modified: the arrays ctx has bound were written behind the library's back.  Every OTHER context that has any of
// them bound gets told (its f16 coefficient copy and the kept state of its last frame are stale too); ctx itself and its
// siblings are handled by the caller.
void scene_modified_elsewhere(lcgs_context* ctx)
{
    const size_t feat       = (size_t)(ctx->sh_deg + 1) * (ctx->sh_deg + 1) * 3;
    const float* arr[5]     = { ctx->pos, ctx->scale, ctx->rotq, ctx->sh, ctx->opacity };
    const size_t floats[5]  = { 3, 3, 4, feat, 1 };
    std::lock_guard<std::mutex> lock(g_registry_mutex);
    for (Published& e : g_registry) {
        if (same_thread_family(ctx, e.ctx)) continue;
        bool overlap = false;
        for (int i = 0; i < 5 && !overlap; ++i) {
            const float* a0 = arr[i];
            const float* a1 = a0 ? a0 + floats[i] * (size_t)std::max(ctx->P, 0) : nullptr;
            for (int j = 0; j < 5 && !overlap; ++j) {
                const float* b0 = e.arr[j];
                const float* b1 = b0 ? b0 + e.arr_floats[j] : nullptr;
                overlap = a0 && b0 && a0 < b1 && b0 < a1;
            }
        }
        if (overlap) e.ctx->foreign_writes.fetch_or(lcgs_context::kSceneModified, std::memory_order_acq_rel);
    }
}
} // namespace abi
} // namespace lcgs

extern "C" {

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
    return refresh_cull_bound(ctx); // (the context's own arrays only; ordered on ctx->stream like the frames that read it)
}

lcgs_status lcgs_scene_declare_static(lcgs_context* ctx, int num_gaussians, const float* d_pos, const float* d_scale,
                                      const float* d_rotq)
{
    LCGS_REQUIRE(ctx != nullptr, "ctx is NULL");
    LCGS_REQUIRE(num_gaussians >= 0 && num_gaussians < (1 << 30), "num_gaussians out of range");
    LCGS_HIP_CHECK(hipSetDevice(ctx->device));
    if (num_gaussians == 0 || !d_pos) { // withdraw the declaration
        ctx->cull_bound = nullptr;
        ctx->cull_key   = {};
        registry_publish(ctx);
        return refresh_cull_bound(ctx); // (a context-owned scene that is bound keeps its own rows)
    }
    LCGS_REQUIRE(d_scale && d_rotq, "NULL device pointer");
    LCGS_REQUIRE((reinterpret_cast<uintptr_t>(d_rotq) & 15) == 0, "rotq must be 16-byte aligned");
    return build_cull_bound(ctx, num_gaussians, d_pos, d_scale, d_rotq);
}

lcgs_status lcgs_scene_modified(lcgs_context* ctx)
{
    LCGS_REQUIRE(ctx != nullptr, "ctx is NULL");
    LCGS_HIP_CHECK(hipSetDevice(ctx->device));
    // whoever keeps rows derived from the bound arrays (this context, its siblings, another context on the same arrays)
    // drops them; this context builds its own again (context-owned arrays only, ordered on its stream behind the
    // caller's writes as far as the caller ordered those before this call)
    scene_arrays_written(ctx, ctx->pos, ctx->scale, ctx->rotq);
    scene_modified_elsewhere(ctx); // other threads' contexts on these arrays: told, they act at their next frame / backward
    for (lcgs_context* c = ctx; c; c = c->twin) {
        c->use_half_sh = false; // the f16 copy of the coefficients (if any) is stale as well
        c->last.valid  = false; // ... and so is the kept state of the last frame
    }
    return refresh_cull_bound(ctx);
}

lcgs_status lcgs_debug_verify_derived(lcgs_context* ctx, int64_t* stale_rows)
{
    LCGS_REQUIRE(ctx && stale_rows, "NULL argument");
    LCGS_HIP_CHECK(hipSetDevice(ctx->device));
    *stale_rows = 0;
    const float4* rows = ctx->cull_rows();
    if (!rows || ctx->P <= 0) return LCGS_OK; // nothing derived is in use for the bound arrays
    LCGS_TRY(ctx->verify_ws.ensure(8));
    LCGS_HIP_CHECK(hipMemsetAsync(ctx->verify_ws.ptr, 0, 8, ctx->stream));
    launch_cull_bound_verify(ctx->P, ctx->pos, ctx->scale, ctx->rotq, rows, ctx->verify_ws.as<unsigned long long>(), ctx->stream);
    LCGS_HIP_CHECK(hipGetLastError());
    unsigned long long n = 0;
    LCGS_HIP_CHECK(hipMemcpyAsync(&n, ctx->verify_ws.ptr, 8, hipMemcpyDeviceToHost, ctx->stream));
    LCGS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    *stale_rows = (int64_t)n;
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

} // extern "C"
