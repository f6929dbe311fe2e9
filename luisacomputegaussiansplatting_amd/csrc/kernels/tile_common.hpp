// tile_common.hpp -- device helpers the forward renderer (render.hip) and the render-backward (backward.hip) must
// share: the workgroup -> tile map, and the conservative "can this splat reach the rect" test.  The backward re-uses the
// forward's per-strip reach masks (strip_masks), so both kernels have to cull identically: one definition, here.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gs_math.hpp"

namespace lcgs
{
namespace tile
{

constexpr int kXcd = 8;

// Conservative "can this splat reach any pixel of the rect?" test, exact up to a safety margin.
// A pixel receives a contribution only if alpha = min(0.99, o * exp(power)) >= 1/255 with power <= 0
// (shader.cpp:256-259), i.e. q(d) = ca dx^2 + 2 cb dx dy + cc dy^2 <= t = 2 ln(255 o).  The minimum of the convex
// quadratic q over the pixel rect [x0,x1] x [y0,y1] is 0 if the mean is inside, else it lies on one of the four
// edges (a clamped 1-D quadratic each).  `t` already carries its margin; the absolute rounding slack scales with
// the magnitude of the terms that cancel in q.  Non-finite inputs keep the entry.
__device__ __forceinline__ bool splat_may_touch_rect(float mx, float my, float ca, float cb, float cc, float t,
                                                     float x0, float y0, float x1, float y1)
{
    if (!(t > 0.0f)) return false; // opacity <= 1/255: alpha < 1/255 everywhere
    const float ex0 = x0 - mx, ex1 = x1 - mx, ey0 = y0 - my, ey1 = y1 - my; // rect relative to the mean
    if (ex0 <= 0.0f && ex1 >= 0.0f && ey0 <= 0.0f && ey1 >= 0.0f) return true;
    if (!(ca > 0.0f) || !(cc > 0.0f)) return true; // degenerate conic: keep
    // 1-ulp hardware reciprocals are enough: a minimiser that is off by delta raises q by cc*delta^2 (or ca*delta^2),
    // ~1e-14 relative -- nine orders below the slack applied at the end.
    const float nb_cc = -cb * __builtin_amdgcn_rcpf(cc), nb_ca = -cb * __builtin_amdgcn_rcpf(ca);
    float best = 3.0e38f, slack = 0.0f;
    // vertical edges dx = ex0 / ex1: dy* = -cb dx / cc clamped to [ey0, ey1]
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const float dx = s ? ex1 : ex0;
        const float dy = fmin_(fmax_(nb_cc * dx, ey0), ey1);
        const float q1 = ca * dx * dx, q2 = 2.0f * cb * dx * dy, q3 = cc * dy * dy;
        const float q  = q1 + q2 + q3;
        if (q < best) {
            best  = q;
            slack = fabsf(q1) + fabsf(q2) + fabsf(q3);
        }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const float dy = s ? ey1 : ey0;
        const float dx = fmin_(fmax_(nb_ca * dy, ex0), ex1);
        const float q1 = ca * dx * dx, q2 = 2.0f * cb * dx * dy, q3 = cc * dy * dy;
        const float q  = q1 + q2 + q3;
        if (q < best) {
            best  = q;
            slack = fabsf(q1) + fabsf(q2) + fabsf(q3);
        }
    }
    if (!(best == best)) return true; // NaN: keep
    return best - 1e-5f * slack <= t;
}

// The same test for the FOUR 16x4 strips of a tile at once (rows y0 + 4k .. y0 + 4k + 3, k = 0..3; columns x0 .. x1): bit k of
// the result = "the splat may reach strip k".  What the renderer's staging pays per list entry (round 5: it was four calls of
// the function above, ~350 vector instructions per lane and round -- a fifth of the forward renderer's instruction stream).
// Two economies, both exact:
//  * q is a positive-definite form with its minimum (0) at the mean, so it increases along every ray from the mean, and the
//    minimum over a rect that does not contain the mean is attained on an edge that FACES the mean: the left edge if the
//    mean lies left of the rect, the right one if right of it, neither if its x is inside the rect's x range; likewise in y.
//    One vertical and one horizontal edge per strip instead of two and two (evaluating an edge that does not face the mean
//    only adds a larger candidate: harmless).  The form's definiteness is checked (ca cc - cb^2 beyond rounding), else the
//    entry is kept for all four strips like any degenerate conic.
//  * the strips share their x range: the facing vertical edge, its minimiser's slope and its dx terms are computed once.
// Same margin (`t` carries its own; 1e-5 of the cancelling terms) and the same treatment of non-finite inputs as above.
__device__ __forceinline__ uint32_t splat_strip_mask(float mx, float my, float ca, float cb, float cc, float t, float x0,
                                                     float y0, float x1)
{
    if (!(t > 0.0f)) return 0u; // opacity <= 1/255: alpha < 1/255 everywhere
    const float det = ca * cc - cb * cb;
    if (!(ca > 0.0f) || !(cc > 0.0f) || !(det > 1e-5f * (ca * cc))) return 0xFu; // degenerate / indefinite / NaN conic: keep
    const float ex0 = x0 - mx, ex1 = x1 - mx;
    const bool  in_x = ex0 <= 0.0f && ex1 >= 0.0f;
    // 1-ulp hardware reciprocals are enough (see above)
    const float nb_cc = -cb * __builtin_amdgcn_rcpf(cc), nb_ca = -cb * __builtin_amdgcn_rcpf(ca);
    const float dxv   = ex0 > 0.0f ? ex0 : ex1; // the vertical edge that faces the mean (either, if the mean's x is inside)
    const float dyv_f = nb_cc * dxv;            // its free minimiser
    const float v1 = ca * dxv * dxv, cb2dx = 2.0f * cb * dxv;
    uint32_t    mask = 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float ey0 = (y0 + 4.0f * k) - my, ey1 = (y0 + 4.0f * k + 3.0f) - my;
        // the vertical edge: dy* clamped to the strip's rows
        const float dyv = fmin_(fmax_(dyv_f, ey0), ey1);
        const float v2 = cb2dx * dyv, v3 = cc * dyv * dyv;
        const float qv = v1 + v2 + v3, sv = fabsf(v1) + fabsf(v2) + fabsf(v3);
        // the horizontal edge that faces the mean: dx* clamped to the tile's columns
        const float dyh = ey0 > 0.0f ? ey0 : ey1;
        const float dxh = fmin_(fmax_(nb_ca * dyh, ex0), ex1);
        const float h1 = ca * dxh * dxh, h2 = 2.0f * cb * dxh * dyh, h3 = cc * dyh * dyh;
        const float qh = h1 + h2 + h3, sh = fabsf(h1) + fabsf(h2) + fabsf(h3);
        const bool  hv   = qh < qv;
        const float best = hv ? qh : qv, slack = hv ? sh : sv;
        const bool  inside = in_x && ey0 <= 0.0f && ey1 >= 0.0f;
        const bool  reach  = inside || !(best == best) || best - 1e-5f * slack <= t; // (NaN: keep)
        mask |= reach ? (1u << k) : 0u;
    }
    return mask;
}

// The same for the four 8x8 QUADRANTS of a tile (bit k: columns x0 + 8 (k & 1) .. + 7, rows y0 + 8 (k >> 1) .. + 7): per quadrant
// one facing vertical and one facing horizontal edge, the vertical edge's terms shared by the two quadrants of a column pair.
__device__ __forceinline__ uint32_t splat_quad_mask(float mx, float my, float ca, float cb, float cc, float t, float x0, float y0)
{
    if (!(t > 0.0f)) return 0u; // opacity <= 1/255: alpha < 1/255 everywhere
    const float det = ca * cc - cb * cb;
    if (!(ca > 0.0f) || !(cc > 0.0f) || !(det > 1e-5f * (ca * cc))) return 0xFu; // degenerate / indefinite / NaN conic: keep
    const float nb_cc = -cb * __builtin_amdgcn_rcpf(cc), nb_ca = -cb * __builtin_amdgcn_rcpf(ca);
    uint32_t    mask = 0u;
#pragma unroll
    for (int jx = 0; jx < 2; ++jx) {
        const float ex0 = (x0 + 8.0f * jx) - mx, ex1 = (x0 + 8.0f * jx + 7.0f) - mx;
        const bool  in_x = ex0 <= 0.0f && ex1 >= 0.0f;
        const float dxv   = ex0 > 0.0f ? ex0 : ex1; // the vertical edge that faces the mean (either, if the mean's x is inside)
        const float dyv_f = nb_cc * dxv;            // its free minimiser
        const float v1 = ca * dxv * dxv, cb2dx = 2.0f * cb * dxv;
#pragma unroll
        for (int jy = 0; jy < 2; ++jy) {
            const float ey0 = (y0 + 8.0f * jy) - my, ey1 = (y0 + 8.0f * jy + 7.0f) - my;
            const float dyv = fmin_(fmax_(dyv_f, ey0), ey1);
            const float v2 = cb2dx * dyv, v3 = cc * dyv * dyv;
            const float qv = v1 + v2 + v3, sv = fabsf(v1) + fabsf(v2) + fabsf(v3);
            const float dyh = ey0 > 0.0f ? ey0 : ey1;
            const float dxh = fmin_(fmax_(nb_ca * dyh, ex0), ex1);
            const float h1 = ca * dxh * dxh, h2 = 2.0f * cb * dxh * dyh, h3 = cc * dyh * dyh;
            const float qh = h1 + h2 + h3, sh = fabsf(h1) + fabsf(h2) + fabsf(h3);
            const bool  hv   = qh < qv;
            const float best = hv ? qh : qv, slack = hv ? sh : sv;
            const bool  inside = in_x && ey0 <= 0.0f && ey1 >= 0.0f;
            const bool  reach  = inside || !(best == best) || best - 1e-5f * slack <= t; // (NaN: keep)
            mask |= reach ? (1u << (2 * jy + jx)) : 0u;
        }
    }
    return mask;
}

// Which 64 pixels of a 16x16 tile wave k of a renderer workgroup owns ("strip k" in the renderers' names), and the reach test
// that goes with it.
//   LCGS_UNIT_QUADS = 1 (default since round 5): the 8x8 quadrant k (lane = 8 row + column)
//   LCGS_UNIT_QUADS = 0 (rounds 1-5a; A/B builds): the 16x4 strip k (lane = 16 row + column)
// A compact unit meets fewer splats: a footprint of d x d pixels reaches (d/16 + 1)(d/4 + 1) strips but (d/8 + 1)^2 quadrants,
// ~10 % fewer (entry, wave) pairs for both renderers -- forward 1 400 -> 1 432 frames/s, render-backward 0.440 -> 0.415 ms,
// forward+backward +3.6 % in same-box A/B (profiles/r05_unit_quads_ab.txt).  Per-pixel arithmetic is untouched: same images bit
// for bit.  The backward's reduction needs nothing new: lanes l, l + 16, l + 32, l + 48 still share their column, so dx stays a
// lane constant through the two permlane folds, and the DPP step sums a 16-lane group = two pixel rows of eight columns.
#ifndef LCGS_UNIT_QUADS
#define LCGS_UNIT_QUADS 1
#endif
__device__ __forceinline__ uint32_t unit_px(uint32_t tx, uint32_t wave, uint32_t lane)
{
    return LCGS_UNIT_QUADS ? tx * kBlockX + 8u * (wave & 1u) + (lane & 7u) : tx * kBlockX + (lane & 15u);
}
__device__ __forceinline__ uint32_t unit_py(uint32_t ty, uint32_t wave, uint32_t lane)
{
    return LCGS_UNIT_QUADS ? ty * kBlockY + 8u * (wave >> 1) + (lane >> 3) : ty * kBlockY + 4u * wave + (lane >> 4);
}
__device__ __forceinline__ uint32_t splat_unit_mask(float mx, float my, float ca, float cb, float cc, float t, float x0, float y0,
                                                    float x1)
{
    return LCGS_UNIT_QUADS ? splat_quad_mask(mx, my, ca, cb, cc, t, x0, y0) : splat_strip_mask(mx, my, ca, cb, cc, t, x0, y0, x1);
}

// Workgroup -> tile map.  Workgroups are dealt round-robin to the 8 XCDs (workgroup b lands on XCD b % 8,
// and is that XCD's (b / 8)-th workgroup).  Tiles are grouped into blocks of 8 x 4 tiles (128 x 64 px: most
// splats live inside one block, so its tiles share their records in one XCD's L2) and the blocks are dealt
// round-robin to the XCDs, so every XCD gets an even mix of dense (image centre) and sparse (border) regions.
// Speed only: any placement gives the same image.
constexpr uint32_t kBlkW = 8, kBlkH = 4;

__device__ __forceinline__ bool tile_of_workgroup(uint32_t b, uint32_t grid_x, uint32_t grid_y, uint32_t& tx,
                                                  uint32_t& ty)
{
    const uint32_t xcd = b % kXcd, seq = b / kXcd;
    const uint32_t bw = (grid_x + kBlkW - 1) / kBlkW;
    const uint32_t blk = xcd + kXcd * (seq / (kBlkW * kBlkH));
    const uint32_t in  = seq % (kBlkW * kBlkH);
    tx = (blk % bw) * kBlkW + in % kBlkW;
    ty = (blk / bw) * kBlkH + in / kBlkW;
    return tx < grid_x && ty < grid_y;
}

__host__ __device__ inline uint32_t render_grid_size(uint32_t grid_x, uint32_t grid_y)
{
    const uint32_t bw = (grid_x + kBlkW - 1) / kBlkW, bh = (grid_y + kBlkH - 1) / kBlkH;
    const uint32_t per_xcd = (bw * bh + kXcd - 1) / kXcd; // blocks per XCD
    return per_xcd * kBlkW * kBlkH * kXcd;
}

} // namespace tile
} // namespace lcgs
