// fused_forward.hip -- the one-submission forward frame: everything between "scene resident in HBM"
// and "sorted per-tile lists ready for the renderer", with no host round trip.
//
// What the reference does per frame (app/main.cpp:266-308 -> sh_preprocessor.cpp, gs_projector/,
// gs_tile_splatter/impl.cpp:63-180): three per-splat passes, three fills, a 64-bit radix sort over
// all L (tile,splat) pairs and five stream synchronisations.  What this file does instead, with
// bit-identical per-tile lists:
//
//   k_cull_compact       one pass over ALL splats: view transform, near cull, covariance projection,
//                        conic/radius/rect and the exact opacity-aware pruning of the rect; each 2048-splat
//                        chunk leaves its survivors IN INDEX ORDER in its own slab, plus the depth sort's
//                        first digit counts (no cross-workgroup dependency; the sort's first pass compacts).
//   k_build_records      one DENSE pass over the V survivors: 192-byte SH fetch + colour (the dominant
//                        stream, paid only for splats that reach the screen) -> packed 48-byte records.
//   depth sort           32-bit radix sort of the V survivors by depth bits (pair_sort.hip) -- the low
//                        32 bits of the reference's 64-bit key, sorted BEFORE duplication, on V ~ L/5 items.
//   k_expand_*           pruned tile counts in depth order -> pair offsets -> wave-cooperative, load-balanced
//                        duplication: every 64 consecutive output pairs are written by 64 consecutive lanes
//                        (coalesced), whatever the splat sizes.
//   tile partition       stable radix passes over only the ceil(log2 G) tile-id bits (2 passes at 1080p).
//                        LSD order (depth digits first, tile digits last) makes the result equal to a
//                        stable sort on the reference's (tile << 32 | depth) key with index-order ties.
//   k_get_ranges_u32     per-tile [start,end) (gs_tile_splatter/shader.cpp:71-100).
//
// Element counts (V, L) never leave the device: kernels read them from d_counts.
#include <algorithm>

#include <hip/hip_fp16.h>

#include <hip/hip_ext.h>

#include "launch.hpp"
#include "stream_access.hpp"
#include "tie_order.hpp"

namespace lcgs
{
namespace
{

constexpr int kThreads = 256;

__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t& total)
{
    const int lane = threadIdx.x & 63;
    uint32_t  inc  = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    total = __shfl(inc, 63, 64);
    return inc - v;
}

// Everything the projector + allocate_tiles compute for one splat (gs_projector/shader.cpp:107-137,
// gs_tile_splatter/shader.cpp:117-157), plus the pruned rect of gs_math.hpp tight_rect.
struct Projected {
    bool     visible;   // emits >= 1 pair after pruning
    int32_t  radius;    // reference radius (0 if near-culled)
    uint32_t ref_tiles; // reference tiles_touched
    float    pix_x, pix_y, depth, conic[3];
    uint32_t rmin[2], rw, rh; // pruned rect: origin + size in LIST BLOCKS (= tiles unless CamParams::list_shift)
    uint32_t trect_xy, trect_wh; // the same rect in tiles, packed like SplatRecord::rect_xy / rect_wh
};

// One splat's inputs, loaded before any of the arithmetic so that all of a lane's loads are in flight together.
struct SplatIn {
    float  px, py, pz, sx, sy, sz;
    float4 q; // stored (r,x,y,z)
};

__device__ __forceinline__ SplatIn load_splat(int64_t idx, const float* __restrict__ pos, const float* __restrict__ scale,
                                              const float* __restrict__ rotq)
{
    SplatIn in;
    in.px = pos[3 * (size_t)idx + 0];
    in.py = pos[3 * (size_t)idx + 1];
    in.pz = pos[3 * (size_t)idx + 2];
    in.sx = scale[3 * (size_t)idx + 0];
    in.sy = scale[3 * (size_t)idx + 1];
    in.sz = scale[3 * (size_t)idx + 2];
    in.q  = *reinterpret_cast<const float4*>(rotq + 4 * (size_t)idx);
    return in;
}

__device__ __forceinline__ Projected project_splat(const CamParams& cp, float scale_modifier, const SplatIn& in,
                                                   float opac)
{
    const float px = in.px, py = in.py, pz = in.pz;
    Projected r;
    r.visible   = false;
    r.radius    = 0;
    r.ref_tiles = 0;
    r.pix_x = r.pix_y = r.depth = 0.0f;
    r.conic[0] = r.conic[1] = r.conic[2] = 0.0f;
    r.rmin[0] = r.rmin[1] = r.rw = r.rh = 0;
    r.trect_xy = r.trect_wh = 0;
    float v[3], ndc[2];
    view_transform(cp, px, py, pz, v);
    if (v[2] < 0.2f) return r; // gs_projector/shader.cpp:121
    ndc_from_view(cp, v, ndc);
    r.depth    = v[2];
    float s[3] = { scale_modifier * in.sx, scale_modifier * in.sy, scale_modifier * in.sz };
    const float4 q = in.q;
    float        Sig[3][3], t[3], cov2d[3], filt[3];
    cov3d_from_scale_rot(s, q.y, q.z, q.w, q.x, Sig);
    cam_clamp(cp, v, t);
    ewa_cov2d(cp, Sig, t, true, cov2d);
    conic_and_radius(cov2d[0], cov2d[1], cov2d[2], true, cp.width, cp.height, r.conic, r.radius, filt);
    if (r.radius < cp.lod_min_radius) { // opt-in footprint cull (lcgs_set_lod; never taken at the default 0)
        r.radius = 0;
        return r;
    }
    r.pix_x = ndc2pix(ndc[0], cp.width);
    r.pix_y = ndc2pix(ndc[1], cp.height);
    uint32_t fmin[2], fmax[2], tmin[2], tmax[2];
    get_rect(r.pix_x, r.pix_y, r.radius, cp.grid_x, cp.grid_y, fmin, fmax);
    r.ref_tiles = (fmax[0] - fmin[0]) * (fmax[1] - fmin[1]);
    if (r.radius <= 0) r.ref_tiles = 0; // radius <= 0 never emits pairs (gs_tile_splatter/shader.cpp:41-42)
    if (r.ref_tiles == 0) return r;
    tight_rect(r.pix_x, r.pix_y, filt[0], filt[1], filt[2], opac, fmin, fmax, tmin, tmax);
    r.trect_xy = tmin[0] | (tmin[1] << 16);
    r.trect_wh = (tmax[0] - tmin[0]) | ((tmax[1] - tmin[1]) << 16);
    if (cp.list_shift != 0u && tmax[0] > tmin[0] && tmax[1] > tmin[1]) { // the same rect in list blocks (half-open)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            tmax[a] = ((tmax[a] - 1u) >> cp.list_shift) + 1u;
            tmin[a] >>= cp.list_shift;
        }
    }
    r.rmin[0] = tmin[0];
    r.rmin[1] = tmin[1];
    r.rw      = tmax[0] - tmin[0];
    r.rh      = tmax[1] - tmin[1];
    r.visible = r.rw * r.rh > 0u;
    return r;
}

// Pass A over ALL splats: project, cull, and collect the survivors IN INDEX ORDER chunk by chunk.
// Writes 16 B per survivor (+ radii on request).
// A workgroup owns a chunk of kCullItems x 512 consecutive splats and works on it in two phases:
//   1. every splat: view transform, the reference's near test (exact), and a CONSERVATIVE screen test -- an upper
//      bound of the reference radius from trace(cov2d) <= (|T0|^2 + |T1|^2) * lambda_max(Sigma), with explicit slack;
//      a splat this rejects provably has an empty tile rect (tiles_touched == 0).  The candidates' chunk-local
//      indices are compacted, in index order, into LDS.
//   2. the candidates only (40 % of the bicycle stand-in's splats; dense lanes): the full projection, radius,
//      rect and opacity-aware pruning, bit for bit the reference's expressions.
// Results are identical to running phase 2 on everything (that is what RADII = true does, for callers who want
// the reference's radii array, which is defined for off-screen splats too).
// Output: nothing here depends on another workgroup.  The chunk's survivors go, in index order, to the chunk's own
// 2048-slot slab as 16-byte {depth bits, splat index, pruned rect}; chunk_info[chunk] = {survivors, reference
// tiles_touched}; and -- the chunk IS the depth sort's first chunk -- counts0[digit][chunk] = how many of its keys
// carry each value of the sort's first digit.  The dense ids come out of the sort's first pass (pair_sort.hip:
// k_rowscan_first scans the chunk counts beside the digit rows, k_scatter_first reads the slabs).  An earlier
// version compacted here through a single-pass chained scan with one ticket per chunk: the pass then took 0.135 ms,
// 0.105 ms with the look-back cut out -- the chain, not the arithmetic or the traffic, was its largest cost.
constexpr int kCullThreads = 512; // 8 waves: the per-lane chain of dependent loads is 4 splats long, not 8
constexpr int kCullWaves   = kCullThreads / 64;
constexpr int kCullItems   = 4;
static_assert(kCullThreads * kCullItems == kCullChunkSplats, "the cull chunk is the depth sort's first chunk");
constexpr int kCullStaged  = 2 * kCullThreads; // candidates whose inputs stay in LDS between the phases
constexpr int kCullChunk   = kCullThreads * kCullItems; // 2048 splats = one chunk of the depth sort's first pass

// Phase-1 test.  Returns false only if the splat certainly emits no pair: behind the near plane (the reference's
// own test) or bound-of-radius disc entirely off the rasterised tiles.  Any NaN makes every comparison false, i.e.
// the splat stays a candidate and phase 2 decides.
struct ScreenBound {
    float lr2, lu2, lf2, drf, duf; // |right|^2, |up|^2, |front|^2, |right.front|, |up.front| (1,1,1,0,0 if orthonormal)
    float xhi, yhi;                // 16 * (grid - 1): first pixel column / row that is never rasterised
};

__device__ __forceinline__ ScreenBound make_screen_bound(const CamParams& cp)
{
    ScreenBound b;
    b.lr2 = cp.right[0] * cp.right[0] + cp.right[1] * cp.right[1] + cp.right[2] * cp.right[2];
    b.lu2 = cp.up[0] * cp.up[0] + cp.up[1] * cp.up[1] + cp.up[2] * cp.up[2];
    b.lf2 = cp.front[0] * cp.front[0] + cp.front[1] * cp.front[1] + cp.front[2] * cp.front[2];
    b.drf = fabsf(cp.right[0] * cp.front[0] + cp.right[1] * cp.front[1] + cp.right[2] * cp.front[2]);
    b.duf = fabsf(cp.up[0] * cp.front[0] + cp.up[1] * cp.front[1] + cp.up[2] * cp.front[2]);
    b.xhi = (float)(kBlockX * (cp.grid_x - 1u));
    b.yhi = (float)(kBlockY * (cp.grid_y - 1u));
    return b;
}

// (px, py, pz): the splat's position; g: an upper bound of sqrt(lambda_max(Sigma)) -- see splat_extent_bound
__device__ __forceinline__ bool may_reach_screen(const CamParams& cp, const ScreenBound& sb, float px_, float py_, float pz_,
                                                 float g)
{
    float v[3], ndc[2];
    view_transform(cp, px_, py_, pz_, v);
    if (v[2] < 0.2f) return false; // gs_projector/shader.cpp:121 (same expression as project_splat)
    ndc_from_view(cp, v, ndc);
    const float px = ndc2pix(ndc[0], cp.width), py = ndc2pix(ndc[1], cp.height); // bit-identical to phase 2
    const float iz = __builtin_amdgcn_rcpf(v[2]);
    const float cx = fminf(fabsf(v[0] * iz), 1.3f * cp.tanfovx) * 1.001f; // |t.x / t.z| after cam_clamp
    const float cy = fminf(fabsf(v[1] * iz), 1.3f * cp.tanfovy) * 1.001f;
    const float a = cp.focalx * iz, b = cp.focaly * iz;                   // |j00|, |j11|; |j02| = a cx, |j12| = b cy
    const float t0 = a * a * (sb.lr2 + cx * (cx * sb.lf2 + 2.0f * sb.drf)); // >= |T0|^2
    const float t1 = b * b * (sb.lu2 + cy * (cy * sb.lf2 + 2.0f * sb.duf)); // >= |T1|^2
    const float tr = (t0 + t1) * (g * g) * 1.01f; // >= cov2d.xx + cov2d.yy, 1 % slack for the rounding of either side
    // filtered: mid = tr/2 + 0.3, det >= 0  =>  lambda1 = mid + sqrt(max(0.1, mid^2 - det)) <= 2 mid + 0.05
    const float rb = 3.0f * __builtin_amdgcn_sqrtf(tr + 0.65f) * 1.001f + 2.0f; // >= ceil(3 sqrt(lambda1)), + 1 px slack
    // get_rect: the rect is empty iff px + r < 1 or px - r >= 16 (grid_x - 1), likewise in y
    if (px + rb < 1.0f) return false;
    if (py + rb < 1.0f) return false;
    if (px - rb >= sb.xhi) return false;
    if (py - rb >= sb.yhi) return false;
    return true;
}

// The camera-independent factor of the bound: lambda_max(Sigma) = |R diag(s)|^2 <= (|R| s_max)^2 with
// R(q) = I + |q|^2 (R(q/|q|) - I), i.e. |R| <= |1 - n| + n for the stored, un-normalised quaternion (n = |q|^2).
// Times |scale_modifier| it is the g of may_reach_screen.  A context-owned scene keeps it beside the position
// (lcgs_context::cull_bound, 16 bytes a splat) so that phase 1 of the cull pass reads 16 instead of 40 bytes per splat.
__device__ __forceinline__ float splat_extent_bound(float sx, float sy, float sz, const float4& q)
{
    const float n = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
    return (fabsf(1.0f - n) + n) * fmaxf(fmaxf(fabsf(sx), fabsf(sy)), fabsf(sz));
}

__device__ __forceinline__ bool may_reach_screen(const CamParams& cp, const ScreenBound& sb, float scale_modifier,
                                                 const SplatIn& in)
{
    return may_reach_screen(cp, sb, in.px, in.py, in.pz, fabsf(scale_modifier) * splat_extent_bound(in.sx, in.sy, in.sz, in.q));
}

__global__ void __launch_bounds__(256) k_cull_bound(int64_t P, const float* __restrict__ pos, const float* __restrict__ scale,
                                                      const float* __restrict__ rotq, float4* __restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < P; i += (int64_t)gridDim.x * 256) {
        const SplatIn in = load_splat(i, pos, scale, rotq);
        out[i]           = make_float4(in.px, in.py, in.pz, splat_extent_bound(in.sx, in.sy, in.sz, in.q));
    }
}

__global__ void __launch_bounds__(256) k_cull_bound_verify(int64_t P, const float* __restrict__ pos, const float* __restrict__ scale,
                                                             const float* __restrict__ rotq, const float4* __restrict__ rows,
                                                             unsigned long long* __restrict__ mismatches)
{
    uint32_t bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < P; i += (int64_t)gridDim.x * 256) {
        const SplatIn in = load_splat(i, pos, scale, rotq);
        const float4  want = make_float4(in.px, in.py, in.pz, splat_extent_bound(in.sx, in.sy, in.sz, in.q)), have = rows[i];
        bad += ((__float_as_uint(want.x) ^ __float_as_uint(have.x)) | (__float_as_uint(want.y) ^ __float_as_uint(have.y)) |
                (__float_as_uint(want.z) ^ __float_as_uint(have.z)) | (__float_as_uint(want.w) ^ __float_as_uint(have.w))) != 0u;
    }
    for (int off = 32; off > 0; off >>= 1) bad += __shfl_xor(bad, off, 64);
    if ((threadIdx.x & 63u) == 0u && bad) atomicAdd(mismatches, (unsigned long long)bad);
}

// BOUND (context-owned scenes, lcgs_context::cull_bound): phase 1 reads ONE 16-byte {position, extent bound} row per splat
// instead of the 40 bytes of position + scale + rotation in 28 loads per lane -- the pass was bound by that load pipeline,
// not by arithmetic or bytes (REJECTED.md, "The frame's cull pass") -- and phase 2 fetches scale / rotation / opacity of
// the candidates only (40 % of the stand-in's splats, in runs of consecutive rows); only positions are staged in LDS.
// (A bounded grid of workgroups striding over the chunks with the next chunk's rows requested ahead -- the "persistent,
// software-pipelined cull" of earlier rounds' to-do lists -- was built on this form and measured SLOWER: REJECTED.md.)
template <bool RADII, bool BOUND>
__global__ void __launch_bounds__(kCullThreads)
k_cull_compact(int P, CamParams cp, float scale_modifier, const FrameParams* __restrict__ fpp,
               const float* __restrict__ pos,
               const float* __restrict__ scale, const float* __restrict__ rotq, const float* __restrict__ opacity,
               int32_t* __restrict__ radii, uint4* __restrict__ slab, uint2* __restrict__ chunk_info,
               uint32_t* __restrict__ counts0, uint32_t stride0, uint32_t mask0, const float4* __restrict__ bound4)
{
    static_assert(!(RADII && BOUND), "the radii of off-screen splats need phase 2 on everything");
    __shared__ uint32_t s_wave_vis[kCullItems][kCullWaves];
    __shared__ uint32_t s_wave_tiles[kCullWaves];
    __shared__ uint32_t s_hist[256];
    __shared__ uint16_t s_cand[kCullChunk]; // chunk-local indices of the phase-1 survivors, in index order
    // the first kCullStaged candidates' inputs (10 floats each, one array per component: conflict-free) so that
    // phase 2 does not fetch them a second time; later candidates (rare: a chunk averages 820) are read again
    __shared__ float s_in[(RADII || BOUND) ? 1 : 10][(RADII || BOUND) ? 1 : kCullStaged];
    __shared__ float s_pos[3][BOUND ? kCullStaged : 1]; // BOUND: only the positions travel through LDS

    if (fpp) { // graph replay: per-call parameters come from device memory
        cp             = fpp->cp;
        scale_modifier = fpp->scale_modifier;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t bid  = blockIdx.x;
    const int64_t  base = (int64_t)bid * kCullChunk;
    if (tid < 256) s_hist[tid] = 0;

    // ---- phase 1: candidates of the chunk -> s_cand (lane t takes splats t, t + 512, ...: coalesced)
    uint32_t ncand;
    if (RADII) {
        const int64_t left = (int64_t)P - base;
        ncand              = left >= kCullChunk ? (uint32_t)kCullChunk : (uint32_t)(left > 0 ? left : 0);
        for (int c = tid; c < kCullChunk; c += kCullThreads) s_cand[c] = (uint16_t)c;
        __syncthreads();
    } else {
        const ScreenBound sb = make_screen_bound(cp);
        uint32_t          rank[kCullItems], cmask = 0;
        SplatIn           in[kCullItems]; // all 28 loads of the lane's four splats are issued before the first test
        float4            pb[kCullItems]; // BOUND: four loads
#pragma unroll
        for (int k = 0; k < kCullItems; ++k) {
            const int64_t idx = base + (int64_t)k * kCullThreads + tid;
            if (BOUND) pb[k] = bound4[idx < P ? idx : (int64_t)P - 1];
            else in[k] = load_splat(idx < P ? idx : (int64_t)P - 1, pos, scale, rotq);
        }
        const float am = fabsf(scale_modifier);
#pragma unroll
        for (int k = 0; k < kCullItems; ++k) {
            const int64_t idx = base + (int64_t)k * kCullThreads + tid;
            const bool    c   = idx < P && (BOUND ? may_reach_screen(cp, sb, pb[k].x, pb[k].y, pb[k].z, am * pb[k].w)
                                                  : may_reach_screen(cp, sb, scale_modifier, in[k]));
            const unsigned long long m = __ballot(c);
            rank[k] = __popcll(m & ((1ull << lane) - 1ull));
            if (c) cmask |= 1u << k;
            if (lane == 0) s_wave_vis[k][wave] = __popcll(m);
        }
        __syncthreads();
        uint32_t run = 0;
#pragma unroll
        for (int k = 0; k < kCullItems; ++k) {
#pragma unroll
            for (int w = 0; w < kCullWaves; ++w) {
                if (w == wave && ((cmask >> k) & 1u)) {
                    const uint32_t slot = run + rank[k];
                    s_cand[slot]        = (uint16_t)(k * kCullThreads + tid);
                    if (BOUND && slot < (uint32_t)kCullStaged) {
                        s_pos[0][slot] = pb[k].x; s_pos[1][slot] = pb[k].y; s_pos[2][slot] = pb[k].z;
                    }
                    if (!BOUND && slot < (uint32_t)kCullStaged) {
                        s_in[0][slot] = in[k].px; s_in[1][slot] = in[k].py; s_in[2][slot] = in[k].pz;
                        s_in[3][slot] = in[k].sx; s_in[4][slot] = in[k].sy; s_in[5][slot] = in[k].sz;
                        s_in[6][slot] = in[k].q.x; s_in[7][slot] = in[k].q.y; s_in[8][slot] = in[k].q.z;
                        s_in[9][slot] = in[k].q.w;
                    }
                }
                run += s_wave_vis[k][w];
            }
        }
        ncand = run;
        __syncthreads(); // s_cand complete; s_wave_vis free for phase 2
    }

    // ---- phase 2: the full projection of the candidates (dense lanes), survivors ranked in index order
    uint32_t vis_mask = 0;        // bit k: candidate k of this lane survives
    uint32_t lv[kCullItems];      // exclusive rank of the survivor inside its wave, per round
    uint32_t gidx[kCullItems];    // its splat index
    float    depth[kCullItems];
    uint2    rect[kCullItems];    // pruned rect: x | y << 16, w | h << 16
    uint32_t tiles_sum = 0;       // reference tiles_touched (for num_rendered)
#pragma unroll
    for (int k = 0; k < kCullItems; ++k) {
        bool visible = false;
        depth[k]     = 0.0f;
        rect[k]      = make_uint2(0u, 0u);
        gidx[k]      = 0u;
        lv[k]        = 0u;
        if ((uint32_t)(k * kCullThreads) < ncand) { // workgroup-uniform
            const uint32_t c = (uint32_t)(k * kCullThreads + tid);
            if (c < ncand) {
                const int64_t idx = base + (int64_t)s_cand[c];
                gidx[k]           = (uint32_t)idx;
                SplatIn in;
                if (BOUND) { // position from phase 1, the rest from the scene's arrays
                    if (k * kCullThreads < kCullStaged) { // (static per round)
                        in.px = s_pos[0][c]; in.py = s_pos[1][c]; in.pz = s_pos[2][c];
                    } else { // (rare: a chunk averages 820 candidates) the same bits from the row
                        const float4 p4 = bound4[idx];
                        in.px = p4.x; in.py = p4.y; in.pz = p4.z;
                    }
                    in.sx = scale[3 * (size_t)idx + 0]; in.sy = scale[3 * (size_t)idx + 1]; in.sz = scale[3 * (size_t)idx + 2];
                    in.q  = *reinterpret_cast<const float4*>(rotq + 4 * (size_t)idx);
                } else if (!RADII && k * kCullThreads < kCullStaged) { // (static per round: rounds 0 and 1 come from LDS)
                    in.px = s_in[0][c]; in.py = s_in[1][c]; in.pz = s_in[2][c];
                    in.sx = s_in[3][c]; in.sy = s_in[4][c]; in.sz = s_in[5][c];
                    in.q  = make_float4(s_in[6][c], s_in[7][c], s_in[8][c], s_in[9][c]);
                } else {
                    in = load_splat(idx, pos, scale, rotq);
                }
                const float     op = opacity[idx];
                // keep the compiler from sinking the scale / rotation / opacity loads below the near test: a
                // candidate always passes it, and one round trip is cheaper than two
                asm volatile("" ::"v"(in.sx), "v"(in.sy), "v"(in.sz), "v"(in.q.x), "v"(in.q.y), "v"(in.q.z), "v"(in.q.w), "v"(op));
                const Projected pr = project_splat(cp, scale_modifier, in, op);
                visible  = pr.visible;
                depth[k] = pr.depth;
                rect[k]  = make_uint2(pr.rmin[0] | (pr.rmin[1] << 16), pr.rw | (pr.rh << 16));
                tiles_sum += pr.ref_tiles;
                if (RADII) radii[idx] = pr.radius;
            }
            const unsigned long long m = __ballot(visible);
            lv[k] = __popcll(m & ((1ull << lane) - 1ull));
            if (visible) vis_mask |= 1u << k;
            if (lane == 0) s_wave_vis[k][wave] = __popcll(m);
        } else if (lane == 0) {
            s_wave_vis[k][wave] = 0u;
        }
    }
    uint32_t wt_total;
    (void)wave_excl_scan(tiles_sum, wt_total);
    if (lane == 0) s_wave_tiles[wave] = wt_total;
    __syncthreads();
    // index order inside the chunk: round k first, then wave, then lane
    uint32_t bv = 0, bt = 0, my_base[kCullItems];
#pragma unroll
    for (int k = 0; k < kCullItems; ++k) {
#pragma unroll
        for (int w = 0; w < kCullWaves; ++w) {
            if (w == wave) my_base[k] = bv;
            bv += s_wave_vis[k][w];
        }
    }
#pragma unroll
    for (int w = 0; w < kCullWaves; ++w) bt += s_wave_tiles[w];

    // ---- the chunk's survivors, in index order, into the chunk's own slab; no cross-workgroup dependency
#pragma unroll
    for (int k = 0; k < kCullItems; ++k) {
        if (!((vis_mask >> k) & 1u)) continue;
        const uint32_t key = __float_as_uint(depth[k]);
        slab[base + my_base[k] + lv[k]] = make_uint4(key, gidx[k], rect[k].x, rect[k].y);
        atomicAdd(&s_hist[key & mask0], 1u);
    }
    if (tid == 0) chunk_info[bid] = make_uint2(bv, bt);
    __syncthreads();
    if (tid < 256) counts0[(size_t)tid * stride0 + bid] = s_hist[tid]; // the depth sort's first count table
}

// Pass B over the V survivors only (dense: every lane does useful work): re-project (40 B), evaluate the SH
// colour (192 B -- the dominant stream, read only for splats that reach the screen) and write the packed
// 48-byte record the expander and the renderer gather.
// The SH block of a splat is 192 contiguous bytes; a lane-per-splat float4 walk would touch 64 different
// cache lines per wave instruction.  Instead the wave loads its 64 splats' 768 float4 chunks cooperatively --
// 12 consecutive lanes cover one splat's 192 B, so an instruction touches ~8 lines -- parks them in LDS with a
// 13-float4 row pitch (208 B: conflict-free for the per-lane ds_read_b128 that follows) and each lane then
// reads its own 48 coefficients back.
// HALF: the coefficients come from the opt-in f16 copy (lcgs_scene_use_half_sh; 96-byte rows, six 16-byte chunks per
// splat, 7-chunk LDS pitch) and are widened to f32 before the same evaluation.
// JAC (frames that keep backward state, degree 3): also stores, per survivor, the 3x3 Jacobian of the un-clamped
// colour w.r.t. the unit view direction with the colour clamp folded in (rows of clamped channels are zero) and
// the clamp mask -- 48 bytes that let the backward derive dL/dsh and the direction part of dL/dpos WITHOUT reading
// the 192-byte coefficient row again.
template <bool HALF, bool JAC>
__global__ void __launch_bounds__(kThreads)
k_build_records(int sh_deg, CamParams cp, float scale_modifier, const FrameParams* __restrict__ fpp,
                const float* __restrict__ pos,
                const float* __restrict__ scale, const float* __restrict__ rotq, const float* __restrict__ sh,
                const float* __restrict__ opacity, const uint32_t* __restrict__ vis_index,
                const uint32_t* __restrict__ d_counts, SplatRecord* __restrict__ recs, float4* __restrict__ shjac)
{
    __shared__ float4 s_sh[kThreads / 64][64 * (HALF ? 7 : 13)];

    if (fpp) {
        cp             = fpp->cp;
        scale_modifier = fpp->scale_modifier;
    }
    const uint32_t V = d_counts[0];
    const int      lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t blk = blockIdx.x; blk * kThreads < V; blk += gridDim.x) {
    const uint32_t vid   = blk * kThreads + threadIdx.x;
    const bool     valid = vid < V;
    const int      idx   = (int)vis_index[valid ? vid : V - 1];
    const bool     staged = sh_deg == 3 && ((reinterpret_cast<uintptr_t>(sh) & 15) == 0);

    if (HALF) {
        const uint32_t wave_first = blk * kThreads + wave * 64;
        const uint32_t nvalid     = wave_first < V ? ((V - wave_first) < 64u ? (V - wave_first) : 64u) : 0u;
        float4         q[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const uint32_t c    = (uint32_t)i * 64u + lane;
            const uint32_t slot = c / 6u, part = c - slot * 6u;
            const int      sidx = __shfl(idx, (int)slot, 64);
            q[i]                = make_float4(0, 0, 0, 0);
            if (slot < nvalid) q[i] = reinterpret_cast<const float4*>(reinterpret_cast<const __half*>(sh) + (size_t)sidx * 48)[part];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const uint32_t c    = (uint32_t)i * 64u + lane;
            const uint32_t slot = c / 6u, part = c - slot * 6u;
            s_sh[wave][slot * 7u + part] = q[i];
        }
    } else if (staged) {
        const uint32_t wave_first = blk * kThreads + wave * 64;
        const uint32_t nvalid     = wave_first < V ? ((V - wave_first) < 64u ? (V - wave_first) : 64u) : 0u;
        float4         q[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const uint32_t c    = (uint32_t)i * 64u + lane;
            const uint32_t slot = c / 12u, part = c - slot * 12u;
            const int      sidx = __shfl(idx, (int)slot, 64);
            q[i]                = make_float4(0, 0, 0, 0);
            // (read once per frame: a streaming load -- builder alone 0.133 -> 0.124 ms; the frame beside it unchanged)
            if (slot < nvalid) q[i] = ld_stream(reinterpret_cast<const float4*>(sh + (size_t)sidx * 48) + part);
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const uint32_t c    = (uint32_t)i * 64u + lane;
            const uint32_t slot = c / 12u, part = c - slot * 12u;
            s_sh[wave][slot * 13u + part] = q[i];
        }
    }
    __syncthreads();
    if (valid) {

    const float px = pos[3 * (size_t)idx + 0], py = pos[3 * (size_t)idx + 1], pz = pos[3 * (size_t)idx + 2];

    // colour (sh_preprocessor.cpp:27-157) first, projection after: the 48 coefficients and the projection's
    // intermediates are never live together
    float raw[3];
    // d raw[c] / d (unit direction)[j] from the same coefficients (only formed when JAC)
    auto colour_jacobian = [&](auto coef) {
        const float dx = px - cp.campos[0], dy = py - cp.campos[1], dz = pz - cp.campos[2];
        const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
        const float x = dx * inv, y = dy * inv, z = dz * inv;
        const float xx = x * x, yy = y * y, zz = z * z;
        float       J[3][3] = { { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 } };
#define LCGS_JAC(k, B, DX, DY, DZ)                                                                                    \
    {                                                                                                                 \
        const float ddx = (DX), ddy = (DY), ddz = (DZ);                                                               \
        _Pragma("unroll") for (int c = 0; c < 3; ++c)                                                                 \
        {                                                                                                             \
            const float ck = coef(k, c);                                                                              \
            J[c][0] += ck * ddx;                                                                                      \
            J[c][1] += ck * ddy;                                                                                      \
            J[c][2] += ck * ddz;                                                                                      \
        }                                                                                                             \
        asm volatile("" ::: "memory"); /* one coefficient triple in registers at a time (they come from LDS) */        \
    }
        LCGS_SH_TERMS(LCGS_JAC)
#undef LCGS_JAC
        uint32_t mask = 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const bool open = raw[c] > 0.0f && raw[c] < 1.0f; // the clamp passes a gradient only inside (0, 1)
            mask |= open ? (1u << c) : 0u;
            if (!open) J[c][0] = J[c][1] = J[c][2] = 0.0f;
        }
        float4* o = shjac + (size_t)vid * 3;
        o[0]      = make_float4(J[0][0], J[0][1], J[0][2], J[1][0]);
        o[1]      = make_float4(J[1][1], J[1][2], J[2][0], J[2][1]);
        o[2]      = make_float4(J[2][2], __uint_as_float(mask), 0.0f, 0.0f);
    };
    if (HALF) {
        float4 q[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) q[k] = s_sh[wave][lane * 7 + k];
        const __half* h = reinterpret_cast<const __half*>(q);
        sh_to_color(3, cp.campos, px, py, pz, [&](int k, int c) { return __half2float(h[k * 3 + c]); }, raw);
    } else if (staged) {
        float4 q[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) q[k] = s_sh[wave][lane * 13 + k];
        const float* f = reinterpret_cast<const float*>(q);
        sh_to_color(3, cp.campos, px, py, pz, [&](int k, int c) { return f[k * 3 + c]; }, raw);
    } else {
        const int    feat_dim = (sh_deg + 1) * (sh_deg + 1);
        const float* s        = sh + (size_t)idx * feat_dim * 3;
        sh_to_color(sh_deg, cp.campos, px, py, pz, [&](int k, int c) { return s[k * 3 + c]; }, raw);
    }

    asm volatile("" ::: "memory");
    SplatIn in = load_splat(idx, pos, scale, rotq);
    in.px = px; in.py = py; in.pz = pz; // already in registers
    const Projected pr = project_splat(cp, scale_modifier, in, opacity[idx]);
    float4* out = reinterpret_cast<float4*>(recs + vid);
    out[0]      = make_float4(pr.pix_x, pr.pix_y, pr.conic[0], pr.conic[1]);
    out[1]      = make_float4(pr.conic[2], opacity[idx], clamp_(raw[0], 0.0f, 1.0f), clamp_(raw[1], 0.0f, 1.0f));
    out[2]      = make_float4(clamp_(raw[2], 0.0f, 1.0f), pr.depth, __uint_as_float(pr.trect_xy), __uint_as_float(pr.trect_wh));
    if (JAC) {
        // last, with the record gone from the registers; the coefficients are read from the LDS row once more
        asm volatile("" ::: "memory");
        if (HALF) {
            const __half* h2 = reinterpret_cast<const __half*>(&s_sh[wave][lane * 7]);
            colour_jacobian([&](int k, int c) { return __half2float(h2[k * 3 + c]); });
        } else {
            const float* f2 = reinterpret_cast<const float*>(&s_sh[wave][lane * 13]);
            colour_jacobian([&](int k, int c) { return f2[k * 3 + c]; });
        }
    }
    }
    __syncthreads(); // the LDS slab is reused by the next iteration
    }
}

// ---------------------------------------------------------------------------------------------
// Duplication (gs_tile_splatter/shader.cpp:26-69 emits the same pairs, one thread per splat) in three launches:
//   k_expand_reduce   per 1024 depth-ordered survivors: sum of their pruned tile counts
//   k_expand_offsets  one workgroup: exclusive scan of those sums; publishes the pair totals
//   k_expand_emit     re-derives the in-block offsets and writes the pairs, load-balanced: a wave owns 64
//                     consecutive survivors whose pairs form one contiguous output run, written 64 pairs per step
//                     (coalesced), each lane locating its source splat by a 6-step search over the wave's
//                     exclusive offsets in LDS.  Order inside a splat is y-outer, x-inner like the reference.
// ---------------------------------------------------------------------------------------------
constexpr int kExpandSub   = 4;                      // 256-splat sub-chunks per workgroup
constexpr int kExpandChunk = kThreads * kExpandSub;  // 1024 survivors

__device__ __forceinline__ uint32_t rect_tiles(uint2 rc) { return (rc.y & 0xFFFFu) * (rc.y >> 16); }

__global__ void __launch_bounds__(kThreads) k_expand_reduce(const uint32_t* __restrict__ d_counts, uint32_t v_cap,
                                                              const uint32_t* __restrict__ order, uint32_t id_mask,
                                                              const uint2* __restrict__ rects,
                                                              uint2* __restrict__ rects_sorted,
                                                              uint32_t* __restrict__ block_sums)
{
    __shared__ uint32_t s_wave[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the first chunk's `order` entries are requested before the survivor count arrives (the buffer holds v_cap)
    uint32_t chunk = blockIdx.x;
    uint32_t src[kExpandSub];
#pragma unroll
    for (int s = 0; s < kExpandSub; ++s) {
        const uint32_t k = chunk * kExpandChunk + s * kThreads + threadIdx.x;
        src[s]           = k < v_cap ? order[k] : 0u;
    }
    const uint32_t V  = d_counts[0];
    const uint32_t nb = (V + kExpandChunk - 1) / kExpandChunk;
    while (chunk < nb) {
        uint2 rc[kExpandSub];
#pragma unroll
        for (int s = 0; s < kExpandSub; ++s) { // the one random gather (four in flight); emit reads the sorted copy
            const uint32_t k = chunk * kExpandChunk + s * kThreads + threadIdx.x;
            rc[s]            = k < V ? rects[src[s] & id_mask] : make_uint2(0u, 0u);
        }
        uint32_t sum = 0;
#pragma unroll
        for (int s = 0; s < kExpandSub; ++s) {
            const uint32_t k = chunk * kExpandChunk + s * kThreads + threadIdx.x;
            if (k < V) {
                rects_sorted[k] = rc[s];
                sum += rect_tiles(rc[s]);
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
        if (lane == 0) s_wave[wave] = sum;
        __syncthreads();
        if (threadIdx.x == 0) block_sums[chunk] = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        chunk += gridDim.x;
        if (chunk >= nb) break;
        __syncthreads();
        for (int s = 0; s < kExpandSub; ++s) {
            const uint32_t k = chunk * kExpandChunk + s * kThreads + threadIdx.x;
            src[s]           = k < V ? order[k] : 0u;
        }
    }
}

// exclusive scan of the block sums in place by one workgroup; d_counts[4] = pairs wanted,
// [2] = pairs emitted (clamped to the workspace capacity), [3] = overflow flag of THIS frame;
// [6] / [7] are sticky across frames (cleared by the host once read back): frames that overflowed since the last clear,
// and the largest pair count any of them wanted -- an asynchronous frame's overflow is not lost when the next frame
// rewrites [3]
__global__ void __launch_bounds__(1024) k_expand_offsets(uint32_t* __restrict__ d_counts,
                                                           uint32_t* __restrict__ block_sums, uint32_t nb_cap,
                                                           uint32_t capacity)
{
    __shared__ uint32_t s_wave[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // thread t owns sums 4t .. 4t+3 (one 16-byte load, requested before the survivor count arrives; the buffer
    // holds nb_cap rounded up to 4 words); more than 4096 chunks take further rounds
    uint32_t i0 = threadIdx.x * 4;
    uint4    a  = i0 < nb_cap ? *reinterpret_cast<const uint4*>(block_sums + i0) : make_uint4(0, 0, 0, 0);
    const uint32_t V  = d_counts[0];
    const uint32_t nb = (V + kExpandChunk - 1) / kExpandChunk;
    uint32_t       carry_in = 0;
    for (uint32_t base = 0; base < nb; base += 4096) {
        if (base > 0) {
            i0 = base + threadIdx.x * 4;
            a  = i0 < nb_cap ? *reinterpret_cast<const uint4*>(block_sums + i0) : make_uint4(0, 0, 0, 0);
        }
        uint32_t v[4] = { a.x, a.y, a.z, a.w }, e[4], tot = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (i0 + j >= nb) v[j] = 0;
            e[j] = tot;
            tot += v[j];
        }
        uint32_t inc = tot;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(inc, off, 64);
            if (lane >= off) inc += o;
        }
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        uint32_t carry = carry_in, round_total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            if (w < wave) carry += s_wave[w];
            round_total += s_wave[w];
        }
        const uint32_t ex = carry + inc - tot;
        if (i0 < nb_cap) *reinterpret_cast<uint4*>(block_sums + i0) = make_uint4(ex + e[0], ex + e[1], ex + e[2], ex + e[3]);
        carry_in += round_total;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const uint32_t L = carry_in;
        d_counts[4]      = L;
        d_counts[2]      = L < capacity ? L : capacity;
        d_counts[3]      = L > capacity ? 1u : 0u;
        if (L > capacity) {
            d_counts[6] += 1u;
            if (L > d_counts[7]) d_counts[7] = L;
        }
    }
}

// Output-balanced emit.  In depth order the first survivors are the nearest -- and by far the largest -- splats
// (hundreds of tiles each), so assigning splats to waves leaves a few waves with 100x the average work.  Instead a
// workgroup owns a fixed window of 4096 OUTPUT pairs: it locates the 1024-survivor chunk(s) overlapping the window
// by binary search over the chunk offsets, rebuilds each chunk's per-splat offsets in LDS (4 KiB scan), and every
// lane finds its source splat by a 10-step search.  Stores stay fully coalesced.
constexpr int kEmitWindow = 4096;

template <bool HIST>
__global__ void __launch_bounds__(kThreads) k_expand_emit(const uint32_t* __restrict__ d_counts, uint32_t grid_x,
                                                            const uint32_t* __restrict__ order, uint32_t id_mask,
                                                            const uint2* __restrict__ rects_sorted,
                                                            const uint32_t* __restrict__ block_offsets,
                                                            uint32_t* __restrict__ pair_keys,
                                                            uint32_t* __restrict__ pair_vals,
                                                            // HIST: also the tile sort's first per-chunk digit counts
                                                            int h_shift, uint32_t h_mask, uint32_t h_kpc,
                                                            uint32_t* __restrict__ h_counts, uint32_t h_stride)
{
    static_assert(kThreads == 256, "thread t owns digit t of the 256-row count table");
    __shared__ uint32_t s_hist[2][256]; // a 4096-pair window is one 4096-key sort chunk or two 2048-key ones
    __shared__ uint32_t s_off[kExpandChunk + 1];
    __shared__ uint32_t s_vid[kExpandChunk];
    __shared__ uint32_t s_xy[kExpandChunk];
    __shared__ uint32_t s_w[kExpandChunk];
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_c0;

    const uint32_t V    = d_counts[0];
    const uint32_t L    = d_counts[2]; // pairs to emit (already clamped to the workspace capacity)
    const uint32_t nb   = (V + kExpandChunk - 1) / kExpandChunk;
    const uint32_t nwin = (L + kEmitWindow - 1) / kEmitWindow;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (uint32_t win = blockIdx.x; win < nwin; win += gridDim.x) {
        const uint32_t p0 = win * kEmitWindow;
        const uint32_t p1 = (L - p0) < (uint32_t)kEmitWindow ? L : p0 + kEmitWindow;
        if (HIST) {
            s_hist[0][tid] = 0;
            s_hist[1][tid] = 0;
        }
        // largest chunk c with block_offsets[c] <= p0 (offsets are non-decreasing, [0] = 0): every thread probes one
        // of 256 evenly spaced entries, then the gap after the best one is probed the same way -- two rounds of
        // independent loads (more only beyond 65 K chunks) instead of a 12-step dependent search by one thread
        if (tid == 0) s_c0 = 0;
        __syncthreads();
        for (uint32_t span = nb; span > 1;) {
            const uint32_t step = (span + kThreads - 1) / kThreads;
            const uint32_t lo   = s_c0;
            const uint32_t idx  = lo + tid * step;
            const bool     hit  = idx < lo + span && idx < nb && block_offsets[idx] <= p0;
            __syncthreads(); // everyone has read s_c0
            if (hit) atomicMax(&s_c0, idx);
            __syncthreads();
            span = step; // the answer lies in [s_c0, s_c0 + step)
        }
        for (uint32_t c = s_c0; c < nb; ++c) {
            const uint32_t base = block_offsets[c];
            if (base >= p1) break;
            // ---- per-splat exclusive offsets of chunk c (lane t owns survivors 4t .. 4t+3 of the chunk)
            uint32_t cnt[4], run = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t j = tid * 4 + i, k = c * kExpandChunk + j;
                uint32_t       count = 0;
                if (k < V) {
                    const uint2 rc = rects_sorted[k];
                    s_vid[j]       = order[k] & id_mask;
                    s_xy[j]        = rc.x;
                    s_w[j]         = rc.y & 0xFFFFu;
                    count          = (rc.y & 0xFFFFu) * (rc.y >> 16);
                } else {
                    s_w[j] = 1;
                }
                cnt[i] = run;
                run += count;
            }
            uint32_t wtotal;
            uint32_t excl = wave_excl_scan(run, wtotal);
            if (lane == 0) s_wave[wave] = wtotal;
            __syncthreads();
            uint32_t carry = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w)
                if (w < wave) carry += s_wave[w];
            const uint32_t chunk_total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
#pragma unroll
            for (int i = 0; i < 4; ++i) s_off[tid * 4 + i] = carry + excl + cnt[i];
            if (tid == 0) s_off[kExpandChunk] = chunk_total;
            __syncthreads();
            // ---- pairs of this chunk that fall inside the window
            const uint32_t lo = p0 > base ? p0 : base;
            const uint32_t end = base + chunk_total;
            const uint32_t hi = p1 < end ? p1 : end;
            for (uint32_t p = lo + tid; p < hi; p += kThreads) {
                const uint32_t rel = p - base;
                // largest l with s_off[l] <= rel (zero-count padding entries share the offset of their successor;
                // the search lands on the last of them, which is always a real survivor because rel < chunk_total)
                uint32_t l = 0;
#pragma unroll
                for (uint32_t step = kExpandChunk / 2; step > 0; step >>= 1)
                    if (s_off[l + step] <= rel) l += step;
                const uint32_t local = rel - s_off[l];
                const uint32_t ww    = s_w[l];
                const uint32_t xy0   = s_xy[l];
                const uint32_t ty    = (xy0 >> 16) + local / ww;
                const uint32_t tx    = (xy0 & 0xFFFFu) + local % ww;
                const uint32_t key   = ty * grid_x + tx;
                pair_keys[p]         = key;
                pair_vals[p]         = s_vid[l];
                if (HIST) atomicAdd(&s_hist[(p - p0) >= h_kpc ? 1 : 0][(key >> h_shift) & h_mask], 1u);
            }
            __syncthreads();
        }
        if (HIST) { // (the chunk loop ended on a barrier: the counts are complete)
            const uint32_t nbs = (L + h_kpc - 1) / h_kpc; // the sort's chunk count for this L
            const uint32_t c0  = p0 / h_kpc;
            h_counts[(size_t)tid * h_stride + c0] = s_hist[0][tid];
            if (h_kpc < (uint32_t)kEmitWindow && c0 + 1 < nbs) h_counts[(size_t)tid * h_stride + c0 + 1] = s_hist[1][tid];
            __syncthreads(); // before the next window clears them
        }
    }
}

// shad_get_ranges (gs_tile_splatter/shader.cpp:71-100) on 32-bit tile keys; ranges zero-filled by the caller.
// A thread takes four consecutive keys (one 16-byte load) and the key before them, requested before the pair
// count arrives (the key buffer holds l_cap entries, l_cap a multiple of 4 or the tail is read key by key).
__global__ void __launch_bounds__(kThreads) k_get_ranges_u32(uint32_t* __restrict__ d_counts, uint32_t l_cap,
                                                               const uint32_t* __restrict__ keys,
                                                               uint32_t* __restrict__ ranges,
                                                               const uint32_t* __restrict__ scan_error_flag)
{
    // (also forwards a time-out flag, if the frame has one, into the counter block the host reads back)
    if (blockIdx.x == 0 && threadIdx.x == 0) d_counts[5] = scan_error_flag ? *scan_error_flag : 0u;
    uint32_t g  = blockIdx.x * kThreads + threadIdx.x; // group of four keys
    uint32_t i0 = g * 4u;
    uint4    k4 = make_uint4(0, 0, 0, 0);
    uint32_t kp = 0;
    if (i0 + 4u <= l_cap) k4 = *reinterpret_cast<const uint4*>(keys + i0);
    else if (i0 < l_cap) {
        k4.x = keys[i0];
        if (i0 + 1 < l_cap) k4.y = keys[i0 + 1];
        if (i0 + 2 < l_cap) k4.z = keys[i0 + 2];
    }
    if (i0 > 0 && i0 - 1 < l_cap) kp = keys[i0 - 1];
    const uint32_t L = d_counts[2];
    for (;;) {
        if (i0 >= L) break;
        const uint32_t kk[5] = { kp, k4.x, k4.y, k4.z, k4.w };
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t idx = i0 + j;
            if (idx >= L) break;
            const uint32_t curr_tile = kk[j + 1];
            if (idx == 0) {
                ranges[2 * (size_t)curr_tile + 0] = 0u;
            } else if (curr_tile != kk[j]) {
                ranges[2 * (size_t)kk[j] + 1]     = idx;
                ranges[2 * (size_t)curr_tile + 0] = idx;
            }
            if (idx == L - 1) ranges[2 * (size_t)curr_tile + 1] = L;
        }
        g += gridDim.x * kThreads;
        i0 = g * 4u;
        if (i0 >= L) break;
        if (i0 + 4u <= l_cap) k4 = *reinterpret_cast<const uint4*>(keys + i0);
        else {
            k4.x = keys[i0];
            k4.y = i0 + 1 < L ? keys[i0 + 1] : 0u;
            k4.z = i0 + 2 < L ? keys[i0 + 2] : 0u;
            k4.w = 0u;
        }
        kp = keys[i0 - 1];
    }
}

// point_list in original splat indices (what the reference's point_list holds), for parity checks
__global__ void __launch_bounds__(kThreads) k_map_to_index(const uint32_t* __restrict__ d_counts,
                                                             const uint32_t* __restrict__ list_vid,
                                                             const uint32_t* __restrict__ vis_index,
                                                             uint32_t* __restrict__ list_idx)
{
    const uint32_t L = d_counts[2];
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i < L) list_idx[i] = vis_index[list_vid[i]];
}

inline unsigned blocks_for(int64_t n) { return (unsigned)((n + kThreads - 1) / kThreads); }

} // namespace

inline unsigned chunks_for(int64_t n) { return (unsigned)((n + kCullChunk - 1) / kCullChunk); }
int    cull_chunk_count(int P) { return (int)chunks_for(P); }

__global__ void k_set_frame_params(FrameParams fp, FrameParams* __restrict__ dst) { *dst = fp; }

void launch_set_frame_params(const FrameParams& fp, FrameParams* d_fp, hipStream_t stream)
{
    hipLaunchKernelGGL(k_set_frame_params, dim3(1), dim3(1), 0, stream, fp, d_fp);
}

void launch_cull_bound(int64_t P, const float* pos, const float* scale, const float* rotq, float4* out, hipStream_t stream)
{
    if (P <= 0) return;
    const int64_t blocks = (P + 255) / 256;
    hipLaunchKernelGGL(k_cull_bound, dim3((unsigned)std::min<int64_t>(blocks, 16384)), dim3(256), 0, stream, P, pos, scale, rotq,
                       out);
}

void launch_cull_bound_verify(int64_t P, const float* pos, const float* scale, const float* rotq, const float4* rows,
                              unsigned long long* mismatches, hipStream_t stream)
{
    if (P <= 0) return;
    const int64_t blocks = (P + 255) / 256;
    hipLaunchKernelGGL(k_cull_bound_verify, dim3((unsigned)std::min<int64_t>(blocks, 16384)), dim3(256), 0, stream, P, pos, scale,
                       rotq, rows, mismatches);
}

void launch_cull_compact(int P, const CamParams& cp, float scale_modifier, const FrameParams* d_fp, const float* pos,
                         const float* scale, const float* rotq, const float* opacity, int32_t* radii, uint4* slab,
                         uint2* chunk_info, const DepthSortFirstPass& first, hipStream_t stream, const float4* bound4)
{
    if (radii)
        hipLaunchKernelGGL((k_cull_compact<true, false>), dim3(chunks_for(P)), dim3(kCullThreads), 0, stream, P, cp,
                           scale_modifier, d_fp, pos, scale, rotq, opacity, radii, slab, chunk_info, first.counts,
                           first.row_stride, first.mask, nullptr);
    else if (bound4)
        hipLaunchKernelGGL((k_cull_compact<false, true>), dim3(chunks_for(P)), dim3(kCullThreads), 0, stream, P, cp,
                           scale_modifier, d_fp, pos, scale, rotq, opacity, radii, slab, chunk_info, first.counts,
                           first.row_stride, first.mask, bound4);
    else
        hipLaunchKernelGGL((k_cull_compact<false, false>), dim3(chunks_for(P)), dim3(kCullThreads), 0, stream, P, cp,
                           scale_modifier, d_fp, pos, scale, rotq, opacity, radii, slab, chunk_info, first.counts,
                           first.row_stride, first.mask, nullptr);
}

void launch_build_records(int P_cap, int sh_deg, const CamParams& cp, float scale_modifier, const FrameParams* d_fp,
                          const float* pos,
                          const float* scale, const float* rotq, const float* sh, const float* opacity,
                          const uint32_t* vis_index, const uint32_t* d_counts, SplatRecord* recs, hipStream_t stream,
                          const uint16_t* sh_half, float4* shjac)
{
    // the Jacobian needs the staged (degree 3, 16-byte aligned) coefficient path
    float4* jac = (shjac && sh_deg == 3 && (sh_half || (reinterpret_cast<uintptr_t>(sh) & 15) == 0)) ? shjac : nullptr;
    const dim3 grid(blocks_for(P_cap)), block(kThreads);
#define LCGS_BUILD(H, J, SRC)                                                                                         \
    hipLaunchKernelGGL((k_build_records<H, J>), grid, block, 0, stream, sh_deg, cp, scale_modifier, d_fp, pos, scale,   \
                       rotq, SRC, opacity, vis_index, d_counts, recs, jac)
    if (sh_half) {
        if (jac) LCGS_BUILD(true, true, reinterpret_cast<const float*>(sh_half));
        else LCGS_BUILD(true, false, reinterpret_cast<const float*>(sh_half));
    } else {
        if (jac) LCGS_BUILD(false, true, sh);
        else LCGS_BUILD(false, false, sh);
    }
#undef LCGS_BUILD
}

bool build_records_writes_jacobian(int sh_deg, const float* sh, bool half)
{
    return sh_deg == 3 && (half || (reinterpret_cast<uintptr_t>(sh) & 15) == 0);
}

__global__ void __launch_bounds__(256) k_sh_to_half(int64_t n, const float* __restrict__ src, __half* __restrict__ dst)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        dst[i] = __float2half_rn(src[i]);
}

void launch_sh_to_half(int64_t n, const float* src, uint16_t* dst, hipStream_t stream)
{
    if (n <= 0) return;
    int64_t b = (n + 255) / 256;
    if (b > 65536) b = 65536;
    hipLaunchKernelGGL(k_sh_to_half, dim3((unsigned)b), dim3(256), 0, stream, n, src, reinterpret_cast<__half*>(dst));
}




// ---- splat ownership
namespace
{
__global__ void __launch_bounds__(256) k_rows_global(const uint32_t* __restrict__ vis, const uint32_t* __restrict__ d_count,
                                                     uint32_t row_first, uint32_t* __restrict__ rows_out)
{
    const uint32_t n = *d_count;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) rows_out[i] = vis[i] + row_first;
}

__global__ void __launch_bounds__(256) k_unpack_records(uint32_t n, const SplatRecord* __restrict__ recs,
                                                        const uint32_t* __restrict__ rows, const uint32_t* __restrict__ perm,
                                                        uint32_t id_bits, uint32_t tag_shift, uint32_t* __restrict__ keys,
                                                        uint32_t* __restrict__ vals, uint2* __restrict__ rects,
                                                        uint32_t* __restrict__ vis_index, uint32_t* __restrict__ d_counts,
                                                        uint32_t P, uint32_t grid_x, uint32_t grid_y)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        d_counts[0] = n;
        d_counts[1] = n; // (non-zero iff anything can be drawn; the reference's num_rendered is not known on this side)
        d_counts[kCountTieUnresolved] = 0u;
    }
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const SplatRecord r = recs[i];
        uint32_t          row = rows[i];
        // The records came from a peer: nothing in them may index past this context's arrays or this frame's grid.  A row
        // beyond the scene or a rect that leaves the grid (a short or mismatched message) becomes an entry that touches no
        // tile (empty rect) at row 0 -- it is sorted and dropped like any culled splat, never dereferenced.
        const uint32_t x0 = r.rect_xy & 0xFFFFu, y0 = r.rect_xy >> 16, w = r.rect_wh & 0xFFFFu, h = r.rect_wh >> 16;
        const bool     ok = row < P && x0 + w <= grid_x && y0 + h <= grid_y;
        if (!ok) row = 0u;
        keys[i]      = __float_as_uint(r.depth);
        vals[i]      = perm ? (i | ((perm[row] >> tag_shift) << id_bits)) : i;
        rects[i]     = ok ? make_uint2(r.rect_xy, r.rect_wh) : make_uint2(0u, 0u);
        vis_index[i] = row;
    }
}
// The same from PADDED segments (the ownership step without a host read-back, comm.cpp): owner o's rows occupy positions
// [segs.off[o], segs.off[o] + count_o) of recs / rows, count_o = table[o * N + view] -- the all-gathered counts, read on the
// device -- clipped to the segment's capacity (a clipped segment raises *overflow: the step is redone).  The sort's input is
// dense (entry d = the d-th valid row in owner order = ascending global rows), its values are POSITIONS: rects, vis_index,
// the records and later the 2-D gradient rows are addressed by position, so nothing is ever compacted or copied.
__global__ void __launch_bounds__(256) k_unpack_records_seg(OwnerSegs segs, const uint32_t* __restrict__ table, uint32_t view,
                                                            const SplatRecord* __restrict__ recs, const uint32_t* __restrict__ rows,
                                                            const uint32_t* __restrict__ perm, uint32_t id_bits, uint32_t tag_shift,
                                                            uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                            uint2* __restrict__ rects, uint32_t* __restrict__ vis_index,
                                                            uint32_t* __restrict__ d_counts, uint32_t* __restrict__ overflow,
                                                            uint32_t P, uint32_t grid_x, uint32_t grid_y)
{
    __shared__ uint32_t s_cnt[kMaxOwnerSegs], s_dense[kMaxOwnerSegs + 1];
    if (threadIdx.x == 0) {
        uint32_t acc = 0u;
        bool     over = false;
        for (uint32_t o = 0; o < segs.n; ++o) {
            const uint32_t want = table[o * segs.n + view], cap = segs.off[o + 1] - segs.off[o];
            over |= want > cap;
            s_cnt[o]   = want < cap ? want : cap;
            s_dense[o] = acc;
            acc += s_cnt[o];
        }
        s_dense[segs.n] = acc;
        if (blockIdx.x == 0) {
            d_counts[0] = acc;
            d_counts[1] = acc; // (non-zero iff anything can be drawn)
            d_counts[kCountTieUnresolved] = 0u;
            if (over) atomicOr(overflow, 1u);
        }
    }
    __syncthreads();
    const uint32_t total = segs.off[segs.n];
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        uint32_t o = 0u;
        while (o + 1u < segs.n && i >= segs.off[o + 1]) ++o;
        const uint32_t j = i - segs.off[o];
        if (j >= s_cnt[o]) continue;
        const uint32_t    d = s_dense[o] + j;
        const SplatRecord r = recs[i];
        uint32_t          row = rows[i];
        const uint32_t x0 = r.rect_xy & 0xFFFFu, y0 = r.rect_xy >> 16, w = r.rect_wh & 0xFFFFu, h = r.rect_wh >> 16;
        const bool     ok = row < P && x0 + w <= grid_x && y0 + h <= grid_y; // (as k_unpack_records: nothing indexes past the scene / grid)
        if (!ok) row = 0u;
        keys[d]      = __float_as_uint(r.depth);
        vals[d]      = perm ? (i | ((perm[row] >> tag_shift) << id_bits)) : i;
        rects[i]     = ok ? make_uint2(r.rect_xy, r.rect_wh) : make_uint2(0u, 0u);
        vis_index[i] = row;
    }
}
} // namespace

void launch_rows_global(const uint32_t* vis, const uint32_t* d_count, uint32_t row_first, uint32_t* rows_out, int64_t hint,
                        hipStream_t stream)
{
    int64_t b = (hint + 255) / 256;
    b         = b < 1 ? 1 : (b > 8192 ? 8192 : b);
    hipLaunchKernelGGL(k_rows_global, dim3((unsigned)b), dim3(256), 0, stream, vis, d_count, row_first, rows_out);
}

void launch_unpack_records(int64_t n, const SplatRecord* recs, const uint32_t* rows, const uint32_t* perm, uint32_t id_bits,
                           uint32_t tag_shift, uint32_t* keys, uint32_t* vals, uint2* rects, uint32_t* vis_index,
                           uint32_t* d_counts, uint32_t P, uint32_t grid_x, uint32_t grid_y, hipStream_t stream)
{
    int64_t b = (n + 255) / 256;
    b         = b < 1 ? 1 : (b > 8192 ? 8192 : b);
    hipLaunchKernelGGL(k_unpack_records, dim3((unsigned)b), dim3(256), 0, stream, (uint32_t)n, recs, rows, perm, id_bits,
                       tag_shift, keys, vals, rects, vis_index, d_counts, P, grid_x, grid_y);
}

void launch_unpack_records_seg(const OwnerSegs& segs, const uint32_t* table, uint32_t view, const SplatRecord* recs,
                               const uint32_t* rows, const uint32_t* perm, uint32_t id_bits, uint32_t tag_shift, uint32_t* keys,
                               uint32_t* vals, uint2* rects, uint32_t* vis_index, uint32_t* d_counts, uint32_t* overflow, uint32_t P,
                               uint32_t grid_x, uint32_t grid_y, hipStream_t stream)
{
    int64_t b = ((int64_t)segs.off[segs.n] + 255) / 256;
    b         = b < 1 ? 1 : (b > 8192 ? 8192 : b);
    hipLaunchKernelGGL(k_unpack_records_seg, dim3((unsigned)b), dim3(256), 0, stream, segs, table, view, recs, rows, perm,
                       id_bits, tag_shift, keys, vals, rects, vis_index, d_counts, overflow, P, grid_x, grid_y);
}

namespace
{
__global__ void k_owner_pair_verdict(const uint32_t* __restrict__ d_counts, uint32_t* __restrict__ overflow)
{
    if (d_counts[3] != 0u) atomicOr(overflow, 2u); // this frame needed more pairs than the workspace holds (truncated)
}
} // namespace
void launch_owner_pair_verdict(const uint32_t* d_counts, uint32_t* overflow, hipStream_t stream)
{
    hipLaunchKernelGGL(k_owner_pair_verdict, dim3(1), dim3(1), 0, stream, d_counts, overflow);
}

size_t expand_ws_bytes(int P_cap) { return (size_t)((P_cap + kExpandChunk - 1) / kExpandChunk + 8) * sizeof(uint32_t); }

bool launch_expand(int P_cap, int64_t v_hint, int64_t l_hint, uint32_t* d_counts, uint32_t grid_x,
                   const uint32_t* order, const uint2* rects, uint2* rects_sorted, uint32_t* pair_keys,
                   uint32_t* pair_vals, uint32_t capacity, uint32_t* ws, hipStream_t stream,
                   const PairSortFirstPass* first_pass, uint32_t id_mask)
{
    int64_t hint   = v_hint > 0 ? v_hint : P_cap;
    int64_t blocks = (hint + kExpandChunk - 1) / kExpandChunk;
    int64_t cap    = ((int64_t)P_cap + kExpandChunk - 1) / kExpandChunk;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_expand_reduce, dim3((unsigned)blocks), dim3(kThreads), 0, stream, d_counts, (uint32_t)P_cap, order,
                       id_mask, rects, rects_sorted, ws);
    hipLaunchKernelGGL(k_expand_offsets, dim3(1), dim3(1024), 0, stream, d_counts, ws, (uint32_t)((cap + 3) & ~(int64_t)3),
                       capacity);
    int64_t lh      = l_hint > 0 ? l_hint : capacity;
    int64_t eblocks = (lh + kEmitWindow - 1) / kEmitWindow;
    if (eblocks > 16384) eblocks = 16384;
    if (eblocks < 1) eblocks = 1;
    const bool hist = first_pass && first_pass->valid &&
                      (first_pass->keys_per_chunk == kEmitWindow || first_pass->keys_per_chunk * 2 == kEmitWindow);
    if (hist)
        hipLaunchKernelGGL(k_expand_emit<true>, dim3((unsigned)eblocks), dim3(kThreads), 0, stream, d_counts, grid_x, order,
                           id_mask, rects_sorted, ws, pair_keys, pair_vals, first_pass->shift, first_pass->mask,
                           (uint32_t)first_pass->keys_per_chunk, first_pass->counts, first_pass->row_stride);
    else
        hipLaunchKernelGGL(k_expand_emit<false>, dim3((unsigned)eblocks), dim3(kThreads), 0, stream, d_counts, grid_x, order,
                           id_mask, rects_sorted, ws, pair_keys, pair_vals, 0, 0u, (uint32_t)kEmitWindow, (uint32_t*)nullptr, 0u);
    return hist;
}

void launch_get_ranges_u32(int64_t L_hint, uint32_t l_cap, uint32_t* d_counts, const uint32_t* keys, uint32_t* ranges,
                           const uint32_t* scan_error_flag, hipStream_t stream, hipEvent_t done)
{
    unsigned blocks = blocks_for((L_hint + 3) / 4); // four keys per thread
    if (blocks > 16384u) blocks = 16384u;
    hipExtLaunchKernelGGL(k_get_ranges_u32, dim3(blocks), dim3(kThreads), 0, stream, nullptr, done, 0, d_counts, l_cap, keys,
                          ranges, scan_error_flag);
}

// d_counts[9] accumulates the members of equal-depth runs beyond the cap (zeroed per frame by k_rowscan_first)
void launch_map_to_index(int64_t L_cap, const uint32_t* d_counts, const uint32_t* list_vid, const uint32_t* vis_index,
                         uint32_t* list_idx, hipStream_t stream)
{
    hipLaunchKernelGGL(k_map_to_index, dim3(blocks_for(L_cap)), dim3(kThreads), 0, stream, d_counts, list_vid,
                       vis_index, list_idx);
}

} // namespace lcgs
