// stage_kernels.hip -- one HIP kernel per reference "shader", operating on the reference's own
// buffer layouts (AoS-packed flat float arrays, lcgs/include/lcgs/proxy.h:21-73).  These back the
// stage-level C-ABI entry points (lcgs_sh_process / lcgs_project_forward / lcgs_tile_splat_forward)
// so that each reference operator can be swapped for its MI355X counterpart on its own.  The fused
// one-submission frame lives in fused_forward.hip and shares gs_math.hpp with these kernels.
//
// All kernels here are HBM-streaming, one splat per lane, 256 lanes per block (4 wave64s).
#include <algorithm>

#include "launch.hpp"
#include "stream_access.hpp"

namespace lcgs
{
namespace
{

constexpr int kThreads = 256;

// shad_sh_process (lcgs/src/sh_preprocessor.cpp:159-166) + mp_compute_color_from_sh (:27-157)
// Degree 3 (the only degree the reference's app uses): a wave's 64 consecutive splats own 12 KiB of contiguous
// coefficients, fetched as 12 fully coalesced 16-byte loads per lane into an LDS slab with a 13-chunk row pitch
// (conflict-free for the per-lane 16-byte reads that follow); other degrees / unaligned buffers read lane-wise.
__global__ void __launch_bounds__(kThreads) k_sh_process(int P, int deg, CamParams cp, const float* __restrict__ pos,
                                                           const float* __restrict__ sh, float* __restrict__ color)
{
    __shared__ float4 s_sh[kThreads / 64][64 * 13];
    const int  idx    = blockIdx.x * kThreads + threadIdx.x;
    const int  lane   = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool staged = deg == 3 && ((reinterpret_cast<uintptr_t>(sh) & 15) == 0);
    if (staged) {
        const int64_t first = (int64_t)blockIdx.x * kThreads + wave * 64; // first splat of this wave
        const int64_t rows  = first < P ? (P - first < 64 ? P - first : 64) : 0;
        const float4* src   = reinterpret_cast<const float4*>(sh + (size_t)first * 48);
        float4        q[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int c = i * 64 + lane; // chunk c of the wave's slab = row c / 12, part c % 12
            q[i]        = c < rows * 12 ? ld_stream(src + c) : make_float4(0, 0, 0, 0); // (1.2 GB read once: streaming loads)
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int c = i * 64 + lane;
            s_sh[wave][(c / 12) * 13 + (c % 12)] = q[i];
        }
        __builtin_amdgcn_wave_barrier(); // a wave reads only what it wrote
    }
    if (idx >= P) return;
    const float px = pos[3 * (size_t)idx + 0], py = pos[3 * (size_t)idx + 1], pz = pos[3 * (size_t)idx + 2];
    float       raw[3];
    if (staged) {
        float4 q[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) q[k] = s_sh[wave][lane * 13 + k];
        const float* f = reinterpret_cast<const float*>(q);
        sh_to_color(3, cp.campos, px, py, pz, [&](int k, int c) { return f[k * 3 + c]; }, raw);
    } else {
        const int    feat_dim = (deg + 1) * (deg + 1);
        const float* s        = sh + (size_t)idx * feat_dim * 3;
        sh_to_color(deg, cp.campos, px, py, pz, [&](int k, int c) { return s[k * 3 + c]; }, raw);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) color[3 * (size_t)idx + c] = clamp_(raw[c], 0.0f, 1.0f); // :153
}

// The same for degree 3 with 16-byte-aligned coefficient rows -- the only shape the reference's app produces -- as a STREAMING
// kernel: a bounded grid of waves, each walking slabs of 64 consecutive splats (12 KiB of coefficients) with a fixed stride and
// issuing the NEXT slab's twelve 16-byte loads (and its positions) before it evaluates the current one from LDS, so that every
// wave always has a slab in flight.  (The one-slab-per-wave form above leaves a wave's memory pipe empty while it computes and
// stores: 5.1 TB/s over the 1.33 GB of this pass on the bicycle stand-in; this form: see profiles/r05_sh_stream_ab.txt.)
__global__ void __launch_bounds__(kThreads) k_sh_process_stream(int P, CamParams cp, const float* __restrict__ pos,
                                                                  const float* __restrict__ sh, float* __restrict__ color)
{
    __shared__ float4 s_sh[kThreads / 64][64 * 13];
    const int     lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t slabs  = ((int64_t)P + 63) / 64;
    const int64_t stride = (int64_t)gridDim.x * (kThreads / 64);
    int64_t       slab   = (int64_t)blockIdx.x * (kThreads / 64) + wave;
    float4        q[12];
    float         px = 0.0f, py = 0.0f, pz = 0.0f;
    auto          fetch = [&](int64_t sl) {
        const int64_t first = sl * 64;
        const int64_t rows  = P - first < 64 ? P - first : 64;
        const float4* src   = reinterpret_cast<const float4*>(sh + (size_t)first * 48);
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int c = i * 64 + lane; // chunk c of the slab = row c / 12, part c % 12
            q[i]        = c < rows * 12 ? ld_stream(src + c) : make_float4(0, 0, 0, 0);
        }
        if (lane < rows) {
            px = pos[3 * (size_t)(first + lane) + 0];
            py = pos[3 * (size_t)(first + lane) + 1];
            pz = pos[3 * (size_t)(first + lane) + 2];
        }
    };
    if (slab < slabs) fetch(slab);
    while (slab < slabs) {
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int c = i * 64 + lane;
            s_sh[wave][(c / 12) * 13 + (c % 12)] = q[i];
        }
        const float   cx = px, cy = py, cz = pz;
        const int64_t idx = slab * 64 + lane;
        __builtin_amdgcn_wave_barrier(); // a wave reads only what it wrote
        const int64_t next = slab + stride;
        if (next < slabs) fetch(next); // in flight while this slab is evaluated
        if (idx < P) {
            float4 r[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) r[k] = s_sh[wave][lane * 13 + k];
            const float* f = reinterpret_cast<const float*>(r);
            float        raw[3];
            sh_to_color(3, cp.campos, cx, cy, cz, [&](int k, int c) { return f[k * 3 + c]; }, raw);
#pragma unroll
            for (int c = 0; c < 3; ++c) color[3 * (size_t)idx + c] = clamp_(raw[c], 0.0f, 1.0f); // :153
        }
        __builtin_amdgcn_wave_barrier(); // the slab's reads are done before the next one's writes
        slab = next;
    }
}

// shad_project_gs_focal / shad_project_gs (lcgs/src/gs_projector/shader.cpp:82-139 / :20-80)
__global__ void __launch_bounds__(kThreads) k_project(int P, CamParams cp, bool use_focal,
                                                        const float* __restrict__ pos, const float* __restrict__ scale,
                                                        const float* __restrict__ rotq, float scale_modifier,
                                                        float* __restrict__ means_2d, float* __restrict__ depth,
                                                        float* __restrict__ covs_2d)
{
    const int idx = blockIdx.x * kThreads + threadIdx.x;
    if (idx >= P) return;
    const float px = pos[3 * (size_t)idx + 0], py = pos[3 * (size_t)idx + 1], pz = pos[3 * (size_t)idx + 2];
    float       v[3], ndc[2];
    view_transform(cp, px, py, pz, v);
    ndc_from_view(cp, v, ndc);
    if (v[2] < 0.2f) return; // :121 -- no writes for near-culled splats
    depth[idx]                    = v[2];
    means_2d[2 * (size_t)idx + 0] = ndc[0];
    means_2d[2 * (size_t)idx + 1] = ndc[1];
    float s[3] = { scale_modifier * scale[3 * (size_t)idx + 0], scale_modifier * scale[3 * (size_t)idx + 1],
                   scale_modifier * scale[3 * (size_t)idx + 2] };
    // stored (r,x,y,z) -> (x,y,z,w) via .yzwx() (:130)
    const float qr = rotq[4 * (size_t)idx + 0], qx = rotq[4 * (size_t)idx + 1], qy = rotq[4 * (size_t)idx + 2],
                qz = rotq[4 * (size_t)idx + 3];
    float Sig[3][3], t[3], cov2d[3];
    cov3d_from_scale_rot(s, qx, qy, qz, qr, Sig);
    cam_clamp(cp, v, t);
    ewa_cov2d(cp, Sig, t, use_focal, cov2d);
    covs_2d[3 * (size_t)idx + 0] = cov2d[0];
    covs_2d[3 * (size_t)idx + 1] = cov2d[1];
    covs_2d[3 * (size_t)idx + 2] = cov2d[2];
}

// shad_allocate_tiles (lcgs/src/gs_tile_splatter/shader.cpp:102-163)
__global__ void __launch_bounds__(kThreads) k_allocate_tiles(int P, CamParams cp, bool use_focal,
                                                               const float* __restrict__ depth,
                                                               float* __restrict__ means_2d,
                                                               float* __restrict__ covs_2d,
                                                               uint32_t* __restrict__ tiles_touched,
                                                               int32_t* __restrict__ radii,
                                                               uint32_t* __restrict__ hole_flag, uint32_t hole_mark,
                                                               uint8_t* __restrict__ flags, int flags_len)
{
    const int idx = blockIdx.x * kThreads + threadIdx.x;
    if (idx >= P) {
        if (flags && idx < flags_len) flags[idx] = 0; // the compaction's padding behind P (no separate fill launch)
        return;
    }
    if (depth[idx] < 0.2f) { // :120-121
        radii[idx]         = 0;
        tiles_touched[idx] = 0u;
        if (flags) flags[idx] = 0;
        return;
    }
    const float ndc_x = means_2d[2 * (size_t)idx + 0], ndc_y = means_2d[2 * (size_t)idx + 1];
    float       conic[3];
    int32_t     radius;
    conic_and_radius(covs_2d[3 * (size_t)idx + 0], covs_2d[3 * (size_t)idx + 1], covs_2d[3 * (size_t)idx + 2],
                     use_focal, cp.width, cp.height, conic, radius);
    const float pix_x = ndc2pix(ndc_x, cp.width), pix_y = ndc2pix(ndc_y, cp.height);
    uint32_t    rmin[2], rmax[2];
    get_rect(pix_x, pix_y, radius, cp.grid_x, cp.grid_y, rmin, rmax);
    const uint32_t tiles          = (rmax[0] - rmin[0]) * (rmax[1] - rmin[1]);
    radii[idx]                    = radius;
    tiles_touched[idx]            = tiles;
    // A splat that claims pair slots (tiles > 0) which copy_with_keys will leave unwritten (it skips radius <= 0,
    // shader.cpp:41-42): only a NaN covariance gets here.  Its slots keep the value of the reference's BufferFiller pass
    // (impl.cpp:117-118) -- the caller learns whether that fill is needed at all this frame.
    if (hole_flag && radius <= 0 && tiles > 0u) *hole_flag = hole_mark; // (a per-frame mark: the word is never zero-filled)
    covs_2d[3 * (size_t)idx + 0]  = conic[0];
    covs_2d[3 * (size_t)idx + 1]  = conic[1];
    covs_2d[3 * (size_t)idx + 2]  = conic[2];
    means_2d[2 * (size_t)idx + 0] = pix_x;
    means_2d[2 * (size_t)idx + 1] = pix_y;
    if (flags) flags[idx] = tiles > 0u ? 1 : 0;
}

// shad_copy_with_keys (lcgs/src/gs_tile_splatter/shader.cpp:26-69).  The reference walks each
// splat's rect with one thread; here the 64 lanes of a wave first handle the small rects one per
// lane, and rects with more than 32 tiles are expanded cooperatively by the whole wave (64
// consecutive pairs per step, coalesced 8-byte and 4-byte stores).
// ORDERED (the splatter's own sorted copy, see lcgs_tile_splat_forward): thread idx handles splat order[idx] and writes at
// the offsets of THAT sequence -- the same pairs, grouped by splat in depth order instead of index order.
template <bool ORDERED>
__global__ void __launch_bounds__(kThreads) k_copy_with_keys(int P, CamParams cp, const float* __restrict__ means_2d,
                                                               const uint32_t* __restrict__ offsets,
                                                               const int32_t* __restrict__ radii,
                                                               const float* __restrict__ depth,
                                                               uint64_t* __restrict__ keys,
                                                               uint32_t* __restrict__ values,
                                                               const uint32_t* __restrict__ order)
{
    const int idx  = blockIdx.x * kThreads + threadIdx.x;
    const int lane = threadIdx.x & 63;
    uint32_t  rmin[2] = { 0, 0 }, rmax[2] = { 0, 0 };
    uint32_t  off = 0, dbits = 0, w = 0, count = 0, sid = 0;
    if (idx < P) {
        sid = ORDERED ? order[idx] : (uint32_t)idx;
        const int32_t radius = radii[sid];
        if (radius > 0) {
            off = idx >= 1 ? offsets[idx - 1] : 0u;
            get_rect(means_2d[2 * (size_t)sid + 0], means_2d[2 * (size_t)sid + 1], radius, cp.grid_x, cp.grid_y, rmin,
                     rmax);
            w     = rmax[0] - rmin[0];
            count = w * (rmax[1] - rmin[1]);
            dbits = __float_as_uint(depth[sid]);
        }
    }
    const bool big = count > 32u;
    if (count > 0 && !big) {
        for (uint32_t j = rmin[1]; j < rmax[1]; ++j)
            for (uint32_t i = rmin[0]; i < rmax[0]; ++i) {
                keys[off]   = ((uint64_t)(i + j * cp.grid_x) << 32) | (uint64_t)dbits;
                values[off] = sid;
                off         = off + 1u;
            }
    }
    unsigned long long big_mask = __ballot(big);
    while (big_mask) {
        const int      src    = __ffsll((long long)big_mask) - 1;
        big_mask &= big_mask - 1;
        const uint32_t b_off   = __shfl(off, src, 64);
        const uint32_t b_count = __shfl(count, src, 64);
        const uint32_t b_w     = __shfl(w, src, 64);
        const uint32_t b_x0    = __shfl(rmin[0], src, 64);
        const uint32_t b_y0    = __shfl(rmin[1], src, 64);
        const uint32_t b_dbits = __shfl(dbits, src, 64);
        const uint32_t b_idx   = __shfl(sid, src, 64);
        for (uint32_t k = lane; k < b_count; k += 64) {
            const uint32_t j = b_y0 + k / b_w, i = b_x0 + k % b_w;
            keys[b_off + k]   = ((uint64_t)(i + j * cp.grid_x) << 32) | (uint64_t)b_dbits;
            values[b_off + k] = b_idx;
        }
    }
}

// The same pairs written OUTPUT-balanced (round 5): the kernel above gives every splat to one lane, so a wave's stores go to
// 64 different runs of the pair arrays (5.4 pairs a splat on average: scattered 8- and 4-byte stores, 100-120 us for the
// stand-in's 13 M pairs).  Here a workgroup owns kCopyWindow consecutive OUTPUT slots: k_window_sources notes, per window,
// the source whose slot range contains the window's first slot; the workgroup compacts the sources that overlap its window
// (rect, depth bits) into LDS, and every lane then finds its slot's source by binary search there -- fully coalesced stores,
// the work per workgroup bounded by the window whatever the sizes of the rects.  Frames with a NaN-covariance splat (slots
// claimed but never written, shader.cpp:41-42) keep the kernel above.
constexpr uint32_t kCopyWindow = 1024;

__global__ void __launch_bounds__(kThreads) k_window_sources(int n, const uint32_t* __restrict__ offsets,
                                                               uint32_t* __restrict__ win_first)
{
    const int s = blockIdx.x * kThreads + threadIdx.x;
    if (s >= n) return;
    const uint32_t start = s >= 1 ? offsets[s - 1] : 0u, end = offsets[s];
    if (end <= start) return;
    for (uint32_t w = (start + kCopyWindow - 1u) / kCopyWindow; w * kCopyWindow < end; ++w) win_first[w] = (uint32_t)s;
}

// sources: the n splats that claim slots (every one with tiles > 0 and a radius > 0), source e = splat order[e], its slots
// [offsets[e - 1], offsets[e])
__global__ void __launch_bounds__(kThreads) k_copy_with_keys_balanced(int n, CamParams cp, const float* __restrict__ means_2d,
                                                                        const uint32_t* __restrict__ offsets,
                                                                        const int32_t* __restrict__ radii,
                                                                        const float* __restrict__ depth,
                                                                        uint64_t* __restrict__ keys,
                                                                        uint32_t* __restrict__ values,
                                                                        const uint32_t* __restrict__ order,
                                                                        const uint32_t* __restrict__ win_first, uint32_t L,
                                                                        uint32_t num_windows,
                                                                        const uint32_t* __restrict__ dbits_of_source)
{
    __shared__ uint32_t s_start[kCopyWindow + 1], s_sid[kCopyWindow + 1], s_xy[kCopyWindow + 1], s_w[kCopyWindow + 1],
        s_db[kCopyWindow + 1];
    const uint32_t tid = threadIdx.x;
    const uint32_t win = blockIdx.x, o0 = win * kCopyWindow, o1 = o0 + kCopyWindow < L ? o0 + kCopyWindow : L;
    const uint32_t first = win_first[win];
    uint32_t       last  = win + 1u < num_windows ? win_first[win + 1u] : (uint32_t)n - 1u;
    if (last - first > kCopyWindow) last = first + kCopyWindow; // (never: every source of a window holds >= 1 of its slots)
    const uint32_t m = last - first + 1u; // sources that overlap the window, ascending = the order of their slots
    for (uint32_t e = tid; e < m; e += kThreads) {
        const uint32_t idx = first + e, sid = order[idx];
        uint32_t       rmin[2], rmax[2];
        get_rect(means_2d[2 * (size_t)sid + 0], means_2d[2 * (size_t)sid + 1], radii[sid], cp.grid_x, cp.grid_y, rmin, rmax);
        s_start[e] = idx >= 1u ? offsets[idx - 1u] : 0u;
        s_sid[e]   = sid;
        s_xy[e]    = rmin[0] | (rmin[1] << 16);
        s_w[e]     = rmax[0] - rmin[0];
        // (depth-ordered sources come with their sorted depth keys: a coalesced read instead of a gather)
        s_db[e]    = dbits_of_source ? dbits_of_source[idx] : __float_as_uint(depth[sid]);
    }
    __syncthreads();
    for (uint32_t slot = o0 + tid; slot < o1; slot += kThreads) {
        uint32_t lo = 0, hi = m - 1u; // the last source whose first slot is <= slot
        while (lo < hi) {
            const uint32_t mid = (lo + hi + 1u) >> 1;
            if (s_start[mid] <= slot) lo = mid;
            else hi = mid - 1u;
        }
        const uint32_t k = slot - s_start[lo], w = s_w[lo], xy = s_xy[lo];
        const uint32_t j = k / w, i = k - j * w;
        const uint32_t tile = ((xy & 0xFFFFu) + i) + ((xy >> 16) + j) * cp.grid_x; // shader.cpp:50-66: rows of the rect, x fastest
        keys[slot]   = ((uint64_t)tile << 32) | (uint64_t)s_db[lo];
        values[slot] = s_sid[lo];
    }
}

// ---- helpers of the splatter's sort-before-duplicate (lcgs_tile_splat_forward) ----
// the depth sort's input: (depth bits, splat index) of the n splats that claim slots, in index order
__global__ void __launch_bounds__(kThreads) k_gather_depth_keys(int n, const uint32_t* __restrict__ vis,
                                                                  const float* __restrict__ depth,
                                                                  uint32_t* __restrict__ keys, uint32_t* __restrict__ vals)
{
    const int j = blockIdx.x * kThreads + threadIdx.x;
    if (j >= n) return;
    const uint32_t sid = vis[j];
    keys[j]            = __float_as_uint(depth[sid]);
    vals[j]            = sid;
}

__global__ void __launch_bounds__(kThreads) k_gather_u32(int n, const uint32_t* __restrict__ order,
                                                           const uint32_t* __restrict__ src, uint32_t* __restrict__ dst)
{
    const int j = blockIdx.x * kThreads + threadIdx.x;
    if (j < n) dst[j] = src[order[j]];
}

// shad_get_ranges (lcgs/src/gs_tile_splatter/shader.cpp:71-100); ranges zero-filled by the caller (impl.cpp:147)
__global__ void __launch_bounds__(kThreads) k_get_ranges_u64(int64_t L, const uint64_t* __restrict__ keys,
                                                               uint32_t* __restrict__ ranges)
{
    const int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (idx >= L) return;
    const uint32_t curr_tile = (uint32_t)(keys[idx] >> 32);
    if (idx == 0) {
        ranges[2 * (size_t)curr_tile + 0] = 0u;
    } else {
        const uint32_t prev_tile = (uint32_t)(keys[idx - 1] >> 32);
        if (curr_tile != prev_tile) {
            ranges[2 * (size_t)prev_tile + 1] = (uint32_t)idx;
            ranges[2 * (size_t)curr_tile + 0] = (uint32_t)idx;
        }
    }
    if (idx == L - 1) ranges[2 * (size_t)curr_tile + 1] = (uint32_t)L;
}

inline unsigned blocks_for(int64_t n) { return (unsigned)((n + kThreads - 1) / kThreads); }

} // namespace

void launch_sh_process(int P, int deg, const CamParams& cp, const float* pos, const float* sh, float* color,
                       hipStream_t stream)
{
    if (P <= 0) return;
    static const bool stream_form = [] { const char* e = getenv("LCGS_STAGE_SH_STREAM"); return !e || e[0] != '0'; }(); // A/B hook
    if (stream_form && deg == 3 && (reinterpret_cast<uintptr_t>(sh) & 15) == 0 && P >= (1 << 16)) {
        // three workgroups per CU is what the 52 KB LDS slab allows; every wave then walks ~ P / 64 / (3072) slabs
        static const int cus = [] { int d = 0, n = 0; (void)hipGetDevice(&d); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d); return n > 0 ? n : 256; }();
        const int grid = std::min<int64_t>(blocks_for(P), (int64_t)cus * 3);
        hipLaunchKernelGGL(k_sh_process_stream, dim3(grid), dim3(kThreads), 0, stream, P, cp, pos, sh, color);
        return;
    }
    hipLaunchKernelGGL(k_sh_process, dim3(blocks_for(P)), dim3(kThreads), 0, stream, P, deg, cp, pos, sh, color);
}

void launch_project(int P, const CamParams& cp, bool use_focal, const float* pos, const float* scale,
                    const float* rotq, float scale_modifier, float* means_2d, float* depth, float* covs_2d,
                    hipStream_t stream)
{
    if (P <= 0) return;
    hipLaunchKernelGGL(k_project, dim3(blocks_for(P)), dim3(kThreads), 0, stream, P, cp, use_focal, pos, scale, rotq,
                       scale_modifier, means_2d, depth, covs_2d);
}

void launch_allocate_tiles(int P, const CamParams& cp, bool use_focal, const float* depth, float* means_2d,
                           float* covs_2d, uint32_t* tiles_touched, int32_t* radii, hipStream_t stream, uint32_t* hole_flag,
                           uint32_t hole_mark, uint8_t* flags, size_t flags_len)
{
    if (P <= 0) return;
    const int64_t threads = flags ? std::max<int64_t>(P, (int64_t)flags_len) : P;
    hipLaunchKernelGGL(k_allocate_tiles, dim3(blocks_for(threads)), dim3(kThreads), 0, stream, P, cp, use_focal, depth,
                       means_2d, covs_2d, tiles_touched, radii, hole_flag, hole_mark, flags, (int)flags_len);
}

void launch_copy_with_keys(int P, const CamParams& cp, const float* means_2d, const uint32_t* offsets,
                           const int32_t* radii, const float* depth, uint64_t* keys, uint32_t* values,
                           hipStream_t stream)
{
    if (P <= 0) return;
    hipLaunchKernelGGL(k_copy_with_keys<false>, dim3(blocks_for(P)), dim3(kThreads), 0, stream, P, cp, means_2d, offsets,
                       radii, depth, keys, values, (const uint32_t*)nullptr);
}

void launch_copy_with_keys_balanced(int n, const CamParams& cp, const float* means_2d, const uint32_t* offsets,
                                    const int32_t* radii, const float* depth, const uint32_t* order, uint64_t* keys,
                                    uint32_t* values, uint32_t L, uint32_t* win_first, hipStream_t stream,
                                    const uint32_t* dbits_of_source)
{
    if (n <= 0 || L == 0u) return;
    const uint32_t windows = (L + kCopyWindow - 1u) / kCopyWindow;
    hipLaunchKernelGGL(k_window_sources, dim3(blocks_for(n)), dim3(kThreads), 0, stream, n, offsets, win_first);
    hipLaunchKernelGGL(k_copy_with_keys_balanced, dim3(windows), dim3(kThreads), 0, stream, n, cp, means_2d, offsets, radii, depth,
                       keys, values, order, win_first, L, windows, dbits_of_source);
}
size_t copy_with_keys_windows_bytes(uint32_t L) { return ((size_t)(L + kCopyWindow - 1u) / kCopyWindow + 1) * 4; }

void launch_copy_with_keys_ordered(int n, const CamParams& cp, const float* means_2d, const uint32_t* offsets_sorted,
                                   const int32_t* radii, const float* depth, const uint32_t* order, uint64_t* keys,
                                   uint32_t* values, hipStream_t stream)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_copy_with_keys<true>, dim3(blocks_for(n)), dim3(kThreads), 0, stream, n, cp, means_2d, offsets_sorted,
                       radii, depth, keys, values, order);
}

void launch_gather_depth_keys(int n, const uint32_t* vis, const float* depth, uint32_t* keys, uint32_t* vals, hipStream_t stream)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_gather_depth_keys, dim3(blocks_for(n)), dim3(kThreads), 0, stream, n, vis, depth, keys, vals);
}

void launch_gather_u32(int n, const uint32_t* order, const uint32_t* src, uint32_t* dst, hipStream_t stream)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_gather_u32, dim3(blocks_for(n)), dim3(kThreads), 0, stream, n, order, src, dst);
}

void launch_get_ranges_u64(int64_t L, const uint64_t* keys, uint32_t* ranges, hipStream_t stream)
{
    if (L <= 0) return;
    hipLaunchKernelGGL(k_get_ranges_u64, dim3(blocks_for(L)), dim3(kThreads), 0, stream, L, keys, ranges);
}

} // namespace lcgs
