// fused_forward.hip -- the one-submission forward frame: everything between "scene resident in HBM"
// and "sorted per-tile lists ready for the renderer", with no host round trip.
//
// What the reference does per frame (app/main.cpp:266-308 -> sh_preprocessor.cpp, gs_projector/,
// gs_tile_splatter/impl.cpp:63-180): three per-splat passes, three fills, a 64-bit radix sort over
// all L (tile,splat) pairs and five stream synchronisations.  What this file does instead, with
// bit-identical per-tile lists:
//
//   k_fused_preprocess   one pass over the splats: view transform, near cull, covariance projection,
//                        conic/radius/rect, and -- only for splats that touch >= 1 tile -- the 192-byte
//                        SH fetch and colour.  Survivors are compacted IN INDEX ORDER into dense 48-byte
//                        records through a single-pass chained scan (decoupled look-back across
//                        workgroups, 8-byte self-validating status words).
//   depth sort           32-bit radix sort of the V survivors by depth bits (radix_sort.hip) -- the low
//                        32 bits of the reference's 64-bit key, sorted BEFORE duplication, on V ~ L/5 items.
//   k_gather_tiles + scan   per-splat tile counts in depth order -> pair offsets.
//   k_expand_pairs       wave-cooperative, load-balanced duplication: every 64 consecutive output pairs
//                        are written by 64 consecutive lanes (coalesced), whatever the splat sizes.
//   tile partition       stable radix passes over only the ceil(log2 G) tile-id bits (2 passes at 1080p).
//                        LSD order (depth digits first, tile digits last) makes the result equal to a
//                        stable sort on the reference's (tile << 32 | depth) key with index-order ties.
//   k_get_ranges_u32     per-tile [start,end) (gs_tile_splatter/shader.cpp:71-100).
//
// Element counts (V, L) never leave the device: kernels read them from d_counts.
#include "launch.hpp"

namespace lcgs
{
namespace
{

constexpr int kThreads = 256;

// ---------------------------------------------------------------------------------------------
// chained-scan status word: [63:62] status, [61:32] visible count, [31:0] tile count
// ---------------------------------------------------------------------------------------------
constexpr uint64_t kStatusInvalid   = 0ull;
constexpr uint64_t kStatusAggregate = 1ull;
constexpr uint64_t kStatusInclusive = 2ull;

__device__ __forceinline__ uint64_t pack_state(uint64_t status, uint32_t vis, uint32_t tiles)
{
    return (status << 62) | ((uint64_t)(vis & 0x3FFFFFFFu) << 32) | (uint64_t)tiles;
}
__device__ __forceinline__ uint64_t state_status(uint64_t s) { return s >> 62; }
__device__ __forceinline__ uint32_t state_vis(uint64_t s) { return (uint32_t)(s >> 32) & 0x3FFFFFFFu; }
__device__ __forceinline__ uint32_t state_tiles(uint64_t s) { return (uint32_t)s; }

__device__ __forceinline__ void state_store(uint64_t* p, uint64_t v)
{
    // one naturally aligned 8-byte agent-scope store: the word validates itself, no fence needed
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t state_load(const uint64_t* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

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

// scan_state: [0] ticket counter (u32 in the low half), [1] error flag, [2..] per-workgroup status words
__global__ void __launch_bounds__(kThreads)
k_fused_preprocess(int P, int sh_deg, CamParams cp, float scale_modifier, const float* __restrict__ pos,
                   const float* __restrict__ scale, const float* __restrict__ rotq, const float* __restrict__ sh,
                   const float* __restrict__ opacity, int32_t* __restrict__ radii, SplatRecord* __restrict__ recs,
                   uint32_t* __restrict__ sort_keys, uint32_t* __restrict__ sort_vals,
                   uint32_t* __restrict__ vis_index, uint64_t* __restrict__ scan_state,
                   uint32_t* __restrict__ d_counts)
{
    __shared__ uint32_t s_ticket;
    __shared__ uint32_t s_wave_vis[4], s_wave_tiles[4];
    __shared__ uint32_t s_prefix_vis, s_prefix_tiles;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_ticket = atomicAdd(reinterpret_cast<uint32_t*>(scan_state), 1u);
    __syncthreads();
    const uint32_t bid     = s_ticket; // tickets are handed out in start order: every predecessor is running
    const uint32_t nblocks = gridDim.x;
    uint64_t*      states  = scan_state + 2;
    const int      idx     = (int)(bid * kThreads + tid);

    // ---- per-splat projection (gs_projector/shader.cpp:107-137, gs_tile_splatter/shader.cpp:117-157)
    bool     visible = false;
    uint32_t tiles   = 0;
    int32_t  radius  = 0;
    float    pix_x = 0, pix_y = 0, depth = 0, conic[3] = { 0, 0, 0 };
    float    px = 0, py = 0, pz = 0;
    uint32_t rmin[2] = { 0, 0 }, rmax[2] = { 0, 0 };
    if (idx < P) {
        px = pos[3 * (size_t)idx + 0];
        py = pos[3 * (size_t)idx + 1];
        pz = pos[3 * (size_t)idx + 2];
        float v[3], ndc[2];
        view_transform(cp, px, py, pz, v);
        if (!(v[2] < 0.2f)) {
            ndc_from_view(cp, v, ndc);
            depth      = v[2];
            float s[3] = { scale_modifier * scale[3 * (size_t)idx + 0], scale_modifier * scale[3 * (size_t)idx + 1],
                           scale_modifier * scale[3 * (size_t)idx + 2] };
            const float4 q = *reinterpret_cast<const float4*>(rotq + 4 * (size_t)idx); // (r,x,y,z)
            float        Sig[3][3], t[3], cov2d[3];
            cov3d_from_scale_rot(s, q.y, q.z, q.w, q.x, Sig);
            cam_clamp(cp, v, t);
            ewa_cov2d(cp, Sig, t, true, cov2d);
            conic_and_radius(cov2d[0], cov2d[1], cov2d[2], true, cp.width, cp.height, conic, radius);
            pix_x = ndc2pix(ndc[0], cp.width);
            pix_y = ndc2pix(ndc[1], cp.height);
            get_rect(pix_x, pix_y, radius, cp.grid_x, cp.grid_y, rmin, rmax);
            tiles   = (rmax[0] - rmin[0]) * (rmax[1] - rmin[1]);
            // radius <= 0 never emits pairs (gs_tile_splatter/shader.cpp:41-42)
            if (radius <= 0) tiles = 0;
            visible = tiles > 0u;
        }
        if (radii) radii[idx] = radius;
    }

    // ---- block-local exclusive scan of (visible, tiles)
    uint32_t wv_total, wt_total;
    const uint32_t lv = wave_excl_scan(visible ? 1u : 0u, wv_total);
    const uint32_t lt = wave_excl_scan(tiles, wt_total);
    if (lane == 0) {
        s_wave_vis[wave]   = wv_total;
        s_wave_tiles[wave] = wt_total;
    }
    __syncthreads();
    uint32_t bv = 0, bt = 0, cv = 0, ct = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w < wave) {
            cv += s_wave_vis[w];
            ct += s_wave_tiles[w];
        }
        bv += s_wave_vis[w];
        bt += s_wave_tiles[w];
    }

    // ---- chained scan across workgroups (decoupled look-back), done by wave 0
    if (wave == 0) {
        uint32_t ex_v = 0, ex_t = 0;
        if (bid == 0) {
            if (lane == 0) state_store(&states[0], pack_state(kStatusInclusive, bv, bt));
        } else {
            if (lane == 0) state_store(&states[bid], pack_state(kStatusAggregate, bv, bt));
            int64_t look = (int64_t)bid - 1;
            bool    found = false;
            uint32_t spins = 0;
            while (!found) {
                const int64_t j = look - lane;
                uint64_t      s = j >= 0 ? state_load(&states[j]) : pack_state(kStatusInclusive, 0u, 0u);
                // wait until no word in the window (up to the first INCLUSIVE) is still invalid
                unsigned long long inv = __ballot(state_status(s) == kStatusInvalid);
                unsigned long long inc = __ballot(state_status(s) == kStatusInclusive);
                const int first_inc = inc ? (__ffsll((long long)inc) - 1) : 64;
                const unsigned long long need = first_inc >= 63 ? ~0ull : ((2ull << first_inc) - 1ull);
                if (inv & need) {
                    if (++spins > (1u << 22)) { // bounded spin: flag the error and bail out
                        if (lane == 0) atomicExch(reinterpret_cast<unsigned int*>(scan_state + 1), 1u);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                    continue;
                }
                const bool take = lane <= first_inc;
                uint32_t   av   = take ? state_vis(s) : 0u;
                uint32_t   at   = take ? state_tiles(s) : 0u;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    av += __shfl_xor(av, off, 64);
                    at += __shfl_xor(at, off, 64);
                }
                ex_v += av;
                ex_t += at;
                if (inc) found = true;
                else look -= 64;
            }
            if (lane == 0) state_store(&states[bid], pack_state(kStatusInclusive, ex_v + bv, ex_t + bt));
        }
        if (lane == 0) {
            s_prefix_vis   = ex_v;
            s_prefix_tiles = ex_t;
            if (bid == nblocks - 1) {
                d_counts[0] = ex_v + bv; // V: splats that touch >= 1 tile
                d_counts[1] = ex_t + bt; // the reference's num_rendered (gs_tile_splatter/impl.cpp:106)
            }
        }
    }
    __syncthreads();

    if (!visible) return;
    const uint32_t vid = s_prefix_vis + cv + lv;

    // ---- colour: only survivors pay for the 192-byte SH fetch (sh_preprocessor.cpp:27-157)
    const int    feat_dim = (sh_deg + 1) * (sh_deg + 1);
    const float* s        = sh + (size_t)idx * feat_dim * 3;
    float        raw[3];
    if (sh_deg == 3 && ((reinterpret_cast<uintptr_t>(s) & 15) == 0)) {
        float4 q[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) q[k] = reinterpret_cast<const float4*>(s)[k];
        const float* f = reinterpret_cast<const float*>(q);
        sh_to_color(3, cp.campos, px, py, pz, [&](int k, int c) { return f[k * 3 + c]; }, raw);
    } else {
        sh_to_color(sh_deg, cp.campos, px, py, pz, [&](int k, int c) { return s[k * 3 + c]; }, raw);
    }

    float4* out = reinterpret_cast<float4*>(recs + vid);
    out[0]      = make_float4(pix_x, pix_y, conic[0], conic[1]);
    out[1]      = make_float4(conic[2], opacity[idx], clamp_(raw[0], 0.0f, 1.0f), clamp_(raw[1], 0.0f, 1.0f));
    out[2]      = make_float4(clamp_(raw[2], 0.0f, 1.0f), depth, __uint_as_float(rmin[0] | (rmin[1] << 16)),
                              __uint_as_float((rmax[0] - rmin[0]) | ((rmax[1] - rmin[1]) << 16)));
    sort_keys[vid] = __float_as_uint(depth);
    sort_vals[vid] = vid;
    vis_index[vid] = (uint32_t)idx;
}

// tile counts of the survivors in depth order
__global__ void __launch_bounds__(kThreads) k_gather_tiles(const uint32_t* __restrict__ d_counts,
                                                             const uint32_t* __restrict__ order,
                                                             const SplatRecord* __restrict__ recs,
                                                             uint32_t* __restrict__ tiles_sorted)
{
    const uint32_t V = d_counts[0];
    const uint32_t k = blockIdx.x * kThreads + threadIdx.x;
    if (k >= V) return;
    const uint32_t wh = reinterpret_cast<const uint32_t*>(recs + order[k])[11];
    tiles_sorted[k]   = (wh & 0xFFFFu) * (wh >> 16);
}

// clamp the pair count to the workspace capacity; d_counts[2] = pairs actually emitted, [3] = overflow flag
__global__ void k_finalize_counts(uint32_t* __restrict__ d_counts, uint32_t capacity)
{
    const uint32_t L = d_counts[1];
    d_counts[2]      = L < capacity ? L : capacity;
    d_counts[3]      = L > capacity ? 1u : 0u;
}

// Load-balanced duplication (gs_tile_splatter/shader.cpp:26-69 emits the same pairs, one thread per
// splat).  Wave w owns 64 consecutive depth-ordered splats; their pairs form one contiguous output run
// that the wave writes 64 pairs per step, each lane locating its source splat by a 6-step search over
// the wave's exclusive offsets in LDS.  Order inside a splat is y-outer, x-inner like the reference.
__global__ void __launch_bounds__(kThreads) k_expand_pairs(const uint32_t* __restrict__ d_counts, uint32_t grid_x,
                                                             const uint32_t* __restrict__ order,
                                                             const uint32_t* __restrict__ offsets_incl,
                                                             const SplatRecord* __restrict__ recs,
                                                             uint32_t* __restrict__ pair_keys,
                                                             uint32_t* __restrict__ pair_vals, uint32_t capacity)
{
    __shared__ uint32_t s_excl[4][64];
    __shared__ uint32_t s_vid[4][64];
    __shared__ uint32_t s_xy[4][64];
    __shared__ uint32_t s_w[4][64];

    const uint32_t V    = d_counts[0];
    const int      lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t k    = blockIdx.x * kThreads + threadIdx.x;
    if (blockIdx.x * kThreads >= V) return;

    uint32_t vid = 0, xy = 0, w = 1, count = 0, incl = 0;
    if (k < V) {
        vid               = order[k];
        const uint32_t* r = reinterpret_cast<const uint32_t*>(recs + vid);
        xy                = r[10];
        const uint32_t wh = r[11];
        w                 = wh & 0xFFFFu;
        count             = w * (wh >> 16);
        incl              = offsets_incl[k];
    }
    uint32_t       total;
    const uint32_t excl      = wave_excl_scan(count, total);
    const uint32_t wave_base = __shfl(incl - count, 0, 64); // global offset of the wave's first pair
    s_excl[wave][lane] = excl;
    s_vid[wave][lane]  = vid;
    s_xy[wave][lane]   = xy;
    s_w[wave][lane]    = w;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0): the wave's own LDS writes have landed

    for (uint32_t p = lane; p < total; p += 64) {
        // largest l with excl[l] <= p (zero-count splats share an offset with their successor: take the last)
        int l = 0;
#pragma unroll
        for (int step = 32; step > 0; step >>= 1) {
            const int cand = l + step;
            if (cand < 64 && s_excl[wave][cand] <= p) l = cand;
        }
        const uint32_t local = p - s_excl[wave][l];
        const uint32_t ww    = s_w[wave][l];
        const uint32_t xy0   = s_xy[wave][l];
        const uint32_t ty    = (xy0 >> 16) + local / ww;
        const uint32_t tx    = (xy0 & 0xFFFFu) + local % ww;
        const uint32_t dst   = wave_base + p;
        if (dst < capacity) {
            pair_keys[dst] = ty * grid_x + tx;
            pair_vals[dst] = s_vid[wave][l];
        }
    }
}

// shad_get_ranges (gs_tile_splatter/shader.cpp:71-100) on 32-bit tile keys; ranges zero-filled by the caller
__global__ void __launch_bounds__(kThreads) k_get_ranges_u32(const uint32_t* __restrict__ d_counts,
                                                               const uint32_t* __restrict__ keys,
                                                               uint32_t* __restrict__ ranges)
{
    const uint32_t L   = d_counts[2];
    const uint32_t idx = blockIdx.x * kThreads + threadIdx.x;
    if (idx >= L) return;
    const uint32_t curr_tile = keys[idx];
    if (idx == 0) {
        ranges[2 * (size_t)curr_tile + 0] = 0u;
    } else {
        const uint32_t prev_tile = keys[idx - 1];
        if (curr_tile != prev_tile) {
            ranges[2 * (size_t)prev_tile + 1] = idx;
            ranges[2 * (size_t)curr_tile + 0] = idx;
        }
    }
    if (idx == L - 1) ranges[2 * (size_t)curr_tile + 1] = L;
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

size_t fused_scan_state_bytes(int P) { return (size_t)(blocks_for(P) + 2) * sizeof(uint64_t); }

void launch_fused_preprocess(int P, int sh_deg, const CamParams& cp, float scale_modifier, const float* pos,
                             const float* scale, const float* rotq, const float* sh, const float* opacity,
                             int32_t* radii, SplatRecord* recs, uint32_t* sort_keys, uint32_t* sort_vals,
                             uint32_t* vis_index, uint64_t* scan_state, uint32_t* d_counts, hipStream_t stream)
{
    hipMemsetAsync(scan_state, 0, fused_scan_state_bytes(P), stream);
    hipLaunchKernelGGL(k_fused_preprocess, dim3(blocks_for(P)), dim3(kThreads), 0, stream, P, sh_deg, cp,
                       scale_modifier, pos, scale, rotq, sh, opacity, radii, recs, sort_keys, sort_vals, vis_index,
                       scan_state, d_counts);
}

void launch_gather_tiles(int P_cap, const uint32_t* d_counts, const uint32_t* order, const SplatRecord* recs,
                         uint32_t* tiles_sorted, hipStream_t stream)
{
    hipLaunchKernelGGL(k_gather_tiles, dim3(blocks_for(P_cap)), dim3(kThreads), 0, stream, d_counts, order, recs,
                       tiles_sorted);
}

void launch_finalize_counts(uint32_t* d_counts, uint32_t capacity, hipStream_t stream)
{
    hipLaunchKernelGGL(k_finalize_counts, dim3(1), dim3(1), 0, stream, d_counts, capacity);
}

void launch_expand_pairs(int P_cap, const uint32_t* d_counts, uint32_t grid_x, const uint32_t* order,
                         const uint32_t* offsets_incl, const SplatRecord* recs, uint32_t* pair_keys,
                         uint32_t* pair_vals, uint32_t capacity, hipStream_t stream)
{
    hipLaunchKernelGGL(k_expand_pairs, dim3(blocks_for(P_cap)), dim3(kThreads), 0, stream, d_counts, grid_x, order,
                       offsets_incl, recs, pair_keys, pair_vals, capacity);
}

void launch_get_ranges_u32(int64_t L_cap, const uint32_t* d_counts, const uint32_t* keys, uint32_t* ranges,
                           hipStream_t stream)
{
    hipLaunchKernelGGL(k_get_ranges_u32, dim3(blocks_for(L_cap)), dim3(kThreads), 0, stream, d_counts, keys, ranges);
}

void launch_map_to_index(int64_t L_cap, const uint32_t* d_counts, const uint32_t* list_vid, const uint32_t* vis_index,
                         uint32_t* list_idx, hipStream_t stream)
{
    hipLaunchKernelGGL(k_map_to_index, dim3(blocks_for(L_cap)), dim3(kThreads), 0, stream, d_counts, list_vid,
                       vis_index, list_idx);
}

} // namespace lcgs
