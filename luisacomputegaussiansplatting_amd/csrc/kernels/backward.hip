// backward.hip -- the backward of the fused forward frame.  The reference has no backward (README.md:70: "only
// forward code"); its only backward artefacts are the unused per-band dL/dSH helpers of
// lcgs/include/lcgs/util/sh.hpp:37-40,53-65,87-117,141-165.  What is differentiated here is exactly the forward of
// render.hip / gs_math.hpp (= gs_tile_splatter/shader.cpp:171-288, gs_projector/shader.cpp:82-158,
// sh_preprocessor.cpp:27-157); the specification is DESIGN.md "Backward" (SURVEY Appendix B).
//
//   k_render_backward      one workgroup per tile, wave k = "strip" k (the tile's 8x8 quadrant k, 16x4 strip k until round 5;
//                          unit_px / unit_py in tile_common.hpp), one pixel per lane (same geometry as the
//                          forward renderer).  The tile list is walked BACK TO FRONT in rounds of 256 entries; per
//                          round a wave lists the entries the forward BLENDED in its strip and walks that list four
//                          entries at a time with no data-dependent branch: six per-pixel terms per entry (the sums'
//                          algebra: see reduce_quad_and_add) go through two permlane-swap folds, become the nine sums
//                          w.r.t. {pixel mean, conic, opacity, rgb} behind them, are finished with DPP row shifts and
//                          added to a per-round LDS accumulator; after the round 16 lanes per entry flush the
//                          workgroup's totals with global float atomics -- 9 atomics per (tile, splat) instead of per
//                          (pixel, splat).  VALU-bound: DESIGN.md 5, profiles/r05_bwd_list_walk_ab.txt.
//   k_preprocess_backward  one lane per surviving splat: 2-D gradients -> dL/d{pos, scale, rotq, sh, opacity};
//                          the 192-byte SH gradient rows leave through LDS as coalesced 16-byte stores.
// Thresholds are constants for the derivative: near cull, alpha < 1/255 skip, T < 1e-4 stop and power > 0 gate the
// sums; the 0.99 alpha cap, the colour clamp and a saturated cam_clamp axis pass no gradient.
#include "launch.hpp"
#include "stream_access.hpp"
#include "tile_common.hpp"

namespace lcgs
{

namespace
{
using namespace tile; // tile_of_workgroup / render_grid_size / splat_strip_mask: shared with render.hip

// ---------------------------------------------------------------------------------------------------------------
// Wave reduction of the per-pixel gradient terms, four list entries at a time.
// Every entry yields per-lane values that must be summed over the 64 pixels of the strip.  Reducing each entry on its
// own costs 6 cross-lane adds per value; instead the sums of FOUR entries (A, B, C, D) are folded together so that
// each cross-lane step works on registers that are full of useful data (costs from tools/microbench/issue_rates):
//   1. v_permlane32_swap + add on (A, B): lanes 0-31 now hold A summed over lane pairs (l, l+32), lanes 32-63 hold B;
//      the same on (C, D)                                                   2 x N x (8.1 + 2.3) cycles
//   2. v_permlane16_swap + add on (AB, CD): the four 16-lane rows hold A, C, B, D, each summed over rows   N x 10.4
//   3. four DPP row-shift adds: lane 15 of every row holds that entry's total                               9 x 4 x 4.2
// and one LDS add instruction per value and FOUR entries (lanes 15, 31, 47 and 63, each to its own entry's accumulator
// row) instead of one per value and entry.  N = 6 values go through steps 1-2, nine through step 3 (reduce_quad_and_add).
// ---------------------------------------------------------------------------------------------------------------
// (inline asm rather than __builtin_amdgcn_permlane32_swap: the builtin returns fresh values, and the compiler copies
// both operands first -- two extra moves per swap; x and y are dead after the call, the swap may clobber them.  The
// s_nop covers the 2-wait-state VALU-write -> permlane-read hazard, which nothing pads inside asm.)
template <int N>
__device__ __forceinline__ void swap32_add(float x[N], float y[N], float out[N])
{
#pragma unroll
    for (int g = 0; g < N; ++g) {
        // lanes 0-31: x = own x, y = partner's x;  lanes 32-63: x = partner's y, y = own y
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x[g]), "+v"(y[g]));
        out[g] = x[g] + y[g];
    }
}
template <int N>
__device__ __forceinline__ void swap16_add(float x[N], float y[N], float out[N])
{
#pragma unroll
    for (int g = 0; g < N; ++g) {
        // rows 1 and 3 of x trade places with rows 0 and 2 of y: after the add the rows hold x.lo, y.lo, x.hi, y.hi
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x[g]), "+v"(y[g]));
        out[g] = x[g] + y[g];
    }
}
// sums over each 16-lane row, result in lane 15 of the row.  The nine chains advance in lockstep, one DPP step each
// per round, so the 2-wait-state VALU-write -> DPP-read hazard of a chain is covered by the eight other chains'
// instructions (hipcc pads nothing inside asm; the leading s_nop covers the producers of v[]).
__device__ __forceinline__ void row_sum9_to_lane15(float v[9])
{
    asm volatile("s_nop 1" ::: "memory");
#define LCGS_DPP_STEP(ctrl)                                                                                         \
    asm volatile("v_add_f32_dpp %0, %0, %0 " ctrl "\n\t"                                                             \
                 "v_add_f32_dpp %1, %1, %1 " ctrl "\n\t"                                                             \
                 "v_add_f32_dpp %2, %2, %2 " ctrl "\n\t"                                                             \
                 "v_add_f32_dpp %3, %3, %3 " ctrl "\n\t"                                                             \
                 "v_add_f32_dpp %4, %4, %4 " ctrl "\n\t"                                                             \
                 "v_add_f32_dpp %5, %5, %5 " ctrl "\n\t"                                                             \
                 "v_add_f32_dpp %6, %6, %6 " ctrl "\n\t"                                                             \
                 "v_add_f32_dpp %7, %7, %7 " ctrl "\n\t"                                                             \
                 "v_add_f32_dpp %8, %8, %8 " ctrl                                                                    \
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),   \
                   "+v"(v[8]))
    LCGS_DPP_STEP("row_shr:1 row_mask:0xf bank_mask:0xf");
    LCGS_DPP_STEP("row_shr:2 row_mask:0xf bank_mask:0xf");
    LCGS_DPP_STEP("row_shr:4 row_mask:0xf bank_mask:0xe");
    LCGS_DPP_STEP("row_shr:8 row_mask:0xf bank_mask:0xc"); // lane 15 of each row holds the row sum
#undef LCGS_DPP_STEP
    asm volatile("s_nop 1" ::: "memory");
}

// Steps 2 and 3 on (AB, CD) and the LDS adds.  rows: lanes 15 / 31 / 47 / 63 hold the LDS byte address of the
// accumulator row (&s_grad[0][idx]) of entry A / C / B / D; value g goes to row + g * 1028 (kRows floats).
// The five geometry sums separate: with q = dL/dopacity per pixel (G dL/dG = opacity x q, the opacity applied at the flush), dx depends
// on the pixel's COLUMN only, and steps 1-2 sum over the strip's four rows of one column, so only (q, q dy, q dy dy) go through
// them; behind step 2 a lane holds those three summed over its column for the entry of its 16-lane row, multiplies by THAT entry's
// dx (dx_row) and step 3 sums the products over the columns:
//   S hx = S_c dx (S_r q)   S hx dx = S_c dx dx (S_r q)   S hx dy = S_c dx (S_r q dy)   S hy = S_c S_r q dy   S hy dy = S_c S_r q dy dy
// and S_c S_r q is dL/dopacity itself -- SIX values through the permlane steps instead of nine, two multiplications per entry
// instead of six.
__device__ __forceinline__ void reduce_quad_and_add(float pair[6], float quad[6], float dx_row, uint32_t rows,
                                                    bool is_row_end)
{
    float c[6], r[9];
    swap16_add<6>(pair, quad, c);
    r[0] = dx_row * c[0]; // S hx     (all five / opacity)
    r[1] = c[1];          // S hy
    r[2] = dx_row * r[0]; // S hx dx
    r[3] = dx_row * c[1]; // S hx dy
    r[4] = c[2];          // S hy dy
    r[5] = c[0];          // dL/dopacity
    r[6] = c[3];
    r[7] = c[4];
    r[8] = c[5];
    row_sum9_to_lane15(r);
    if (is_row_end) {
        // a raw ds_add_f32: hipcc's atomic optimiser would wrap a C++ atomicAdd in a per-lane scan loop
#pragma unroll
        for (int g = 0; g < 9; ++g)
            asm volatile("ds_add_f32 %0, %1 offset:%2" ::"v"(rows), "v"(r[g]), "n"(g * 257 * 4) : "memory");
    }
}

constexpr uint32_t kNullEntry  = 256u;                // LDS row of the entry that blends nowhere
constexpr uint32_t kRows       = 257u;                // rows per slab / accumulator columns: a round's 256 entries + the null entry
constexpr uint32_t kSlab       = kRows * 16u;         // bytes per slab of s_rows
constexpr uint32_t kListStride = 264u;                // 256 entries + 3 of padding, a multiple of four (8-byte rows)
constexpr int kG2D = 12; // floats per splat in the 2-D gradient buffer: mean(2) conic(3) opacity(1) rgb(3) pad(3)

#ifdef LCGS_BWD_STATS // (measuring builds only, tools/gpu/bwd_stats.py: what the render-backward's waves actually walk)
// [0] (entry, strip) pairs walked  [1] ... that pass the wave-level candidate test  [2] lanes that blend, summed
// [3] strips with at least one walked entry  [4] staging rounds  [5] tiles  [6] list entries staged  [7] entries fetched
// [8 + b] strips whose walked count n satisfies 2^(b-1) < n <= 2^b (b = 0: n <= 1 incl. 0 ... b = 15)
__device__ unsigned long long g_bwd_stats[32];
#define LCGS_STAT(i, n) (st_[(i)] += (n))
#else
#define LCGS_STAT(i, n)
#endif

// (six waves per SIMD: 80 VGPRs with 8 spilled registers per lane instead of 92 and five waves -- render-backward 0.71 ->
//  0.67 ms, +1.6 % on the whole forward+backward step in same-box A/B runs; seven waves spill 19 and lose it again)
// KNOWN: the forward kept every entry's strip bits (it always does when it keeps backward state; the other instantiation
// repeats the four strip tests and exists for callers of the launcher that pass no masks)
// PERSIST: a bounded grid pulling tiles from *work_counter (zeroed by k_zero_grads2d), as in render.hip: while another view's
// forward is in flight (lcgs_fit_views) the cap leaves wave slots, registers and LDS free on every CU for its sort chain.
#ifndef LCGS_BWD_WAVES
#define LCGS_BWD_WAVES 6
#endif
#ifndef LCGS_BWD_KO // knock-out builds (tools/gpu/bwd_knockout.sh): 1 no reduction, 2 no evaluation, 3 no flush -- wrong results, timing only
#define LCGS_BWD_KO 0
#endif
// fill: the dense per-splat gradient rows' zero-fill as a side job (launch.hpp DenseFill): slot s clears its share of the five
// arrays' 16-byte-aligned interiors with fire-and-forget stores before it turns to its tile -- 1.45 GB through a memory
// system this VALU-bound kernel leaves idle, instead of 0.33 ms of memset kernels beside it that cost it 0.13 ms.
template <bool KNOWN, bool PERSIST>
__global__ void __launch_bounds__(256, LCGS_BWD_WAVES) k_render_backward(CamParams cp, float bg0, float bg1, float bg2,
                                                           const uint32_t* __restrict__ ranges,
                                                           const uint32_t* __restrict__ point_list,
                                                           const SplatRecord* __restrict__ recs,
                                                           const float* __restrict__ final_T,
                                                           const uint32_t* __restrict__ n_contrib,
                                                           const float* __restrict__ dL_dimg,
                                                           float* __restrict__ grads2d,
                                                           const uint32_t* __restrict__ tile_order,
                                                           const uint8_t* __restrict__ strip_masks,
                                                           const uint32_t* __restrict__ d_counts,
                                                           uint32_t* __restrict__ work_counter, DenseFill fill)
{
    // one 16-byte row per entry in each of three slabs (ONE address register serves an entry's three reads, as in the forward):
    // [0] mean.x, mean.y, -conic.x / 2, conic.y   [1] -conic.z / 2, opacity, r, g   [2] b, power floor, -, -
    // (the halved, negated diagonal: `power` then needs no multiplication by -1/2 -- a power of two commutes with rounding, so
    //  the value is the forward's bit for bit)
    // (row kNullEntry of every slab is the null entry that pads a strip's list to a multiple of four: opacity 0, blends nowhere;
    //  its accumulator column is never read)
    __shared__ float4             s_rows[3][kRows];
    __shared__ uint2              s_flush[256]; // the round's entries that reach a strip at all, compacted: (LDS row, splat)
    __shared__ uint32_t           s_fcnt[4];    // ... how many of them each staging wave holds
    __shared__ float              s_grad[9][kRows];
    __shared__ __attribute__((aligned(8))) uint16_t s_list[4][kListStride]; // [strip]: the round's entries to walk, back to front
    __shared__ unsigned long long s_mask[4][4]; // [staging wave][strip]
    __shared__ uint32_t           s_max[4];
    __shared__ uint32_t           s_slot;

    // a frame that drew nothing left final_T / n_contrib untouched (the forward returns before writing them, like
    // gs_tile_splatter/impl.cpp:109): there is nothing to differentiate, and nothing valid to read
    const bool     nothing_drawn = d_counts && d_counts[1] == 0u;
    if (nothing_drawn && fill.n[3] == 0u) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t slots = tile_order ? cp.grid_x * cp.grid_y : render_grid_size(cp.grid_x, cp.grid_y);
    uint32_t       slot  = blockIdx.x;
#ifdef LCGS_BWD_STATS
    unsigned long long st_[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    uint32_t           sub_[13];
#endif
  for (;;) { // (one pass unless PERSIST)
    if (PERSIST) {
        __syncthreads(); // the previous tile's last flush has read s_flush / s_grad; nobody still reads s_slot
        if (tid == 0) s_slot = atomicAdd(work_counter, 1u);
        __syncthreads();
        slot = s_slot;
    }
    if (slot >= slots) return; // persistent grids: the exit every workgroup reaches
    if (fill.n[3] != 0u) { // (wave-uniform) the dense rows' zero-fill: this slot's share of each array
        // (streaming stores: 1.45 GB that nothing reads before the preprocess pass overwrites 39 % of it -- together with
        //  that pass's streaming row stores +2.2 % forward+backward in same-box A/B, profiles/r04_nt_accesses_ab.txt)
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f z = { 0.0f, 0.0f, 0.0f, 0.0f };
#define LCGS_FILL_STORE(P) __builtin_nontemporal_store(z, reinterpret_cast<v4f*>(P))
        // per array: up to 3 floats in front of the first 16-byte boundary and behind the last one go to slot 0, the aligned
        // interior is shared out as float4 stores
#define LCGS_FILL(A)                                                                                     \
    {                                                                                                    \
        float*         base = fill.b##A;                                                                 \
        const uint32_t n    = fill.n[A];                                                                 \
        uint32_t       head = (uint32_t)((16u - (uint32_t)(reinterpret_cast<uintptr_t>(base) & 15u)) & 15u) >> 2; \
        head                = head < n ? head : n;                                                       \
        const uint32_t n4 = (n - head) >> 2, tail = n - head - 4u * n4;                                   \
        float4*        mid = reinterpret_cast<float4*>(base + head);                                     \
        for (uint32_t i = slot * 256u + tid; i < n4; i += slots * 256u) LCGS_FILL_STORE(mid + i); /* 4 KB per slot and step */ \
        if (slot == 0u) {                                                                                \
            if (tid < head) base[tid] = 0.0f;                                                            \
            if (tid < tail) base[head + 4u * n4 + tid] = 0.0f;                                           \
        }                                                                                                \
    }
        LCGS_FILL(0) LCGS_FILL(1) LCGS_FILL(2) LCGS_FILL(3) LCGS_FILL(4)
#undef LCGS_FILL
#undef LCGS_FILL_STORE
    }
    if (nothing_drawn) {
        if (PERSIST) continue;
        return;
    }
#ifdef LCGS_BWD_STATS
    for (int i = 0; i < 13; ++i) sub_[i] = 0;
#endif
    uint32_t tx, ty;
    if (tile_order) { // longest-list-first schedule of the forward (scheduling hint only)
        const uint32_t t = tile_order[slot];
        tx = t % cp.grid_x;
        ty = t / cp.grid_x;
    } else if (!tile_of_workgroup(slot, cp.grid_x, cp.grid_y, tx, ty)) {
        if (PERSIST) continue;
        return;
    }
    const uint32_t tile = ty * cp.grid_x + tx;
    const uint32_t px = unit_px(tx, wave, lane);
    const uint32_t py = unit_py(ty, wave, lane);
    const float    pxf = (float)px, pyf = (float)py;
    const bool     inside = (px < cp.width) && (py < cp.height);
    const size_t   hw  = (size_t)cp.width * cp.height;
    const size_t   pix = (size_t)px + (size_t)cp.width * py;

    const uint32_t last    = inside ? n_contrib[pix] : 0u; // 1-based list position of the last contributor
    const float    T_final = inside ? final_T[pix] : 0.0f;
    float          dpr = 0.0f, dpg = 0.0f, dpb = 0.0f;
    if (inside) {
        dpr = dL_dimg[pix];
        dpg = dL_dimg[pix + hw];
        dpb = dL_dimg[pix + 2 * hw];
    }

    uint32_t wmax = last;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = __shfl_xor(wmax, off, 64);
        wmax             = o > wmax ? o : wmax;
    }
    if (lane == 0) s_max[wave] = wmax;
    __syncthreads();
    uint32_t hi = s_max[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) hi = s_max[w] > hi ? s_max[w] : hi;
    const uint32_t range_start = ranges[2 * (size_t)tile + 0];
    {   // never walk past the tile's own list, whatever n_contrib holds
        const uint32_t len = ranges[2 * (size_t)tile + 1] - range_start;
        hi                 = hi < len ? hi : len;
    }
    hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)hi); // (uniform by construction: keeps the round bounds on the scalar side)

    // per-pixel recurrences, walked back to front
    // Qr: product of (1 - alpha) over the entries walked so far, over T_final (a pixel outside the image: infinity, T = 0)
    // Bd: (colour composited behind the current splat) . dL/dpixel, the background being the last layer
    float             Qr = 1.0f / T_final, Bd = bg0 * dpr + bg1 * dpg + bg2 * dpb;
    const bool        is_row_end = (lane & 15u) == 15u;
    const uint32_t    grad_base  = (uint32_t)(uintptr_t)&s_grad[0][0]; // low half of a flat LDS address = LDS offset

    if (tid == 0u) { // the null entry (visible to every wave behind the first round's barriers)
        s_rows[0][kNullEntry] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        s_rows[1][kNullEntry] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        s_rows[2][kNullEntry] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    while (hi > 0u) {
        const uint32_t lo   = hi > 256u ? hi - 256u : 0u;
        const uint32_t e    = lo + tid;
        const bool     have = e < hi;
        // ---- stage entries [lo, hi): strip tests, per-strip ballots, slab; clear the round's accumulators
        float4   a = make_float4(0, 0, 0, 0), b = a;
        float    c = 0.0f, t = -1.0f;
        uint32_t vid = 0, kmask = 0;
        if (have) {
            vid = point_list[range_start + e];
            // the forward kept every entry's strip bits: entries that reach no strip are not even fetched
            if (KNOWN) kmask = strip_masks[range_start + e];
            if (!KNOWN || kmask != 0u) {
                const float4* p = reinterpret_cast<const float4*>(recs + vid);
                a = p[0];                                           // mx, my, ca, cb
                b = p[1];                                           // cc, opacity, r, g
                c = reinterpret_cast<const float*>(recs + vid)[8];  // b
                t = (2.0f * __logf(255.0f * b.y)) * 1.0001f + 2e-4f;
            }
            if (!KNOWN) {
                const float rx0 = (float)(tx * kBlockX), ry0 = (float)(ty * kBlockY), rx1 = rx0 + (float)(kBlockX - 1);
                kmask = splat_unit_mask(a.x, a.y, a.z, a.w, b.x, t, rx0, ry0, rx1);
            }
        }
        __syncthreads(); // previous round fully flushed
        LCGS_STAT(4, wave == 0u ? 1u : 0u);
        LCGS_STAT(6, (unsigned)__popcll(__ballot(have)));
        LCGS_STAT(7, (unsigned)__popcll(__ballot(have && kmask != 0u)));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned long long m = __ballot((kmask >> k) & 1u);
            if (lane == 0) s_mask[wave][k] = m;
        }
        s_rows[0][tid] = make_float4(a.x, a.y, -0.5f * a.z, a.w);
        s_rows[1][tid] = make_float4(-0.5f * b.x, b.y, b.z, b.w);
        *reinterpret_cast<float2*>(&s_rows[2][tid]) = make_float2(c, fmax_(-0.5f * t, kBlendExpMin));
        const unsigned long long fm = __ballot(have && kmask != 0u); // entries with something to flush
        if (lane == 0) s_fcnt[wave] = (uint32_t)__popcll(fm);
#pragma unroll
        for (int g = 0; g < 9; ++g) s_grad[g][tid] = 0.0f;
        __syncthreads();

        // ---- the flush list: the entries some strip will add to, compacted (a round holds ~150 of 256 on the bench frames, and
        // the flush spends 16 lanes on each)
        uint32_t n_flush;
        {
            const uint32_t c0 = s_fcnt[0], c1 = s_fcnt[1], c2 = s_fcnt[2], c3 = s_fcnt[3];
            const uint32_t before = (wave > 0u ? c0 : 0u) + (wave > 1u ? c1 : 0u) + (wave > 2u ? c2 : 0u);
            n_flush               = c0 + c1 + c2 + c3;
            if (have && kmask != 0u) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
                s_flush[before + rank] = make_uint2(tid, vid);
            }
        }

        // ---- this strip's entries of the round as a list, back to front: the four ballots' set bits, highest first, padded to
        // a multiple of four with the null entry.  The walk below then has no data-dependent branch at all: one 8-byte LDS
        // read names four entries, their evaluations are straight-line code the scheduler can overlap (what a scalar bit
        // walk with a branch per entry cost: profiles/r05_bwd_list_walk_ab.txt).
        uint32_t n_walk = 0u; // (scalar)
        {
            uint16_t* list = &s_list[wave][0];
#pragma unroll
            for (int w = 3; w >= 0; --w) {
                const unsigned long long mm  = s_mask[w][wave];
                const uint32_t           mlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)mm);
                const uint32_t           mhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(mm >> 32));
                const uint32_t below = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u)); // set bits under my lane
                const uint32_t total = (uint32_t)__builtin_popcount(mlo) + (uint32_t)__builtin_popcount(mhi);
                const uint32_t mine  = lane < 32u ? mlo >> lane : mhi >> (lane - 32u);
                if (mine & 1u) list[n_walk + (total - 1u - below)] = (uint16_t)((uint32_t)w * 64u + lane);
                n_walk += total;
            }
            if (lane < 4u) list[n_walk + lane] = (uint16_t)kNullEntry;
            asm volatile("" ::: "memory"); // (the list is read back as 8-byte words below)
        }

        // ---- walk them, four at a time (see the reduction notes above)
        {
            // the nine per-pixel terms of entry idx (an LDS row of this round, or the null entry)
            // (q = dL/dopacity = G dL/dalpha; q dy; q dy dy; the three colour terms.  G dL/dG = opacity x q: the entry-uniform
            //  opacity joins the other entry-uniform factors at the flush)
            auto evaluate = [&](const uint32_t idx, float v[6], uint32_t& row) {
                const uint32_t pos = lo + idx; // 0-based list position (the null entry: >= hi, never below `last`)
                LCGS_STAT(0, idx < 256u ? 1u : 0u);
#if LCGS_BWD_KO == 2 // (measuring builds only: no evaluation -- the walk and the reduction alone)
                for (int g = 0; g < 6; ++g) v[g] = pxf + (float)pos;
                row = grad_base + idx * 4u;
                return;
#endif
                uint32_t roff = idx * 16u; // (byte offset of the entry's row in every slab; pinned: one scalar-to-vector move)
                asm("" : "+v"(roff));
                const char*    rows = reinterpret_cast<const char*>(&s_rows[0][0]) + roff;
                const float4   ea = *reinterpret_cast<const float4*>(rows);
                const float4   eb = *reinterpret_cast<const float4*>(rows + kSlab);
                const float2   ec = *reinterpret_cast<const float2*>(rows + 2 * kSlab);
                // the forward's own expression and evaluation order: the same splats pass the same thresholds
                const float dx    = ea.x - pxf;
                const float dy    = ea.y - pyf;
                const float power = (ea.z * dx * dx + eb.x * dy * dy) - ea.w * dx * dy; // (ea.z, eb.x: -ca / 2, -cc / 2)
                const bool  c_pos = pos < last, c_neg = !(power > 0.0f), c_flr = power >= ec.y;
                const bool  cand  = c_pos & c_neg & c_flr;
                // (three ballots of plain compares are the compares' own lane masks; a ballot of the conjunction costs a
                //  select and another compare -- two vector instructions per walked entry)
                LCGS_STAT(1, 1u);
                // exp(power): the hardware's v_exp_f32 (1 ulp; two instructions) since round 5, not the forward's DEFINED
                // function (ten).  The forward needs that one for bit-identical images; here the tolerance is 1e-3 and what
                // the two differ by -- ~1e-7 relative in alpha, an entry within that of alpha = 1/255 blended on one side
                // only (its weight is <= T / 255) -- is far inside it: every gradient test, the f64 checks at full size and
                // the soak's error distribution are unchanged.  render-backward 0.66-0.68 -> 0.63-0.64 ms, forward+backward
                // +2.2 % in same-box A/B (profiles/r05_bwd_hw_exp_ab.txt; -DLCGS_BWD_DEFINED_EXP builds the old form).
#ifdef LCGS_BWD_DEFINED_EXP
                const float G     = blend_exp(power);
#else
                const float G     = __builtin_amdgcn_exp2f(power * kExpLog2e);
#endif
                const float oG    = eb.y * G;
                const float alpha = __builtin_fminf(0.99f, oG);
                const bool  valid = cand & !(alpha < 1.0f / 255.0f);
                LCGS_STAT(2, (unsigned)__popcll(__builtin_amdgcn_ballot_w64(valid)));
#ifdef LCGS_BWD_STATS
                {   // which sub-blocks of the wave's unit does this entry blend into?  (Lane groups as named for the 16x4 strip of
                    // rounds 1-5a, lane = 16 * row + column; on the 8x8 quadrant the same lane groups are other pixel blocks.)
                    const unsigned long long vb = __builtin_amdgcn_ballot_w64(valid);
                    sub_[0] += (vb & 0x00FF00FF00FF00FFull) != 0, sub_[1] += (vb & 0xFF00FF00FF00FF00ull) != 0; // 8x4 halves
                    sub_[2] += (vb & 0x00000000FFFFFFFFull) != 0, sub_[3] += (vb & 0xFFFFFFFF00000000ull) != 0; // 16x2 halves
                    for (int q = 0; q < 4; ++q) {
                        sub_[4 + q] += (vb & (0x000F000F000F000Full << (4 * q))) != 0; // 4x4 blocks
                        sub_[8 + q] += (vb & (0xFFFFull << (16 * q))) != 0;            // 16x1 rows
                    }
                    sub_[12] += vb != 0;
                }
#endif
                // A lane that does not blend this entry carries alpha 0 through the recurrences: the product keeps its
                // value, Bd + 0 * d leaves the colour behind alone, all terms come out 0.
                // (No second wave-level skip: the staging floor already implies alpha >= 1/255 somewhere.)
                const float a   = valid ? alpha : 0.0f;
                // T in front of this splat = T_final / prod(1 - a) over this entry and everything behind it.  The product (over
                // T_final: Qr) is carried (one rounded multiplication per entry: errors of either sign, ~sqrt(n) half-ulps at the front of
                // n entries) and divided out ONCE per entry with v_rcp_f32 (1 ulp, not carried).  Dividing T itself entry
                // by entry with v_rcp_f32 is what drifted in round 3 (its bias was seen by every splat in front: a
                // screen-filling splat, whose geometry gradients are sums of ~1e5 cancelling per-pixel terms, amplified it
                // to a few 1e-3); rounds 4-5 paid a Newton step on that quotient (two FMAs) -- this form needs neither.
                Qr = Qr * (1.0f - a);
                const float Tn  = __builtin_amdgcn_rcpf(Qr); // the forward's T in front of this splat
                const float wgt = a * Tn;
                // colour behind this splat (B, the background included: it is the last layer, with weight T_final) enters
                // dL/dalpha = T (c - B) . dL/dpixel, then B absorbs the splat: B <- B + a (c - B).  Only B . dL/dpixel is ever
                // used, and the recurrence is linear: ONE carried value (Bd) instead of three colours
                const float d         = __builtin_fmaf(eb.z, dpr, __builtin_fmaf(eb.w, dpg, ec.x * dpb)) - Bd; // (c - B) . dL/dpixel
                const float dL_dalpha = d * Tn;
                Bd                    = __builtin_fmaf(a, d, Bd);
                // the 0.99 cap passes no gradient to G / opacity
                // (selected AFTER the product: on a lane that is not a candidate `power` lies outside blend_exp's
                //  domain and G is arbitrary bits, possibly NaN -- it must not meet a multiplication by 0)
                v[0] = (valid & (oG < 0.99f)) ? G * dL_dalpha : 0.0f; // q = dL/dopacity
                v[1] = v[0] * dy;             // the entry-uniform factors (opacity, conic, -1, -0.5) are applied once per entry
                v[2] = v[1] * dy;             // when the round is flushed, dx behind the row sums (reduce_quad_and_add)
                v[3] = wgt * dpr;
                v[4] = wgt * dpg;
                v[5] = wgt * dpb;
                row  = grad_base + idx * 4u;
            };
            uint32_t rows = grad_base;
            // rows 0 .. 3 of the wave hold entries A, C, B, D of a group = list slots 0, 2, 1, 3 (bytes into the group's four u16)
            const uint32_t my_slot = 2u * ((((lane >> 4) & 1u) << 1) | (lane >> 5));
#if LCGS_BWD_KO == 1 // (measuring builds only: no reduction -- the walk and the evaluation alone)
            float sink = 0.0f;
#endif
            for (uint32_t i = 0u; i < n_walk; i += 4u) {
                const uint2    four = *reinterpret_cast<const uint2*>(&s_list[wave][i]);
                const uint32_t p0   = (uint32_t)__builtin_amdgcn_readfirstlane((int)four.x);
                const uint32_t p1   = (uint32_t)__builtin_amdgcn_readfirstlane((int)four.y);
                float          A[6], B[6], pair[6], quad[6];
                uint32_t       row;
                // the entry my 16-lane row will hold behind the two swaps, and its dx at my column
                const uint32_t mine   = *reinterpret_cast<const uint16_t*>(reinterpret_cast<const char*>(&s_list[wave][i]) + my_slot);
                const float    dx_row = s_rows[0][mine].x - pxf;
                evaluate(p0 & 0xFFFFu, A, row);
                asm("v_writelane_b32 %0, %1, 15" : "+v"(rows) : "s"(row));
                evaluate(p0 >> 16, B, row);
                asm("v_writelane_b32 %0, %1, 47" : "+v"(rows) : "s"(row));
#if LCGS_BWD_KO == 1
                for (int g = 0; g < 6; ++g) sink += A[g] + B[g];
#else
                swap32_add<6>(A, B, pair);
#endif
                evaluate(p1 & 0xFFFFu, A, row);
                asm("v_writelane_b32 %0, %1, 31" : "+v"(rows) : "s"(row));
                evaluate(p1 >> 16, B, row);
                asm("v_writelane_b32 %0, %1, 63" : "+v"(rows) : "s"(row));
#if LCGS_BWD_KO == 1
                for (int g = 0; g < 6; ++g) sink += A[g] + B[g];
                sink += __builtin_bit_cast(float, rows) + dx_row;
#else
                swap32_add<6>(A, B, quad);
                reduce_quad_and_add(pair, quad, dx_row, rows, is_row_end);
#endif
            }
#if LCGS_BWD_KO == 1
            if (sink == 12345.678f) s_grad[0][tid] = sink;
#endif
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the raw LDS adds above have landed
        __syncthreads();
        // ---- flush the round.  Global float atomics run at full rate only when a wave instruction covers
        // contiguous bytes (MI355X_MICROARCH "Global float atomics": 64 lanes in 64 different rows are ~17x
        // slower), so consecutive lanes own one entry's gradient row instead of one lane per entry: 16 lanes per
        // entry (9 of them active; shifts and masks instead of a division by 12 in a loop that runs every round).
#if LCGS_BWD_KO == 3 // (measuring builds only: no flush)
        if (lo == 0xFFFFFFFFu)
#endif
        for (uint32_t cidx = tid; cidx < n_flush * 16u; cidx += 256u) {
            const uint32_t g   = cidx & 15u;
            const uint2    fe  = s_flush[cidx >> 4];
            const uint32_t idx = fe.x, v = fe.y;
            if (g < 9u) {
                // sums -> gradients: d/dmean = -(conic . (S hx, S hy)), d/dconic = (-1/2, -1, -1/2) (S hx dx, ...), the five of them
                // times the opacity (the sums were formed from q = h / opacity)
                float s = s_grad[g][idx];
                if (g < 5u) {
                    const float4 eb = s_rows[1][idx]; // -cc / 2, opacity, ..
                    if (g < 2u) {
                        const float4 ea = s_rows[0][idx];
                        const float  ca = -2.0f * ea.z, cc = -2.0f * eb.x, s0 = s_grad[0][idx], s1 = s_grad[1][idx];
                        s = (g == 0u) ? -(ca * s0 + ea.w * s1) : -(cc * s1 + ea.w * s0);
                    } else {
                        s *= (g == 3u) ? -1.0f : -0.5f;
                    }
                    s *= eb.y;
                }
                if (s != 0.0f) atomicAdd(&grads2d[(size_t)v * kG2D + g], s);
            }
        }
        hi = lo;
    }
#ifdef LCGS_BWD_STATS
    if (lane == 0u) {
        const unsigned long long n = st_[0] - st_[3]; // this tile's walked count for my strip (st_[3] = walked before this tile)
        uint32_t b = 0;
        while (b < 15u && (1ull << b) < n) ++b;
        atomicAdd(&g_bwd_stats[8 + b], 1ull);
        atomicAdd(&g_bwd_stats[3], n ? 1ull : 0ull);
        if (wave == 0u) atomicAdd(&g_bwd_stats[5], 1ull);
        // [24..27] sum over strips of the LONGEST sub-block stream (8x4 halves, 16x2 halves, 4x4 blocks, 16x1 rows);
        // [28] entries that blend anywhere; [29..31] sums of the sub-block streams (8x4, 4x4, 16x1)
        auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
        atomicAdd(&g_bwd_stats[24], (unsigned long long)mx(sub_[0], sub_[1]));
        atomicAdd(&g_bwd_stats[25], (unsigned long long)mx(sub_[2], sub_[3]));
        atomicAdd(&g_bwd_stats[26], (unsigned long long)mx(mx(sub_[4], sub_[5]), mx(sub_[6], sub_[7])));
        atomicAdd(&g_bwd_stats[27], (unsigned long long)mx(mx(sub_[8], sub_[9]), mx(sub_[10], sub_[11])));
        atomicAdd(&g_bwd_stats[28], (unsigned long long)sub_[12]);
        atomicAdd(&g_bwd_stats[29], (unsigned long long)(sub_[0] + sub_[1]));
        atomicAdd(&g_bwd_stats[30], (unsigned long long)(sub_[4] + sub_[5] + sub_[6] + sub_[7]));
        atomicAdd(&g_bwd_stats[31], (unsigned long long)(sub_[8] + sub_[9] + sub_[10] + sub_[11]));
    }
    st_[3] = st_[0];
#endif
    if (!PERSIST) break;
  }
#ifdef LCGS_BWD_STATS
    if (lane == 0u) {
        atomicAdd(&g_bwd_stats[0], st_[0]);
        atomicAdd(&g_bwd_stats[1], st_[1]);
        atomicAdd(&g_bwd_stats[2], st_[2]);
        if (wave == 0u) atomicAdd(&g_bwd_stats[4], st_[4]);
        atomicAdd(&g_bwd_stats[6], st_[6]);
        atomicAdd(&g_bwd_stats[7], st_[7]);
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// 2-D gradients -> parameter gradients, one lane per surviving splat.
// ---------------------------------------------------------------------------------------------------------------
// dL/d{pixel mean (gmx, gmy), conic (gA, gB, gC)} -> dL/d{pos (gp), scale (gs), rotq (gq: r,x,y,z)} through the EWA
// projection, in precision R.  The forward quantities are recomputed (the order of gs_math.hpp is irrelevant for the
// derivative).  cov_trace = a + c of the filtered 2-D covariance (>= its larger eigenvalue: the splat's footprint).
template <typename FP>
__device__ __forceinline__ void geom_backward_t(const CamParams& cp, FP scale_modifier, FP px, FP py, FP pz, FP sc0, FP sc1, FP sc2,
                                                FP qw_, FP qx_, FP qy_, FP qz_, FP gmx, FP gmy, FP gA, FP gB, FP gC, FP gp[3], FP gs[3],
                                                FP gq[4], FP& cov_trace)
{
    // ---- geometry: recompute the forward quantities (gs_math.hpp order is irrelevant for the derivative)
    FP v[3];
    v[0] = FP((cp.right[0])) * px + FP((cp.right[1])) * py + FP((cp.right[2])) * pz + FP(cp.tx);
    v[1] = FP((cp.up[0])) * px + FP((cp.up[1])) * py + FP((cp.up[2])) * pz + FP(cp.ty);
    v[2] = FP((cp.front[0])) * px + FP((cp.front[1])) * py + FP((cp.front[2])) * pz + FP(cp.tz);
    const FP limx = FP(1.3) * FP(cp.tanfovx), limy = FP(1.3) * FP(cp.tanfovy);
    const FP rx = v[0] / v[2], ry = v[1] / v[2];
    const int   clx = (rx < -limx) ? -1 : (rx > limx ? 1 : 0);
    const int   cly = (ry < -limy) ? -1 : (ry > limy ? 1 : 0);
    const FP tx = (clx ? FP(clx) * limx : rx) * v[2];
    const FP ty = (cly ? FP(cly) * limy : ry) * v[2];
    const FP tz = v[2];
    const FP sc[3] = { scale_modifier * sc0, scale_modifier * sc1, scale_modifier * sc2 };
    const FP x = qx_, y = qy_, z = qz_, w = qw_;
    FP Rm[3][3];
    Rm[0][0] = FP(1.0) - FP(2.0) * y * y - FP(2.0) * z * z; Rm[0][1] = FP(2.0) * x * y - FP(2.0) * z * w; Rm[0][2] = FP(2.0) * x * z + FP(2.0) * y * w;
    Rm[1][0] = FP(2.0) * x * y + FP(2.0) * z * w; Rm[1][1] = FP(1.0) - FP(2.0) * x * x - FP(2.0) * z * z; Rm[1][2] = FP(2.0) * y * z - FP(2.0) * x * w;
    Rm[2][0] = FP(2.0) * x * z - FP(2.0) * y * w; Rm[2][1] = FP(2.0) * y * z + FP(2.0) * x * w; Rm[2][2] = FP(1.0) - FP(2.0) * x * x - FP(2.0) * y * y;
    FP M[3][3], Sig[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) M[r][k] = Rm[r][k] * sc[k];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) Sig[r][k] = M[r][0] * M[k][0] + M[r][1] * M[k][1] + M[r][2] * M[k][2];
    const FP j00 = FP(cp.focalx) / tz, j11 = FP(cp.focaly) / tz, j02 = -FP(cp.focalx) * tx / (tz * tz),
                j12 = -FP(cp.focaly) * ty / (tz * tz);
    FP T0[3], T1[3], ST0[3], ST1[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        T0[r] = FP(cp.right[r]) * j00 + FP(cp.front[r]) * j02;
        T1[r] = FP(cp.up[r]) * j11 + FP(cp.front[r]) * j12;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        ST0[r] = Sig[r][0] * T0[0] + Sig[r][1] * T0[1] + Sig[r][2] * T0[2];
        ST1[r] = Sig[r][0] * T1[0] + Sig[r][1] * T1[1] + Sig[r][2] * T1[2];
    }
    const FP a = T0[0] * ST0[0] + T0[1] * ST0[1] + T0[2] * ST0[2] + FP(0.3);
    const FP b = T1[0] * ST0[0] + T1[1] * ST0[1] + T1[2] * ST0[2];
    const FP c = T1[0] * ST1[0] + T1[1] * ST1[1] + T1[2] * ST1[2] + FP(0.3);
    const FP D = a * c - b * b + FP(1e-6);
    const FP iD2 = FP(1.0) / (D * D);
    const FP g00 = (-c * c * gA + b * c * gB + (D - a * c) * gC) * iD2;
    const FP g11 = ((D - a * c) * gA + a * b * gB - a * a * gC) * iD2;
    const FP g01 = (FP(2.0) * b * c * gA - (D + FP(2.0) * b * b) * gB + FP(2.0) * a * b * gC) * iD2;
    FP Gm[3][3], dT0[3], dT1[3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) Gm[r][k] = g00 * T0[r] * T0[k] + g01 * T1[r] * T0[k] + g11 * T1[r] * T1[k];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        dT0[r] = FP(2.0) * g00 * ST0[r] + g01 * ST1[r];
        dT1[r] = FP(2.0) * g11 * ST1[r] + g01 * ST0[r];
    }
    const FP dj00 = FP(cp.right[0]) * dT0[0] + FP(cp.right[1]) * dT0[1] + FP(cp.right[2]) * dT0[2];
    const FP dj02 = FP(cp.front[0]) * dT0[0] + FP(cp.front[1]) * dT0[1] + FP(cp.front[2]) * dT0[2];
    const FP dj11 = FP(cp.up[0]) * dT1[0] + FP(cp.up[1]) * dT1[1] + FP(cp.up[2]) * dT1[2];
    const FP dj12 = FP(cp.front[0]) * dT1[0] + FP(cp.front[1]) * dT1[1] + FP(cp.front[2]) * dT1[2];
    const FP itz2 = FP(1.0) / (tz * tz), itz3 = itz2 / tz;
    const FP dtx = dj02 * (-FP(cp.focalx) * itz2);
    const FP dty = dj12 * (-FP(cp.focaly) * itz2);
    const FP dtz = dj00 * (-FP(cp.focalx) * itz2) + dj11 * (-FP(cp.focaly) * itz2) + dj02 * (FP(2.0) * FP(cp.focalx) * tx * itz3) +
                      dj12 * (FP(2.0) * FP(cp.focaly) * ty * itz3);
    FP dv[3];
    dv[0] = clx ? FP(0.0) : dtx;
    dv[1] = cly ? FP(0.0) : dty;
    dv[2] = dtz + (clx ? dtx * FP(clx) * limx : FP(0.0)) + (cly ? dty * FP(cly) * limy : FP(0.0));
    const FP pw = FP(1.0) / (v[2] + FP(1e-6));
    dv[0] += gmx * FP(cp.focalx) * pw;
    dv[1] += gmy * FP(cp.focaly) * pw;
    dv[2] += -(gmx * FP(cp.focalx) * v[0] + gmy * FP(cp.focaly) * v[1]) * pw * pw;
#pragma unroll
    for (int i = 0; i < 3; ++i) gp[i] = FP(cp.right[i]) * dv[0] + FP(cp.up[i]) * dv[1] + FP(cp.front[i]) * dv[2];

    FP dM[3][3], dRm[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k)
            dM[r][k] = (Gm[r][0] + Gm[0][r]) * M[0][k] + (Gm[r][1] + Gm[1][r]) * M[1][k] + (Gm[r][2] + Gm[2][r]) * M[2][k];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        gs[k] = scale_modifier * (dM[0][k] * Rm[0][k] + dM[1][k] * Rm[1][k] + dM[2][k] * Rm[2][k]);
#pragma unroll
        for (int r = 0; r < 3; ++r) dRm[r][k] = dM[r][k] * sc[k];
    }
    const FP gx_ = FP(2.0) * (y * (dRm[0][1] + dRm[1][0]) + z * (dRm[0][2] + dRm[2][0]) + w * (dRm[2][1] - dRm[1][2])) - FP(4.0) * x * (dRm[1][1] + dRm[2][2]);
    const FP gy_ = FP(2.0) * (x * (dRm[0][1] + dRm[1][0]) + z * (dRm[1][2] + dRm[2][1]) + w * (dRm[0][2] - dRm[2][0])) - FP(4.0) * y * (dRm[0][0] + dRm[2][2]);
    const FP gz_ = FP(2.0) * (x * (dRm[0][2] + dRm[2][0]) + y * (dRm[1][2] + dRm[2][1]) + w * (dRm[1][0] - dRm[0][1])) - FP(4.0) * z * (dRm[0][0] + dRm[1][1]);
    const FP gw_ = FP(2.0) * (z * (dRm[1][0] - dRm[0][1]) + y * (dRm[0][2] - dRm[2][0]) + x * (dRm[2][1] - dRm[1][2]));

    gq[0] = gw_; gq[1] = gx_; gq[2] = gy_; gq[3] = gz_; // (r, x, y, z)
    cov_trace = a + c;
}

// A footprint beyond this (trace of the 2-D covariance, px^2: radius ~ 3 sqrt(lambda_max) > 64 px) takes the algebra in f64.
constexpr float kGiantCovTrace = 455.0f;
// (Both precisions are inlined: the kernels' register count doubles and their occupancy halves -- preprocess-backward
//  0.181 -> 0.185 ms on the bicycle stand-in.  Holding them to the f32 occupancy with __launch_bounds__ spills the f64 branch
//  to scratch and costs 0.05 ms: measured, gpurun_out/r4_ab_f64.log.)

// The f32 algebra, and -- for screen-filling splats only -- the same algebra again in f64.  Their 2-D covariance is ~1e5 and
// their conic ~1e-6: conic -> covariance -> Sigma -> scale / quaternion multiplies sums that cancel to a 1e-3..1e-5 of their
// terms, and ANY f32 evaluation loses them (the f32 CPU restatement is off by up to 1.7e-1 on such rows; the 2-D gradients
// feeding this step are good to ~1e-4: profiles/r04_gradient_error_survey.txt).  0.4 % of the on-screen splats of the
// bicycle stand-in qualify; the kernels calling this are HBM-bound, the divergent f64 pass hides under their stores.
__device__ __forceinline__ void geom_backward(const CamParams& cp, float scale_modifier, float px, float py, float pz,
                                              float sc0, float sc1, float sc2, float4 q, float gmx, float gmy, float gA,
                                              float gB, float gC, float gp[3], float gs[3], float4& gq)
{
    float dp[3], q4[4], tr;
    geom_backward_t<float>(cp, scale_modifier, px, py, pz, sc0, sc1, sc2, q.x, q.y, q.z, q.w, gmx, gmy, gA, gB, gC, dp, gs, q4, tr);
    if (tr > kGiantCovTrace) {
        double dpd[3], gsd[3], q4d[4], trd;
        geom_backward_t<double>(cp, scale_modifier, px, py, pz, sc0, sc1, sc2, q.x, q.y, q.z, q.w, gmx, gmy, gA, gB, gC, dpd, gsd,
                                q4d, trd);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            dp[i] = (float)dpd[i];
            gs[i] = (float)gsd[i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) q4[i] = (float)q4d[i];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) gp[i] += dp[i];
    gq = make_float4(q4[0], q4[1], q4[2], q4[3]);
}

// One lane per surviving splat (dense ids).  The splat's 48 SH coefficients arrive through the wave's LDS slab
// (cooperative 16-byte loads, 12 lanes per 192-byte row), the SH gradient row is written back IN PLACE into the
// slab as it is produced (so it never lives in registers next to the geometry Jacobians) and leaves as coalesced
// 16-byte stores.
// the four geometry rows of one splat; accumulate: add to what the arrays hold (a further view of a multi-view batch)
__device__ __forceinline__ void store_geometry_rows(size_t orow, bool accumulate, const float gp[3], const float gs[3],
                                                    float4 gq, float gop, float* __restrict__ dL_dpos,
                                                    float* __restrict__ dL_dscale, float* __restrict__ dL_drotq,
                                                    float* __restrict__ dL_dopacity)
{
    float4* rq = reinterpret_cast<float4*>(dL_drotq + 4 * orow); // (r,x,y,z)
    if (accumulate) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            dL_dpos[3 * orow + i] += gp[i];
            dL_dscale[3 * orow + i] += gs[i];
        }
        const float4 o = *rq;
        *rq            = make_float4(o.x + gq.x, o.y + gq.y, o.z + gq.z, o.w + gq.w);
        dL_dopacity[orow] += gop;
    } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            dL_dpos[3 * orow + i]   = gp[i];
            dL_dscale[3 * orow + i] = gs[i];
        }
        *rq               = gq;
        dL_dopacity[orow] = gop;
    }
}

__global__ void __launch_bounds__(256)
k_preprocess_backward(int sh_deg, CamParams cp, float scale_modifier, const float* __restrict__ pos,
                      const float* __restrict__ scale, const float* __restrict__ rotq, const float* __restrict__ sh,
                      const uint32_t* __restrict__ vis_index, const uint32_t* __restrict__ d_counts,
                      const float* __restrict__ grads2d, float* __restrict__ dL_dpos, float* __restrict__ dL_dscale,
                      float* __restrict__ dL_drotq, float* __restrict__ dL_dsh, float* __restrict__ dL_dopacity,
                      int mode, const uint32_t* __restrict__ slice_bounds, int slice)
{
    __shared__ float4 s_sh[4][64 * 13];
    const bool compact = (mode & 1) != 0, accumulate = (mode & 2) != 0; // (see launch_preprocess_backward)
    // the survivors (dense ids) this launch covers: all of them, or slice `slice` of a splat-range split (k_slice_bounds)
    const uint32_t v0 = slice_bounds ? slice_bounds[slice] : 0u;
    const uint32_t V  = slice_bounds ? slice_bounds[slice + 1] : d_counts[0];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int feat = (sh_deg + 1) * (sh_deg + 1);
    const bool staged = sh_deg == 3 && ((reinterpret_cast<uintptr_t>(sh) & 15) == 0) &&
                        ((reinterpret_cast<uintptr_t>(dL_dsh) & 15) == 0);
    for (uint32_t blk = blockIdx.x; v0 + blk * 256u < V; blk += gridDim.x) {
        const uint32_t vid   = v0 + blk * 256u + threadIdx.x;
        const bool     valid = vid < V;
        const int      idx   = (int)vis_index[valid ? vid : V - 1];
        const uint32_t wave_first = v0 + blk * 256u + wave * 64u;
        const uint32_t nvalid     = wave_first < V ? ((V - wave_first) < 64u ? (V - wave_first) : 64u) : 0u;
        float*         row = reinterpret_cast<float*>(&s_sh[wave][lane * 13]); // this lane's 48 floats (+4 pad)
        // Every global operand of this splat is requested here, together with the SH rows below: one memory round
        // trip per block instead of one per use (lanes past V read the last survivor's rows and discard them).
        const float* g2 = grads2d + (size_t)(valid ? vid : V - 1) * kG2D;
        const float4 q0 = reinterpret_cast<const float4*>(g2)[0], q1 = reinterpret_cast<const float4*>(g2)[1];
        const float  gcol2 = g2[8];
        const float  px = pos[3 * (size_t)idx + 0], py = pos[3 * (size_t)idx + 1], pz = pos[3 * (size_t)idx + 2];
        const float  sc0 = scale[3 * (size_t)idx + 0], sc1 = scale[3 * (size_t)idx + 1], sc2 = scale[3 * (size_t)idx + 2];
        const float4 q = *reinterpret_cast<const float4*>(rotq + 4 * (size_t)idx); // (r,x,y,z)

        // ---- stage the SH rows (coalesced), or fetch them lane-wise for other degrees
        __syncthreads();
        if (staged) {
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const uint32_t c    = (uint32_t)i * 64u + lane;
                const uint32_t slot = c / 12u, part = c - slot * 12u;
                const int      sidx = __shfl(idx, (int)slot, 64);
                if (slot < nvalid) s_sh[wave][slot * 13u + part] = reinterpret_cast<const float4*>(sh + (size_t)sidx * 48)[part];
            }
        } else if (valid) {
            const float* s = sh + (size_t)idx * feat * 3;
            for (int k = 0; k < 48; ++k) row[k] = k < feat * 3 ? s[k] : 0.0f;
        }
        __syncthreads();

        if (valid) {
            const float  gmx = q0.x, gmy = q0.y, gA = q0.z, gB = q0.w, gC = q1.x, gop = q1.y;
            const float  gcol[3] = { q1.z, q1.w, gcol2 };
            float        gp[3] = { 0.0f, 0.0f, 0.0f };

            // ---- colour -> SH coefficients and position (through the view direction)
            {
                const float dx = px - cp.campos[0], dy = py - cp.campos[1], dz = pz - cp.campos[2];
                const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
                const float x = dx * inv, y = dy * inv, z = dz * inv;
                const float xx = x * x, yy = y * y, zz = z * z;
                float raw[3] = { 0.5f, 0.5f, 0.5f };
#define LCGS_RAW(k, B, DX, DY, DZ)                                                                                    \
    if (k < feat) {                                                                                                   \
        const float bk = (B);                                                                                         \
        raw[0] += bk * row[k * 3 + 0];                                                                                \
        raw[1] += bk * row[k * 3 + 1];                                                                                \
        raw[2] += bk * row[k * 3 + 2];                                                                                \
    }
                LCGS_SH_TERMS(LCGS_RAW)
#undef LCGS_RAW
                float g[3];
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) g[ch] = (raw[ch] > 0.0f && raw[ch] < 1.0f) ? gcol[ch] : 0.0f; // clamp mask
                float ddx = 0.0f, ddy = 0.0f, ddz = 0.0f;
#define LCGS_GRAD(k, B, DX, DY, DZ)                                                                                   \
    {                                                                                                                 \
        const float c0 = row[k * 3 + 0], c1 = row[k * 3 + 1], c2 = row[k * 3 + 2];                                    \
        const float bk = (k < feat) ? (B) : 0.0f;                                                                     \
        const float wk = (k < feat) ? g[0] * c0 + g[1] * c1 + g[2] * c2 : 0.0f;                                       \
        ddx += wk * (DX);                                                                                             \
        ddy += wk * (DY);                                                                                             \
        ddz += wk * (DZ);                                                                                             \
        row[k * 3 + 0] = bk * g[0];                                                                                   \
        row[k * 3 + 1] = bk * g[1];                                                                                   \
        row[k * 3 + 2] = bk * g[2];                                                                                   \
    }
                LCGS_SH_TERMS(LCGS_GRAD)
#undef LCGS_GRAD
                const float dd = x * ddx + y * ddy + z * ddz;
                gp[0] += (ddx - x * dd) * inv;
                gp[1] += (ddy - y * dd) * inv;
                gp[2] += (ddz - z * dd) * inv;
            }

            float  gs[3];
            float4 gq;
            geom_backward(cp, scale_modifier, px, py, pz, sc0, sc1, sc2, q, gmx, gmy, gA, gB, gC, gp, gs, gq);

            const size_t orow = compact ? (size_t)vid : (size_t)idx; // compact: row = dense id (see launch.hpp)
            store_geometry_rows(orow, accumulate, gp, gs, gq, gop, dL_dpos, dL_dscale, dL_drotq, dL_dopacity);
        }

        // ---- SH gradient rows: 12 consecutive lanes write one splat's 192 contiguous bytes
        __syncthreads();
        if (staged) {
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const uint32_t cidx = (uint32_t)i * 64u + lane;
                const uint32_t slot = cidx / 12u, part = cidx - slot * 12u;
                const int      sidx = __shfl(idx, (int)slot, 64);
                const size_t   orow = compact ? (size_t)(wave_first + slot) : (size_t)sidx;
                if (slot < nvalid) {
                    float4* dst = reinterpret_cast<float4*>(dL_dsh + orow * 48) + part;
                    float4  v   = s_sh[wave][slot * 13u + part];
                    if (accumulate) {
                        const float4 o = *dst;
                        v = make_float4(o.x + v.x, o.y + v.y, o.z + v.z, o.w + v.w);
                    }
                    *dst = v;
                }
            }
        } else if (valid) {
            float* o = dL_dsh + (compact ? (size_t)vid : (size_t)idx) * feat * 3;
            for (int k = 0; k < feat * 3; ++k) o[k] = accumulate ? o[k] + row[k] : row[k];
        }
    }
}

// The same step for frames whose forward kept the colour Jacobian (build_records<.., JAC>; degree 3): nothing of the
// 192-byte coefficient row is read again.  dL/dsh[k][c] = basis_k(direction) * dL/dcolour[c] (clamp-masked) is an
// outer product of 16 + 3 numbers per splat: each lane parks those 19 floats in LDS and the wave writes the 192-byte
// gradient rows cooperatively (12 lanes x 16 B per row), forming the products on the way out.  The direction part of
// dL/dpos comes from the kept 3x3 Jacobian.  No 53 KB slab, a third of the registers: more workgroups in flight on
// a kernel that is bound by bytes in flight.
constexpr int kJacPitch = 19; // floats per splat in the LDS slab (odd: conflict-free per-lane writes)

__global__ void __launch_bounds__(256)
k_preprocess_backward_jac(CamParams cp, float scale_modifier, const float* __restrict__ pos,
                          const float* __restrict__ scale, const float* __restrict__ rotq,
                          const uint32_t* __restrict__ vis_index, const uint32_t* __restrict__ d_counts,
                          const float* __restrict__ grads2d, const float4* __restrict__ shjac,
                          float* __restrict__ dL_dpos, float* __restrict__ dL_dscale, float* __restrict__ dL_drotq,
                          float* __restrict__ dL_dsh, float* __restrict__ dL_dopacity, int mode,
                          const uint32_t* __restrict__ slice_bounds, int slice)
{
    __shared__ float s_outer[4][64 * kJacPitch];
    const bool compact = (mode & 1) != 0, accumulate = (mode & 2) != 0;
    const uint32_t v0 = slice_bounds ? slice_bounds[slice] : 0u; // (see k_preprocess_backward)
    const uint32_t V  = slice_bounds ? slice_bounds[slice + 1] : d_counts[0];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t blk = blockIdx.x; v0 + blk * 256u < V; blk += gridDim.x) {
        const uint32_t vid   = v0 + blk * 256u + threadIdx.x;
        const bool     valid = vid < V;
        const uint32_t vsafe = valid ? vid : V - 1;
        const int      idx   = (int)vis_index[vsafe];
        const uint32_t wave_first = v0 + blk * 256u + wave * 64u;
        const uint32_t nvalid     = wave_first < V ? ((V - wave_first) < 64u ? (V - wave_first) : 64u) : 0u;
        const float4* g2 = reinterpret_cast<const float4*>(grads2d + (size_t)vsafe * kG2D);
        const float4  q0 = g2[0], q1 = g2[1];
        const float   gcol2 = reinterpret_cast<const float*>(g2)[8];
        const float4  j0 = shjac[(size_t)vsafe * 3 + 0], j1 = shjac[(size_t)vsafe * 3 + 1], j2 = shjac[(size_t)vsafe * 3 + 2];
        const float   px = pos[3 * (size_t)idx + 0], py = pos[3 * (size_t)idx + 1], pz = pos[3 * (size_t)idx + 2];
        const float   sc0 = scale[3 * (size_t)idx + 0], sc1 = scale[3 * (size_t)idx + 1], sc2 = scale[3 * (size_t)idx + 2];
        const float4  q = *reinterpret_cast<const float4*>(rotq + 4 * (size_t)idx); // (r,x,y,z)
        float*        mine = &s_outer[wave][lane * kJacPitch];
        if (valid) {
            const float    gmx = q0.x, gmy = q0.y, gA = q0.z, gB = q0.w, gC = q1.x, gop = q1.y;
            const float    gcol[3] = { q1.z, q1.w, gcol2 };
            const uint32_t mask    = __float_as_uint(j2.y);
            float          gp[3];
            {
                const float dx = px - cp.campos[0], dy = py - cp.campos[1], dz = pz - cp.campos[2];
                const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
                const float x = dx * inv, y = dy * inv, z = dz * inv;
                const float xx = x * x, yy = y * y, zz = z * z;
#define LCGS_BASIS(k, B, DX, DY, DZ) mine[k] = (B);
                LCGS_SH_TERMS(LCGS_BASIS)
#undef LCGS_BASIS
#pragma unroll
                for (int c = 0; c < 3; ++c) mine[16 + c] = ((mask >> c) & 1u) ? gcol[c] : 0.0f; // clamp mask
                // J rows of clamped channels are already zero: no mask needed here
                const float ddx = gcol[0] * j0.x + gcol[1] * j0.w + gcol[2] * j1.z;
                const float ddy = gcol[0] * j0.y + gcol[1] * j1.x + gcol[2] * j1.w;
                const float ddz = gcol[0] * j0.z + gcol[1] * j1.y + gcol[2] * j2.x;
                const float dd  = x * ddx + y * ddy + z * ddz;
                gp[0] = (ddx - x * dd) * inv;
                gp[1] = (ddy - y * dd) * inv;
                gp[2] = (ddz - z * dd) * inv;
            }
            float  gs[3];
            float4 gq;
            geom_backward(cp, scale_modifier, px, py, pz, sc0, sc1, sc2, q, gmx, gmy, gA, gB, gC, gp, gs, gq);
            const size_t orow = compact ? (size_t)vid : (size_t)idx; // compact: row = dense id (see launch.hpp)
            store_geometry_rows(orow, accumulate, gp, gs, gq, gop, dL_dpos, dL_dscale, dL_drotq, dL_dopacity);
        }
        __syncthreads();
        // ---- SH gradient rows: 12 consecutive lanes write one splat's 192 contiguous bytes
#pragma unroll 1
        for (int i = 0; i < 12; ++i) {
            const uint32_t cidx = (uint32_t)i * 64u + lane;
            const uint32_t slot = cidx / 12u, part = cidx - slot * 12u;
            const int      sidx = __shfl(idx, (int)slot, 64);
            if (slot < nvalid) {
                const float* o = &s_outer[wave][slot * kJacPitch];
                float        v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t f = part * 4u + (uint32_t)e, k = f / 3u, ch = f - 3u * k; // row[k * 3 + ch]
                    v[e]             = o[k] * o[16u + ch];
                }
                const size_t orow = compact ? (size_t)(wave_first + slot) : (size_t)sidx;
                float4*      dst  = reinterpret_cast<float4*>(dL_dsh + orow * 48) + part;
                if (accumulate) {
                    const float4 o = *dst;
                    v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w;
                }
                // written once, read by the optimiser / the all-reduce a kernel later: a streaming store
                st_stream(dst, make_float4(v[0], v[1], v[2], v[3]));
            }
        }
        __syncthreads(); // the slab is reused by the next iteration
    }
}

// The same per-splat pass with the optimiser folded in (single-GPU training steps: lcgs_render_backward_adam).  The
// gradients of a splat exist only in this lane's registers (geometry, opacity) and in the wave's LDS slab (the 16 + 3
// factors of the SH outer product): the on-screen-only Adam update of train.hip -- the same adam_update, the same chain
// rules, the same order of operations, so the result is train.hip's bit for bit -- is applied right here and NO gradient
// row is ever written or read back (2 x 236 bytes per on-screen splat less; the separate step was 0.56 GB written by this
// kernel + 0.56 GB read by the optimiser's on the bicycle stand-in).  Each lane owns its splat's rows in raw / m / v /
// activated: read, updated, written by the same lane, so the in-place update of the arrays this very kernel reads
// (the activated pos / scale / rotq are the renderer's scene arrays) needs no synchronisation.
__global__ void __launch_bounds__(256)
k_preprocess_backward_adam(CamParams cp, float scale_modifier, const float* pos, const float* scale, const float* rotq,
                           const uint32_t* __restrict__ vis_index, const uint32_t* __restrict__ d_counts,
                           const float* __restrict__ grads2d, const float4* __restrict__ shjac, AdamArrays raw, AdamArrays am,
                           AdamArrays av, AdamArrays act, AdamRates lr, AdamStep a)
{
    __shared__ float s_outer[4][64 * kJacPitch];
    const uint32_t V = d_counts[0];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t blk = blockIdx.x; blk * 256u < V; blk += gridDim.x) {
        const uint32_t vid   = blk * 256u + threadIdx.x;
        const bool     valid = vid < V;
        const uint32_t vsafe = valid ? vid : V - 1;
        const int      idx   = (int)vis_index[vsafe];
        const uint32_t wave_first = blk * 256u + wave * 64u;
        const uint32_t nvalid     = wave_first < V ? ((V - wave_first) < 64u ? (V - wave_first) : 64u) : 0u;
        const float4* g2 = reinterpret_cast<const float4*>(grads2d + (size_t)vsafe * kG2D);
        const float4  q0 = g2[0], q1 = g2[1];
        const float   gcol2 = reinterpret_cast<const float*>(g2)[8];
        const float4  j0 = shjac[(size_t)vsafe * 3 + 0], j1 = shjac[(size_t)vsafe * 3 + 1], j2 = shjac[(size_t)vsafe * 3 + 2];
        const float   px = pos[3 * (size_t)idx + 0], py = pos[3 * (size_t)idx + 1], pz = pos[3 * (size_t)idx + 2];
        const float   sc0 = scale[3 * (size_t)idx + 0], sc1 = scale[3 * (size_t)idx + 1], sc2 = scale[3 * (size_t)idx + 2];
        const float4  q = *reinterpret_cast<const float4*>(rotq + 4 * (size_t)idx); // (r,x,y,z)
        float*        mine = &s_outer[wave][lane * kJacPitch];
        if (valid) {
            const float    gmx = q0.x, gmy = q0.y, gA = q0.z, gB = q0.w, gC = q1.x, gop = q1.y;
            const float    gcol[3] = { q1.z, q1.w, gcol2 };
            const uint32_t mask    = __float_as_uint(j2.y);
            float          gp[3];
            {
                const float dx = px - cp.campos[0], dy = py - cp.campos[1], dz = pz - cp.campos[2];
                const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
                const float x = dx * inv, y = dy * inv, z = dz * inv;
                const float xx = x * x, yy = y * y, zz = z * z;
#define LCGS_BASIS(k, B, DX, DY, DZ) mine[k] = (B);
                LCGS_SH_TERMS(LCGS_BASIS)
#undef LCGS_BASIS
#pragma unroll
                for (int c = 0; c < 3; ++c) mine[16 + c] = ((mask >> c) & 1u) ? gcol[c] : 0.0f; // clamp mask
                const float ddx = gcol[0] * j0.x + gcol[1] * j0.w + gcol[2] * j1.z;
                const float ddy = gcol[0] * j0.y + gcol[1] * j1.x + gcol[2] * j1.w;
                const float ddz = gcol[0] * j0.z + gcol[1] * j1.y + gcol[2] * j2.x;
                const float dd  = x * ddx + y * ddy + z * ddz;
                gp[0] = (ddx - x * dd) * inv;
                gp[1] = (ddy - y * dd) * inv;
                gp[2] = (ddz - z * dd) * inv;
            }
            float  gs[3];
            float4 gq;
            geom_backward(cp, scale_modifier, px, py, pz, sc0, sc1, sc2, q, gmx, gmy, gA, gB, gC, gp, gs, gq);
            // ---- Adam on this splat's geometry rows (train.hip: k_adam_rows<3,0>, <3,1>, k_adam_rot, <1,2>)
            const size_t i3 = 3 * (size_t)idx;
#pragma unroll
            for (int c = 0; c < 3; ++c) { // pos: raw == activated value
                float       mm = am.pos[i3 + c], vv = av.pos[i3 + c];
                const float x  = raw.pos[i3 + c] - adam_update(gp[c], mm, vv, lr.pos, a);
                am.pos[i3 + c]  = mm;
                av.pos[i3 + c]  = vv;
                raw.pos[i3 + c] = x;
                if (act.pos != raw.pos) act.pos[i3 + c] = x;
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) { // scale = exp(raw): g_raw = g * s
                const float g  = gs[c] * act.scale[i3 + c];
                float       mm = am.scale[i3 + c], vv = av.scale[i3 + c];
                const float x  = raw.scale[i3 + c] - adam_update(g, mm, vv, lr.scale, a);
                am.scale[i3 + c]  = mm;
                av.scale[i3 + c]  = vv;
                raw.scale[i3 + c] = x;
                act.scale[i3 + c] = expf(x);
            }
            { // rotq = raw / |raw|: g_raw = (g - q (q . g)) / |raw|   (gq is (r,x,y,z), like the rows)
                float4*      rr = reinterpret_cast<float4*>(raw.rotq) + idx;
                float4*      rm = reinterpret_cast<float4*>(am.rotq) + idx;
                float4*      rv = reinterpret_cast<float4*>(av.rotq) + idx;
                float4*      ra = reinterpret_cast<float4*>(act.rotq) + idx;
                const float4 g = gq, qa = *ra;
                float4       x = *rr, mm = *rm, vv = *rv;
                const float  inv_norm = 1.0f / sqrtf(x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w);
                const float  qg       = qa.x * g.x + qa.y * g.y + qa.z * g.z + qa.w * g.w;
                x.x -= adam_update((g.x - qa.x * qg) * inv_norm, mm.x, vv.x, lr.rot, a);
                x.y -= adam_update((g.y - qa.y * qg) * inv_norm, mm.y, vv.y, lr.rot, a);
                x.z -= adam_update((g.z - qa.z * qg) * inv_norm, mm.z, vv.z, lr.rot, a);
                x.w -= adam_update((g.w - qa.w * qg) * inv_norm, mm.w, vv.w, lr.rot, a);
                const float n2 = 1.0f / sqrtf(x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w);
                *rr = x;
                *rm = mm;
                *rv = vv;
                *ra = make_float4(x.x * n2, x.y * n2, x.z * n2, x.w * n2);
            }
            { // opacity = sigmoid(raw): g_raw = g o (1 - o)
                const float o  = act.opacity[idx];
                const float g  = gop * o * (1.0f - o);
                float       mm = am.opacity[idx], vv = av.opacity[idx];
                const float x  = raw.opacity[idx] - adam_update(g, mm, vv, lr.opacity, a);
                am.opacity[idx]  = mm;
                av.opacity[idx]  = vv;
                raw.opacity[idx] = x;
                act.opacity[idx] = 1.0f / (1.0f + expf(-x));
            }
        }
        __syncthreads();
        // ---- SH rows: 12 consecutive lanes own one splat's 192 contiguous bytes of raw / m / v (train.hip: k_adam_sh48)
#pragma unroll 1
        for (int i = 0; i < 12; ++i) {
            const uint32_t cidx = (uint32_t)i * 64u + lane;
            const uint32_t slot = cidx / 12u, part = cidx - slot * 12u;
            const int      sidx = __shfl(idx, (int)slot, 64);
            if (slot < nvalid) {
                const float* o = &s_outer[wave][slot * kJacPitch];
                float        g[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t f = part * 4u + (uint32_t)e, k = f / 3u, ch = f - 3u * k; // row[k * 3 + ch]
                    g[e]             = o[k] * o[16u + ch];
                }
                const size_t r4 = (size_t)sidx * 12 + part;
                float4*      xr = reinterpret_cast<float4*>(raw.sh) + r4;
                float4*      xm = reinterpret_cast<float4*>(am.sh) + r4;
                float4*      xv = reinterpret_cast<float4*>(av.sh) + r4;
                float4       x = *xr, mm = *xm, vv = *xv;
                const float  l = part == 0u ? lr.sh_dc : lr.sh_rest; // floats 0..2 of a row are the dc band
                x.x -= adam_update(g[0], mm.x, vv.x, l, a);
                x.y -= adam_update(g[1], mm.y, vv.y, l, a);
                x.z -= adam_update(g[2], mm.z, vv.z, l, a);
                x.w -= adam_update(g[3], mm.w, vv.w, lr.sh_rest, a);
                *xr = x;
                *xm = mm;
                *xv = vv;
                if (act.sh != raw.sh) reinterpret_cast<float4*>(act.sh)[r4] = x;
            }
        }
        __syncthreads(); // the slab is reused by the next iteration
    }
}

} // namespace

size_t grads2d_bytes(int64_t V_cap) { return (size_t)V_cap * kG2D * sizeof(float); }

namespace
{
__global__ void __launch_bounds__(256) k_zero_grads2d(const uint32_t* __restrict__ d_counts, float4* __restrict__ g,
                                                      uint32_t* __restrict__ bwd_counter)
{
    if (bwd_counter && blockIdx.x == 0 && threadIdx.x == 0) *bwd_counter = 0u; // the persistent render-backward's tile counter
    const size_t n = (size_t)d_counts[0] * (kG2D / 4);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        g[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}
} // namespace

void launch_zero_grads2d(const uint32_t* d_counts, float* grads2d, hipStream_t stream, uint32_t* bwd_counter)
{
    hipLaunchKernelGGL(k_zero_grads2d, dim3(2048), dim3(256), 0, stream, d_counts, reinterpret_cast<float4*>(grads2d), bwd_counter);
}

void launch_render_backward(const CamParams& cp, const float bg[3], const uint32_t* ranges, const uint32_t* point_list,
                            const SplatRecord* recs, const float* final_T, const uint32_t* n_contrib,
                            const float* dL_dimg, float* grads2d, const uint32_t* tile_order, hipStream_t stream,
                            const uint8_t* strip_masks, const uint32_t* d_counts, uint32_t* work_counter,
                            uint32_t persistent_wgs, const DenseFill* fill_)
{
    if (cp.grid_x * cp.grid_y == 0) return;
    const DenseFill fill = fill_ ? *fill_ : DenseFill{};
    const uint32_t full = render_grid_size(cp.grid_x, cp.grid_y);
#define LCGS_LAUNCH_BWD(KNOWN_, PERSIST_, GRID_)                                                                             \
    hipLaunchKernelGGL((k_render_backward<KNOWN_, PERSIST_>), dim3(GRID_), dim3(256), 0, stream, cp, bg[0], bg[1], bg[2], ranges, \
                       point_list, recs, final_T, n_contrib, dL_dimg, grads2d, tile_order, strip_masks, d_counts, work_counter, fill)
    if (work_counter && persistent_wgs > 0 && persistent_wgs < full) {
        if (strip_masks) LCGS_LAUNCH_BWD(true, true, persistent_wgs);
        else LCGS_LAUNCH_BWD(false, true, persistent_wgs);
    } else {
        work_counter = nullptr;
        if (strip_masks) LCGS_LAUNCH_BWD(true, false, full);
        else LCGS_LAUNCH_BWD(false, false, full);
    }
#undef LCGS_LAUNCH_BWD
}

// Splat-range slices of the survivors (for the chunked gradient all-reduce, lcgs_grads_allreduce): the dense ids are
// handed out in ascending splat index, so the survivors whose splat index lies in [k P / K, (k + 1) P / K) are the
// dense ids [bounds[k], bounds[k + 1]).  One lane per boundary: lower_bound over vis_index[0, V).
__global__ void __launch_bounds__(64) k_slice_bounds(const uint32_t* __restrict__ vis_index,
                                                     const uint32_t* __restrict__ d_counts, uint32_t P, int slices,
                                                     uint32_t* __restrict__ bounds)
{
    const int k = threadIdx.x;
    if (k > slices) return;
    const uint32_t V = d_counts[0];
    uint32_t       lo = 0, hi = V; // first dense id whose splat index >= target
    if (k == slices) lo = V;
    else {
        const uint32_t target = (uint32_t)(((uint64_t)P * (uint64_t)k) / (uint64_t)slices);
        while (lo < hi) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (vis_index[mid] < target) lo = mid + 1;
            else hi = mid;
        }
    }
    bounds[k] = lo;
}

void launch_slice_bounds(const uint32_t* vis_index, const uint32_t* d_counts, int64_t P, int slices, uint32_t* bounds,
                         hipStream_t stream)
{
    hipLaunchKernelGGL(k_slice_bounds, dim3(1), dim3(64), 0, stream, vis_index, d_counts, (uint32_t)P, slices, bounds);
}

void launch_preprocess_backward(int64_t v_hint, int sh_deg, const CamParams& cp, float scale_modifier, const float* pos,
                                const float* scale, const float* rotq, const float* sh, const uint32_t* vis_index,
                                const uint32_t* d_counts, const float* grads2d, float* dL_dpos, float* dL_dscale,
                                float* dL_drotq, float* dL_dsh, float* dL_dopacity, hipStream_t stream,
                                const float4* shjac, bool compact, const uint32_t* slice_bounds, int slice, int slices,
                                bool accumulate)
{
    const int mode = (compact ? 1 : 0) | (accumulate ? 2 : 0);
    // (a slice's launch is sized for its share of the hint; larger live counts are strided)
    int64_t blocks = (v_hint / (slice_bounds ? slices : 1) + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 65536) blocks = 65536;
    if (shjac && sh_deg == 3 && (reinterpret_cast<uintptr_t>(dL_dsh) & 15) == 0) {
        hipLaunchKernelGGL(k_preprocess_backward_jac, dim3((unsigned)blocks), dim3(256), 0, stream, cp, scale_modifier, pos,
                           scale, rotq, vis_index, d_counts, grads2d, shjac, dL_dpos, dL_dscale, dL_drotq, dL_dsh,
                           dL_dopacity, mode, slice_bounds, slice);
        return;
    }
    hipLaunchKernelGGL(k_preprocess_backward, dim3((unsigned)blocks), dim3(256), 0, stream, sh_deg, cp, scale_modifier,
                       pos, scale, rotq, sh, vis_index, d_counts, grads2d, dL_dpos, dL_dscale, dL_drotq, dL_dsh,
                       dL_dopacity, mode, slice_bounds, slice);
}

void launch_preprocess_backward_adam(int64_t v_hint, const CamParams& cp, float scale_modifier, const float* pos,
                                     const float* scale, const float* rotq, const uint32_t* vis_index, const uint32_t* d_counts,
                                     const float* grads2d, const float4* shjac, const AdamArrays& raw, const AdamArrays& m,
                                     const AdamArrays& v, const AdamArrays& act, const AdamRates& lr, const AdamStep& a,
                                     hipStream_t stream)
{
    int64_t blocks = (v_hint + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_preprocess_backward_adam, dim3((unsigned)blocks), dim3(256), 0, stream, cp, scale_modifier, pos, scale,
                       rotq, vis_index, d_counts, grads2d, shjac, raw, m, v, act, lr, a);
}

} // namespace lcgs

#ifdef LCGS_BWD_STATS
// (measuring builds only; not declared in include/lcgs_hip.h)
extern "C" __attribute__((visibility("default"))) int lcgs_debug_bwd_stats(unsigned long long* out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(lcgs::g_bwd_stats), sizeof(unsigned long long) * 32) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[32] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(lcgs::g_bwd_stats), z, sizeof z) != hipSuccess) return 1;
    }
    return 0;
}
#endif
