// render.hip -- per-tile front-to-back alpha compositing
// (m_forward_render_shader, lcgs/src/gs_tile_splatter/shader.cpp:171-288).
//
// CDNA4 shape (not the reference's 256-thread block in which every thread walks every entry):
//   * k_render_forward_b (the renderer in use): one workgroup of four wave64s per 16x16 tile, wave k owns the 8x8
//     pixel quadrant k, one pixel per lane (rounds 1-5a: the 16x4 strip k -- "strip" in names and comments below means a
//     wave's 64 pixels; a compact unit meets ~10 % fewer splats: tile_common.hpp, profiles/r05_unit_quads_ab.txt).  A round stages 256 list entries: lane l gathers entry l's 36-byte record
//     with wide loads, tests it against the four strips (exact "can this splat reach the strip" test) and parks it
//     in three 16-byte-pitch LDS slabs; the per-strip ballots go to LDS and wave k then walks only the set bits of
//     "its" masks with scalar bit scans, reading each entry back as wave-uniform (broadcast, conflict-free) LDS
//     reads.  The colour comes from LDS too -- the reference re-fetches it from global memory per pixel per
//     contributing splat (shader.cpp:268-269).  Details and measurements: DESIGN.md 4.
//   * strip-level early out: a finished pixel carries a NaN coordinate (it fails every later test by itself); a
//     strip whose pixels are all finished retires, a tile whose strips are all retired stops staging (the
//     reference's "collect num_done" at shader.cpp:229 has no code behind it, so every tile walks its whole list).
//   * workgroup -> tile: longest-list-first (k_tile_order, from the previous frame's list lengths) when the caller
//     supplies an order, else XCD-aware blocks of 8x4 tiles dealt round-robin to the 8 XCDs (tile_of_workgroup).
//   (A one-wave64-per-tile variant was kept as a tuning hook through round 1; measured slower at every size, removed.)
// Numerics: the per-pixel expressions keep the reference's evaluation order with no FMA contraction
// (-ffp-contract=off); exp() is blend_exp (gs_math.hpp): a fixed sequence of IEEE operations the tests' CPU
// restatement repeats, so the two images are equal bit for bit (v_exp_f32 differs from any CPU exp by an ulp or two, and an ulp flips the
// hard thresholds behind it).
#include <stdlib.h>

#include <hip/hip_ext.h>

#include "launch.hpp"
#include "tile_common.hpp"

namespace lcgs
{
namespace
{

struct FetchAoS {
    const float* __restrict__ means_2d; // 2P, pixel
    const float* __restrict__ conic;    // 3P
    const float* __restrict__ opacity;  // P
    const float* __restrict__ color;    // 3P
    __device__ __forceinline__ void operator()(uint32_t id, float4& a, float4& b, float& c) const
    {
        a = make_float4(means_2d[2 * (size_t)id], means_2d[2 * (size_t)id + 1], conic[3 * (size_t)id],
                        conic[3 * (size_t)id + 1]);
        b = make_float4(conic[3 * (size_t)id + 2], opacity[id], color[3 * (size_t)id], color[3 * (size_t)id + 1]);
        c = color[3 * (size_t)id + 2];
    }
    __device__ __forceinline__ uint2 rect(uint32_t) const { return make_uint2(0u, 0xFFFFFFFFu); } // (no list blocks on this path)
};

struct FetchRec {
    const SplatRecord* __restrict__ recs;
    __device__ __forceinline__ void operator()(uint32_t id, float4& a, float4& b, float& c) const
    {
        const float4* p = reinterpret_cast<const float4*>(recs + id);
        a               = p[0];
        b               = p[1];
        c               = reinterpret_cast<const float*>(recs + id)[8];
    }
    // the splat's pruned rect in tiles (origin, size): what its per-tile lists would hold it for
    __device__ __forceinline__ uint2 rect(uint32_t id) const { return make_uint2(recs[id].rect_xy, recs[id].rect_wh); }
};

using namespace tile;

// ---------------------------------------------------------------------------------------------
// One workgroup (4 wave64s) per tile, wave k owns "strip" k = the tile's 8x8 quadrant k (unit_px / unit_py), one pixel per lane.
// The heaviest tiles set the kernel's critical path; splitting a tile over four waves shortens it 4x and gives
// the dispatcher four times as many independent wave-sized work items to balance.  A round stages 256 list
// entries: every lane fetches one entry, tests it against the four strips, and the per-strip ballots (one
// 64-bit mask per staging wave and strip) go to LDS; wave k then walks the set bits of "its" four masks with
// scalar bit-scan instructions -- entries irrelevant to a strip cost that strip nothing, order is preserved.
// ---------------------------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2))); // arithmetic on it lowers to v_pk_{add,mul}_f32 (IEEE per lane)

// PERSIST: a bounded grid of workgroups that take tiles from a device counter (work_counter, zeroed with the frame's tile
// ranges) in schedule order until it runs past the last one -- every workgroup reaches that exit.  Same image (which
// workgroup renders which tile never mattered); what it buys is a cap on the wave slots the renderer holds per CU, so that
// ANOTHER frame's short sort-chain kernels find free slots on every CU the moment they are dispatched instead of queueing
// behind 8160 resident-or-pending tile workgroups (camera batches, lcgs_fit_views: DESIGN.md 9).
// COMPACT (frames that keep backward state on PER-BLOCK lists, round 6): a tile's workgroup walks its block's list and, while
// it stages, writes the entries that can reach one of its four units -- in list order -- to a segment of its own (keep_list;
// tile q of a block whose list is [s, e) owns [4 s + q (e - s), ...): room for the whole block list, no allocation), with the
// kept masks beside them and its range in keep_ranges; the last-contributor positions are positions in THAT list.  The
// backward then walks a per-tile list exactly as before -- a denser one than the reference's (only entries that reach a
// unit) -- while duplication, tile partition and range pass ran on 0.6 x the pairs.
template <typename Fetch, bool KEEP, bool PERSIST, bool COMPACT>
__global__ void __launch_bounds__(256) k_render_forward_b(CamParams cp, float bg0, float bg1, float bg2,
                                                            const FrameParams* __restrict__ fpp,
                                                            const uint32_t* __restrict__ ranges,
                                                            const uint32_t* __restrict__ point_list, Fetch fetch,
                                                            float* __restrict__ img, float* __restrict__ final_T,
                                                            uint32_t* __restrict__ n_contrib,
                                                            const uint32_t* __restrict__ d_counts,
                                                            const uint32_t* __restrict__ tile_order,
                                                            uint8_t* __restrict__ strip_masks,
                                                            uint32_t* __restrict__ work_counter,
                                                            float4* __restrict__ g2d_zero,
                                                            uint32_t* __restrict__ bwd_counters,
                                                            uint32_t* __restrict__ keep_list,
                                                            uint32_t* __restrict__ keep_ranges)
{
    // one 16-byte row per entry in each of three slabs: a single address register serves all three reads
    // s_rows[0]: mean.x, mean.y, -conic.x / 2, -conic.z / 2;  [1]: conic.y, power floor (-t/2), -, - (with [0], all the cull
    // test needs);  [2]: opacity, r, g, b (read only by entries that pass it)
    __shared__ float4             s_rows[3][256];
    __shared__ unsigned long long s_mask[4][4]; // [staging wave][strip]
    // KEEP: [strip][staging wave] the entries of the round that passed the strip's wave-level candidate test while it was walked --
    // what the backward has to walk (the reach test at staging is conservative, and pixels finish: a quarter of the masks' set bits
    // blend nothing; of the entries that pass the candidate test 99.99 % are blended by some pixel)
    __shared__ unsigned long long s_blend[4][4];
    __shared__ uint32_t           s_live_waves;
    __shared__ uint32_t           s_slot;
    __shared__ uint32_t           s_cnt[4]; // COMPACT: entries of the round each staging wave keeps

    if (fpp) { // graph replay: per-call parameters come from device memory
        cp  = fpp->cp;
        bg0 = fpp->bg[0];
        bg1 = fpp->bg[1];
        bg2 = fpp->bg[2];
    }
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (d_counts && d_counts[1] == 0u) { // image untouched (gs_tile_splatter/impl.cpp:109)
        // ... but the side job below is owed all the same: the host has already noted the 2-D gradient rows and the
        // backward's counter block as cleared by this launch (abi_frame.cpp g2d_zeroed)
        if (KEEP && g2d_zero) {
            const uint32_t n4 = d_counts[0] * 3u;
            for (uint32_t i = blockIdx.x * 256u + tid; i < n4; i += gridDim.x * 256u) g2d_zero[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (blockIdx.x == 0u && tid < 32u && bwd_counters) bwd_counters[tid] = 0u;
        }
        return;
    }
    const uint32_t slots = tile_order ? cp.grid_x * cp.grid_y : render_grid_size(cp.grid_x, cp.grid_y);
    uint32_t       slot  = blockIdx.x;
  for (;;) { // (one pass unless PERSIST)
    if (PERSIST) {
        __syncthreads(); // nobody still reads the previous s_slot (a padding slot's `continue` passes no other barrier)
        if (tid == 0) s_slot = atomicAdd(work_counter, 1u);
        __syncthreads();
        slot = s_slot;
    }
    if (slot >= slots) return; // persistent grids: the exit every workgroup reaches
    if (KEEP && g2d_zero) {
        // A side job of frames that keep backward state: slot s clears its share of the 2-D gradient rows the render-backward
        // will add to (12 floats per on-screen splat) -- stores into a memory system this VALU-bound kernel leaves idle,
        // instead of a separate launch on the auxiliary stream and a cross-stream wait (~10 us) in front of the backward.
        const uint32_t n4 = d_counts[0] * 3u; // float4s; the slots stride over them, 4 KB per slot and step
        for (uint32_t i = slot * 256u + tid; i < n4; i += slots * 256u) g2d_zero[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (slot == 0u && tid < 32u && bwd_counters) bwd_counters[tid] = 0u; // (see k_zero_grads2d)
    }
    uint32_t tx, ty;
    if (tile_order) { // scheduling hint only: which workgroup takes which tile never changes the image
        const uint32_t t = tile_order[slot];
        tx = t % cp.grid_x;
        ty = t / cp.grid_x;
    } else if (!tile_of_workgroup(slot, cp.grid_x, cp.grid_y, tx, ty)) {
        if (PERSIST) continue; // a padding slot of the XCD-aware map
        return;
    }
    const uint32_t tile = ty * cp.grid_x + tx;
    const uint32_t px = unit_px(tx, wave, lane);
    const uint32_t py = unit_py(ty, wave, lane);
    const float rx0 = (float)(tx * kBlockX), ry0 = (float)(ty * kBlockY), rx1 = rx0 + (float)(kBlockX - 1);
    const bool  inside = (px < cp.width) && (py < cp.height);
    // A finished pixel (saturated, or outside the image) gets a NaN y coordinate: its power is NaN, fails
    // "power >= floor", and the pixel drops out of every later test without a compare of its own.
    v2f pxy = {(float)px, inside ? (float)py : __builtin_nanf("")};
    asm volatile("" : "+v"(pxy));

    const unsigned long long lane_bit = 1ull << lane;
    float    T = 1.0f;
    float    Cr = 0.0f;
    v2f      Cgb = {0.0f, 0.0f};
    uint32_t last_contrib = 0u;
    bool     alive = __builtin_amdgcn_ballot_w64(inside) != 0ull; // wave-uniform: the strip has unfinished pixels
    if (tid == 0) s_live_waves = 0u;
    __syncthreads();
    if (lane == 0 && alive) atomicAdd(&s_live_waves, 1u);

    const uint32_t lblock      = cp.list_shift ? list_block_of_tile(cp, tx, ty) : tile; // whose list this tile walks
    const uint32_t range_start = ranges[2 * (size_t)lblock + 0];
    const uint32_t range_end   = ranges[2 * (size_t)lblock + 1];

    float4 na = make_float4(0, 0, 0, 0), nb = make_float4(0, 0, 0, 0);
    float  nc = 0.0f;
    uint2  nrect = make_uint2(0u, 0xFFFFFFFFu); // (coarse lists only)
    uint32_t nid = 0u;                           // (COMPACT: the staged entry's dense id goes into the tile's own list)
    if (range_start + tid < range_end) {
        const uint32_t id = point_list[range_start + tid];
        fetch(id, na, nb, nc);
        if (cp.list_shift) nrect = fetch.rect(id);
        nid = id;
    }
    // COMPACT: this tile's segment of keep_list / strip_masks and how much of it is filled
    const uint32_t seg        = COMPACT ? 4u * range_start + ((tx & 1u) | ((ty & 1u) << 1)) * (range_end - range_start) : 0u;
    uint32_t       round_base = 0u;
    __syncthreads();

    for (uint32_t base = range_start; base < range_end; base += 256u) {
        if (s_live_waves == 0u) break; // every pixel of the tile is finished
        // ---- stage 256 entries: strip tests, per-strip ballots, slab
        const uint32_t e    = base + tid;
        const bool     have = e < range_end;
        const float4   a = na, b = nb;
        const float    c = nc;
        const float    t = have ? (2.0f * __logf(255.0f * b.y)) * 1.0001f + 2e-4f : -1.0f;
        uint32_t       kmask = have ? splat_unit_mask(a.x, a.y, a.z, a.w, b.x, t, rx0, ry0, rx1) : 0u;
        if (cp.list_shift) {
            // a block's list holds every splat that reaches ANY of its tiles: this tile takes those whose pruned rect covers it --
            // exactly the entries of its own list (the reference's 3-sigma rect cuts footprints the alpha test alone would keep)
            const uint32_t dx_ = tx - (nrect.x & 0xFFFFu), dy_ = ty - (nrect.x >> 16);
            if (!(dx_ < (nrect.y & 0xFFFFu) && dy_ < (nrect.y >> 16))) kmask = 0u;
        }
        __syncthreads(); // previous round's readers are done with the slab and the masks
        if (KEEP && lane < 4u) s_blend[wave][lane] = 0ull; // (strips that are finished, or finish mid-round, blend nothing)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned long long m = __ballot((kmask >> k) & 1u);
            if (lane == 0) s_mask[wave][k] = m;
        }
        uint32_t rank = 0u; // COMPACT: the entry's place among its staging wave's kept entries
        if (COMPACT) {
            const unsigned long long km = __ballot(kmask != 0u);
            rank = (uint32_t)__popcll(km & (lane_bit - 1ull));
            if (lane == 0) s_cnt[wave] = (uint32_t)__popcll(km);
        }
        if (kmask) {
            s_rows[0][tid] = make_float4(a.x, a.y, -0.5f * a.z, -0.5f * b.x);
            if (COMPACT) s_rows[1][tid] = make_float4(a.w, fmax_(-0.5f * t, kBlendExpMin), __uint_as_float(rank), 0.0f);
            else *reinterpret_cast<float2*>(&s_rows[1][tid]) = make_float2(a.w, fmax_(-0.5f * t, kBlendExpMin));
            s_rows[2][tid] = make_float4(b.y, b.z, b.w, c);
        }
        const uint32_t id_now = nid;
        const uint32_t en = e + 256u;
        if (en < range_end) {
            const uint32_t id = point_list[en];
            fetch(id, na, nb, nc);
            if (cp.list_shift) nrect = fetch.rect(id);
            nid = id;
        }
        __syncthreads();
        uint32_t my_pos = 0u, kept = 0u; // COMPACT: where this lane's entry goes in the tile's list; the round's total
        if (COMPACT) {
            const uint32_t c0 = s_cnt[0], c1 = s_cnt[1], c2 = s_cnt[2], c3 = s_cnt[3];
            kept   = c0 + c1 + c2 + c3;
            my_pos = seg + round_base + (wave > 0u ? c0 : 0u) + (wave > 1u ? c1 : 0u) + (wave > 2u ? c2 : 0u) + rank;
            if (kmask) keep_list[my_pos] = id_now;
        }

        if (alive) {
            for (uint32_t w = 0; w < 4u && alive; ++w) {
                unsigned long long m  = s_mask[w][wave];
                // readfirstlane returns int: cast through uint32_t so the low half is not sign-extended
                m = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(m >> 32)) << 32) |
                    (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)m);
                // KEEP: (scalar) the entries of staging wave w that pass this strip's wave-level candidate test -- on the bench
                // frames 99.99 % of them are blended by some pixel (4 195 020 against 4 194 674 pairs, profiles/r05_bwd_walk_stats.txt),
                // and ONE scalar instruction per passing entry replaces the four that an exact "some pixel took it into its sum" bit
                // cost (the backward only needs a superset).  (On the passing path only: an instruction on the `continue` path
                // makes the compiler merge the two paths through seven register copies per entry.)
                unsigned long long bm = 0ull;
                // COMPACT: list position (1-based, within the tile's own list) of staging wave w's first kept entry
                uint32_t wbase = 0u;
                if (COMPACT) {
                    wbase = round_base + 1u;
                    for (uint32_t q = 0; q < w; ++q) wbase += s_cnt[q];
                    wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)wbase);
                }
                while (m != 0ull) { // scalar loop control
                    const uint32_t l   = (uint32_t)__ffsll((long long)m) - 1u;
                    asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(l)); // one scalar op instead of add/addc/and
                    const uint32_t idx = w * 64u + l;
                    // ONE address register for the entry's three rows (4 KB apart: immediate offsets), pinned so that the
                    // read behind the wave-level test does not pay for a second scalar-to-vector move
                    uint32_t row = idx * 16u; // (byte offset: the shift stays on the scalar side)
                    asm volatile("" : "+v"(row));
                    const char*    rows = reinterpret_cast<const char*>(&s_rows[0][0]) + row;
                    const float4   ea = *reinterpret_cast<const float4*>(rows);
                    const float2   eb = *reinterpret_cast<const float2*>(rows + 4096);
                    // power = -0.5 (ca dx dx + cc dy dy) - cb dx dy, products left to right (shader.cpp:256)
                    // (plain, not packed, arithmetic: on gfx950 a packed op costs two plain ones AND drags a wait state behind it
                    //  -- tools/microbench/issue_rates.hip; the library is built with -fno-slp-vectorize for that reason)
                    float dx = ea.x - pxy.x, dy = ea.y - pxy.y;
                    // ea.z / ea.w hold -0.5 ca / -0.5 cc (scaled by a power of two when staged: the same bits as scaling
                    // the sum afterwards, one multiplication fewer per entry and strip)
                    float qx = (ea.z * dx) * dx, cross = eb.x * dx;
                    float qy = (ea.w * dy) * dy;
                    float half = qx + qy;
                    const float power = half - cross * dy;
                    // candidate lanes: !(power > 0) and power >= the staged floor --
                    // (lane masks from ballots of plain compares, combined with scalar ANDs: a ballot of a compound
                    //  condition is lowered through a select and a second compare)
                    const unsigned long long cmask =
                        __builtin_amdgcn_ballot_w64(!(power > 0.0f)) & __builtin_amdgcn_ballot_w64(power >= eb.y);
                    if (cmask == 0ull) continue; // scalar test of the lane mask
                    if (KEEP) asm("s_bitset1_b64 %0, %1" : "+s"(bm) : "s"(l));
                    const float4 ec    = *reinterpret_cast<const float4*>(rows + 8192); // one 16-byte read for the survivors
                    // (alpha is never NaN on a candidate lane, so the hardware minimum equals min(0.99, x); on a lane where
                    //  it does not hold, power may lie outside blend_exp's domain and alpha is arbitrary bits -- masked)
                    const float alpha  = __builtin_fminf(0.99f, ec.x * blend_exp(power));
                    // Lanes that skip the entry sit the update out under the EXEC mask: what the arithmetic produces on
                    // them is never written anywhere, and the select that used to zero their alpha is gone.  (Written as
                    // one asm block because the compiler turns `if (valid) { ... }` back into four selects.)
                    const unsigned long long vmask = __builtin_amdgcn_ballot_w64(!(alpha < 1.0f / 255.0f)) & cmask;
                    const float test_T = T * (1.0f - alpha);
                    float       wgt = T * alpha;
                    float       nT  = test_T;
                    // T >= 1e-4 holds for every lane (a saturating update is never applied)
                    const unsigned long long satm = __builtin_amdgcn_ballot_w64(test_T < 0.0001f) & vmask;
                    // KEEP: the pixels that take this entry into their sum (blend it and do not saturate on it)
                    const unsigned long long live = vmask & ~satm;
                    if (satm != 0ull) { // rare: some pixel of the strip just saturated
                        const bool sat = (satm & lane_bit) != 0ull;
                        wgt   = sat ? 0.0f : wgt; // shader.cpp:268-272: the saturating entry is not blended
                        nT    = sat ? T : nT;
                        pxy.y = sat ? __builtin_nanf("") : pxy.y;
                        if (__builtin_amdgcn_ballot_w64(pxy.y == pxy.y) == 0ull) {
                            alive = false; // the whole strip is finished
                            m     = 0ull;
                        }
                    }
                    if (KEEP) { // the same block + the position of the pixel's last contributor (1-based), all of it under `live`:
                        // a pixel that saturates on this entry sits it out altogether (its weight was zeroed above anyway), so ONE
                        // exec region serves the sums and the position (two regions cost 6 us of the 0.24 ms)
                        float              t0, t1, t2;
                        unsigned long long sv;
                        // (COMPACT: the entry's rank among its staging wave's kept entries lies in its second slab row)
                        const uint32_t     lc = COMPACT ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(
                                                              wbase + __float_as_uint(*reinterpret_cast<const float*>(rows + 4096 + 8))))
                                                        : base - range_start + idx + 1u;
                        asm volatile("s_and_saveexec_b64 %[sv], %[lv]\n\t"
                                     "v_mul_f32 %[t0], %[w], %[cr]\n\t"
                                     "v_mul_f32 %[t1], %[w], %[cg]\n\t"
                                     "v_mul_f32 %[t2], %[w], %[cb]\n\t"
                                     "v_add_f32 %[Cr], %[Cr], %[t0]\n\t"
                                     "v_add_f32 %[Cg], %[Cg], %[t1]\n\t"
                                     "v_add_f32 %[Cb], %[Cb], %[t2]\n\t"
                                     "v_mov_b32 %[T], %[nT]\n\t"
                                     "v_mov_b32 %[lcv], %[lc]\n\t"
                                     "s_mov_b64 exec, %[sv]"
                                     : [Cr] "+v"(Cr), [Cg] "+v"(Cgb.x), [Cb] "+v"(Cgb.y), [T] "+v"(T), [t0] "=&v"(t0),
                                       [t1] "=&v"(t1), [t2] "=&v"(t2), [sv] "=&s"(sv), [lcv] "+v"(last_contrib)
                                     : [w] "v"(wgt), [cr] "v"(ec.y), [cg] "v"(ec.z), [cb] "v"(ec.w), [nT] "v"(nT),
                                       [lv] "s"(live), [lc] "s"(lc)
                                     : "scc");
                    } else {
                        float              t0, t1, t2;
                        unsigned long long sv;
                        asm volatile("s_and_saveexec_b64 %[sv], %[vm]\n\t"
                                     "v_mul_f32 %[t0], %[w], %[cr]\n\t"
                                     "v_mul_f32 %[t1], %[w], %[cg]\n\t"
                                     "v_mul_f32 %[t2], %[w], %[cb]\n\t"
                                     "v_add_f32 %[Cr], %[Cr], %[t0]\n\t"
                                     "v_add_f32 %[Cg], %[Cg], %[t1]\n\t"
                                     "v_add_f32 %[Cb], %[Cb], %[t2]\n\t"
                                     "v_mov_b32 %[T], %[nT]\n\t"
                                     "s_mov_b64 exec, %[sv]"
                                     : [Cr] "+v"(Cr), [Cg] "+v"(Cgb.x), [Cb] "+v"(Cgb.y), [T] "+v"(T), [t0] "=&v"(t0),
                                       [t1] "=&v"(t1), [t2] "=&v"(t2), [sv] "=&s"(sv)
                                     : [w] "v"(wgt), [cr] "v"(ec.y), [cg] "v"(ec.z), [cb] "v"(ec.w), [nT] "v"(nT),
                                       [vm] "s"(vmask)
                                     : "scc");
                    }
                }
                if (KEEP && lane == 0) s_blend[wave][w] = bm;
            }
            if (!alive && lane == 0) atomicSub(&s_live_waves, 1u);
        }
        __syncthreads();
        if (COMPACT) {
            if (kmask) {
                uint32_t kref = 0u;
#pragma unroll
                for (int k = 0; k < 4; ++k) kref |= (uint32_t)((s_blend[k][wave] >> lane) & 1ull) << k;
                strip_masks[my_pos] = (uint8_t)kref;
            }
            round_base += kept;
        } else if (KEEP && strip_masks && have) {
            // The backward walks the same list positions through these masks (it need not repeat the strip tests) -- and only
            // the entries a strip's pixels could blend when it walked them: bit k = the entry passed strip k's candidate test.  Entries
            // that merely could reach a strip contribute exact zeros to every gradient; a quarter of the backward's walks were those.
            uint32_t kref = 0u;
#pragma unroll
            for (int k = 0; k < 4; ++k) kref |= (uint32_t)((s_blend[k][wave] >> lane) & 1ull) << k;
            strip_masks[e] = (uint8_t)kref;
        }
    }

    if (COMPACT && tid == 0) {
        keep_ranges[2 * (size_t)tile + 0] = seg;
        keep_ranges[2 * (size_t)tile + 1] = seg + round_base;
    }
    if (inside) {
        const size_t hw  = (size_t)cp.width * cp.height;
        const float  Tk  = T;
        const size_t pix = (size_t)px + (size_t)cp.width * py;
        img[pix]          = bg0 * Tk + Cr;
        img[pix + hw]     = bg1 * Tk + Cgb.x;
        img[pix + 2 * hw] = bg2 * Tk + Cgb.y;
        if (KEEP) {
            if (final_T) final_T[pix] = Tk;
            if (n_contrib) n_contrib[pix] = last_contrib;
        }
    }
    if (!PERSIST) return;
  }
}

// Longest-list-first tile schedule (a scheduling hint: any order gives the same image).  Tile work is roughly
// proportional to list length and varies >10x across a frame; dispatching the long tiles first removes the tail in
// which a few late heavy tiles keep the kernel alive (measured: -16 % renderer time on the bicycle stand-in).
// One workgroup: counting sort of the tiles into 1024 length buckets (descending); order inside a bucket is
// whatever the LDS atomics produce.
// (grid_x, shift: the tiles' lists are their blocks' -- CamParams::list_shift; shift 0: ranges are per tile)
// Round 6, per-block lists: the BLOCKS of 2 x 2 tiles are sorted by their list's length, and the (up to) four
// tiles of a block go to slots that are 8 apart -- workgroups are dealt round-robin over the 8 XCDs, so slots s, s + 8, s + 16,
// s + 24 share an XCD and its L2: a block's list, which all four tile workgroups stage, is fetched into ONE L2 instead of four
// (renderer FETCH_SIZE 204 -> 86 MB per launch), and neighbouring tiles' lists name the same splats (scratch: the sorted
// blocks behind the G slots of `order`).
__global__ void __launch_bounds__(1024) k_tile_order(const uint32_t* __restrict__ ranges, uint32_t G,
                                                       uint32_t* __restrict__ order, uint32_t grid_x, uint32_t shift)
{
    // (A/B hook LCGS_TILE_ORDER_XCD=0: `shift` arrives with bit 8 set -> the tiles themselves are sorted, slots in sorted order)
    // Per-TILE lists keep the plain longest-tile-first order: the block-interleaved one was measured on them too (round 6) and
    // loses 2 % on forward+backward (5 % in file order) -- four neighbouring tiles that add to the same gradient rows at the
    // same time, and a coarser longest-first.  (The weight() of that experiment stays below for the record.)
    const uint32_t sh  = shift & 0xFFu;
    const bool     by_block = !(shift & 0x100u) && grid_x > 1u && sh != 0u;
    const uint32_t lgx = (grid_x + (1u << sh) - 1u) >> sh;
    auto           list_of = [&](uint32_t t) { return sh ? ((t / grid_x) >> sh) * lgx + ((t % grid_x) >> sh) : t; };
    __shared__ uint32_t s_bucket[1024];
    __shared__ uint32_t s_wave[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t grid_y = G / grid_x;
    const uint32_t bgx = (grid_x + 1u) >> 1, bgy = (grid_y + 1u) >> 1; // blocks of 2 x 2 tiles
    const uint32_t n_items = by_block ? bgx * bgy : G;                 // what is sorted: blocks, or tiles
    // a block's weight: its list's length (per-block lists), or the longest of its tiles' lists (per-tile lists)
    auto weight = [&](uint32_t t) -> uint32_t {
        if (!by_block) {
            const uint32_t lb = list_of(t);
            return ranges[2 * (size_t)lb + 1] - ranges[2 * (size_t)lb];
        }
        if (sh) return ranges[2 * (size_t)t + 1] - ranges[2 * (size_t)t]; // (block t IS list t: same 2 x 2 grid)
        const uint32_t bx = t % bgx, by = t / bgx;
        uint32_t       w = 0u;
        for (uint32_t j = 0; j < 4u; ++j) {
            const uint32_t tx = bx * 2u + (j & 1u), ty = by * 2u + (j >> 1);
            if (tx < grid_x && ty < grid_y) {
                const uint32_t l = ranges[2 * (size_t)(ty * grid_x + tx) + 1] - ranges[2 * (size_t)(ty * grid_x + tx)];
                w = l > w ? l : w;
            }
        }
        return w;
    };
    s_bucket[tid] = 0;
    __syncthreads();
    for (uint32_t t = tid; t < n_items; t += 1024u) {
        const uint32_t len = weight(t);
        const uint32_t b   = 1023u - ((len >> 3) < 1023u ? (len >> 3) : 1023u);
        atomicAdd(&s_bucket[b], 1u);
    }
    __syncthreads();
    const uint32_t own = s_bucket[tid];
    uint32_t       inc = own;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(inc, off, 64);
        if (lane >= (uint32_t)off) inc += o;
    }
    if (lane == 63u) s_wave[wave] = inc;
    __syncthreads();
    uint32_t carry = 0;
    for (uint32_t w = 0; w < wave; ++w) carry += s_wave[w];
    __syncthreads();
    s_bucket[tid] = carry + inc - own; // exclusive start of the bucket
    __syncthreads();
    uint32_t* sorted = by_block ? order + G : order; // by block: the sorted BLOCKS go to the scratch behind the slots
    for (uint32_t t = tid; t < n_items; t += 1024u) {
        const uint32_t len = weight(t);
        const uint32_t b   = 1023u - ((len >> 3) < 1023u ? (len >> 3) : 1023u);
        sorted[atomicAdd(&s_bucket[b], 1u)] = t;
    }
    if (!by_block) return;
    __threadfence_block();
    __syncthreads();
    // candidate slot c = 32 g + 8 j + k holds tile j (0..3: x + 2 y inside the block) of the (8 g + k)-th block; tiles that do
    // not exist (odd grids) are squeezed out -- the four tiles of a block stay a multiple of 8 apart unless a hole falls
    // between them (edge blocks only)
    const uint32_t n_cand = ((n_items + 7u) / 8u) * 32u;
    uint32_t       base = 0u;
    for (uint32_t c0 = 0; c0 < n_cand; c0 += 1024u) {
        const uint32_t c = c0 + tid;
        uint32_t       tile = 0xFFFFFFFFu;
        if (c < n_cand) {
            const uint32_t g = c >> 5, j = (c >> 3) & 3u, k = c & 7u, bi = 8u * g + k;
            if (bi < n_items) {
                const uint32_t blk = sorted[bi];
                const uint32_t tx = (blk % bgx) * 2u + (j & 1u), ty = (blk / bgx) * 2u + (j >> 1);
                if (tx < grid_x && ty < grid_y) tile = ty * grid_x + tx;
            }
        }
        const bool               valid = tile != 0xFFFFFFFFu;
        const unsigned long long m     = __ballot(valid);
        const uint32_t           rank  = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        __syncthreads(); // (s_wave of the previous chunk has been read)
        if (lane == 0) s_wave[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = 0u, total = 0u;
        for (uint32_t w = 0; w < 16u; ++w) {
            const uint32_t n = s_wave[w];
            before += w < wave ? n : 0u;
            total += n;
        }
        if (valid) order[base + before + rank] = tile;
        base += total;
    }
}

// diagnostics (lcgs_debug_blend_exp): the loop's exp on its own
__global__ void __launch_bounds__(256) k_blend_exp(const float* __restrict__ x, float* __restrict__ out, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = blend_exp(x[i]);
}

template <typename Fetch>
void launch_render(const CamParams& cp, const float bg[3], const uint32_t* ranges, const uint32_t* point_list,
                   Fetch fetch, float* img, float* final_T, uint32_t* n_contrib, const uint32_t* d_counts,
                   const FrameParams* d_fp, const uint32_t* tile_order, hipStream_t stream,
                   uint8_t* strip_masks = nullptr, hipEvent_t done = nullptr, uint32_t* work_counter = nullptr,
                   uint32_t persistent_wgs = 0, float* g2d_zero = nullptr, uint32_t* bwd_counters = nullptr,
                   uint32_t* keep_list = nullptr, uint32_t* keep_ranges = nullptr)
{
    if (cp.grid_x * cp.grid_y == 0) return;
    const uint32_t full = render_grid_size(cp.grid_x, cp.grid_y);
    const bool     keep = final_T || n_contrib;
    // a keep-state frame on per-block lists writes every tile's own list while it stages (COMPACT)
    const bool     compact = keep && cp.list_shift != 0u && keep_list && keep_ranges && strip_masks;
    // (`done`, when given, is carried by the dispatch packet: no separate event-record packet behind the kernel)
#define LCGS_LAUNCH_RENDER(KEEP_, PERSIST_, COMPACT_, GRID_)                                                                \
    hipExtLaunchKernelGGL((k_render_forward_b<Fetch, KEEP_, PERSIST_, COMPACT_>), dim3(GRID_), dim3(256), 0, stream, nullptr, done, 0, cp, \
                          bg[0], bg[1], bg[2], d_fp, ranges, point_list, fetch, img, final_T, n_contrib, d_counts, tile_order, \
                          KEEP_ ? strip_masks : (uint8_t*)nullptr, work_counter, reinterpret_cast<float4*>(g2d_zero), bwd_counters, \
                          keep_list, keep_ranges)
    if (work_counter && persistent_wgs > 0 && persistent_wgs < full) { // a bounded grid that pulls tiles from the counter
        if (compact) LCGS_LAUNCH_RENDER(true, true, true, persistent_wgs);
        else if (keep) LCGS_LAUNCH_RENDER(true, true, false, persistent_wgs);
        else LCGS_LAUNCH_RENDER(false, true, false, persistent_wgs);
    } else {
        work_counter = nullptr;
        if (compact) LCGS_LAUNCH_RENDER(true, false, true, full);
        else if (keep) LCGS_LAUNCH_RENDER(true, false, false, full);
        else LCGS_LAUNCH_RENDER(false, false, false, full); // forward only: the last-contributor bookkeeping is compiled out
    }
#undef LCGS_LAUNCH_RENDER
}

} // namespace

void launch_blend_exp(const float* x, float* out, int64_t n, hipStream_t stream)
{
    if (n <= 0) return;
    const int64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_blend_exp, dim3((uint32_t)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, stream, x, out, n);
}

void launch_tile_order(const uint32_t* ranges, uint32_t G, uint32_t* order, hipStream_t stream, uint32_t grid_x, uint32_t list_shift)
{
    if (G == 0) return;
    static const bool xcd_aware = [] { const char* e = getenv("LCGS_TILE_ORDER_XCD"); return !(e && e[0] == '0'); }(); // A/B hook
    // (grid_x == 0: a caller that does not say how the tiles lie -- sorted as tiles)
    hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(1024), 0, stream, ranges, G, order, grid_x ? grid_x : 1u,
                       (grid_x ? list_shift : 0u) | ((!grid_x || !xcd_aware) ? 0x100u : 0u));
}

void launch_render_forward_aos(const CamParams& cp, const float bg[3], const uint32_t* ranges,
                               const uint32_t* point_list, const float* means_2d, const float* conic,
                               const float* opacity, const float* color, float* img, float* final_T,
                               uint32_t* n_contrib, hipStream_t stream)
{
    launch_render(cp, bg, ranges, point_list, FetchAoS{ means_2d, conic, opacity, color }, img, final_T, n_contrib,
                  nullptr, nullptr, nullptr, stream);
}

void launch_render_forward_rec(const CamParams& cp, const float bg[3], const uint32_t* ranges,
                               const uint32_t* point_list, const SplatRecord* recs, float* img, float* final_T,
                               uint32_t* n_contrib, const uint32_t* d_counts, const FrameParams* d_fp,
                               const uint32_t* tile_order, hipStream_t stream, uint8_t* strip_masks, hipEvent_t done,
                               uint32_t* work_counter, uint32_t persistent_wgs, float* g2d_zero, uint32_t* bwd_counters,
                               uint32_t* keep_list, uint32_t* keep_ranges)
{
    launch_render(cp, bg, ranges, point_list, FetchRec{ recs }, img, final_T, n_contrib, d_counts, d_fp, tile_order,
                  stream, strip_masks, done, work_counter, persistent_wgs, g2d_zero, bwd_counters, keep_list, keep_ranges);
}

// the forward renderer fills strip_masks whenever it keeps backward state
bool render_forward_writes_strip_masks() { return true; }

} // namespace lcgs
