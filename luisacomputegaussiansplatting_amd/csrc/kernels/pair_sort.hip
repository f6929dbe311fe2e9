// pair_sort.hip -- the stable LSD radix sort of (key, u32 value) pairs: u32 keys twice per fused frame (survivors by
// depth bits; pairs by tile id, element count read from device memory), u64 keys for the stage-level operator and
// the exported primitive (lcpp DeviceRadixSort::SortPairs<ulong, uint>, call site
// lcgs/src/gs_tile_splatter/impl.cpp:135-143).  Three launches per digit:
//   k_hist     per-chunk digit counts -> counts[digit][chunk]            (reads the keys)
//              (each kernel requests everything its first chunk needs before it reads the live element count)
//   k_rowscan  one workgroup per digit: exclusive scan of its row + the row total
//   k_scatter  wave64-ballot stable ranking, LDS staging into chunk-sorted order, coalesced runs out;
//              the global digit bases are re-derived per workgroup from the 256 row totals
// A decoupled-look-back single-pass variant was measured and rejected on this part: per-digit look-back chains
// are latency-bound here (agent-scope sc1 loads are ~1 us each under load, one ticket atomic saturates at
// ~88/us), 1.5-2x slower than these three plain launches at V ~ 2.4 M / L ~ 7.5 M.
// Workgroups stride over chunks, so the grid is bounded (no launch of capacity-sized empty grids).
#include <hip/hip_ext.h>

#include "launch.hpp"
#include "tie_order.hpp"

namespace lcgs
{
namespace
{

constexpr int kThreads = 256;
constexpr int kWaves   = kThreads / 64;
constexpr int kRadix   = 256;

// Layout of the count table: counts[digit * row_stride + chunk], row_stride = the chunk CAPACITY rounded up to 8 --
// host-known, so every address is known before the live element count (*d_n) has been read, and each kernel issues
// all the loads of its first chunk at once, ahead of that read: these launches are short and latency-bound, one
// round trip each instead of three or four is most of their running time.
// d_n == NULL: the element count is the host-known n_cap (stage-level callers).  Buffers hold n_cap elements.
template <int kItems, typename K>
__global__ void __launch_bounds__(kThreads) k_hist(const K* __restrict__ keys, const uint32_t* __restrict__ d_n,
                                                     uint32_t n_cap, int shift, uint32_t mask,
                                                     uint32_t* __restrict__ counts, uint32_t row_stride)
{
    constexpr int kKPB = kThreads * kItems;
    __shared__ uint32_t s_hist[kRadix];
    uint32_t chunk = blockIdx.x;
    K        key[kItems];
#pragma unroll
    for (int r = 0; r < kItems; ++r) {
        const uint32_t i = chunk * kKPB + r * kThreads + threadIdx.x;
        key[r]           = i < n_cap ? keys[i] : (K)0;
    }
    s_hist[threadIdx.x] = 0;
    const uint32_t n  = d_n ? *d_n : n_cap;
    const uint32_t nb = (n + kKPB - 1) / kKPB;
    __syncthreads();
    while (chunk < nb) {
#pragma unroll
        for (int r = 0; r < kItems; ++r) {
            const uint32_t i = chunk * kKPB + r * kThreads + threadIdx.x;
            if (i < n) atomicAdd(&s_hist[(uint32_t)(key[r] >> shift) & mask], 1u);
        }
        __syncthreads();
        counts[(size_t)threadIdx.x * row_stride + chunk] = s_hist[threadIdx.x];
        chunk += gridDim.x;
        if (chunk >= nb) break;
        __syncthreads();
        s_hist[threadIdx.x] = 0;
#pragma unroll
        for (int r = 0; r < kItems; ++r) {
            const uint32_t i = chunk * kKPB + r * kThreads + threadIdx.x;
            key[r]           = i < n ? keys[i] : (K)0;
        }
        __syncthreads();
    }
}

// one workgroup per digit: counts[d][0..nb) -> exclusive prefix in place; totals[d] = row sum.
// Thread t owns 8 consecutive chunks (two 16-byte loads); rows longer than 2048 chunks take further rounds.
template <int kItems>
__global__ void __launch_bounds__(kThreads) k_rowscan(uint32_t* __restrict__ counts, const uint32_t* __restrict__ d_n,
                                                        uint32_t n_cap, uint32_t* __restrict__ totals,
                                                        uint32_t row_stride)
{
    constexpr int kKPB = kThreads * kItems;
    constexpr int kPer = 8;
    __shared__ uint32_t s_wave[kWaves];
    uint32_t*      row  = counts + (size_t)blockIdx.x * row_stride;
    const int      lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint4          a = make_uint4(0, 0, 0, 0), b = a;
    uint32_t       i0 = threadIdx.x * kPer;
    if (i0 < row_stride) { // row_stride is a multiple of 8: the whole group is inside the row
        a = *reinterpret_cast<const uint4*>(row + i0);
        b = *reinterpret_cast<const uint4*>(row + i0 + 4);
    }
    const uint32_t n  = d_n ? *d_n : n_cap;
    const uint32_t nb = (n + kKPB - 1) / kKPB;
    uint32_t       carry_in = 0;
    for (uint32_t base = 0; base < nb; base += kThreads * kPer) {
        if (base > 0) {
            i0 = base + threadIdx.x * kPer;
            a = b = make_uint4(0, 0, 0, 0);
            if (i0 < row_stride) {
                a = *reinterpret_cast<const uint4*>(row + i0);
                b = *reinterpret_cast<const uint4*>(row + i0 + 4);
            }
        }
        uint32_t v[kPer] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w };
        uint32_t e[kPer], tot = 0;
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            if (i0 + j >= nb) v[j] = 0; // stale words beyond the live chunks
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
        uint32_t carry = carry_in, block_total = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            if (w < wave) carry += s_wave[w];
            block_total += s_wave[w];
        }
        const uint32_t ex = carry + inc - tot;
        if (i0 < row_stride) {
            *reinterpret_cast<uint4*>(row + i0)     = make_uint4(ex + e[0], ex + e[1], ex + e[2], ex + e[3]);
            *reinterpret_cast<uint4*>(row + i0 + 4) = make_uint4(ex + e[4], ex + e[5], ex + e[6], ex + e[7]);
        }
        carry_in += block_total;
        __syncthreads(); // s_wave is reused by the next round
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry_in;
}

// The depth sort's first pass reads the cull pass's per-chunk slabs instead of a dense array: workgroup c takes
// chunk c's chunk_info[c].x survivors ({key, splat index, rect} in 16-byte slots, index order), whose dense ids are
// chunk_base[c] + slot -- the ids the single-pass compaction used to hand out -- and, besides the (key, id) pairs in
// sorted position, writes the dense vis_index[id] / rects[id] the later stages gather from.
template <int kItems>
__global__ void __launch_bounds__(kThreads) k_scatter_first(const uint4* __restrict__ slab,
                                                              const uint2* __restrict__ chunk_info,
                                                              const uint32_t* __restrict__ chunk_base,
                                                              uint32_t* __restrict__ keys_out,
                                                              uint32_t* __restrict__ vals_out,
                                                              uint32_t* __restrict__ vis_index,
                                                              uint2* __restrict__ rects, uint32_t mask, int bits,
                                                              const uint32_t* __restrict__ row_excl,
                                                              const uint32_t* __restrict__ totals, uint32_t row_stride,
                                                              // re-ordered scenes (perm[splat index] = file index):
                                                              // the bits of a value above id_bits carry the top bits of
                                                              // the splat's file index (see k_fix_equal_depth_order)
                                                              const uint32_t* __restrict__ perm, uint32_t id_bits,
                                                              uint32_t tag_shift)
{ // (perm == NULL: plain dense ids)
    constexpr int kKPB = kThreads * kItems;
    __shared__ uint32_t s_wave_hist[kWaves][kRadix];
    __shared__ uint32_t s_global_delta[kRadix];
    __shared__ uint32_t s_digit_base[kRadix];
    __shared__ uint32_t s_scan[kWaves];
    __shared__ uint32_t s_keys[kKPB];
    __shared__ uint32_t s_vals[kKPB];

    const int      tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t chunk = blockIdx.x;
    // the chunk's count first (a slab is 40 % full on average: reading only the live slots is worth the round trip),
    // then everything else at once
    const uint32_t n_c  = chunk_info[chunk].x;
    const uint32_t own  = totals[tid];
    const uint32_t rex  = row_excl[(size_t)tid * row_stride + chunk];
    const uint32_t vid0 = chunk_base[chunk];
    uint4          rec[kItems];
#pragma unroll
    for (int r = 0; r < kItems; ++r) {
        const uint32_t li = wave * 64 * kItems + r * 64 + lane;
        rec[r]            = li < n_c ? slab[(size_t)chunk * kKPB + li] : make_uint4(0u, 0u, 0u, 0u);
    }

    {
        uint32_t inc = own;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(inc, off, 64);
            if (lane >= off) inc += o;
        }
        if (lane == 63) s_scan[wave] = inc;
        __syncthreads();
        uint32_t carry = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w)
            if (w < wave) carry += s_scan[w];
        s_digit_base[tid] = carry + inc - own;
    }
#pragma unroll
    for (int w = 0; w < kWaves; ++w) s_wave_hist[w][tid] = 0;
    __syncthreads();

    uint32_t           rank[kItems];
    volatile uint32_t* my_hist = s_wave_hist[wave];
#pragma unroll
    for (int r = 0; r < kItems; ++r) {
        const uint32_t li    = wave * 64 * kItems + r * 64 + lane;
        rank[r]              = 0;
        if ((uint32_t)(wave * 64 * kItems + r * 64) >= n_c) continue; // wave-uniform: the slab is filled from the front
        const bool     valid = li < n_c;
        const uint32_t d     = valid ? (rec[r].x & mask) : 0u;
        unsigned long long peers = __ballot(valid);
        for (int b = 0; b < bits; ++b) {
            const bool               bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        const uint32_t below = __popcll(peers & ((1ull << lane) - 1ull));
        const uint32_t count = __popcll(peers);
        uint32_t       prev  = 0;
        if (valid) prev = my_hist[d];
        __builtin_amdgcn_wave_barrier();
        if (valid && below == 0) my_hist[d] = prev + count;
        __builtin_amdgcn_wave_barrier();
        rank[r] = prev + below;
        if (valid) { // dense copies for the stages that gather by id (coalesced: ids are consecutive in li)
            vis_index[vid0 + li] = rec[r].y;
            rects[vid0 + li]     = make_uint2(rec[r].z, rec[r].w);
            // (ascending splat indices: a sparse sequential read; .y is free once vis_index is written)
            rec[r].y = perm ? (perm[rec[r].y] >> tag_shift) << id_bits : 0u;
        }
    }
    __syncthreads();

    uint32_t wave_off[kWaves];
    uint32_t total = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
        wave_off[w] = total;
        total += s_wave_hist[w][tid];
    }
    uint32_t inc = total;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane == 63) s_scan[wave] = inc;
    __syncthreads();
    uint32_t carry = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w)
        if (w < wave) carry += s_scan[w];
    const uint32_t start = carry + inc - total;
    s_global_delta[tid]  = s_digit_base[tid] + rex - start;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) s_wave_hist[w][tid] = start + wave_off[w];
    __syncthreads();

#pragma unroll
    for (int r = 0; r < kItems; ++r) {
        const uint32_t li = wave * 64 * kItems + r * 64 + lane;
        if (li < n_c) {
            const uint32_t pos = s_wave_hist[wave][rec[r].x & mask] + rank[r];
            s_keys[pos]        = rec[r].x;
            s_vals[pos]        = (vid0 + li) | rec[r].y;
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < n_c; i += kThreads) {
        const uint32_t k   = s_keys[i];
        const uint32_t dst = s_global_delta[k & mask] + i;
        keys_out[dst]      = k;
        vals_out[dst]      = s_vals[i];
    }
}

// The row scan of the first pass, plus one more workgroup (blockIdx == 256) that turns the per-chunk survivor
// counts into dense-id bases and publishes V and the reference's num_rendered.  nb is the host-known chunk count.
__global__ void __launch_bounds__(kThreads) k_rowscan_first(uint32_t* __restrict__ counts, uint32_t nb,
                                                              uint32_t* __restrict__ totals, uint32_t row_stride,
                                                              const uint2* __restrict__ chunk_info,
                                                              uint32_t* __restrict__ chunk_base,
                                                              uint32_t* __restrict__ d_counts)
{
    constexpr int kPer = 16; // one round covers 4096 chunks = 8.4 M splats
    __shared__ uint32_t s_wave[kWaves], s_wave2[kWaves];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool chunks = blockIdx.x == (uint32_t)kRadix;
    uint32_t*  row    = counts + (size_t)blockIdx.x * row_stride;
    uint32_t   carry_in = 0, carry2_in = 0;
    for (uint32_t base = 0; base < nb; base += kThreads * kPer) {
        const uint32_t i0 = base + threadIdx.x * kPer;
        uint32_t       v[kPer], w2[kPer];
        if (chunks) {
#pragma unroll
            for (int j = 0; j < kPer; ++j) {
                const uint2 ci = i0 + j < nb ? chunk_info[i0 + j] : make_uint2(0u, 0u);
                v[j]           = ci.x;
                w2[j]          = ci.y;
            }
        } else {
#pragma unroll
            for (int q = 0; q < kPer / 4; ++q) { // row_stride is a multiple of 8: groups of 4 are whole or absent
                const uint4 a = i0 + 4 * q < row_stride ? *reinterpret_cast<const uint4*>(row + i0 + 4 * q) : make_uint4(0, 0, 0, 0);
                v[4 * q + 0] = a.x; v[4 * q + 1] = a.y; v[4 * q + 2] = a.z; v[4 * q + 3] = a.w;
            }
#pragma unroll
            for (int j = 0; j < kPer; ++j) {
                if (i0 + j >= nb) v[j] = 0;
                w2[j] = 0;
            }
        }
        uint32_t e[kPer], tot = 0, tot2 = 0;
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            e[j] = tot;
            tot += v[j];
            tot2 += w2[j];
        }
        uint32_t inc = tot;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(inc, off, 64);
            if (lane >= off) inc += o;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) tot2 += __shfl_xor(tot2, off, 64);
        if (lane == 63) s_wave[wave] = inc;
        if (lane == 0) s_wave2[wave] = tot2;
        __syncthreads();
        uint32_t carry = carry_in, round_total = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            if (w < wave) carry += s_wave[w];
            round_total += s_wave[w];
            carry2_in += s_wave2[w];
        }
        const uint32_t ex = carry + inc - tot;
        if (chunks) {
#pragma unroll
            for (int j = 0; j < kPer; ++j)
                if (i0 + j < nb) chunk_base[i0 + j] = ex + e[j];
        } else {
#pragma unroll
            for (int q = 0; q < kPer / 4; ++q)
                if (i0 + 4 * q < row_stride)
                    *reinterpret_cast<uint4*>(row + i0 + 4 * q) =
                        make_uint4(ex + e[4 * q], ex + e[4 * q + 1], ex + e[4 * q + 2], ex + e[4 * q + 3]);
        }
        carry_in += round_total;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (chunks) {
            d_counts[0] = carry_in;  // V: splats that emit >= 1 pair
            d_counts[1] = carry2_in; // the reference's num_rendered (gs_tile_splatter/impl.cpp:106)
            d_counts[kCountTieUnresolved] = 0u; // (tie_order.hpp: k_fix_equal_depth_order adds to it)
        } else {
            totals[blockIdx.x] = carry_in;
        }
    }
}

template <int kItems, typename K>
__global__ void __launch_bounds__(kThreads) k_scatter(const K* __restrict__ keys_in,
                                                        const uint32_t* __restrict__ vals_in,
                                                        K* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                        const uint32_t* __restrict__ d_n, uint32_t n_cap, int shift,
                                                        uint32_t mask,
                                                        int bits, const uint32_t* __restrict__ row_excl,
                                                        const uint32_t* __restrict__ totals, uint32_t row_stride)
{
    constexpr int kKPB = kThreads * kItems;
    __shared__ uint32_t s_wave_hist[kWaves][kRadix];
    __shared__ uint32_t s_global_delta[kRadix];
    __shared__ uint32_t s_digit_base[kRadix];
    __shared__ uint32_t s_scan[kWaves];
    __shared__ K        s_keys[kKPB];
    __shared__ uint32_t s_vals[kKPB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // everything the first chunk needs is requested before the element count arrives (capacity-guarded)
    uint32_t       chunk = blockIdx.x;
    const uint32_t own   = totals[tid];
    uint32_t       rex   = chunk < row_stride ? row_excl[(size_t)tid * row_stride + chunk] : 0u;
    K              key[kItems];
    uint32_t       val[kItems];
#pragma unroll
    for (int r = 0; r < kItems; ++r) {
        const uint32_t i = chunk * kKPB + wave * 64 * kItems + r * 64 + lane;
        key[r]           = i < n_cap ? keys_in[i] : (K)0;
        val[r]           = i < n_cap ? vals_in[i] : 0u;
    }
    const uint32_t n  = d_n ? *d_n : n_cap;
    const uint32_t nb = (n + kKPB - 1) / kKPB;
    if (blockIdx.x >= nb) return;

    // global exclusive base of each digit from the 256 row totals
    {
        uint32_t inc = own;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(inc, off, 64);
            if (lane >= off) inc += o;
        }
        if (lane == 63) s_scan[wave] = inc;
        __syncthreads();
        uint32_t carry = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w)
            if (w < wave) carry += s_scan[w];
        s_digit_base[tid] = carry + inc - own;
        __syncthreads();
    }

    for (;;) {
        const uint32_t block_base = chunk * kKPB;
        const uint32_t wave_base  = block_base + wave * 64 * kItems;
        const uint32_t in_block   = (n - block_base) < (uint32_t)kKPB ? (n - block_base) : (uint32_t)kKPB;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) s_wave_hist[w][tid] = 0;
        __syncthreads();

        uint32_t rank[kItems];
        volatile uint32_t* my_hist = s_wave_hist[wave];
        // ---- per-wave stable ranking, 64 keys per round in memory order
#pragma unroll
        for (int r = 0; r < kItems; ++r) {
            const uint32_t i     = wave_base + r * 64 + lane;
            const bool     valid = i < n;
            if (!valid) key[r] = (K)0;
            const uint32_t d     = (uint32_t)(key[r] >> shift) & mask;
            unsigned long long peers = __ballot(valid);
            for (int b = 0; b < bits; ++b) {
                const bool               bit = (d >> b) & 1u;
                const unsigned long long bal = __ballot(bit);
                peers &= bit ? bal : ~bal;
            }
            const uint32_t below = __popcll(peers & ((1ull << lane) - 1ull));
            const uint32_t count = __popcll(peers);
            uint32_t       prev  = 0;
            if (valid) prev = my_hist[d];
            __builtin_amdgcn_wave_barrier();
            if (valid && below == 0) my_hist[d] = prev + count;
            __builtin_amdgcn_wave_barrier();
            rank[r] = prev + below;
        }
        __syncthreads();

        // ---- thread d owns digit d: combine the waves; chunk-local start; global position
        uint32_t wave_off[kWaves];
        uint32_t total = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            wave_off[w] = total;
            total += s_wave_hist[w][tid];
        }
        uint32_t inc = total;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(inc, off, 64);
            if (lane >= off) inc += o;
        }
        if (lane == 63) s_scan[wave] = inc;
        __syncthreads();
        uint32_t carry = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w)
            if (w < wave) carry += s_scan[w];
        const uint32_t start = carry + inc - total;
        s_global_delta[tid]  = s_digit_base[tid] + rex - start;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) s_wave_hist[w][tid] = start + wave_off[w];
        __syncthreads();

        // ---- stage into chunk-sorted order in LDS, then scatter each digit's run coalesced
#pragma unroll
        for (int r = 0; r < kItems; ++r) {
            const uint32_t i = wave_base + r * 64 + lane;
            if (i < n) {
                const uint32_t d   = (uint32_t)(key[r] >> shift) & mask;
                const uint32_t pos = s_wave_hist[wave][d] + rank[r];
                s_keys[pos]        = key[r];
                s_vals[pos]        = val[r];
            }
        }
        __syncthreads();
        for (uint32_t i = tid; i < in_block; i += kThreads) {
            const K        k   = s_keys[i];
            const uint32_t dst = s_global_delta[(uint32_t)(k >> shift) & mask] + i;
            keys_out[dst]      = k;
            vals_out[dst]      = s_vals[i];
        }
        chunk += gridDim.x;
        if (chunk >= nb) break;
        __syncthreads();
        rex = row_excl[(size_t)tid * row_stride + chunk];
        for (int r = 0; r < kItems; ++r) { // (constant trip count: unrolled without being asked)
            const uint32_t i = chunk * kKPB + wave * 64 * kItems + r * 64 + lane;
            key[r]           = i < n ? keys_in[i] : (K)0;
            val[r]           = i < n ? vals_in[i] : 0u;
        }
    }
}

// Equal depths in a re-ordered scene (tie_order.hpp): puts every run of equal keys of the sorted survivors back into
// file order, in place.  Only the first member of a run does anything: a lane reads kTieLane consecutive keys and their
// neighbours (requested before the survivor count arrives), and values only if one of its elements starts a run.
//   run of 2 (nearly all)      compare the tags, swap in place
//   run of 3 .. kTieRunShort   selection sort in place by the first member's lane
//   longer, up to kTieRunCap   the first member's workgroup ranks the members by file index through LDS
//   longer still               left in the context's order and counted in d_counts[kCountTieUnresolved]
constexpr uint32_t kTieLane = 8;                   // elements per lane: the whole frame is resident at once
constexpr uint32_t kTieSpan = kThreads * kTieLane; // elements per workgroup step

// A run of MORE than kTieRunCap equal keys (a plane seen head-on: thousands of splats at exactly one depth), put into
// file order by its first member's workgroup: a stable LSD radix sort on the members' file indices (8-bit digits, all
// id_bits of them) through global scratch -- (scratch_k1, vals) and (scratch_k0, scratch_v) ping-pong inside the run's own
// slot [lo, lo + m), which no other workgroup touches (only a run's first member acts, and a member's value is never
// read by anybody else; the KEYS of the run are left alone: neighbouring workgroups compare them).  One workgroup, three
// barriers per 256 members and pass: milliseconds for a run of 100 K -- the price of a pathological frame, paid only by
// it; every shorter run keeps the paths above.  `s_lds` is the kTieRunCap-word LDS array of the ranked path.
__device__ void sort_long_run_by_file_index(uint32_t* vals, uint32_t lo, uint32_t m, const TieOrder& tie, uint32_t* s_lds)
{
    uint32_t* s_base = s_lds;           // [256] running start of each digit
    uint32_t* s_hist = s_lds + 256;     // [256]
    uint32_t* s_wcnt = s_lds + 512;     // [kWaves][256] per-wave digit counts of the current block
    uint32_t* s_scan = s_lds + 512 + kWaves * 256; // [kWaves]
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t id_mask = (1u << tie.id_bits) - 1u;
    const int      n_pass  = (int)((tie.id_bits + 7u) / 8u);
    uint32_t*      k0 = tie.scratch_k0 + lo;
    uint32_t*      k1 = tie.scratch_k1 + lo;
    uint32_t*      sv = tie.scratch_v + lo;
    uint32_t*      rv = vals + lo;
    __syncthreads();
    for (uint32_t j = tid; j < m; j += kThreads) k1[j] = tie.perm[tie.vis_index[rv[j] & id_mask]]; // the sort keys
    for (int p = 0; p < n_pass; ++p) {
        const uint32_t* src_k = (p & 1) ? k0 : k1;
        const uint32_t* src_v = (p & 1) ? sv : rv;
        uint32_t*       dst_k = (p & 1) ? k1 : k0;
        uint32_t*       dst_v = (p & 1) ? rv : sv;
        const int       shift = 8 * p;
        s_hist[tid] = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) s_wcnt[w * 256 + tid] = 0;
        __syncthreads(); // (also: the previous pass's stores are visible to the whole workgroup)
        for (uint32_t j = tid; j < m; j += kThreads) atomicAdd(&s_hist[(src_k[j] >> shift) & 255u], 1u);
        __syncthreads();
        { // exclusive scan of the 256 digit counts: thread d owns digit d
            const uint32_t own = s_hist[tid];
            uint32_t       inc = own;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t o = __shfl_up(inc, off, 64);
                if (lane >= (uint32_t)off) inc += o;
            }
            if (lane == 63u) s_scan[wave] = inc;
            __syncthreads();
            uint32_t carry = 0;
            for (uint32_t w = 0; w < wave; ++w) carry += s_scan[w];
            s_base[tid] = carry + inc - own;
        }
        __syncthreads();
        for (uint32_t base = 0; base < m; base += kThreads) { // members in order, 256 at a time: a STABLE scatter
            const uint32_t j     = base + tid;
            const bool     valid = j < m;
            const uint32_t f = valid ? src_k[j] : 0u, v = valid ? src_v[j] : 0u;
            const uint32_t d = (f >> shift) & 255u;
            unsigned long long peers = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const bool               bit = (d >> b) & 1u;
                const unsigned long long bal = __ballot(bit);
                peers &= bit ? bal : ~bal;
            }
            const uint32_t below = __popcll(peers & ((1ull << lane) - 1ull));
            if (valid && below == 0u) s_wcnt[wave * 256 + d] = (uint32_t)__popcll(peers);
            __syncthreads();
            if (valid) {
                uint32_t at = s_base[d] + below;
                for (uint32_t w = 0; w < wave; ++w) at += s_wcnt[w * 256 + d];
                dst_k[at] = f;
                dst_v[at] = v;
            }
            __syncthreads();
            { // digit tid: advance its start by what this block placed, clear the per-wave counts
                uint32_t add = 0;
#pragma unroll
                for (int w = 0; w < kWaves; ++w) {
                    add += s_wcnt[w * 256 + tid];
                    s_wcnt[w * 256 + tid] = 0;
                }
                s_base[tid] += add;
            }
            __syncthreads();
        }
    }
    if (n_pass & 1) { // the last pass landed in the scratch pair
        __syncthreads();
        for (uint32_t j = tid; j < m; j += kThreads) rv[j] = sv[j];
    }
    __syncthreads();
}

__global__ void __launch_bounds__(kThreads) k_fix_equal_depth_order(const uint32_t* __restrict__ keys, uint32_t* vals,
                                                                      uint32_t v_cap, TieOrder tie)
{
    __shared__ uint32_t s_n, s_len;
    __shared__ uint32_t s_head[kTieSpan / (kTieRunShort + 1u) + 4u]; // long runs start >= kTieRunShort + 1 apart
    __shared__ uint32_t s_f[kTieRunCap];
    const uint32_t tid     = threadIdx.x;
    const uint32_t id_mask = (1u << tie.id_bits) - 1u;
    // keys i0 - 1 .. i0 + kTieLane + 1 around the lane's elements; v_cap is a multiple of 4 or the tail goes one by one
    auto load_keys = [&](uint32_t i0, uint32_t limit, uint32_t (&kk)[kTieLane + 3]) {
#pragma unroll
        for (uint32_t g = 0; g < kTieLane; g += 4) {
            if (i0 + g + 4u <= limit) {
                const uint4 k4 = *reinterpret_cast<const uint4*>(keys + i0 + g);
                kk[g + 1] = k4.x, kk[g + 2] = k4.y, kk[g + 3] = k4.z, kk[g + 4] = k4.w;
            } else {
#pragma unroll
                for (uint32_t e = 0; e < 4; ++e) kk[g + e + 1] = i0 + g + e < limit ? keys[i0 + g + e] : 0u;
            }
        }
        kk[0]            = (i0 > 0u && i0 <= limit) ? keys[i0 - 1u] : 0u;
        kk[kTieLane + 1] = i0 + kTieLane < limit ? keys[i0 + kTieLane] : 0u;
        kk[kTieLane + 2] = i0 + kTieLane + 1u < limit ? keys[i0 + kTieLane + 1u] : 0u;
    };
    uint32_t base = blockIdx.x * kTieSpan;
    uint32_t kk[kTieLane + 3];
    load_keys(base + tid * kTieLane, v_cap, kk); // (entries beyond the live count are never compared)
    const uint32_t V = tie.d_counts[0];
    for (; base < V; base += gridDim.x * kTieSpan) { // (workgroup-uniform)
        const uint32_t i0 = base + tid * kTieLane;
        if (base != blockIdx.x * kTieSpan) load_keys(i0, V, kk);
        if (tid == 0) {
            s_n   = 0;
            s_len = 0xFFFFFFFFu;
        }
        __syncthreads();
        // element e starts a run: its left neighbour differs, its right one does not
        uint32_t heads = 0, more = 0;
#pragma unroll
        for (uint32_t e = 0; e < kTieLane; ++e) {
            const uint32_t i = i0 + e;
            const bool     h = i + 1u < V && kk[e + 1] == kk[e + 2] && !(i > 0u && kk[e] == kk[e + 1]);
            const bool     t = i + 2u < V && kk[e + 2] == kk[e + 3]; // a third member
            heads |= (h ? 1u : 0u) << e;
            more |= (h && t ? 1u : 0u) << e;
        }
        if (heads) {
            // the values of the lane's elements and of the one after them: all a run of two needs
            uint32_t vv[kTieLane + 1];
#pragma unroll
            for (uint32_t g = 0; g < kTieLane; g += 4) {
                if (i0 + g + 4u <= V) {
                    const uint4 v4 = *reinterpret_cast<const uint4*>(vals + i0 + g);
                    vv[g] = v4.x, vv[g + 1] = v4.y, vv[g + 2] = v4.z, vv[g + 3] = v4.w;
                } else {
#pragma unroll
                    for (uint32_t e = 0; e < 4; ++e) vv[g + e] = i0 + g + e < V ? vals[i0 + g + e] : 0u;
                }
            }
            vv[kTieLane] = i0 + kTieLane < V ? vals[i0 + kTieLane] : 0u;
#pragma unroll
            for (uint32_t e = 0; e < kTieLane; ++e) { // two splats whose depths round to the same float: the usual run
                if (((heads & ~more) >> e) & 1u) {
                    if (tie_before(vv[e + 1], vv[e], tie.id_bits, tie.vis_index, tie.perm)) {
                        vals[i0 + e]      = vv[e + 1];
                        vals[i0 + e + 1u] = vv[e];
                    }
                }
            }
            while (more) { // (rare) three members or more
                const uint32_t e = (uint32_t)__ffs((int)more) - 1u;
                more &= more - 1u;
                const uint32_t i = i0 + e, k = keys[i];
                uint32_t       hi = i + 2u;
                while (hi + 1u < V && hi - i < kTieRunShort && keys[hi + 1u] == k) ++hi;
                const uint32_t m = hi - i + 1u;
                if (m > kTieRunShort) { // (at least that long) left to the workgroup, below
                    s_head[atomicAdd(&s_n, 1u)] = i;
                    continue;
                }
                for (uint32_t j = 0; j + 1u < m; ++j) { // (one lane, same addresses: its loads see its stores)
                    const uint32_t vj   = vals[i + j];
                    uint32_t       best = j, vb = vj;
                    for (uint32_t q = j + 1u; q < m; ++q) {
                        const uint32_t vc = vals[i + q];
                        if (tie_before(vc, vb, tie.id_bits, tie.vis_index, tie.perm)) best = q, vb = vc;
                    }
                    if (best != j) {
                        vals[i + j]    = vb;
                        vals[i + best] = vj;
                    }
                }
            }
        }
        __syncthreads();
        const uint32_t n_long = s_n;
        for (uint32_t r = 0; r < n_long; ++r) { // (rare) runs of more than kTieRunShort members: all 256 lanes on each
            const uint32_t lo = s_head[r], k = keys[lo];
            for (uint32_t off = 1u;; off += kThreads) { // the run's length: the first position that is not a member
                const uint32_t j    = off + tid;
                const bool     stop = lo + j >= V || keys[lo + j] != k;
                if (stop) atomicMin(&s_len, j);
                if (__syncthreads_or(stop)) break;
            }
            const uint32_t m = s_len;
            if (m > kTieRunCap) {
                if (tie.scratch_k0 && tie.scratch_k1 && tie.scratch_v) sort_long_run_by_file_index(vals, lo, m, tie, s_f);
                else if (tid == 0) atomicAdd(&tie.d_counts[kCountTieUnresolved], m); // (no scratch given: reported)
            } else {
                uint32_t own[kTieRunCap / kThreads]; // the members this lane ranks (read before anything is written)
#pragma unroll
                for (uint32_t t = 0; t < kTieRunCap / kThreads; ++t) {
                    const uint32_t j = tid + t * kThreads;
                    own[t]           = j < m ? vals[lo + j] : 0u;
                    if (j < m) s_f[j] = tie.perm[tie.vis_index[own[t] & id_mask]];
                }
                __syncthreads();
#pragma unroll
                for (uint32_t t = 0; t < kTieRunCap / kThreads; ++t) {
                    const uint32_t j = tid + t * kThreads;
                    if (j >= m) break;
                    const uint32_t mine = s_f[j];
                    uint32_t       rank = 0;
                    for (uint32_t q = 0; q < m; ++q) rank += s_f[q] < mine ? 1u : 0u; // (file indices are distinct)
                    vals[lo + rank] = own[t];
                }
            }
            __syncthreads();
            if (tid == 0) s_len = 0xFFFFFFFFu;
            __syncthreads();
        }
    }
}

} // namespace

// Keys per chunk.  Small inputs (the survivors' depth sort, V ~ 2.4 M) are latency-bound: more, smaller chunks keep
// more workgroups in flight (2048: 0.149 ms vs 0.158 ms at 4096, 0.189 ms at 1024); the pair partition (L ~ 7.5 M) prefers the longer
// per-bucket runs of 4096-key chunks (0.141 vs 0.150 ms).
constexpr int64_t kSmallInput = 4 << 20;
inline int items_for(int64_t expected) { return expected <= kSmallInput ? 8 : 16; }

inline int64_t row_stride_for(int64_t n_cap, int items)
{
    const int64_t nb = (n_cap + kThreads * items - 1) / (kThreads * items);
    return (nb + 7) & ~(int64_t)7; // rows start 32-byte aligned (k_rowscan's 16-byte accesses)
}

size_t pair_sort_ws_bytes(int64_t n_cap)
{
    const int64_t stride = row_stride_for(n_cap, 8); // the smaller chunk size bounds the table
    return (size_t)(stride * kRadix + kRadix + 64) * sizeof(uint32_t);
}

// What a producer needs to leave pass 0's per-chunk digit counts behind (so that the sort skips its first k_hist):
// counts[digit * nb + chunk] for all 256 digits, nb = ceil(n / keys_per_chunk), digit = (key >> shift) & mask.
PairSortFirstPass pair_sort_first_pass(int64_t n_cap, int64_t grid_hint, int begin_bit, int end_bit, void* ws)
{
    PairSortFirstPass fp;
    const int n_pass  = (end_bit - begin_bit + 7) / 8;
    const int bits    = n_pass > 0 ? (end_bit - begin_bit + n_pass - 1) / n_pass : 0;
    fp.shift          = begin_bit;
    fp.mask           = (1u << bits) - 1u;
    fp.keys_per_chunk = kThreads * items_for(grid_hint > 0 ? grid_hint : n_cap);
    fp.counts         = reinterpret_cast<uint32_t*>(ws);
    fp.row_stride     = (uint32_t)row_stride_for(n_cap, fp.keys_per_chunk / kThreads);
    fp.valid          = n_pass > 0 && n_cap > 0;
    return fp;
}

// The depth sort fed by the cull pass's chunk slabs (see k_scatter_first).  Chunks are the cull pass's 2048 splats.
DepthSortFirstPass depth_sort_first_pass(int64_t P, void* ws)
{
    DepthSortFirstPass fp;
    fp.mask       = 0xFFu; // 32 key bits in 4 passes of 8
    fp.row_stride = (uint32_t)row_stride_for(P, 8);
    fp.counts     = reinterpret_cast<uint32_t*>(ws);
    return fp;
}

namespace
{
// pass p reads (src_k[p], src_v[p]) and writes (dst_k[p], dst_v[p])
template <int kItems, typename K>
void run_passes(const K* const* src_k, const uint32_t* const* src_v, K* const* dst_k, uint32_t* const* dst_v, int n_pass,
                const uint32_t* d_n, int64_t n_cap, int64_t grid_hint, int begin_bit, int end_bit, void* ws_,
                hipStream_t stream, bool first_hist_done = false)
{
    constexpr int kKPB   = kThreads * kItems;
    const int64_t nb_cap = (n_cap + kKPB - 1) / kKPB;
    const uint32_t stride = (uint32_t)row_stride_for(n_cap, kItems);
    uint32_t*     counts = reinterpret_cast<uint32_t*>(ws_);
    uint32_t*     totals = counts + (size_t)stride * kRadix;
    int64_t       hint   = grid_hint > 0 ? grid_hint : n_cap;
    int64_t       blocks = (hint + kKPB - 1) / kKPB;
    if (blocks > nb_cap) blocks = nb_cap;
    if (blocks < 1) blocks = 1;
    const uint32_t n_host = (uint32_t)n_cap;
    int            shift  = begin_bit;
    for (int p = 0; p < n_pass; ++p) {
        // the live bits are split evenly over the passes (13 tile bits: 7 + 6, not 8 + 5): fewer buckets per pass
        // means longer contiguous runs per bucket in the scatter's stores
        const int      bits = (end_bit - shift + (n_pass - p) - 1) / (n_pass - p);
        const uint32_t mask = (1u << bits) - 1u;
        if (!(p == 0 && first_hist_done)) // the producer of the keys may have left pass 0's chunk counts in `counts`
            hipLaunchKernelGGL((k_hist<kItems, K>), dim3((unsigned)blocks), dim3(kThreads), 0, stream, src_k[p], d_n,
                               n_host, shift, mask, counts, stride);
        hipLaunchKernelGGL(k_rowscan<kItems>, dim3(kRadix), dim3(kThreads), 0, stream, counts, d_n, n_host, totals, stride);
        hipLaunchKernelGGL((k_scatter<kItems, K>), dim3((unsigned)blocks), dim3(kThreads), 0, stream, src_k[p], src_v[p],
                           dst_k[p], dst_v[p], d_n, n_host, shift, mask, bits, counts, totals, stride);
        shift += bits;
    }
}
} // namespace

// Ping-pongs a -> b -> a ...; returns 0 if the result ended in (keys_a, vals_a), 1 if in (keys_b, vals_b).
// grid_hint: expected element count (bounds the launch; larger live counts are handled by chunk striding).
int launch_pair_sort_u32(uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a, uint32_t* vals_b, const uint32_t* d_n,
                         int64_t n_cap, int64_t grid_hint, int begin_bit, int end_bit, void* ws_, hipStream_t stream,
                         bool first_hist_done)
{
    if (n_cap <= 0) return 0;
    const int n_pass = (end_bit - begin_bit + 7) / 8;
    if (n_pass <= 0) return 0;
    const uint32_t* sk[8];
    const uint32_t* sv[8];
    uint32_t*       dk[8];
    uint32_t*       dv[8];
    uint32_t*       kb[2] = { keys_a, keys_b };
    uint32_t*       vb[2] = { vals_a, vals_b };
    for (int p = 0; p < n_pass; ++p) {
        sk[p] = kb[p & 1];
        sv[p] = vb[p & 1];
        dk[p] = kb[(p & 1) ^ 1];
        dv[p] = vb[(p & 1) ^ 1];
    }
    if (items_for(grid_hint > 0 ? grid_hint : n_cap) == 8)
        run_passes<8, uint32_t>(sk, sv, dk, dv, n_pass, d_n, n_cap, grid_hint, begin_bit, end_bit, ws_, stream,
                                first_hist_done);
    else
        run_passes<16, uint32_t>(sk, sv, dk, dv, n_pass, d_n, n_cap, grid_hint, begin_bit, end_bit, ws_, stream,
                                 first_hist_done);
    return n_pass & 1;
}

// Survivors by depth bits: pass 0 from the slabs into (keys_b, vals_b), passes 1-3 ping-pong; the result ends in
// (keys_a, vals_a).  d_counts[0] / [1] (V, num_rendered) are written by pass 0's row-scan launch; `fork`, if given,
// is signalled when pass 0's scatter (which writes vis_index / rects) completes.
void launch_depth_sort_from_chunks(int64_t P, int64_t v_hint, const uint4* slab, const uint2* chunk_info,
                                   uint32_t* chunk_base, uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a,
                                   uint32_t* vals_b, uint32_t* vis_index, uint2* rects, uint32_t* d_counts, void* ws_,
                                   hipStream_t stream, hipEvent_t fork, const TieOrder* tie, bool first_pass_only)
{
    if (P <= 0) return;
    constexpr int  kItems = 8;
    static_assert(kThreads * kItems == kCullChunkSplats, "one workgroup of the first pass per cull chunk");
    const uint32_t nb     = (uint32_t)((P + kThreads * kItems - 1) / (kThreads * kItems));
    const uint32_t stride = (uint32_t)row_stride_for(P, kItems);
    uint32_t*      counts = reinterpret_cast<uint32_t*>(ws_);
    uint32_t*      totals = counts + (size_t)stride * kRadix;
    hipLaunchKernelGGL(k_rowscan_first, dim3(kRadix + 1), dim3(kThreads), 0, stream, counts, nb, totals, stride, chunk_info,
                       chunk_base, d_counts);
    hipExtLaunchKernelGGL(k_scatter_first<kItems>, dim3(nb), dim3(kThreads), 0, stream, nullptr, fork, 0, slab, chunk_info,
                          chunk_base, keys_b, vals_b, vis_index, rects, 0xFFu, 8, counts, totals, stride,
                          tie ? tie->perm : nullptr, tie ? tie->id_bits : 32u, tie ? tie->tag_shift : 0u);
    if (first_pass_only) return; // (the compaction: dense vis_index / rects, V and num_rendered -- splat-ownership owners)
    const uint32_t* sk[3] = { keys_b, keys_a, keys_b };
    const uint32_t* sv[3] = { vals_b, vals_a, vals_b };
    uint32_t*       dk[3] = { keys_a, keys_b, keys_a };
    uint32_t*       dv[3] = { vals_a, vals_b, vals_a };
    run_passes<kItems, uint32_t>(sk, sv, dk, dv, 3, d_counts, P, v_hint, 8, 32, ws_, stream);
    if (tie) {
        int64_t blocks = ((v_hint > 0 ? v_hint : P) + kTieSpan - 1) / kTieSpan; // (larger live counts are strided)
        if (blocks > 16384) blocks = 16384;
        TieOrder t = *tie; // the other half of the ping-pong is free behind the last pass: scratch for runs beyond the cap
        t.scratch_k0 = keys_b;
        t.scratch_v  = vals_b;
        hipLaunchKernelGGL(k_fix_equal_depth_order, dim3((unsigned)std::max<int64_t>(blocks, 1)), dim3(kThreads), 0, stream,
                           keys_a, vals_a, (uint32_t)P, t);
    }
}

// The equal-depth pass on its own (a depth sort that did not come through launch_depth_sort_from_chunks): (keys, vals) sorted,
// (scratch_k, scratch_v) the free half of the sort's ping-pong.
void launch_fix_equal_depth_order(uint32_t* keys, uint32_t* vals, uint32_t* scratch_k, uint32_t* scratch_v, int64_t n_cap,
                                  int64_t v_hint, const TieOrder& tie, hipStream_t stream)
{
    if (n_cap <= 0) return;
    int64_t blocks = ((v_hint > 0 ? v_hint : n_cap) + kTieSpan - 1) / kTieSpan;
    if (blocks > 16384) blocks = 16384;
    TieOrder t   = tie;
    t.scratch_k0 = scratch_k;
    t.scratch_v  = scratch_v;
    hipLaunchKernelGGL(k_fix_equal_depth_order, dim3((unsigned)std::max<int64_t>(blocks, 1)), dim3(kThreads), 0, stream, keys, vals,
                       (uint32_t)n_cap, t);
}

// The stage-level sort (lcpp DeviceRadixSort::SortPairs<ulong, uint>, call site gs_tile_splatter/impl.cpp:135-143):
// n host-known pairs, inputs left intact, result in (keys_out, vals_out); (keys_tmp, vals_tmp) is scratch of n elements.
void launch_pair_sort_u64_preserve(const uint64_t* keys_in, const uint32_t* vals_in, uint64_t* keys_out,
                                   uint32_t* vals_out, uint64_t* keys_tmp, uint32_t* vals_tmp, int64_t n, int begin_bit,
                                   int end_bit, void* ws_, hipStream_t stream)
{
    if (n <= 0) return;
    const int n_pass = (end_bit - begin_bit + 7) / 8;
    if (n_pass <= 0) { // nothing to sort on: the result is a copy
        (void)hipMemcpyAsync(keys_out, keys_in, (size_t)n * 8, hipMemcpyDeviceToDevice, stream);
        (void)hipMemcpyAsync(vals_out, vals_in, (size_t)n * 4, hipMemcpyDeviceToDevice, stream);
        return;
    }
    const uint64_t* sk[8];
    const uint32_t* sv[8];
    uint64_t*       dk[8];
    uint32_t*       dv[8];
    // the last pass must land in `out`: destinations alternate backwards from it
    for (int p = 0; p < n_pass; ++p) {
        const bool to_out = ((n_pass - 1 - p) & 1) == 0;
        dk[p]             = to_out ? keys_out : keys_tmp;
        dv[p]             = to_out ? vals_out : vals_tmp;
        sk[p]             = p == 0 ? keys_in : dk[p - 1];
        sv[p]             = p == 0 ? vals_in : dv[p - 1];
    }
    run_passes<8, uint64_t>(sk, sv, dk, dv, n_pass, nullptr, n, n, begin_bit, end_bit, ws_, stream);
}

} // namespace lcgs
