// tie_order.hpp -- the blend order of splats whose depth keys are exactly equal, in a scene the context re-ordered.
//
// The reference sorts on (tile << 32 | depth bits) with a stable sort, so splats of equal depth are blended in ascending
// FILE index (gs_tile_splatter/impl.cpp:135-143).  The depth sort here is stable in dense-id order = ascending index of
// the scene AS THE CONTEXT HOLDS IT; once lcgs_scene_reorder_spatial has permuted the scene that is no longer the
// file's order, and alpha compositing does not commute (about one survivor in six shares its depth bits with another
// in a 5 M-splat frame).  The sort therefore puts every run of equal keys back into file order:
//
//   * the sorted VALUES carry, above the id_bits of the dense id, the top (32 - id_bits) bits of the splat's file index
//     (written by the first pass, carried by the others for nothing): two members of a run are almost always ordered
//     by those tags alone; only equal tags (one pair in 2^(32 - id_bits)) gather the full index perm[vis_index[id]];
//   * one pass over the sorted keys (k_fix_equal_depth_order, pair_sort.hip) finds the runs; only the first member of a
//     run does anything, in place: runs of two -- nearly all -- are a compare and a swap, runs of up to kTieRunShort a
//     selection sort by one lane, runs of up to kTieRunCap members are ranked by their workgroup through LDS; longer
//     ones -- thousands of splats at exactly one depth: a plane seen head-on by an axis-aligned camera -- are radix
//     sorted on their file indices through global scratch by the same workgroup (round 3; until then they stayed in the
//     context's order and were counted in lcgs_frame_stats.equal_depth_unresolved, which is now always 0).
// Readers of the sorted order mask the values with (1 << id_bits) - 1.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace lcgs
{

constexpr uint32_t kTieRunCap   = 4096; // longest run that is put back into file order
constexpr uint32_t kTieRunShort = 16;   // runs up to this length are resolved by their first member alone

constexpr int kCountTieUnresolved = 9; // d_counts slot: members of runs left in the context's order

struct TieOrder { // (all device pointers)
    uint32_t*       d_counts  = nullptr;
    const uint32_t* vis_index = nullptr; // dense id -> index of the splat in the context's arrays
    const uint32_t* perm      = nullptr; // that index -> file index
    uint32_t        id_bits   = 32;
    uint32_t        tag_shift = 0; // tag = file index >> tag_shift
    // runs of more than kTieRunCap members are sorted through global scratch (three arrays of >= V words each; the depth
    // sort supplies the free half of its ping-pong for two of them, the context a third)
    uint32_t *scratch_k0 = nullptr, *scratch_k1 = nullptr, *scratch_v = nullptr;
};

// does the splat behind sorted value a come before the one behind b in the file?
__device__ __forceinline__ bool tie_before(uint32_t a, uint32_t b, uint32_t id_bits, const uint32_t* __restrict__ vis_index,
                                           const uint32_t* __restrict__ perm)
{
    const uint32_t ta = a >> id_bits, tb = b >> id_bits;
    if (ta != tb) return ta < tb;
    const uint32_t id_mask = (1u << id_bits) - 1u; // (equal tags: the full file indices, which are distinct)
    return perm[vis_index[a & id_mask]] < perm[vis_index[b & id_mask]];
}

} // namespace lcgs
