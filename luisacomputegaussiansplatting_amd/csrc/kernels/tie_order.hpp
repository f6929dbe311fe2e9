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
//   * the LAST radix pass holds each chunk in LDS in fully sorted order, so runs of up to kTieRunShort members that lie
//     inside one chunk are fixed there, before they are written out, at no memory traffic;
//   * what is left -- longer runs, and runs that may continue in the neighbouring chunk -- goes onto a list of output
//     positions that one small launch works through (k_fix_listed_runs): runs of up to kTieRunCap members are ranked
//     through LDS; longer ones stay in the context's order and are counted (lcgs_frame_stats.equal_depth_unresolved):
//     only a scene with thousands of splats at exactly one depth -- a plane seen head-on by an axis-aligned camera --
//     gets there, and LCGS_ORDER_FILE is exact for those.
// Readers of the sorted order mask the values with (1 << id_bits) - 1.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace lcgs
{

constexpr uint32_t kTieRunCap   = 4096; // longest run that is put back into file order
constexpr uint32_t kTieRunShort = 16;   // runs up to this length are resolved inside the last radix pass

// d_counts slots of the frame's counter block used here
constexpr int kCountTieUnresolved = 9;  // members of runs left in the context's order
constexpr int kCountTieListed     = 10; // entries on the shared tail of the run list

// the run list: one line of kTieChunkLine words per chunk of the last radix pass -- {count, positions ...} -- and a
// shared tail for chunks that list more than a line holds
constexpr uint32_t kTieChunkLine  = 16;
constexpr uint32_t kTieChunkSlots = kTieChunkLine - 1;

struct TieOrder { // (all device pointers)
    uint32_t*       d_counts     = nullptr;
    uint32_t*       list         = nullptr; // output positions of the first member (in its chunk) of a run to look at again
    uint32_t*       overflow     = nullptr; // the shared tail; d_counts[kCountTieListed] counts its entries
    uint32_t        overflow_cap = 0;
    const uint32_t* vis_index = nullptr; // dense id -> index of the splat in the context's arrays
    const uint32_t* perm      = nullptr; // that index -> file index
    uint32_t        id_bits   = 32;
    uint32_t        tag_shift = 0; // tag = file index >> tag_shift
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
