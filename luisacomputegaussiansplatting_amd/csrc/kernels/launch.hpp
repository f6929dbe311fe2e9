// launch.hpp -- host-callable launchers of every HIP kernel in the library.
// All launchers only enqueue work on `stream`; none allocates or synchronises.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gs_math.hpp"

namespace lcgs
{

// ---- stage_kernels.hip : one kernel per reference shader (buffers in the reference's layouts) ----
void launch_sh_process(int P, int deg, const CamParams& cp, const float* pos, const float* sh, float* color,
                       hipStream_t stream);
void launch_project(int P, const CamParams& cp, bool use_focal, const float* pos, const float* scale,
                    const float* rotq, float scale_modifier, float* means_2d, float* depth, float* covs_2d,
                    hipStream_t stream);
// hole_flag (optional, device word, cleared by the caller): set when a splat claims pair slots that copy_with_keys leaves
// unwritten (radius <= 0 with tiles > 0: a NaN covariance) -- only then does the reference's zero-fill of the pair buffers matter
// hole_flag (nullable): set to hole_mark (a value the caller has not used before: no zero-fill per frame) when a splat claims pair
// slots its copy_with_keys will not write.  flags (nullable, flags_len >= P bytes): flags[i] = splat i claims pair slots
// (tiles_touched[i] > 0), zeros behind P -- the splatter's compaction input
void launch_allocate_tiles(int P, const CamParams& cp, bool use_focal, const float* depth, float* means_2d,
                           float* covs_2d, uint32_t* tiles_touched, int32_t* radii, hipStream_t stream,
                           uint32_t* hole_flag = nullptr, uint32_t hole_mark = 1u, uint8_t* flags = nullptr, size_t flags_len = 0);
void launch_copy_with_keys(int P, const CamParams& cp, const float* means_2d, const uint32_t* offsets,
                           const int32_t* radii, const float* depth, uint64_t* keys, uint32_t* values,
                           hipStream_t stream);
// the splatter's sort-before-duplicate (abi_stages.cpp lcgs_tile_splat_forward): which splats claim pair slots, their depth
// keys, a gather through the sorted order, and copy_with_keys over the splats IN THAT ORDER
void launch_gather_depth_keys(int n, const uint32_t* vis, const float* depth, uint32_t* keys, uint32_t* vals, hipStream_t stream);
void launch_gather_u32(int n, const uint32_t* order, const uint32_t* src, uint32_t* dst, hipStream_t stream);
void launch_copy_with_keys_ordered(int n, const CamParams& cp, const float* means_2d, const uint32_t* offsets_sorted,
                                   const int32_t* radii, const float* depth, const uint32_t* order, uint64_t* keys,
                                   uint32_t* values, hipStream_t stream);
// the same pairs as launch_copy_with_keys_ordered (with order = the ascending list of the splats that claim slots: as
// launch_copy_with_keys), written by workgroups that own 1024 consecutive OUTPUT slots (coalesced stores, bounded work per
// workgroup).  The n sources are the splats that CLAIM slots (tiles > 0, radius > 0), source e = splat order[e]; offsets: the
// inclusive sums of their tile counts, L = offsets[n - 1] (known on the host).  win_first: copy_with_keys_windows_bytes(L).
void   launch_copy_with_keys_balanced(int n, const CamParams& cp, const float* means_2d, const uint32_t* offsets,
                                      const int32_t* radii, const float* depth, const uint32_t* order, uint64_t* keys,
                                      uint32_t* values, uint32_t L, uint32_t* win_first, hipStream_t stream,
                                      const uint32_t* dbits_of_source = nullptr); // (nullable) source e's depth bits, in source order
size_t copy_with_keys_windows_bytes(uint32_t L);
void launch_get_ranges_u64(int64_t L, const uint64_t* keys, uint32_t* ranges, hipStream_t stream);

// ---- scan.hip : DeviceScan::InclusiveSum<uint> ----
size_t scan_temp_bytes(int64_t n);
void   launch_inclusive_sum_u32(const uint32_t* in, uint32_t* out, int64_t n, void* temp, hipStream_t stream);
// element count read from device memory (*d_n <= n_cap)
void   launch_inclusive_sum_u32_dyn(const uint32_t* in, uint32_t* out, int64_t n_cap, const uint32_t* d_n, void* temp,
                                    hipStream_t stream);

// ---- render.hip : per-tile front-to-back compositing (gs_tile_splatter/shader.cpp:171-288) ----
// Reference-layout inputs (means_2d[2P] pixel, conic[3P], opacity[P], color[3P]).
void launch_render_forward_aos(const CamParams& cp, const float bg[3], const uint32_t* ranges,
                               const uint32_t* point_list, const float* means_2d, const float* conic,
                               const float* opacity, const float* color, float* img, float* final_T,
                               uint32_t* n_contrib, hipStream_t stream);

// ---- fused_forward.hip : the one-submission frame ----
// Packed per-visible-splat record written by the fused preprocess and gathered by the renderer.
struct __attribute__((aligned(16))) SplatRecord {
    float    mx, my;   // pixel mean                       (float4 #0)
    float    ca, cb;   // conic xx, xy
    float    cc, opac; // conic yy, opacity                (float4 #1)
    float    r, g;     // colour
    float    b;        //                                  (float4 #2)
    float    depth;    // view-space z
    uint32_t rect_xy;  // rect_min.x | rect_min.y << 16    (tile units)
    uint32_t rect_wh;  // width | height << 16             (tile units); tiles touched = w * h
};
static_assert(sizeof(SplatRecord) == 48, "SplatRecord must be 48 bytes");

// ---- the cull pass and the depth sort it feeds
// Where the cull pass leaves the depth sort's first per-chunk digit counts (pair_sort.hip depth_sort_first_pass).
// splats per cull workgroup = slots per slab = keys per chunk of the depth sort's first pass (both files assert it)
constexpr int kCullChunkSplats = 2048;
struct TieOrder; // tie_order.hpp
struct DepthSortFirstPass {
    uint32_t  mask;       // first digit = key & mask
    uint32_t  row_stride; // counts[digit * row_stride + chunk]
    uint32_t* counts;
};
DepthSortFirstPass depth_sort_first_pass(int64_t P, void* sort_ws);
int    cull_chunk_count(int P); // 2048-splat chunks: slab = 2048 x 16 B per chunk, chunk_info / chunk_base one entry each
void launch_set_frame_params(const FrameParams& fp, FrameParams* d_fp, hipStream_t stream);
// d_fp (nullable): when non-NULL the kernels take camera / bg / scale_modifier from device memory (graph replay)
// slab[chunk * 2048 + r] = {depth bits, splat index, pruned rect} of the chunk's r-th survivor (index order);
// chunk_info[chunk] = {survivors, reference tiles_touched}
// bound4 (nullable; ignored when radii are asked for): the scene's {position, extent bound} rows (launch_cull_bound) --
// phase 1 then reads 16 instead of 40 bytes per splat; same survivors, same slabs
void launch_cull_compact(int P, const CamParams& cp, float scale_modifier, const FrameParams* d_fp, const float* pos,
                         const float* scale, const float* rotq, const float* opacity, int32_t* radii, uint4* slab,
                         uint2* chunk_info, const DepthSortFirstPass& first, hipStream_t stream,
                         const float4* bound4 = nullptr);
// out[i] = {pos[i], |R(q_i)|-bound x max |scale_i|}: the camera-independent inputs of the cull pass's phase-1 test
void launch_cull_bound(int64_t P, const float* pos, const float* scale, const float* rotq, float4* out, hipStream_t stream);
// diagnostics: *mismatches += rows of `rows` that are not what launch_cull_bound would write for the arrays now (bit compare)
void launch_cull_bound_verify(int64_t P, const float* pos, const float* scale, const float* rotq, const float4* rows,
                              unsigned long long* mismatches, hipStream_t stream);
// d_counts: [0] V (splats emitting >= 1 pair), [1] reference num_rendered (both written by the depth sort's first
//           row-scan launch), [2] pairs emitted, [3] overflow flag, [4] pairs wanted (before clamping to capacity)
void launch_depth_sort_from_chunks(int64_t P, int64_t v_hint, const uint4* slab, const uint2* chunk_info,
                                   uint32_t* chunk_base, uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a,
                                   uint32_t* vals_b, uint32_t* vis_index, uint2* rects, uint32_t* d_counts, void* sort_ws,
                                   hipStream_t stream, hipEvent_t fork = nullptr,
                                   // re-ordered scenes: equal depths come out in FILE order (tie_order.hpp); the sorted
                                   // values then carry a file-index tag above their id_bits -- readers mask it off
                                   const TieOrder* tie = nullptr,
                                   // first_pass_only: stop behind the compaction (vis_index, rects, V, num_rendered)
                                   bool first_pass_only = false);
void launch_fix_equal_depth_order(uint32_t* keys, uint32_t* vals, uint32_t* scratch_k, uint32_t* scratch_v, int64_t n_cap,
                                  int64_t v_hint, const TieOrder& tie, hipStream_t stream);
// ---- splat ownership: a frame from RECEIVED records instead of the context's own cull pass
// rows_out[i] = vis[i] + row_first for i < *d_count
void launch_rows_global(const uint32_t* vis, const uint32_t* d_count, uint32_t row_first, uint32_t* rows_out, int64_t hint,
                        hipStream_t stream);
// n records (ascending global row order) -> the depth sort's input (keys = depth bits, vals = position [| file-index tag]),
// rects, vis_index = rows; d_counts[0] = n, [1] = the sum of the pruned rects' tiles (non-zero iff anything is drawn)
void launch_unpack_records(int64_t n, const SplatRecord* recs, const uint32_t* rows, const uint32_t* perm, uint32_t id_bits,
                           uint32_t tag_shift, uint32_t* keys, uint32_t* vals, uint2* rects, uint32_t* vis_index,
                           uint32_t* d_counts, uint32_t P, uint32_t grid_x, uint32_t grid_y, hipStream_t stream);
// ... the same from PADDED per-owner segments whose true counts are on the device (fused_forward.hip k_unpack_records_seg)
constexpr int kMaxOwnerSegs = 16; // = LCGS_MAX_OWNER_VIEWS (one view slot per rank of an ownership step)
struct OwnerSegs {
    uint32_t n = 0;                      // owners
    uint32_t off[kMaxOwnerSegs + 1] = {}; // owner o's segment = positions [off[o], off[o + 1])
};
void launch_unpack_records_seg(const OwnerSegs& segs, const uint32_t* table, uint32_t view, const SplatRecord* recs,
                               const uint32_t* rows, const uint32_t* perm, uint32_t id_bits, uint32_t tag_shift, uint32_t* keys,
                               uint32_t* vals, uint2* rects, uint32_t* vis_index, uint32_t* d_counts, uint32_t* overflow, uint32_t P,
                               uint32_t grid_x, uint32_t grid_y, hipStream_t stream);
// *overflow |= 2 when the frame's pair buffers were too small (d_counts[3])
void launch_owner_pair_verdict(const uint32_t* d_counts, uint32_t* overflow, hipStream_t stream);
struct PairSortFirstPass;
size_t expand_ws_bytes(int P_cap);
// v_hint: expected survivor count (bounds the launch; larger live counts are handled by chunk striding)
// returns true iff the first-pass counts were written
bool launch_expand(int P_cap, int64_t v_hint, int64_t l_hint, uint32_t* d_counts, uint32_t grid_x,
                   const uint32_t* order, const uint2* rects, uint2* rects_sorted, uint32_t* pair_keys,
                   uint32_t* pair_vals, uint32_t capacity, uint32_t* ws, hipStream_t stream,
                   const PairSortFirstPass* first_pass = nullptr, // non-NULL: also leave the tile sort's first counts
                   uint32_t id_mask = 0xFFFFFFFFu); // the bits of an `order` entry that are the dense id

// ---- pair_sort.hip : stable radix sort of (u32 key, u32 value) pairs, count in device memory ----
size_t pair_sort_ws_bytes(int64_t n_cap);
// ping-pongs a -> b -> a ...; returns 0 if the result ended in (keys_a, vals_a), 1 if in (keys_b, vals_b)
// first_hist_done: pass 0's chunk counts are already in ws (see pair_sort_first_pass)
int launch_pair_sort_u32(uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a, uint32_t* vals_b, const uint32_t* d_n,
                         int64_t n_cap, int64_t grid_hint, int begin_bit, int end_bit, void* ws, hipStream_t stream,
                         bool first_hist_done = false);
// Lets the kernel that writes the keys also count them: counts[digit * row_stride + chunk] over all 256 digits,
// chunk = position / keys_per_chunk, digit = (key >> shift) & mask -- exactly what the sort's first k_hist would write
// for the same (n_cap, grid_hint, begin_bit, end_bit, ws).
struct PairSortFirstPass {
    int       shift;
    uint32_t  mask;
    int       keys_per_chunk; // 2048 or 4096
    uint32_t* counts;
    uint32_t  row_stride;
    bool      valid;
};
PairSortFirstPass pair_sort_first_pass(int64_t n_cap, int64_t grid_hint, int begin_bit, int end_bit, void* ws);
// stage-level 64-bit sort on the same kernels: n host-known, inputs intact, (keys_tmp, vals_tmp) = scratch of n elements,
// ws = pair_sort_ws_bytes(n)
void launch_pair_sort_u64_preserve(const uint64_t* keys_in, const uint32_t* vals_in, uint64_t* keys_out,
                                   uint32_t* vals_out, uint64_t* keys_tmp, uint32_t* vals_tmp, int64_t n, int begin_bit,
                                   int end_bit, void* ws, hipStream_t stream);
void launch_build_records(int P_cap, int sh_deg, const CamParams& cp, float scale_modifier, const FrameParams* d_fp,
                          const float* pos,
                          const float* scale, const float* rotq, const float* sh, const float* opacity,
                          const uint32_t* vis_index, const uint32_t* d_counts, SplatRecord* recs, hipStream_t stream,
                          const uint16_t* sh_half = nullptr, // non-NULL: f16 coefficient rows (degree 3 only)
                          float4* shjac = nullptr);          // non-NULL: 3 x float4 per survivor for the backward
bool build_records_writes_jacobian(int sh_deg, const float* sh, bool half);
void launch_sh_to_half(int64_t n, const float* src, uint16_t* dst, hipStream_t stream);
// L_hint sizes the launch (larger live counts are strided); l_cap = entries the key buffer holds
void launch_get_ranges_u32(int64_t L_hint, uint32_t l_cap, uint32_t* d_counts, const uint32_t* keys, uint32_t* ranges,
                           const uint32_t* scan_error_flag, hipStream_t stream, hipEvent_t done = nullptr);
void launch_map_to_index(int64_t L_cap, const uint32_t* d_counts, const uint32_t* list_vid, const uint32_t* vis_index,
                         uint32_t* list_idx, hipStream_t stream);

// longest-list-first tile schedule for the renderers (order[G], a scheduling hint only)
// optimiser step (train.hip): five attribute arrays each for gradients, raw parameters, Adam moments, activated values
struct AdamArrays {
    float *pos, *scale, *rotq, *sh, *opacity;
};
struct AdamRates {
    float pos, sh_dc, sh_rest, opacity, scale, rot;
};
// the step's scalars, and the update itself: ONE definition for train.hip's kernels and for the fused
// preprocess-backward + Adam kernel of backward.hip (the two must agree bit for bit)
struct AdamStep {
    float b1, b2, eps, inv_bc1, inv_sqrt_bc2;
};
AdamStep make_adam_step(float beta1, float beta2, float eps, int step);
__device__ __forceinline__ float adam_update(float g, float& m, float& v, float lr, const AdamStep& a)
{
    m = a.b1 * m + (1.0f - a.b1) * g;
    v = a.b2 * v + (1.0f - a.b2) * g * g;
    return (lr * a.inv_bc1) * m / (sqrtf(v) * a.inv_sqrt_bc2 + a.eps);
}
// row_list != NULL: only rows row_list[0 .. *d_row_count) are updated (launch sized for row_hint rows)
// grad_compact (row_list only): gradient row r belongs to splat row_list[r] (lcgs_render_backward_compact's layout)
void launch_adam_step(int64_t P, int sh_floats, const uint32_t* row_list, const uint32_t* d_row_count, int64_t row_hint,
                      const AdamArrays& grad, const AdamArrays& raw, const AdamArrays& m, const AdamArrays& v,
                      const AdamArrays& act, const AdamRates& lr, float beta1, float beta2, float eps, int step,
                      hipStream_t stream, bool grad_compact = false);
// backward.hip: the per-splat half of the backward with the on-screen-only Adam update applied where the gradients are
// formed (degree 3, frames that kept the colour Jacobian): no gradient rows are written at all
void launch_preprocess_backward_adam(int64_t v_hint, const CamParams& cp, float scale_modifier, const float* pos,
                                     const float* scale, const float* rotq, const uint32_t* vis_index, const uint32_t* d_counts,
                                     const float* grads2d, const float4* shjac, const AdamArrays& raw, const AdamArrays& m,
                                     const AdamArrays& v, const AdamArrays& act, const AdamRates& lr, const AdamStep& a,
                                     hipStream_t stream);
// byte offsets, inside one vertex record, of the 59 wanted float columns (pos3 dc3 rest45 opacity scale3 rot4)
struct PlyColumns {
    uint32_t offset[59];
};
// records [first, first + count) of a PLY payload chunk already in device memory -> activated scene arrays
void launch_ply_activate(const unsigned char* raw, int64_t first, int64_t count, uint32_t stride, const PlyColumns& cols,
                         float* pos, float* scale, float* rotq, float* sh, float* opacity, hipStream_t stream);
// spatial (Morton) order of a scene, ingest.hip: position moments (7 doubles per block: sums, sums of squares, count),
// 30-bit Morton keys + identity values, row gather through a permutation
int  pos_moment_blocks();
void launch_pos_moments(int64_t P, const float* pos, double* partial, hipStream_t stream);
void launch_morton_keys(int64_t P, const float* pos, const float lo[3], const float cells_per_unit[3], uint32_t* keys,
                        uint32_t* vals, hipStream_t stream);
void launch_gather_rows(int64_t rows, int row_floats, const uint32_t* perm, const float* src, float* dst, hipStream_t stream);
void launch_tile_order(const uint32_t* ranges, uint32_t G, uint32_t* order, hipStream_t stream, uint32_t grid_x = 0,
                       uint32_t list_shift = 0);
void launch_blend_exp(const float* x, float* out, int64_t n, hipStream_t stream);
void launch_render_forward_rec(const CamParams& cp, const float bg[3], const uint32_t* ranges,
                               const uint32_t* point_list, const SplatRecord* recs, float* img, float* final_T,
                               uint32_t* n_contrib, const uint32_t* d_counts, const FrameParams* d_fp,
                               const uint32_t* tile_order, hipStream_t stream, uint8_t* strip_masks = nullptr,
                               hipEvent_t done = nullptr, uint32_t* work_counter = nullptr, uint32_t persistent_wgs = 0,
                               // frames that keep backward state: the renderer also clears the 2-D gradient rows
                               // (12 floats x d_counts[0]) and the backward's counter block as a side job
                               float* g2d_zero = nullptr, uint32_t* bwd_counters = nullptr,
                               // frames that keep backward state on PER-BLOCK lists (cp.list_shift): every tile's own list
                               // (entries that reach one of its units, in order; 4 x the pair capacity), with strip_masks
                               // at the same positions, and its range -- what the backward then walks (render.hip COMPACT)
                               uint32_t* keep_list = nullptr, uint32_t* keep_ranges = nullptr);
// work_counter + persistent_wgs: a bounded grid of persistent_wgs workgroups that pull tiles from *work_counter (which
// must be zero when the kernel starts) instead of one workgroup per tile -- caps the wave slots the renderer holds
// strip_masks[list position] = the four per-strip reach bits of that entry (written when final_T / n_contrib are
// kept); the backward takes them instead of repeating the tests
bool render_forward_writes_strip_masks();

// ---- comm_pack.hip : the opt-in f16 transport of the gradient all-reduce (host/comm.cpp) ----
// |max| of n floats, atomically max-ed into *d_out_bits (float bits; zero it first)
void launch_absmax(const float* x, size_t n, uint32_t* d_out_bits, hipStream_t stream);
// dst[i] = srcs[0][i] + ... + srcs[n - 1][i] (n <= 16; dst may be one of the sources): the loopback transport's reductions
void launch_sum_sources(const float* const* srcs, int n, size_t count, float* dst, hipStream_t stream);
// five power-of-two scales (and their inverses) from the five all-reduced magnitudes: max * scale <= 16384 / world
void launch_transport_scales(const float* d_absmax5, int world, float* d_scale5, float* d_inv5, hipStream_t stream);
void launch_pack_f16(const float* x, size_t n, const float* d_scale, uint16_t* out, hipStream_t stream);
void launch_unpack_f16(const uint16_t* in, size_t n, const float* d_inv, float* x, hipStream_t stream);
// comm_sparse.hip: the sparse gradient exchange (touched-row flags -> ascending row list -> per-owner messages)
size_t   sparse_flag_bytes(int64_t P);
uint32_t sparse_flag_chunks(int64_t P);
void     launch_mark_rows(const uint32_t* vis_index, const uint32_t* d_counts, uint8_t* flags, int64_t P, int64_t hint_V,
                          hipStream_t stream);
void     launch_compact_flags(const uint8_t* flags, int64_t P, uint32_t* chunk_ws, uint32_t* rows, uint32_t* d_total,
                              hipStream_t stream, const uint32_t* copy_src = nullptr, uint32_t* copy_dst = nullptr,
                              const uint32_t* box_word0 = nullptr, uint32_t* host_box = nullptr, uint32_t serial = 0);
void     launch_owner_bounds(const uint32_t* rows, const uint32_t* d_total, int64_t shard, int world, uint32_t* d_bounds,
                             hipStream_t stream);
int64_t  sparse_message_words(int64_t count, int sh_degree);
void     launch_sparse_pack(float* const grads[5], int sh_degree, const uint32_t* rows, int64_t count, float* msg,
                            hipStream_t stream);
void     launch_sparse_accumulate(float* const grads[5], int sh_degree, const float* msg, int64_t count, int64_t row_first,
                                  int64_t row_count, hipStream_t stream);

// ---- backward.hip ----
size_t grads2d_bytes(int64_t V_cap);
// (bwd_counter, when given, is zeroed too: the persistent render-backward's tile counter)
void   launch_zero_grads2d(const uint32_t* d_counts, float* grads2d, hipStream_t stream, uint32_t* bwd_counter = nullptr);
// The dense gradient arrays' zero-fill as a side job of the render-backward: the five arrays (pos, scale, rotq, sh, opacity)
// and their lengths in floats; n[3] == 0: no fill.  (Plain pointer members: a dynamically indexed kernel argument would live in
// scratch memory.)
struct DenseFill {
    float *b0 = nullptr, *b1 = nullptr, *b2 = nullptr, *b3 = nullptr, *b4 = nullptr;
    uint32_t n[5] = { 0, 0, 0, 0, 0 };
};
void   launch_render_backward(const CamParams& cp, const float bg[3], const uint32_t* ranges, const uint32_t* point_list,
                              const SplatRecord* recs, const float* final_T, const uint32_t* n_contrib,
                              const float* dL_dimg, float* grads2d, const uint32_t* tile_order, hipStream_t stream,
                              const uint8_t* strip_masks = nullptr, const uint32_t* d_counts = nullptr,
                              uint32_t* work_counter = nullptr, uint32_t persistent_wgs = 0,
                              const DenseFill* fill = nullptr);
void   launch_preprocess_backward(int64_t v_hint, int sh_deg, const CamParams& cp, float scale_modifier, const float* pos,
                                  const float* scale, const float* rotq, const float* sh, const uint32_t* vis_index,
                                  const uint32_t* d_counts, const float* grads2d, float* dL_dpos, float* dL_dscale,
                                  float* dL_drotq, float* dL_dsh, float* dL_dopacity, hipStream_t stream,
                                  const float4* shjac = nullptr, // the forward's colour Jacobian rows, if kept
                                  // compact: output row r belongs to the frame's r-th on-screen splat (dense id r,
                                  // splat vis_index[r], ascending) instead of row = splat index: consecutive rows, every
                                  // one written -- no zero-fill, no 39 %-dense store pattern
                                  bool compact = false,
                                  // slice_bounds != NULL: only the survivors [slice_bounds[slice], slice_bounds[slice + 1])
                                  // (a splat-range slice, see launch_slice_bounds; `slices` sizes the launch)
                                  const uint32_t* slice_bounds = nullptr, int slice = 0, int slices = 1,
                                  // accumulate: ADD the rows to what the arrays hold (dense rows of a further view)
                                  bool accumulate = false);
// bounds[0 .. slices]: dense-id boundaries of the splat-index ranges [k P / slices, (k + 1) P / slices), k < slices <= 63
void   launch_slice_bounds(const uint32_t* vis_index, const uint32_t* d_counts, int64_t P, int slices, uint32_t* bounds,
                           hipStream_t stream);

} // namespace lcgs
