// train.hip -- the optimiser step behind the backward (the reference only names "training" on its
// roadmap, doc/roadmap.md:4).  One fused pass per attribute turns the gradients w.r.t. the ACTIVATED parameters
// (what lcgs_render_backward produces, possibly all-reduced over views) into an Adam update of the RAW parameters
// and refreshes the activated arrays the renderer reads:
//   pos, sh      raw == activated                         g_raw = g
//   scale        s = exp(raw)                             g_raw = g * s
//   opacity      o = sigmoid(raw)                         g_raw = g * o * (1 - o)
//   rotq         q = raw / |raw|                          g_raw = (g - q (q . g)) / |raw|
// Adam as torch.optim.Adam (no weight decay, no amsgrad): m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g g;
// raw -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps).
// `act` may alias `raw` for pos / sh (identity activation); when the renderer's arrays are separate buffers they are
// rewritten too.
// HBM-bound elementwise work: every array is touched once, 16-byte accesses where the layout allows.  With a row
// list (the dense ids -> splat index map of the last forward) only the splats that reached the screen are updated
// ("sparse Adam": their moments are the only ones that change; 2.5x fewer bytes on the bicycle stand-in).
#include "launch.hpp"
#include "stream_access.hpp"

namespace lcgs
{
namespace
{

// Gradients and moments are touched once per step and are far larger than the caches: streaming (non-temporal) accesses
// (dense step 1.95 -> 1.87 ms, 5.18 -> 5.42 TB/s in same-box A/B, profiles/r04_nt_accesses_ab.txt).
// rows of ROW floats; columns below `split` use lr0, the others lr1 (SH: dc vs rest).  MODE 0 plain, 1 exp, 2 sigmoid.
template <int ROW, int MODE>
__global__ void __launch_bounds__(256) k_adam_rows(int64_t rows, const uint32_t* __restrict__ row_list,
                                                   const uint32_t* __restrict__ d_row_count,
                                                   const float* __restrict__ grad, float* __restrict__ raw,
                                                   float* __restrict__ m, float* __restrict__ v, float* act /* may alias raw */,
                                                   int split, float lr0, float lr1, AdamStep a, int grad_compact)
{
    const int64_t n_rows = d_row_count ? (int64_t)*d_row_count : rows;
    const int64_t total  = n_rows * ROW;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / ROW;
        const int     c = (int)(e - r * ROW);
        const int64_t i = (row_list ? (int64_t)row_list[r] : r) * ROW + c;
        float         g = ld_stream(grad + (grad_compact ? e : i)); // compact gradients: row r of the list, not row of the splat
        if (MODE == 1) g *= act[i];
        if (MODE == 2) {
            const float o = act[i];
            g             = g * o * (1.0f - o);
        }
        float       mm = ld_stream(m + i), vv = ld_stream(v + i);
        const float x  = raw[i] - adam_update(g, mm, vv, c < split ? lr0 : lr1, a);
        st_stream(m + i, mm);
        st_stream(v + i, vv);
        raw[i] = x;
        if (MODE == 0 && act != raw) act[i] = x; // (wave-uniform: the renderer's array is a separate buffer)
        if (MODE == 1) act[i] = expf(x);
        if (MODE == 2) act[i] = 1.0f / (1.0f + expf(-x));
    }
}

// degree-3 SH rows (48 floats, 16-byte aligned): twelve 16-byte accesses per row and array instead of 48 4-byte ones
__global__ void __launch_bounds__(256) k_adam_sh48(int64_t rows, const uint32_t* __restrict__ row_list,
                                                   const uint32_t* __restrict__ d_row_count,
                                                   const float4* __restrict__ grad, float4* __restrict__ raw,
                                                   float4* __restrict__ m, float4* __restrict__ v,
                                                   float4* act /* may alias raw */, float lr_dc, float lr_rest, AdamStep a,
                                                   int grad_compact)
{
    const int64_t n_rows = d_row_count ? (int64_t)*d_row_count : rows;
    const int64_t total  = n_rows * 12;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / 12;
        const int     c = (int)(e - r * 12);
        const int64_t i = (row_list ? (int64_t)row_list[r] : r) * 12 + c;
        const float4  g = ld_stream(grad + (grad_compact ? e : i));
        float4        x = raw[i], mm = ld_stream(m + i), vv = ld_stream(v + i);
        const float   l = c == 0 ? lr_dc : lr_rest; // floats 0..2 of a row are the dc band
        x.x -= adam_update(g.x, mm.x, vv.x, l, a);
        x.y -= adam_update(g.y, mm.y, vv.y, l, a);
        x.z -= adam_update(g.z, mm.z, vv.z, l, a);
        x.w -= adam_update(g.w, mm.w, vv.w, lr_rest, a);
        raw[i] = x;
        st_stream(m + i, mm);
        st_stream(v + i, vv);
        if (act != raw) act[i] = x;
    }
}

// quaternions: one lane per splat, 16-byte rows
__global__ void __launch_bounds__(256) k_adam_rot(int64_t rows, const uint32_t* __restrict__ row_list,
                                                  const uint32_t* __restrict__ d_row_count,
                                                  const float4* __restrict__ grad, float4* __restrict__ raw,
                                                  float4* __restrict__ m, float4* __restrict__ v, float4* __restrict__ act,
                                                  float lr, AdamStep a, int grad_compact)
{
    const int64_t n_rows = d_row_count ? (int64_t)*d_row_count : rows;
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * 256) {
        const int64_t i  = row_list ? (int64_t)row_list[r] : r;
        const float4  g  = ld_stream(grad + (grad_compact ? r : i)), q = act[i];
        float4        x  = raw[i], mm = ld_stream(m + i), vv = ld_stream(v + i);
        const float   inv_norm = 1.0f / sqrtf(x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w);
        const float   qg       = q.x * g.x + q.y * g.y + q.z * g.z + q.w * g.w;
        x.x -= adam_update((g.x - q.x * qg) * inv_norm, mm.x, vv.x, lr, a);
        x.y -= adam_update((g.y - q.y * qg) * inv_norm, mm.y, vv.y, lr, a);
        x.z -= adam_update((g.z - q.z * qg) * inv_norm, mm.z, vv.z, lr, a);
        x.w -= adam_update((g.w - q.w * qg) * inv_norm, mm.w, vv.w, lr, a);
        const float n2 = 1.0f / sqrtf(x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w);
        raw[i] = x;
        st_stream(m + i, mm);
        st_stream(v + i, vv);
        act[i] = make_float4(x.x * n2, x.y * n2, x.z * n2, x.w * n2);
    }
}

unsigned grid_for(int64_t elements)
{
    int64_t b = (elements + 255) / 256;
    if (b < 1) b = 1;
    if (b > 65536) b = 65536; // grid-stride beyond
    return (unsigned)b;
}

} // namespace

AdamStep make_adam_step(float beta1, float beta2, float eps, int step)
{
    AdamStep a;
    a.b1           = beta1;
    a.b2           = beta2;
    a.eps          = eps;
    a.inv_bc1      = (float)(1.0 / (1.0 - pow((double)beta1, (double)step)));
    a.inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)beta2, (double)step)));
    return a;
}

void launch_adam_step(int64_t P, int sh_floats, const uint32_t* row_list, const uint32_t* d_row_count, int64_t row_hint,
                      const AdamArrays& grad, const AdamArrays& raw, const AdamArrays& m, const AdamArrays& v,
                      const AdamArrays& act, const AdamRates& lr, float beta1, float beta2, float eps, int step,
                      hipStream_t stream, bool grad_compact)
{
    const int64_t rows = row_list ? row_hint : P;
    const int     gc   = (grad_compact && row_list) ? 1 : 0;
    if (rows <= 0 && !d_row_count) return;
    const AdamStep a = make_adam_step(beta1, beta2, eps, step);
    const int64_t launch_rows = std::max<int64_t>(rows, 1);
    hipLaunchKernelGGL((k_adam_rows<3, 0>), dim3(grid_for(launch_rows * 3)), dim3(256), 0, stream, rows, row_list, d_row_count,
                       grad.pos, raw.pos, m.pos, v.pos, act.pos, 3, lr.pos, lr.pos, a, gc);
    hipLaunchKernelGGL((k_adam_rows<3, 1>), dim3(grid_for(launch_rows * 3)), dim3(256), 0, stream, rows, row_list, d_row_count,
                       grad.scale, raw.scale, m.scale, v.scale, act.scale, 3, lr.scale, lr.scale, a, gc);
    hipLaunchKernelGGL(k_adam_rot, dim3(grid_for(launch_rows)), dim3(256), 0, stream, rows, row_list, d_row_count,
                       reinterpret_cast<const float4*>(grad.rotq), reinterpret_cast<float4*>(raw.rotq),
                       reinterpret_cast<float4*>(m.rotq), reinterpret_cast<float4*>(v.rotq),
                       reinterpret_cast<float4*>(act.rotq), lr.rot, a, gc);
    const bool sh_aligned = ((reinterpret_cast<uintptr_t>(grad.sh) | reinterpret_cast<uintptr_t>(raw.sh) |
                              reinterpret_cast<uintptr_t>(m.sh) | reinterpret_cast<uintptr_t>(v.sh) |
                              reinterpret_cast<uintptr_t>(act.sh)) & 15) == 0;
    if (sh_floats == 48 && sh_aligned)
        hipLaunchKernelGGL(k_adam_sh48, dim3(grid_for(launch_rows * 12)), dim3(256), 0, stream, rows, row_list, d_row_count,
                           reinterpret_cast<const float4*>(grad.sh), reinterpret_cast<float4*>(raw.sh),
                           reinterpret_cast<float4*>(m.sh), reinterpret_cast<float4*>(v.sh),
                           reinterpret_cast<float4*>(act.sh), lr.sh_dc, lr.sh_rest, a, gc);
    else if (sh_floats == 48)
        hipLaunchKernelGGL((k_adam_rows<48, 0>), dim3(grid_for(launch_rows * 48)), dim3(256), 0, stream, rows, row_list,
                           d_row_count, grad.sh, raw.sh, m.sh, v.sh, act.sh, 3, lr.sh_dc, lr.sh_rest, a, gc);
    else if (sh_floats == 27)
        hipLaunchKernelGGL((k_adam_rows<27, 0>), dim3(grid_for(launch_rows * 27)), dim3(256), 0, stream, rows, row_list,
                           d_row_count, grad.sh, raw.sh, m.sh, v.sh, act.sh, 3, lr.sh_dc, lr.sh_rest, a, gc);
    else if (sh_floats == 12)
        hipLaunchKernelGGL((k_adam_rows<12, 0>), dim3(grid_for(launch_rows * 12)), dim3(256), 0, stream, rows, row_list,
                           d_row_count, grad.sh, raw.sh, m.sh, v.sh, act.sh, 3, lr.sh_dc, lr.sh_rest, a, gc);
    else
        hipLaunchKernelGGL((k_adam_rows<3, 0>), dim3(grid_for(launch_rows * 3)), dim3(256), 0, stream, rows, row_list,
                           d_row_count, grad.sh, raw.sh, m.sh, v.sh, act.sh, 3, lr.sh_dc, lr.sh_rest, a, gc);
    hipLaunchKernelGGL((k_adam_rows<1, 2>), dim3(grid_for(launch_rows)), dim3(256), 0, stream, rows, row_list, d_row_count,
                       grad.opacity, raw.opacity, m.opacity, v.opacity, act.opacity, 1, lr.opacity, lr.opacity, a, gc);
}

} // namespace lcgs
