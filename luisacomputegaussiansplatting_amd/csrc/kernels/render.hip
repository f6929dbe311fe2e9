// render.hip -- per-tile front-to-back alpha compositing
// (m_forward_render_shader, lcgs/src/gs_tile_splatter/shader.cpp:171-288).
//
// CDNA4 shape (not the reference's 256-thread block with three block barriers per round):
//   * one wave64 owns one 16x16 tile; lane l covers column x = l & 15 and the four rows
//     y = (l >> 4) + {0,4,8,12}.  The four pixels of a lane share dx, cx*dx*dx and cy*dx, so the
//     per-entry cost is ~8 VALU ops/pixel instead of ~12, and there is no cross-wave barrier at all.
//   * a round stages 64 list entries: lane l gathers entry l's 36-byte record (mean, conic, opacity,
//     rgb) with wide loads and parks it in a 3 KiB LDS slab; the inner loop then reads each entry
//     back with wave-uniform (broadcast, conflict-free) ds_read_b128s.  The colour therefore comes
//     from LDS too -- the reference re-fetches it from global memory per pixel per contributing
//     splat (shader.cpp:268-269).
//   * wave-level early out: a 64-bit ballot of "all four pixels done" ends the tile as soon as every
//     pixel is saturated (the reference's "collect num_done" at shader.cpp:229 has no code behind it,
//     so every tile walks its whole list).
//   * workgroup -> tile mapping is XCD-aware: workgroups are dealt round-robin to the 8 XCDs, so
//     workgroup b takes tile (b % 8) * ceil(G/8) + b / 8 -- each XCD's private L2 sees one
//     contiguous band of tile rows and neighbouring tiles (which share most of their splat records)
//     hit the same L2.
// Numerics: the per-pixel expressions keep the reference's evaluation order with no FMA contraction
// (-ffp-contract=off); exp() is the hardware v_exp_f32 path (__expf).
#include "launch.hpp"

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
};

constexpr int kXcd = 8;

template <typename Fetch>
__global__ void __launch_bounds__(64) k_render_forward(CamParams cp, float bg0, float bg1, float bg2,
                                                         const uint32_t* __restrict__ ranges,
                                                         const uint32_t* __restrict__ point_list, Fetch fetch,
                                                         float* __restrict__ img, float* __restrict__ final_T,
                                                         uint32_t* __restrict__ n_contrib,
                                                         const uint32_t* __restrict__ d_counts)
{
    __shared__ float4 s_a[64];
    __shared__ float4 s_b[64];
    __shared__ float  s_c[64];

    const uint32_t G    = cp.grid_x * cp.grid_y;
    const uint32_t per  = (G + kXcd - 1) / kXcd;
    const uint32_t tile = (blockIdx.x % kXcd) * per + blockIdx.x / kXcd;
    if (tile >= G) return;
    // num_rendered == 0: nothing is drawn and the image is left untouched (gs_tile_splatter/impl.cpp:109)
    if (d_counts && d_counts[1] == 0u) return;

    const uint32_t lane = threadIdx.x;
    const uint32_t tx = tile % cp.grid_x, ty = tile / cp.grid_x;
    const uint32_t px  = tx * kBlockX + (lane & 15u);
    const uint32_t py0 = ty * kBlockY + (lane >> 4);
    const float    pxf = (float)px;

    float    pyf[4];
    bool     inside[4], done[4];
    float    T[4], Cr[4], Cg[4], Cb[4];
    uint32_t last_contrib[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t py = py0 + 4u * k;
        pyf[k]            = (float)py;
        inside[k]         = (px < cp.width) && (py < cp.height);
        done[k]           = !inside[k];
        T[k]              = 1.0f;
        Cr[k] = Cg[k] = Cb[k] = 0.0f;
        last_contrib[k]       = 0u;
    }

    const uint32_t range_start = ranges[2 * (size_t)tile + 0];
    const uint32_t range_end   = ranges[2 * (size_t)tile + 1];

    for (uint32_t base = range_start; base < range_end; base += 64u) {
        if (__all(done[0] && done[1] && done[2] && done[3])) break;
        const uint32_t e = base + lane;
        if (e < range_end) {
            const uint32_t id = point_list[e];
            float4         a, b;
            float          c;
            fetch(id, a, b, c);
            s_a[lane] = a;
            s_b[lane] = b;
            s_c[lane] = c;
        }
        __syncthreads();
        const uint32_t cnt = (range_end - base) < 64u ? (range_end - base) : 64u;
        for (uint32_t j = 0; j < cnt; ++j) {
            if (__all(done[0] && done[1] && done[2] && done[3])) break;
            const float4   a           = s_a[j]; // mean.x, mean.y, conic.x, conic.y
            const float4   b           = s_b[j]; // conic.z, opacity, r, g
            const float    cb          = s_c[j]; // b
            const uint32_t contributor = base - range_start + j + 1u;
            const float    dx          = a.x - pxf;
            const float    cxdxdx      = a.z * dx * dx; // con_o.x * d.x * d.x
            const float    cydx        = a.w * dx;      // con_o.y * d.x
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (done[k]) continue;
                const float dy    = a.y - pyf[k];
                const float power = -0.5f * (cxdxdx + b.x * dy * dy) - cydx * dy; // shader.cpp:256
                if (power > 0.0f) continue;
                const float alpha = fmin_(0.99f, b.y * __expf(power));
                if (alpha < 1.0f / 255.0f) continue;
                const float test_T = T[k] * (1.0f - alpha);
                if (test_T < 0.0001f) {
                    done[k] = true;
                    continue;
                }
                const float w = T[k] * alpha;
                Cr[k]         = Cr[k] + w * b.z;
                Cg[k]         = Cg[k] + w * b.w;
                Cb[k]         = Cb[k] + w * cb;
                T[k]          = test_T;
                last_contrib[k] = contributor;
            }
        }
        __syncthreads();
    }

    const size_t hw = (size_t)cp.width * cp.height;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (!inside[k]) continue;
        const size_t pix = (size_t)px + (size_t)cp.width * (py0 + 4u * k);
        img[pix]          = bg0 * T[k] + Cr[k]; // shader.cpp:279-286, planar CHW
        img[pix + hw]     = bg1 * T[k] + Cg[k];
        img[pix + 2 * hw] = bg2 * T[k] + Cb[k];
        if (final_T) final_T[pix] = T[k];
        if (n_contrib) n_contrib[pix] = last_contrib[k];
    }
}

template <typename Fetch>
void launch_render(const CamParams& cp, const float bg[3], const uint32_t* ranges, const uint32_t* point_list,
                   Fetch fetch, float* img, float* final_T, uint32_t* n_contrib, const uint32_t* d_counts,
                   hipStream_t stream)
{
    const uint32_t G   = cp.grid_x * cp.grid_y;
    const uint32_t per = (G + kXcd - 1) / kXcd;
    if (G == 0) return;
    hipLaunchKernelGGL(k_render_forward<Fetch>, dim3(per * kXcd), dim3(64), 0, stream, cp, bg[0], bg[1], bg[2], ranges,
                       point_list, fetch, img, final_T, n_contrib, d_counts);
}

} // namespace

void launch_render_forward_aos(const CamParams& cp, const float bg[3], const uint32_t* ranges,
                               const uint32_t* point_list, const float* means_2d, const float* conic,
                               const float* opacity, const float* color, float* img, float* final_T,
                               uint32_t* n_contrib, hipStream_t stream)
{
    launch_render(cp, bg, ranges, point_list, FetchAoS{ means_2d, conic, opacity, color }, img, final_T, n_contrib,
                  nullptr, stream);
}

void launch_render_forward_rec(const CamParams& cp, const float bg[3], const uint32_t* ranges,
                               const uint32_t* point_list, const SplatRecord* recs, float* img, float* final_T,
                               uint32_t* n_contrib, const uint32_t* d_counts, hipStream_t stream)
{
    launch_render(cp, bg, ranges, point_list, FetchRec{ recs }, img, final_T, n_contrib, d_counts, stream);
}

} // namespace lcgs
