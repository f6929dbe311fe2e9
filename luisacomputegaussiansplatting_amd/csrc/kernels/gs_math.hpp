// gs_math.hpp -- per-splat math of the lcgs hot path, written once for every kernel that needs it.
//
// Each function states the reference lines it implements (paths relative to the reference repo).
// The arithmetic is spelled out term by term in the reference's evaluation order and every
// translation unit that includes this header is compiled with -ffp-contract=off, so the per-splat
// outputs (colour, NDC/pixel mean, depth, cov/conic, radius, rect, tile count) are reproducible
// bit for bit -- which is what makes the integer half of the pipeline (keys, sorted lists, ranges)
// exactly comparable with a CPU restatement.
//
// LuisaCompute conventions assumed (LC source is not part of the reference tree):
//   column-major matrices, M*v summed left to right over columns, zero-initialised locals,
//   clamp(v,lo,hi) = min(max(v,lo),hi), saturating float->int conversion.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define LCGS_HD __host__ __device__ __forceinline__

namespace lcgs
{

constexpr uint32_t kBlockX = 16u; // lcgs/include/lcgs/module.h:17  (m_blocks = {16,16})
constexpr uint32_t kBlockY = 16u;

// lcgs/include/lcgs/util/sh.hpp:12-28
constexpr float SH_C0    = 0.28209479177387814f;
constexpr float SH_C1    = 0.4886025119029199f;
constexpr float SH_C2_0  = 1.0925484305920792f;
constexpr float SH_C2_1  = -1.0925484305920792f;
constexpr float SH_C2_2  = 0.31539156525252005f;
constexpr float SH_C2_3  = -1.0925484305920792f;
constexpr float SH_C2_4  = 0.5462742152960396f;
constexpr float SH_C3_0  = -0.5900435899266435f;
constexpr float SH_C3_1  = 2.890611442640554f;
constexpr float SH_C3_2  = -0.4570457994644658f;
constexpr float SH_C3_3  = 0.3731763325901154f;
constexpr float SH_C3_4  = -0.4570457994644658f;
constexpr float SH_C3_5  = 1.445305721320277f;
constexpr float SH_C3_6  = -0.5900435899266435f;

// Per-frame camera constants, computed once on the host exactly as
// lcgs/src/gs_projector/impl.cpp:34-42 does and passed to kernels by value.
struct CamParams {
    float campos[3];
    float right[3], up[3], front[3]; // rows of the view rotation (camera.h:38-51)
    float tx, ty, tz;                // view translation: -dot(position, axis)
    float inv_tanx, inv_tany;        // projection_matrix fx, fy (camera.h:58-59)
    float tanfovx, tanfovy;
    float focalx, focaly;
    uint32_t width, height;
    uint32_t grid_x, grid_y;
    // opt-in footprint cull of the fused frame (lcgs_set_lod; 0 = off = the reference's behaviour): a splat whose reference
    // radius (pixels, gs_tile_splatter/shader.cpp:148) is below this is treated as if it touched no tile
    int32_t lod_min_radius;
    // The granularity of the fused frame's pair lists: lists are kept per block of (1 << list_shift)^2 tiles (0: per tile, like the
    // reference; 1: per 32 x 32 pixels).  A splat's footprint meets fewer coarse blocks than tiles (7.53 M -> 4.47 M pairs on the
    // bench frame), and duplication, the tile partition and the range pass scale with the pairs; the renderer's workgroups still
    // own one 16 x 16 tile each and walk their block's list -- the reach masks of their staging drop what belongs to the block's
    // other tiles.  Frames that keep backward state use 0 (the backward walks per-tile lists).
    uint32_t list_shift;
};
LCGS_HD uint32_t list_grid_x(const CamParams& cp) { return (cp.grid_x + (1u << cp.list_shift) - 1u) >> cp.list_shift; }
LCGS_HD uint32_t list_grid_y(const CamParams& cp) { return (cp.grid_y + (1u << cp.list_shift) - 1u) >> cp.list_shift; }
LCGS_HD uint32_t list_block_of_tile(const CamParams& cp, uint32_t tx, uint32_t ty)
{
    return (ty >> cp.list_shift) * list_grid_x(cp) + (tx >> cp.list_shift);
}

// Per-call parameters of the fused frame, kept in DEVICE memory so that a captured hipGraph of the frame stays
// valid when the camera moves: one tiny eager kernel refreshes this block, then the whole frame replays.
struct FrameParams {
    CamParams cp;
    float     bg[3];
    float     scale_modifier;
};

LCGS_HD float fmin_(float a, float b) { return a < b ? a : b; }
LCGS_HD float fmax_(float a, float b) { return a > b ? a : b; }
LCGS_HD float clamp_(float v, float lo, float hi) { return fmin_(fmax_(v, lo), hi); }

// float -> u32 / i32 with the saturating semantics of cvt.rzi / v_cvt_*32_f32 (NaN -> 0)
LCGS_HD uint32_t f2u_sat(float x)
{
    if (!(x > 0.0f)) return 0u;
    if (x >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)x;
}
LCGS_HD int32_t f2i_sat(float x)
{
    if (x != x) return 0;
    if (x >= 2147483648.0f) return 2147483647;
    if (x <= -2147483648.0f) return (int32_t)(-2147483647 - 1);
    return (int32_t)x;
}

// ---------------------------------------------------------------------------------------------
// blend_exp -- THE exp() of the compositing loop (`exp(power)`, gs_tile_splatter/shader.cpp:258).
//
// The reference's exp is whatever LuisaCompute's JIT hands its backend (unpinned: CUDA expf is documented to 2 ulp,
// its fast-math __expf to 2 + floor(|1.16 x|) ulp), and the three hard thresholds behind it (alpha < 1/255,
// T < 1e-4, and through T every later entry of the pixel) turn a 1-ulp difference into a visibly different pixel.
// So this library DEFINES the function, as a fixed sequence of IEEE-754 binary32 operations (fma, add, mul, one
// integer add) that a CPU and a GPU evaluate to the same bits: the tests' CPU restatement of the reference runs the
// same sequence in C, and the images are then equal bit for bit, not within a tolerance.
//   t = fma(x, log2e, 1.5 * 2^23)        the low mantissa bits of t hold n = rint(x log2e)
//   f = fma(x, log2e, -(t - 1.5 * 2^23))  x log2e - n with ONE rounding: |f| <= 1/2, no argument-scaling error
//   2^f = E(f^2) + f O(f^2)               degree-6 minimax of 2^f on [-1/2, 1/2] with P(0) = 1 (relative error 2.6e-9),
//                                         even and odd halves side by side (the device runs them as v_pk_fma_f32)
//   result = 2^f with n added to the exponent field
// Measured over EVERY binary32 in [-6, 0] (the blend's range: alpha >= 1/255 needs power >= -ln 255 = -5.54):
// <= 2.73 ulp; over [-87, 0]: <= 20.8 ulp (the constant log2e itself is only good to 2^-26 x); exp(0) = 1 exactly.
// Domain: kBlendExpMin = -86 <= x <= 0 (no overflow / denormal handling: the kernels gate on a power floor that is
// clamped to it; the C version returns 0 below it -- there alpha < 1/255 for any opacity below 1e34 -- and
// passes NaN through).
// ---------------------------------------------------------------------------------------------
constexpr float kExpLog2e = 0x1.715476p+0f;  // binary32 nearest to log2(e)
constexpr float kExpMagic = 12582912.0f;     // 1.5 * 2^23
constexpr float kBlendExpMin = -86.0f;       // below this blend_exp is not evaluated (defined as 0)
constexpr float kExpC1 = 0x1.62e432p-1f, kExpC2 = 0x1.ebfbe2p-3f, kExpC3 = 0x1.c6ae72p-5f, kExpC4 = 0x1.3b270ep-7f,
                kExpC5 = 0x1.5f7276p-10f, kExpC6 = 0x1.470b4ap-13f;

LCGS_HD float blend_exp(float x)
{
    const float t  = __builtin_fmaf(x, kExpLog2e, kExpMagic);
    const float n  = t - kExpMagic;
    const float f  = __builtin_fmaf(x, kExpLog2e, -n);
    const float f2 = f * f;
#if defined(__HIP_DEVICE_COMPILE__)
    typedef float v2f_ __attribute__((ext_vector_type(2)));
    const v2f_ ff = { f2, f2 };
    v2f_       eo = { __builtin_fmaf(kExpC6, f2, kExpC4), kExpC5 };
    eo            = __builtin_elementwise_fma(eo, ff, (v2f_){ kExpC2, kExpC3 });
    eo            = __builtin_elementwise_fma(eo, ff, (v2f_){ 1.0f, kExpC1 });
    const float p = __builtin_fmaf(eo.y, f, eo.x);
#else
    float E = __builtin_fmaf(kExpC6, f2, kExpC4);
    E       = __builtin_fmaf(E, f2, kExpC2);
    E       = __builtin_fmaf(E, f2, 1.0f);
    float O = __builtin_fmaf(kExpC5, f2, kExpC3);
    O       = __builtin_fmaf(O, f2, kExpC1);
    const float p = __builtin_fmaf(O, f, E);
#endif
    return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, p) + (__builtin_bit_cast(uint32_t, t) << 23));
}

// The 16 SH basis terms and their direction gradients, signs as composed by sh_preprocessor.cpp:49-147.
// X(k, basis, d/dx, d/dy, d/dz) with x, y, z, xx, yy, zz in scope.
#define LCGS_SH_TERMS(X)                                                                                              \
    X(0, SH_C0, 0.0f, 0.0f, 0.0f)                                                                                     \
    X(1, -SH_C1 * y, 0.0f, -SH_C1, 0.0f)                                                                              \
    X(2, SH_C1 * z, 0.0f, 0.0f, SH_C1)                                                                                \
    X(3, -SH_C1 * x, -SH_C1, 0.0f, 0.0f)                                                                              \
    X(4, SH_C2_0 * x * y, SH_C2_0 * y, SH_C2_0 * x, 0.0f)                                                             \
    X(5, SH_C2_1 * y * z, 0.0f, SH_C2_1 * z, SH_C2_1 * y)                                                             \
    X(6, SH_C2_2 * (2.0f * zz - xx - yy), SH_C2_2 * (-2.0f * x), SH_C2_2 * (-2.0f * y), SH_C2_2 * (4.0f * z))         \
    X(7, SH_C2_3 * z * x, SH_C2_3 * z, 0.0f, SH_C2_3 * x)                                                             \
    X(8, SH_C2_4 * (xx - yy), SH_C2_4 * 2.0f * x, SH_C2_4 * -2.0f * y, 0.0f)                                          \
    X(9, SH_C3_0 * y * (3.0f * xx - yy), SH_C3_0 * 6.0f * x * y, SH_C3_0 * (3.0f * xx - 3.0f * yy), 0.0f)             \
    X(10, SH_C3_1 * x * y * z, SH_C3_1 * y * z, SH_C3_1 * x * z, SH_C3_1 * x * y)                                     \
    X(11, SH_C3_2 * y * (4.0f * zz - xx - yy), SH_C3_2 * (-2.0f * x * y), SH_C3_2 * (4.0f * zz - xx - 3.0f * yy),     \
      SH_C3_2 * 8.0f * y * z)                                                                                         \
    X(12, SH_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy), SH_C3_3 * (-6.0f * x * z), SH_C3_3 * (-6.0f * y * z),    \
      SH_C3_3 * (6.0f * zz - 3.0f * xx - 3.0f * yy))                                                                  \
    X(13, SH_C3_4 * x * (4.0f * zz - xx - yy), SH_C3_4 * (4.0f * zz - 3.0f * xx - yy), SH_C3_4 * (-2.0f * x * y),     \
      SH_C3_4 * 8.0f * x * z)                                                                                         \
    X(14, SH_C3_5 * z * (xx - yy), SH_C3_5 * 2.0f * x * z, SH_C3_5 * -2.0f * y * z, SH_C3_5 * (xx - yy))              \
    X(15, SH_C3_6 * x * (xx - 3.0f * yy), SH_C3_6 * (3.0f * xx - 3.0f * yy), SH_C3_6 * (-6.0f * x * y), 0.0f)


// ---------------------------------------------------------------------------------------------
// SH colour: lcgs/src/sh_preprocessor.cpp:27-157 with util/sh.hpp:31-34,43-50,68-84,120-138.
// `sh` points at this splat's (deg+1)^2 x 3 coefficients; sh_at(k, c) abstracts the fetch so the
// caller can feed registers, LDS or global memory.
// ---------------------------------------------------------------------------------------------
template <typename ShAt>
LCGS_HD void sh_to_color(int deg, const float campos[3], float px, float py, float pz, ShAt sh_at,
                         float out_raw[3])
{
    float result[3] = { sh_at(0, 0), sh_at(0, 1), sh_at(0, 2) };
    if (deg > -1) {
#pragma unroll
        for (int c = 0; c < 3; ++c) result[c] = sh_at(0, c) * SH_C0;
        if (deg > 0) {
            float dx = px - campos[0], dy = py - campos[1], dz = pz - campos[2];
            float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
            float x = dx * inv, y = dy * inv, z = dz * inv;
#pragma unroll
            for (int c = 0; c < 3; ++c)
                result[c] = result[c] + (-SH_C1) * (sh_at(1, c) * y - sh_at(2, c) * z + sh_at(3, c) * x);
            if (deg > 1) {
                float xx = x * x, yy = y * y, yz = y * z, zz = z * z, zx = z * x, xy = x * y;
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    result[c] = result[c] +
                                (SH_C2_0 * xy * sh_at(4, c) + SH_C2_1 * yz * sh_at(5, c) +
                                 SH_C2_2 * (2.0f * zz - xx - yy) * sh_at(6, c) + SH_C2_3 * zx * sh_at(7, c) +
                                 SH_C2_4 * (xx - yy) * sh_at(8, c));
                if (deg > 2) {
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        result[c] = result[c] +
                                    (SH_C3_0 * y * (3.0f * xx - yy) * sh_at(9, c) + SH_C3_1 * xy * z * sh_at(10, c) +
                                     SH_C3_2 * y * (4.0f * zz - xx - yy) * sh_at(11, c) +
                                     SH_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh_at(12, c) +
                                     SH_C3_4 * x * (4.0f * zz - xx - yy) * sh_at(13, c) +
                                     SH_C3_5 * z * (xx - yy) * sh_at(14, c) +
                                     SH_C3_6 * x * (xx - 3.0f * yy) * sh_at(15, c));
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) result[c] = result[c] + 0.5f;
    }
    out_raw[0] = result[0];
    out_raw[1] = result[1];
    out_raw[2] = result[2];
}

// ---------------------------------------------------------------------------------------------
// View / projection: lcgs/src/gs_projector/shader.cpp:107-121 with the matrices of
// util/camera.h:38-72.  view = [right up front]^T | t ; proj = diag(fx, fy, a) with w = z.
// Terms multiplied by the matrices' structural zeros are dropped (they add +-0).
// ---------------------------------------------------------------------------------------------
LCGS_HD void view_transform(const CamParams& cp, float px, float py, float pz, float v[3])
{
    v[0] = cp.right[0] * px + cp.right[1] * py + cp.right[2] * pz + cp.tx;
    v[1] = cp.up[0] * px + cp.up[1] * py + cp.up[2] * pz + cp.ty;
    v[2] = cp.front[0] * px + cp.front[1] * py + cp.front[2] * pz + cp.tz;
}

LCGS_HD void ndc_from_view(const CamParams& cp, const float v[3], float ndc[2])
{
    float p_w = 1.0f / (v[2] + 1e-6f); // p_proj_hom.w = v.z (camera.h:69, zsign = 1)
    ndc[0]    = (cp.inv_tanx * v[0]) * p_w;
    ndc[1]    = (cp.inv_tany * v[1]) * p_w;
}

// R_from_qvec (util/transform.hpp:188-212), q = (x,y,z,w); calc_cov (util/gaussian.hpp:15-28):
// M = R * diag(s), Sigma = M * M^T.  Column-major: M[c][r] = R[c][r] * s[c].
LCGS_HD void cov3d_from_scale_rot(const float s[3], float qx, float qy, float qz, float qw, float Sig[3][3] /*[c][r]*/)
{
    float x = qx, y = qy, z = qz, w = qw;
    float R[3][3];
    R[0][0] = 1.0f - 2.0f * y * y - 2.0f * z * z;
    R[0][1] = 2.0f * x * y + 2.0f * z * w;
    R[0][2] = 2.0f * x * z - 2.0f * y * w;
    R[1][0] = 2.0f * x * y - 2.0f * z * w;
    R[1][1] = 1.0f - 2.0f * x * x - 2.0f * z * z;
    R[1][2] = 2.0f * y * z + 2.0f * x * w;
    R[2][0] = 2.0f * x * z + 2.0f * y * w;
    R[2][1] = 2.0f * y * z - 2.0f * x * w;
    R[2][2] = 1.0f - 2.0f * x * x - 2.0f * y * y;
    float M[3][3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) M[c][r] = R[c][r] * s[c];
        // Sigma[c][r] = M[0][r]*M[0][c] + M[1][r]*M[1][c] + M[2][r]*M[2][c]
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) Sig[c][r] = M[0][r] * M[0][c] + M[1][r] * M[1][c] + M[2][r] * M[2][c];
}

// mp_cam_clamp (gs_projector/shader.cpp:146-158)
LCGS_HD void cam_clamp(const CamParams& cp, const float v[3], float t[3])
{
    float limx = 1.3f * cp.tanfovx;
    float limy = 1.3f * cp.tanfovy;
    float txtz = v[0] / v[2];
    float tytz = v[1] / v[2];
    t[0]       = clamp_(txtz, -limx, limx) * v[2];
    t[1]       = clamp_(tytz, -limy, limy) * v[2];
    t[2]       = v[2];
}

// ewasplat_cov_focal / ewasplat_cov (util/gaussian.hpp:52-70 / :31-49): T = W^T... with
// J (column-major) J[0]=(j00,0,j02), J[1]=(0,j11,j12), J[2]=0 and W columns = right, up, front:
//   T[0] = right*j00 + front*j02,  T[1] = up*j11 + front*j12,  T[2] = 0
//   A[c][r] = dot(T[r], Sigma[c]),  cov[c][r] = A[0][r]*T[c][0] + A[1][r]*T[c][1] + A[2][r]*T[c][2]
// Output (cov[0][0], cov[0][1], cov[1][1]).
LCGS_HD void ewa_cov2d(const CamParams& cp, const float Sig[3][3], const float t[3], bool use_focal, float cov2d[3])
{
    float j00, j11, j02, j12;
    if (use_focal) {
        j00 = cp.focalx / t[2];
        j11 = cp.focaly / t[2];
        j02 = (-cp.focalx * t[0]) / (t[2] * t[2]);
        j12 = (-cp.focaly * t[1]) / (t[2] * t[2]);
    } else {
        j00 = 1.0f / t[2];
        j11 = 1.0f / t[2];
        j02 = (-t[0]) / (t[2] * t[2]);
        j12 = (-t[1]) / (t[2] * t[2]);
    }
    float T0[3], T1[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        T0[r] = cp.right[r] * j00 + cp.front[r] * j02;
        T1[r] = cp.up[r] * j11 + cp.front[r] * j12;
    }
    // A[c][0] = dot(T0, Sig[c]); A[c][1] = dot(T1, Sig[c])
    float A0[3], A1[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        A0[c] = T0[0] * Sig[c][0] + T0[1] * Sig[c][1] + T0[2] * Sig[c][2];
        A1[c] = T1[0] * Sig[c][0] + T1[1] * Sig[c][1] + T1[2] * Sig[c][2];
    }
    cov2d[0] = A0[0] * T0[0] + A0[1] * T0[1] + A0[2] * T0[2]; // cov[0][0]
    cov2d[1] = A1[0] * T0[0] + A1[1] * T0[1] + A1[2] * T0[2]; // cov[0][1]
    cov2d[2] = A1[0] * T1[0] + A1[1] * T1[1] + A1[2] * T1[2]; // cov[1][1]
    if (!use_focal) { // gs_projector/shader.cpp:73-76
        cov2d[0] = cov2d[0] * 1.0f / (cp.tanfovx * cp.tanfovx);
        cov2d[1] = cov2d[1] * 1.0f / (cp.tanfovx * cp.tanfovy);
        cov2d[2] = cov2d[2] * 1.0f / (cp.tanfovy * cp.tanfovy);
    }
}

// mp_ndc2pix (lcgs/src/module.cpp:18-20)
LCGS_HD float ndc2pix(float v, uint32_t S) { return ((v + 1.0f) * (float)S - 1.0f) * 0.5f; }

// mp_get_rect (lcgs/src/module.cpp:22-36): both ends clamped to [0, grids-1] -- the last tile
// row/column never receives splats; this is reference behaviour and is reproduced.
LCGS_HD void get_rect(float px, float py, int32_t max_radius, uint32_t grid_x, uint32_t grid_y, uint32_t rmin[2],
                      uint32_t rmax[2])
{
    float    r  = (float)max_radius;
    uint32_t ax = f2u_sat((px - r) / (float)kBlockX);
    uint32_t ay = f2u_sat((py - r) / (float)kBlockY);
    uint32_t bx = f2u_sat(px + r + (float)kBlockX - 1.0f) / kBlockX;
    uint32_t by = f2u_sat(py + r + (float)kBlockY - 1.0f) / kBlockY;
    uint32_t hx = grid_x - 1u, hy = grid_y - 1u;
    rmin[0]     = ax < hx ? ax : hx;
    rmin[1]     = ay < hy ? ay : hy;
    rmax[0]     = bx < hx ? bx : hx;
    rmax[1]     = by < hy ? by : hy;
}

// shad_allocate_tiles body (gs_tile_splatter/shader.cpp:123-157): low-pass, conic, radius.
// `cov` is (xx, xy, yy) as produced by the projector; for use_focal == false the caller's
// resolution scaling of shader.cpp:130-135 (including the res.y*res.x factor on yy) is applied.
LCGS_HD void conic_and_radius(float cx, float cy, float cz, bool use_focal, uint32_t res_x, uint32_t res_y,
                              float conic[3], int32_t& radius, float* filtered = nullptr)
{
    if (!use_focal) {
        cx = cx * (float)res_x * (float)res_x * 0.25f;
        cy = cy * (float)res_x * (float)res_y * 0.25f;
        cz = cz * (float)res_y * (float)res_x * 0.25f;
    }
    cx += 0.3f;
    cz += 0.3f;
    if (filtered) {
        filtered[0] = cx;
        filtered[1] = cy;
        filtered[2] = cz;
    }
    float det     = cx * cz - cy * cy;
    float inv_det = 1.0f / (det + 1e-6f);
    conic[0]      = inv_det * cz;
    conic[1]      = inv_det * (-cy);
    conic[2]      = inv_det * cx;
    float mid     = 0.5f * (cx + cz);
    float lambda1 = mid + sqrtf(fmax_(0.1f, mid * mid - det));
    float lambda2 = mid - sqrtf(fmax_(0.1f, mid * mid - det));
    radius        = f2i_sat(ceilf(3.0f * sqrtf(fmax_(lambda1, lambda2))));
}

// ---------------------------------------------------------------------------------------------
// Opacity-aware tight tile rect (NOT in the reference; an exact pruning of its pair list).
// The reference emits a (tile, splat) pair for every tile of the square rect of half-width
// radius = ceil(3 sqrt(lambda_max)).  A pixel can only receive a contribution from the splat if
// alpha = min(0.99, o exp(power)) >= 1/255 (gs_tile_splatter/shader.cpp:257-259), i.e. if
// q(d) = d^T Q d <= t = 2 ln(255 o) with Q the conic.  The axis-aligned extent of that ellipse is
// |dx| <= sqrt(t * Qyy / det Q), |dy| <= sqrt(t * Qxx / det Q); with Q = (cz, -cy, cx) / (det_f + 1e-6)
// built from the filtered covariance (cx, cy, cz) this is sqrt(t * cx * (det_f + 1e-6) / det) -- computed
// with explicit slack for the cancellation in det, the rounding of log/exp and the pixel grid.  Tiles of
// the reference rect outside this box cannot contain a contributing pixel, so dropping their pairs leaves
// every pixel's blend sequence -- and therefore the image -- bit-identical.  Returns the pruned rect
// (subset of [rmin, rmax)); an empty rect means the splat contributes nowhere (e.g. opacity <= 1/255).
// ---------------------------------------------------------------------------------------------
LCGS_HD void tight_rect(float pix_x, float pix_y, float cx, float cy, float cz /* filtered cov */, float opacity,
                        const uint32_t rmin[2], const uint32_t rmax[2], uint32_t tmin[2], uint32_t tmax[2])
{
    tmin[0] = rmin[0]; tmin[1] = rmin[1]; tmax[0] = rmax[0]; tmax[1] = rmax[1];
    // This bound only has to be conservative (explicit slack below), so on the device the 1-ulp hardware
    // log / sqrt / reciprocal replace the IEEE library forms.
#if defined(__HIP_DEVICE_COMPILE__)
#define LCGS_FAST_LOG(x) __logf(x)
#define LCGS_FAST_SQRT(x) __builtin_amdgcn_sqrtf(x)
#define LCGS_FAST_RCP(x) __builtin_amdgcn_rcpf(x)
#else
#define LCGS_FAST_LOG(x) logf(x)
#define LCGS_FAST_SQRT(x) sqrtf(x)
#define LCGS_FAST_RCP(x) (1.0f / (x))
#endif
    const float t0 = 2.0f * LCGS_FAST_LOG(255.0f * opacity);
    if (!(t0 == t0)) return;                 // NaN opacity: keep the reference rect
    const float t = t0 * 1.0001f + 2e-4f;    // same margin as the renderer's per-tile test
    if (!(t > 0.0f)) {                       // alpha < 1/255 everywhere
        tmax[0] = tmin[0];
        tmax[1] = tmin[1];
        return;
    }
    const float det  = cx * cz - cy * cy;
    const float err  = 4e-7f * (fabsf(cx * cz) + cy * cy); // rounding of the two products that cancel in det
    const float dlow = det - err;
    if (!(dlow > 1e-12f)) return;            // ill-conditioned / non-finite: keep the reference rect
    const float s  = (det + 1e-6f + err) * LCGS_FAST_RCP(dlow);
    const float hx = LCGS_FAST_SQRT(t * cx * s) * 1.0001f + 0.01f;
    const float hy = LCGS_FAST_SQRT(t * cz * s) * 1.0001f + 0.01f;
#undef LCGS_FAST_LOG
#undef LCGS_FAST_SQRT
#undef LCGS_FAST_RCP
    if (!(hx == hx) || !(hy == hy)) return;
    // tiles whose pixel span [16 tx, 16 tx + 15] meets [m - h, m + h]:
    //   16 tx + 15 >= m - h  <=>  tx >= ceil((m - h - 15) / 16);   16 tx <= m + h  <=>  tx <= floor((m + h) / 16)
    const uint32_t ax = f2u_sat(ceilf((pix_x - hx - (float)(kBlockX - 1)) / (float)kBlockX));
    const uint32_t ay = f2u_sat(ceilf((pix_y - hy - (float)(kBlockY - 1)) / (float)kBlockY));
    const float    fx = floorf((pix_x + hx) / (float)kBlockX), fy = floorf((pix_y + hy) / (float)kBlockY);
    const uint32_t bx = fx < 0.0f ? 0u : f2u_sat(fx + 1.0f); // exclusive
    const uint32_t by = fy < 0.0f ? 0u : f2u_sat(fy + 1.0f);
    tmin[0] = ax > rmin[0] ? ax : rmin[0];
    tmin[1] = ay > rmin[1] ? ay : rmin[1];
    tmax[0] = bx < rmax[0] ? bx : rmax[0];
    tmax[1] = by < rmax[1] ? by : rmax[1];
    if (tmax[0] < tmin[0]) tmax[0] = tmin[0];
    if (tmax[1] < tmin[1]) tmax[1] = tmin[1];
}

} // namespace lcgs
