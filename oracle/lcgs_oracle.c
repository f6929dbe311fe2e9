/*
 * lcgs_oracle.c -- CPU restatement of the reference forward path.  TEST INFRASTRUCTURE ONLY
 * (see lcgs_oracle.h for the usage rule and the pinning status of each stage).
 *
 * Every function cites the reference file:line it restates (paths relative to /root/reference).
 * Conventions restated from LuisaCompute (source absent, see DESIGN.md "parity unpinned"):
 *   - float3x3 / float4x4 are column-major, m[c][r]; here m[c*N + r].
 *   - M * v = m[0]*v.x + m[1]*v.y + m[2]*v.z (+ m[3]*v.w), summed left to right.
 *   - A * B is column-wise A * B[c].
 *   - DSL locals are zero-initialised (gaussian.hpp:20-23,55-59 rely on it).
 *   - dot(a,b) = a.x*b.x + a.y*b.y + a.z*b.z, left to right; normalize(v) = v * (1/sqrt(dot(v,v))).
 *   - clamp(v,lo,hi) = min(max(v,lo),hi).
 *   - float -> (u)int conversion saturates, NaN -> 0 (CUDA cvt.rzi / AMD v_cvt_*32_f32).
 * Compile with -ffp-contract=off: no fused multiply-add is formed anywhere.
 */
#include "lcgs_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifdef ORC_IS_DOUBLE
#define R_SQRT sqrt
#define R_EXP exp
#define R_CEIL ceil
#define R_TAN tan
#define R_FABS fabs
#else
#define R_SQRT sqrtf
#define R_EXP orc_blend_exp_sel
#define R_CEIL ceilf
#define R_TAN tanf
#define R_FABS fabsf
#endif
#define RC(x) ((real)(x))

#ifndef ORC_IS_DOUBLE
/* The blend's exp (`exp(power)`, gs_tile_splatter/shader.cpp:258).  The reference's own exp is whatever LuisaCompute's
 * JIT maps it to on its backend -- unpinned (CUDA documents expf to 2 ulp and the fast-math __expf to
 * 2 + floor(|1.16 x|) ulp) -- and the hard thresholds behind it (alpha < 1/255, T < 1e-4) turn a 1-ulp difference into
 * a different pixel.  The build therefore DEFINES the function as a fixed sequence of IEEE-754 binary32 operations:
 *   t = fma(x, log2e, 1.5 * 2^23);  n = t - 1.5 * 2^23 = rint(x log2e);  f = fma(x, log2e, -n)   (|f| <= 1/2)
 *   2^f = E(f^2) + f O(f^2): degree-6 minimax of 2^f on [-1/2, 1/2] with P(0) = 1;  result = 2^f * 2^n by exponent add.
 * Checked over EVERY binary32 in [-6, 0] (alpha >= 1/255 needs power >= -5.54): <= 2.73 ulp from the true value; exp(0)
 * is exactly 1.  Below -86 the function is defined as 0 (true value < 5e-38), NaN passes through.  The HIP kernels run
 * the same sequence (csrc/kernels/gs_math.hpp::blend_exp), so images are comparable bit for bit.  orc_set_blend_exp(1)
 * switches this file back to libm's expf, which tests use to show that the definition moves no pixel by more than
 * rounding noise unless a threshold flips.  (The f64 build, used for finite differences only, keeps libm's exp.) */
static int g_exp_libm = 0;
void orc_set_blend_exp(int use_libm) { g_exp_libm = use_libm; }

float orc_blend_exp(float x)
{
    if (!(x >= -86.0f)) return x < -86.0f ? 0.0f : x;
    const float t  = __builtin_fmaf(x, 0x1.715476p+0f, 12582912.0f);
    const float n  = t - 12582912.0f;
    const float f  = __builtin_fmaf(x, 0x1.715476p+0f, -n);
    const float f2 = f * f;
    float       E  = __builtin_fmaf(0x1.470b4ap-13f, f2, 0x1.3b270ep-7f);
    E              = __builtin_fmaf(E, f2, 0x1.ebfbe2p-3f);
    E              = __builtin_fmaf(E, f2, 1.0f);
    float O        = __builtin_fmaf(0x1.5f7276p-10f, f2, 0x1.c6ae72p-5f);
    O              = __builtin_fmaf(O, f2, 0x1.62e432p-1f);
    const float p  = __builtin_fmaf(O, f, E);
    uint32_t    pb, tb;
    memcpy(&pb, &p, 4);
    memcpy(&tb, &t, 4);
    pb += tb << 23;
    float r;
    memcpy(&r, &pb, 4);
    return r;
}

float orc_blend_exp_sel(float x) { return g_exp_libm ? expf(x) : orc_blend_exp(x); }
void  orc_blend_exp_array(int64_t n, const float* x, float* out)
{
    for (int64_t i = 0; i < n; ++i) out[i] = orc_blend_exp(x[i]);
}
#else
void orc_set_blend_exp(int use_libm) { (void)use_libm; }
#endif

/* ---- numerics VARIANTS (comparison runs only; the parity oracle is the default: everything off, -ffp-contract=off) ----
 * The reference's kernels are JIT-compiled by LuisaCompute for its backend; on CUDA that means NVRTC, which contracts
 * a*b+c into FMAs by default and, under fast-math, turns a/b into a*rcp(b) and sqrt / normalize into rsqrt forms.  None of
 * that is knowable offline (LuisaCompute is absent), so the restatement is ALSO built and run under those perturbations --
 * samples of what such a compiler may do, not replicas -- to measure how far the frame moves and to check that every
 * moved pixel is attributable to a rounding-sensitive decision the oracle itself can name (oracle/numerics.py):
 *   - contraction: this file compiled with -ffp-contract=fast (Makefile target liblcgs_oracle_f32_contract.so, macro
 *     ORC_CONTRACTED); the HOST-side camera code (util/camera.h, gs_projector/impl.cpp:34-42 run in C++ on the host, not
 *     in the JIT) stays uncontracted in every build: the `#pragma GCC optimize` region below.
 *   - ORC_NUM_RCP_DIV: every device-side a / b becomes a * (1 / b);
 *   - ORC_NUM_RSQRT:  normalize(v) = v * rsqrt(dot) with a single-rounding rsqrt, sqrt(x) = x * rsqrt(x);
 *   - ORC_NUM_REASSOC: dot products and matrix-vector sums added right to left (LuisaCompute's own order is assumed, not
 *     known: header above);
 *   - ORC_NUM_REASSOC2: the third grouping, (a + c) + b -- kept OUT of the ensemble that measures the uncertainties
 *     (oracle/numerics.py ENSEMBLE), as the independent check of them.
 * The blend's exp stays the defined sequence (explicit fmaf builtins) unless orc_set_blend_exp(1) asks for libm's. */
static int g_num = 0;
void orc_set_numerics(int flags) { g_num = flags; }
int  orc_get_numerics(void) { return g_num; }
int  orc_build_contracted(void)
{
#ifdef ORC_CONTRACTED
    return 1;
#else
    return 0;
#endif
}

static int g_threads = 0;
/* FD-validation aid: when non-zero the two hard blend thresholds (alpha < 1/255 skip, T < 1e-4 stop) are disabled in
 * the forward AND the backward, which makes the image a smooth function of the parameters so that central
 * differences of the f64 build validate the analytic gradients to ~1e-7.  Never set for parity work. */
int orc_g_smooth = 0;
void orc_set_smooth(int on) { orc_g_smooth = on; }

/* BUILD-DEFINED opt-in rule (no reference counterpart; doc/roadmap.md:8 only names LOD): orc_render treats a splat whose
 * radius (:148) is below this many pixels as touching no tile.  0 = off = the reference's behaviour. */
static int g_lod_min_radius = 0;
void orc_set_lod_min_radius(int px) { g_lod_min_radius = px; }

int orc_sizeof_real(void) { return (int)sizeof(real); }

void orc_set_threads(int n)
{
    g_threads = n;
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    else omp_set_num_threads(omp_get_num_procs());
#endif
}

int orc_get_threads(void)
{
#ifdef _OPENMP
    return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------ small vector helpers */
static inline real r_min(real a, real b) { return a < b ? a : b; }
static inline real r_max(real a, real b) { return a > b ? a : b; }
static inline real r_clamp(real v, real lo, real hi) { return r_min(r_max(v, lo), hi); }
static inline real dot3(const real a[3], const real b[3])
{
    if (g_num & ORC_NUM_REASSOC) return a[0] * b[0] + (a[1] * b[1] + a[2] * b[2]);
    if (g_num & ORC_NUM_REASSOC2) return (a[0] * b[0] + a[2] * b[2]) + a[1] * b[1];
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
static inline void cross3(const real a[3], const real b[3], real o[3])
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
/* device-side division / square root / normalize: the reference's expressions unless a numerics variant is switched on */
static inline real d_div(real a, real b) { return (g_num & ORC_NUM_RCP_DIV) ? a * (RC(1.0f) / b) : a / b; }
static inline real d_rsqrt1(real x) { return (real)(1.0 / sqrt((double)x)); } /* one rounding, like a hardware rsqrt */
static inline real d_sqrt(real x)
{
    if ((g_num & ORC_NUM_RSQRT) && x > RC(0.0f) && x < (real)INFINITY) return x * d_rsqrt1(x);
    return R_SQRT(x);
}
static inline void normalize3(const real v[3], real o[3])
{
    real inv = (g_num & ORC_NUM_RSQRT) ? d_rsqrt1(dot3(v, v)) : RC(1.0f) / R_SQRT(dot3(v, v));
    o[0] = v[0] * inv;
    o[1] = v[1] * inv;
    o[2] = v[2] * inv;
}
/* saturating conversions (CUDA cvt.rzi.{u32,s32}.f32 / AMD v_cvt_{u32,i32}_f32) */
static inline uint32_t f2u_sat(real x)
{
    if (!(x > RC(0.0f))) return 0u; /* negative, zero, NaN */
    if (x >= RC(4294967296.0)) return 0xFFFFFFFFu;
    return (uint32_t)x;
}
static inline int32_t f2i_sat(real x)
{
    if (x != x) return 0;
    if (x >= RC(2147483648.0)) return 2147483647;
    if (x <= RC(-2147483648.0)) return (int32_t)(-2147483647 - 1);
    return (int32_t)x;
}

/* column-major 3x3: m[c*3+r].  LC order: out = m[0]*v.x + m[1]*v.y + m[2]*v.z */
static inline void m3_mul_v3(const real m[9], const real v[3], real o[3])
{
    if (g_num & ORC_NUM_REASSOC) {
        for (int r = 0; r < 3; ++r) o[r] = m[0 * 3 + r] * v[0] + (m[1 * 3 + r] * v[1] + m[2 * 3 + r] * v[2]);
        return;
    }
    if (g_num & ORC_NUM_REASSOC2) {
        for (int r = 0; r < 3; ++r) o[r] = (m[0 * 3 + r] * v[0] + m[2 * 3 + r] * v[2]) + m[1 * 3 + r] * v[1];
        return;
    }
    for (int r = 0; r < 3; ++r) o[r] = m[0 * 3 + r] * v[0] + m[1 * 3 + r] * v[1] + m[2 * 3 + r] * v[2];
}
static inline void m3_mul(const real a[9], const real b[9], real o[9])
{
    real t[9];
    for (int c = 0; c < 3; ++c) m3_mul_v3(a, &b[c * 3], &t[c * 3]);
    memcpy(o, t, sizeof(t));
}
static inline void m3_transpose(const real a[9], real o[9])
{
    real t[9];
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) t[c * 3 + r] = a[r * 3 + c];
    memcpy(o, t, sizeof(t));
}

void orc_mat4_mul_vec4(const real m[16], const real v[4], real out[4])
{
    if (g_num & ORC_NUM_REASSOC) {
        for (int r = 0; r < 4; ++r)
            out[r] = m[0 * 4 + r] * v[0] + (m[1 * 4 + r] * v[1] + (m[2 * 4 + r] * v[2] + m[3 * 4 + r] * v[3]));
        return;
    }
    if (g_num & ORC_NUM_REASSOC2) {
        for (int r = 0; r < 4; ++r)
            out[r] = (m[0 * 4 + r] * v[0] + m[2 * 4 + r] * v[2]) + (m[1 * 4 + r] * v[1] + m[3 * 4 + r] * v[3]);
        return;
    }
    for (int r = 0; r < 4; ++r)
        out[r] = m[0 * 4 + r] * v[0] + m[1 * 4 + r] * v[1] + m[2 * 4 + r] * v[2] + m[3 * 4 + r] * v[3];
}

/* ------------------------------------------------------------------ camera */
/* HOST-side code of the reference (plain C++ in util/camera.h and gs_projector/impl.cpp, never seen by the JIT): compiled
 * without contraction in every build, with its own copies of the small vector helpers so that nothing is inlined across
 * the boundary. */
#pragma GCC push_options
#pragma GCC optimize("fp-contract=off")
static real h_dot3(const real a[3], const real b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void h_cross3(const real a[3], const real b[3], real o[3])
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
static void h_normalize3(const real v[3], real o[3])
{
    real inv = RC(1.0f) / R_SQRT(h_dot3(v, v));
    o[0] = v[0] * inv;
    o[1] = v[1] * inv;
    o[2] = v[2] * inv;
}
/* lcgs/include/lcgs/util/camera.h:74-82 */
void orc_get_lookat_cam(const real pos[3], const real target[3], const real world_up[3], orc_camera* cam)
{
    real d[3] = { target[0] - pos[0], target[1] - pos[1], target[2] - pos[2] };
    real c[3];
    memcpy(cam->position, pos, 3 * sizeof(real));
    h_normalize3(d, cam->front);
    h_cross3(cam->front, world_up, c);
    h_normalize3(c, cam->right);
    h_cross3(cam->right, cam->front, c);
    h_normalize3(c, cam->up);
    /* defaults, camera.h:21-24 */
    cam->fov          = RC(60.0f);
    cam->aspect_ratio = RC(1.0f);
    cam->width        = 512;
    cam->height       = 512;
}

/* camera.h:27-36 */
void orc_local_to_world_matrix(const orc_camera* cam, real m[16])
{
    for (int r = 0; r < 3; ++r) {
        m[0 * 4 + r] = cam->right[r];
        m[1 * 4 + r] = cam->up[r];
        m[2 * 4 + r] = cam->front[r];
        m[3 * 4 + r] = cam->position[r];
    }
    m[0 * 4 + 3] = m[1 * 4 + 3] = m[2 * 4 + 3] = RC(0.0f);
    m[3 * 4 + 3] = RC(1.0f);
}

/* camera.h:38-51 */
void orc_world_to_local_matrix(const orc_camera* cam, real m[16])
{
    real tx = -h_dot3(cam->position, cam->right);
    real ty = -h_dot3(cam->position, cam->up);
    real tz = -h_dot3(cam->position, cam->front);
    for (int c = 0; c < 3; ++c) {
        m[c * 4 + 0] = cam->right[c];
        m[c * 4 + 1] = cam->up[c];
        m[c * 4 + 2] = cam->front[c];
        m[c * 4 + 3] = RC(0.0f);
    }
    m[3 * 4 + 0] = tx;
    m[3 * 4 + 1] = ty;
    m[3 * 4 + 2] = tz;
    m[3 * 4 + 3] = RC(1.0f);
}

/* camera.h:54-72 */
void orc_projection_matrix(real tanfovx, real tanfovy, real znear, real zfar, real m[16])
{
    real zsign   = RC(1.0f);
    real fx      = RC(1.0f) / tanfovx;
    real fy      = RC(1.0f) / tanfovy;
    real z_range = zfar - znear;
    real a       = zfar / z_range;
    real b       = -zfar * znear / z_range;
    memset(m, 0, 16 * sizeof(real));
    m[0 * 4 + 0] = fx;
    m[1 * 4 + 1] = fy;
    m[2 * 4 + 2] = a * zsign;
    m[2 * 4 + 3] = zsign;
    m[3 * 4 + 2] = b;
}

/* host-side camera scalars, lcgs/src/gs_projector/impl.cpp:34-42 */
typedef struct cam_params {
    real tanfovx, tanfovy, focalx, focaly;
    real view[16], proj[16];
} cam_params;

static void make_cam_params(const orc_camera* cam, cam_params* cp)
{
    real fovy   = cam->fov / RC(180.0f) * RC(3.1415926536f);
    cp->tanfovy = R_TAN(fovy * RC(0.5f));
    cp->tanfovx = cp->tanfovy * cam->aspect_ratio;
    orc_world_to_local_matrix(cam, cp->view);
    orc_projection_matrix(cp->tanfovx, cp->tanfovy, RC(0.1f), RC(100.0f), cp->proj);
    cp->focalx = (real)cam->width / (RC(2.0f) * cp->tanfovx);
    cp->focaly = (real)cam->height / (RC(2.0f) * cp->tanfovy);
}
#pragma GCC pop_options

/* ------------------------------------------------------------------ SH */
/* lcgs/include/lcgs/util/sh.hpp:12-28 */
static const float SH_C0   = 0.28209479177387814f;
static const float SH_C1   = 0.4886025119029199f;
static const float SH_C2[5] = { 1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                -1.0925484305920792f, 0.5462742152960396f };
static const float SH_C3[7] = { -0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                -0.5900435899266435f };

/* The band sums of sh_preprocessor.cpp:49-150 for a given unit direction (no +0.5, no clamp).
 * Exposed so the band evaluators can be checked against the reference's own sh.hpp (oracle/_ref). */
void orc_sh_eval_dir(int deg, const real dir[3], const real* shs, real result[3])
{
    /* sh.hpp:31-34 */
    for (int ch = 0; ch < 3; ++ch) result[ch] = shs[ch] * RC(SH_C0);
    if (deg > 0) {
        real x = dir[0], y = dir[1], z = dir[2];
        const real* sh1 = shs + 1 * 3;
        const real* sh2 = shs + 2 * 3;
        const real* sh3 = shs + 3 * 3;
        /* sh.hpp:43-50: -SH_C1 * (sh_10 * y - sh_11 * z + sh_12 * x) */
        for (int ch = 0; ch < 3; ++ch)
            result[ch] = result[ch] + (-RC(SH_C1)) * (sh1[ch] * y - sh2[ch] * z + sh3[ch] * x);
        if (deg > 1) {
            real xx = x * x, yy = y * y, yz = y * z, zz = z * z, zx = z * x, xy = x * y;
            const real* s = shs + 4 * 3;
            /* sh.hpp:68-84 */
            for (int ch = 0; ch < 3; ++ch)
                result[ch] = result[ch] +
                             (RC(SH_C2[0]) * xy * s[0 * 3 + ch] + RC(SH_C2[1]) * yz * s[1 * 3 + ch] +
                              RC(SH_C2[2]) * (RC(2.0f) * zz - xx - yy) * s[2 * 3 + ch] +
                              RC(SH_C2[3]) * zx * s[3 * 3 + ch] + RC(SH_C2[4]) * (xx - yy) * s[4 * 3 + ch]);
            if (deg > 2) {
                const real* t = shs + 9 * 3;
                /* sh.hpp:120-138 */
                for (int ch = 0; ch < 3; ++ch)
                    result[ch] =
                        result[ch] +
                        (RC(SH_C3[0]) * y * (RC(3.0f) * xx - yy) * t[0 * 3 + ch] +
                         RC(SH_C3[1]) * xy * z * t[1 * 3 + ch] +
                         RC(SH_C3[2]) * y * (RC(4.0f) * zz - xx - yy) * t[2 * 3 + ch] +
                         RC(SH_C3[3]) * z * (RC(2.0f) * zz - RC(3.0f) * xx - RC(3.0f) * yy) * t[3 * 3 + ch] +
                         RC(SH_C3[4]) * x * (RC(4.0f) * zz - xx - yy) * t[4 * 3 + ch] +
                         RC(SH_C3[5]) * z * (xx - yy) * t[5 * 3 + ch] +
                         RC(SH_C3[6]) * x * (xx - RC(3.0f) * yy) * t[6 * 3 + ch]);
            }
        }
    }
}

/* sh_preprocessor.cpp:27-157 (callable mp_compute_color_from_sh) for one splat.
 * shs points at this splat's (deg+1)^2 x 3 coefficients. */
static void sh_color_one(int deg, const real campos[3], const real pos[3], const real* shs, real out_raw[3])
{
    real result[3] = { shs[0], shs[1], shs[2] }; /* :42-49 */
    if (deg > -1) {
        real dir[3] = { RC(0.0f), RC(0.0f), RC(0.0f) };
        if (deg > 0) {
            real d[3] = { pos[0] - campos[0], pos[1] - campos[1], pos[2] - campos[2] };
            normalize3(d, dir); /* :55-56 */
        }
        orc_sh_eval_dir(deg, dir, shs, result);
        for (int ch = 0; ch < 3; ++ch) result[ch] = result[ch] + RC(0.5f); /* :150 */
    }
    out_raw[0] = result[0];
    out_raw[1] = result[1];
    out_raw[2] = result[2];
}

/* sh_preprocessor.cpp:159-188 */
void orc_sh_process(int P, int channel, int deg, const real campos[3],
                    const real* xyz, const real* sh, real* color, real* color_raw)
{
    (void)channel;
    int feat_dim = (deg + 1) * (deg + 1); /* :40 */
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; ++idx) {
        real raw[3];
        sh_color_one(deg, campos, &xyz[3 * (size_t)idx], &sh[(size_t)idx * feat_dim * 3], raw);
        for (int ch = 0; ch < 3; ++ch) {
            if (color_raw) color_raw[3 * (size_t)idx + ch] = raw[ch];
            color[3 * (size_t)idx + ch] = r_clamp(raw[ch], RC(0.0f), RC(1.0f)); /* :153 */
        }
    }
}

/* ------------------------------------------------------------------ projection */
/* lcgs/include/lcgs/util/transform.hpp:188-212, q = (x,y,z,w), column-major R[c*3+r] */
static void R_from_qvec(const real q[4], real R[9])
{
    real x = q[0], y = q[1], z = q[2], w = q[3];
    R[0 * 3 + 0] = RC(1.0f) - RC(2.0f) * y * y - RC(2.0f) * z * z;
    R[0 * 3 + 1] = RC(2.0f) * x * y + RC(2.0f) * z * w;
    R[0 * 3 + 2] = RC(2.0f) * x * z - RC(2.0f) * y * w;
    R[1 * 3 + 0] = RC(2.0f) * x * y - RC(2.0f) * z * w;
    R[1 * 3 + 1] = RC(1.0f) - RC(2.0f) * x * x - RC(2.0f) * z * z;
    R[1 * 3 + 2] = RC(2.0f) * y * z + RC(2.0f) * x * w;
    R[2 * 3 + 0] = RC(2.0f) * x * z + RC(2.0f) * y * w;
    R[2 * 3 + 1] = RC(2.0f) * y * z - RC(2.0f) * x * w;
    R[2 * 3 + 2] = RC(1.0f) - RC(2.0f) * x * x - RC(2.0f) * y * y;
}

/* lcgs/include/lcgs/util/gaussian.hpp:15-28 */
static void calc_cov(const real scale[3], const real qvec[4], real cov[9])
{
    real R[9], S[9], M[9], Mt[9];
    R_from_qvec(qvec, R);
    memset(S, 0, sizeof(S));
    S[0 * 3 + 0] = scale[0];
    S[1 * 3 + 1] = scale[1];
    S[2 * 3 + 2] = scale[2];
    m3_mul(R, S, M);
    m3_transpose(M, Mt);
    m3_mul(M, Mt, cov);
}

/* gaussian.hpp:31-49 (focal = 0) and :52-70 (focal = 1) */
static void ewasplat_cov(const real cov3d[9], const real t[3], const real view[16], int focal,
                         real focalx, real focaly, real out[9])
{
    real J[9], W[9], T[9], Tt[9], A[9];
    memset(J, 0, sizeof(J));
    if (focal) {
        J[0 * 3 + 0] = d_div(focalx, t[2]);
        J[1 * 3 + 1] = d_div(focaly, t[2]);
        J[0 * 3 + 2] = d_div(-focalx * t[0], t[2] * t[2]);
        J[1 * 3 + 2] = d_div(-focaly * t[1], t[2] * t[2]);
    } else {
        J[0 * 3 + 0] = RC(1.0f) / t[2];
        J[1 * 3 + 1] = RC(1.0f) / t[2];
        J[0 * 3 + 2] = d_div(-t[0], t[2] * t[2]);
        J[1 * 3 + 2] = d_div(-t[1], t[2] * t[2]);
    }
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) W[c * 3 + r] = view[c * 4 + r];
    m3_transpose(W, W);
    m3_mul(W, J, T);
    m3_transpose(T, Tt);
    m3_mul(Tt, cov3d, A);
    m3_mul(A, T, out);
}

/* gs_projector/shader.cpp:146-158 */
static void cam_clamp(const real p[3], real tanfovx, real tanfovy, real t[3])
{
    real limx = RC(1.3f) * tanfovx;
    real limy = RC(1.3f) * tanfovy;
    real txtz = d_div(p[0], p[2]);
    real tytz = d_div(p[1], p[2]);
    t[0] = r_clamp(txtz, -limx, limx) * p[2];
    t[1] = r_clamp(tytz, -limy, limy) * p[2];
    t[2] = p[2];
}

/* gs_projector/shader.cpp:20-80 (use_focal=0) and :82-139 (use_focal=1) */
static void project_one(const cam_params* cp, const real mean[3], const real s[3], const real rotq[4],
                        real scale_modifier, int use_focal, int* culled, real xy_ndc[2], real* depth,
                        real cov2d[3])
{
    real p_hom[4] = { mean[0], mean[1], mean[2], RC(1.0f) };
    real p_view_hom[4], p_proj_hom[4];
    orc_mat4_mul_vec4(cp->view, p_hom, p_view_hom);
    orc_mat4_mul_vec4(cp->proj, p_view_hom, p_proj_hom);
    real p_w  = RC(1.0f) / (p_proj_hom[3] + RC(1e-6f));
    xy_ndc[0] = p_proj_hom[0] * p_w;
    xy_ndc[1] = p_proj_hom[1] * p_w;
    if (p_view_hom[2] < RC(0.2f)) { /* :55 / :121 */
        *culled = 1;
        return;
    }
    *culled = 0;
    *depth  = p_view_hom[2];
    real scale[3] = { scale_modifier * s[0], scale_modifier * s[1], scale_modifier * s[2] };
    real qvec[4]  = { rotq[1], rotq[2], rotq[3], rotq[0] }; /* rotq.yzwx(): rxyz -> xyzw */
    real cov3d[9], t[3], cov[9];
    calc_cov(scale, qvec, cov3d);
    cam_clamp(p_view_hom, cp->tanfovx, cp->tanfovy, t);
    ewasplat_cov(cov3d, t, cp->view, use_focal, cp->focalx, cp->focaly, cov);
    cov2d[0] = cov[0 * 3 + 0];
    cov2d[1] = cov[0 * 3 + 1];
    cov2d[2] = cov[1 * 3 + 1];
    if (!use_focal) { /* :73-76 */
        cov2d[0] = d_div(cov2d[0] * RC(1.0f), cp->tanfovx * cp->tanfovx);
        cov2d[1] = d_div(cov2d[1] * RC(1.0f), cp->tanfovx * cp->tanfovy);
        cov2d[2] = d_div(cov2d[2] * RC(1.0f), cp->tanfovy * cp->tanfovy);
    }
}

void orc_project_gs(int P, const real* pos, const real* scale, const real* rotq, real scale_modifier,
                    real* means_2d, real* depth, real* covs_2d, const orc_camera* cam, int use_focal)
{
    cam_params cp;
    make_cam_params(cam, &cp);
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; ++idx) {
        int  culled;
        real ndc[2], d = RC(0.0f), c2[3];
        project_one(&cp, &pos[3 * (size_t)idx], &scale[3 * (size_t)idx], &rotq[4 * (size_t)idx],
                    scale_modifier, use_focal, &culled, ndc, &d, c2);
        if (culled) continue; /* reference returns before any write */
        depth[idx]                     = d;
        means_2d[2 * (size_t)idx + 0]  = ndc[0];
        means_2d[2 * (size_t)idx + 1]  = ndc[1];
        covs_2d[3 * (size_t)idx + 0]   = c2[0];
        covs_2d[3 * (size_t)idx + 1]   = c2[1];
        covs_2d[3 * (size_t)idx + 2]   = c2[2];
    }
}

/* ------------------------------------------------------------------ tiles */
#define BLOCK_X 16u /* lcgs/include/lcgs/module.h:17 */
#define BLOCK_Y 16u

/* lcgs/src/module.cpp:18-20 */
static inline real ndc2pix(real v, uint32_t S) { return ((v + RC(1.0f)) * (real)S - RC(1.0f)) * RC(0.5f); }

/* lcgs/src/module.cpp:22-36 */
static inline void get_rect(const real p[2], int32_t max_radius, uint32_t rect_min[2], uint32_t rect_max[2],
                            const uint32_t grids[2])
{
    real     r      = (real)max_radius;
    uint32_t ax     = f2u_sat((p[0] - r) / (real)BLOCK_X);
    uint32_t ay     = f2u_sat((p[1] - r) / (real)BLOCK_Y);
    uint32_t bx     = f2u_sat(p[0] + r + (real)BLOCK_X - RC(1.0f)) / BLOCK_X;
    uint32_t by     = f2u_sat(p[1] + r + (real)BLOCK_Y - RC(1.0f)) / BLOCK_Y;
    uint32_t hx     = grids[0] - 1u, hy = grids[1] - 1u;
    rect_min[0]     = ax < hx ? ax : hx; /* clamp(.,0,grids-1) on unsigned */
    rect_min[1]     = ay < hy ? ay : hy;
    rect_max[0]     = bx < hx ? bx : hx;
    rect_max[1]     = by < hy ? by : hy;
}

static inline void make_grids(int width, int height, uint32_t grids[2])
{
    /* gs_tile_splatter/impl.cpp:76-79 */
    grids[0] = ((uint32_t)width + BLOCK_X - 1u) / BLOCK_X;
    grids[1] = ((uint32_t)height + BLOCK_Y - 1u) / BLOCK_Y;
}

/* gs_tile_splatter/shader.cpp:102-163 */
void orc_allocate_tiles(int P, int width, int height, const real* depth, real* means_2d, real* covs_2d,
                        uint32_t* tiles_touched, int32_t* radii, int use_focal)
{
    uint32_t grids[2];
    uint32_t res[2] = { (uint32_t)width, (uint32_t)height };
    make_grids(width, height, grids);
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; ++idx) {
        radii[idx]         = 0;
        tiles_touched[idx] = 0u;
        if (depth[idx] < RC(0.2f)) continue; /* :120-121 */
        real ndc[2] = { means_2d[2 * (size_t)idx], means_2d[2 * (size_t)idx + 1] };
        real cx = covs_2d[3 * (size_t)idx], cy = covs_2d[3 * (size_t)idx + 1], cz = covs_2d[3 * (size_t)idx + 2];
        if (!use_focal) { /* :130-135, including the res.y*res.x factor on the zz term */
            cx = cx * (real)res[0] * (real)res[0] * RC(0.25f);
            cy = cy * (real)res[0] * (real)res[1] * RC(0.25f);
            cz = cz * (real)res[1] * (real)res[0] * RC(0.25f);
        }
        cx += RC(0.3f);
        cz += RC(0.3f);
        real det     = cx * cz - cy * cy;
        real inv_det = RC(1.0f) / (det + RC(1e-6f));
        real conic[3] = { inv_det * cz, inv_det * (-cy), inv_det * cx };
        real mid     = RC(0.5f) * (cx + cz);
        real lambda1 = mid + d_sqrt(r_max(RC(0.1f), mid * mid - det));
        real lambda2 = mid - d_sqrt(r_max(RC(0.1f), mid * mid - det));
        int32_t my_radius = f2i_sat(R_CEIL(RC(3.0f) * d_sqrt(r_max(lambda1, lambda2))));
        real     pix[2] = { ndc2pix(ndc[0], res[0]), ndc2pix(ndc[1], res[1]) };
        uint32_t rmin[2], rmax[2];
        get_rect(pix, my_radius, rmin, rmax, grids);
        uint32_t n = (rmax[0] - rmin[0]) * (rmax[1] - rmin[1]);
        radii[idx]                    = my_radius;
        tiles_touched[idx]            = n;
        covs_2d[3 * (size_t)idx + 0]  = conic[0];
        covs_2d[3 * (size_t)idx + 1]  = conic[1];
        covs_2d[3 * (size_t)idx + 2]  = conic[2];
        means_2d[2 * (size_t)idx + 0] = pix[0];
        means_2d[2 * (size_t)idx + 1] = pix[1];
    }
}

/* lcpp DeviceScan::InclusiveSum; call site gs_tile_splatter/impl.cpp:104 (u32 wrap-around) */
void orc_inclusive_sum(int n, const uint32_t* in, uint32_t* out)
{
    uint32_t acc = 0u;
    for (int i = 0; i < n; ++i) {
        acc += in[i];
        out[i] = acc;
    }
}

static inline uint32_t depth_bits(real depth)
{
    float    f = (float)depth;
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

/* gs_tile_splatter/shader.cpp:26-69 */
void orc_copy_with_keys(int P, int width, int height, const real* means_2d, const uint32_t* offsets,
                        const int32_t* radii, const real* depth, uint64_t* keys, uint32_t* values)
{
    uint32_t grids[2];
    make_grids(width, height, grids);
#pragma omp parallel for schedule(dynamic, 4096)
    for (int idx = 0; idx < P; ++idx) {
        int32_t radius = radii[idx];
        if (radius <= 0) continue;
        uint32_t off  = idx >= 1 ? offsets[idx - 1] : 0u;
        real     p[2] = { means_2d[2 * (size_t)idx], means_2d[2 * (size_t)idx + 1] };
        uint32_t rmin[2], rmax[2];
        get_rect(p, radius, rmin, rmax, grids);
        for (uint32_t j = rmin[1]; j < rmax[1]; ++j) {
            for (uint32_t i = rmin[0]; i < rmax[0]; ++i) {
                uint64_t key = (uint64_t)(i + j * grids[0]);
                key <<= 32;
                key |= (uint64_t)depth_bits(depth[idx]) & 0x00000000FFFFFFFFull;
                keys[off]   = key;
                values[off] = (uint32_t)idx;
                off         = off + 1u;
            }
        }
    }
}

/* lcpp DeviceRadixSort::SortPairs<ulong,uint> stands for "ascending stable sort" (SURVEY 8c).
 * Restated as a byte-wise LSD counting sort (stable by construction). */
void orc_sort_pairs(int64_t n, const uint64_t* keys_in, const uint32_t* vals_in, uint64_t* keys_out,
                    uint32_t* vals_out)
{
    if (n <= 0) return;
    uint64_t* kt = (uint64_t*)malloc((size_t)n * sizeof(uint64_t));
    uint32_t* vt = (uint32_t*)malloc((size_t)n * sizeof(uint32_t));
    uint64_t *ka = keys_out, *kb = kt;
    uint32_t *va = vals_out, *vb = vt;
    memcpy(ka, keys_in, (size_t)n * sizeof(uint64_t));
    memcpy(va, vals_in, (size_t)n * sizeof(uint32_t));
    uint64_t ored = 0, anded = ~0ull;
    for (int64_t i = 0; i < n; ++i) {
        ored |= ka[i];
        anded &= ka[i];
    }
    for (int pass = 0; pass < 8; ++pass) {
        int shift = pass * 8;
        if ((((ored ^ anded) >> shift) & 0xFFull) == 0) continue; /* all keys share this digit */
        size_t count[257];
        memset(count, 0, sizeof(count));
        for (int64_t i = 0; i < n; ++i) count[((ka[i] >> shift) & 0xFF) + 1]++;
        for (int d = 0; d < 256; ++d) count[d + 1] += count[d];
        for (int64_t i = 0; i < n; ++i) {
            size_t dst = count[(ka[i] >> shift) & 0xFF]++;
            kb[dst]    = ka[i];
            vb[dst]    = va[i];
        }
        uint64_t* tk = ka; ka = kb; kb = tk;
        uint32_t* tv = va; va = vb; vb = tv;
    }
    if (ka != keys_out) {
        memcpy(keys_out, ka, (size_t)n * sizeof(uint64_t));
        memcpy(vals_out, va, (size_t)n * sizeof(uint32_t));
    }
    free(kt);
    free(vt);
}

/* gs_tile_splatter/shader.cpp:71-100 */
void orc_get_ranges(int64_t L, const uint64_t* keys, uint32_t* ranges)
{
    for (int64_t idx = 0; idx < L; ++idx) {
        uint32_t curr_tile = (uint32_t)(keys[idx] >> 32);
        if (idx == 0) {
            ranges[2 * (size_t)curr_tile + 0] = 0u;
        } else {
            uint32_t prev_tile = (uint32_t)(keys[idx - 1] >> 32);
            if (curr_tile != prev_tile) {
                ranges[2 * (size_t)prev_tile + 1] = (uint32_t)idx;
                ranges[2 * (size_t)curr_tile + 0] = (uint32_t)idx;
            }
        }
        if (idx == L - 1) ranges[2 * (size_t)curr_tile + 1] = (uint32_t)L;
    }
}

/* ------------------------------------------------------------------ render */
static inline int near_rel(real a, real b, real eps)
{
    real m = r_max(R_FABS(a), R_FABS(b));
    return R_FABS(a - b) <= eps * m;
}

/* gs_tile_splatter/shader.cpp:171-288.  The reference stages entries through shared memory in
 * rounds of 256; that only changes where the values are read from, not their values or order,
 * so the restatement walks the tile list directly.
 *
 * The optional outputs beyond the reference's image name what makes a pixel sensitive to rounding (oracle/numerics.py):
 *   cls[H*W]   ORC_CLS_THRESHOLD  some `power > 0`, `alpha < 1/255` or `T < 1e-4` decision of the pixel was within
 *                                 ambig_eps (relative) of flipping (the `ambig` flag of rounds 1-5) -- the window widened
 *                                 by window_factor x what the entry's own uncertainty does to its alpha, and, for T, by
 *                                 the accumulated uncertainty of the factors in front;
 *              ORC_CLS_DEPTH      two entries that BOTH contribute to the pixel lie closer in depth than the sum of their
 *                                 depth uncertainties (depth_tol[], absolute): the stable sort may order them either way
 *                                 (equal depth bits included: a perturbed evaluation separates them);
 *              (ORC_CLS_RECT is set by orc_mark_rect_uncertain below);
 *              a bit is set when the flip could move the pixel by more than impact_floor;
 *   flip[H*W]  the sum of what each ambiguous decision could move the pixel by if it went the other way: an alpha skip
 *              T alpha, a `T < 1e-4` stop T, a swap of two adjacent contributors T_i alpha_i alpha_j;
 *   sens[H*W]  first-order bound of what the per-splat uncertainties drec[P][n_var][5] (signed distances of n_var other
 *              evaluations of the splat's record; |d power| = the largest any of them gives at the pixel) / dcolor[P] can do
 *              to the pixel: sum over contributing entries of T alpha (|d power| + |d colour|) -- a perturbed alpha_k
 *              moves the pixel by at most T_k |d alpha_k| (colours lie in [0, 1], the entries behind scale with
 *              1 / (1 - alpha_k));
 *   sens_rss[H*W]  the same terms combined as independent errors: sqrt(sum of squares). */
static void render_forward_impl(int width, int height, const real bg[3], const uint32_t* ranges,
                                const uint32_t* point_list, const real* means_2d, const real* conic,
                                const real* opacity, const real* color, real* img, real* final_T,
                                uint32_t* n_contrib, uint8_t* ambig, real ambig_eps, const real* depth,
                                const real* depth_tol, int n_var, const real* drec, const real* dcolor,
                                real eval_eps, real mean_eps, real window_factor, real impact_floor, real* sens, real* sens_rss, real* flip)
{
    uint32_t grids[2];
    make_grids(width, height, grids);
    const size_t hw      = (size_t)width * (size_t)height;
    const int    n_tiles = (int)(grids[0] * grids[1]);
    const int    want_depth = ambig && depth && depth_tol;
    const int    want_sens  = n_var > 0 && drec;
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile_id = 0; tile_id < n_tiles; ++tile_id) {
        uint32_t tx = (uint32_t)tile_id % grids[0], ty = (uint32_t)tile_id / grids[0];
        uint32_t range_start = ranges[2 * (size_t)tile_id + 0];
        uint32_t range_end   = ranges[2 * (size_t)tile_id + 1];
        for (uint32_t ly = 0; ly < BLOCK_Y; ++ly) {
            for (uint32_t lx = 0; lx < BLOCK_X; ++lx) {
                uint32_t x = tx * BLOCK_X + lx, y = ty * BLOCK_Y + ly;
                if (!(x < (uint32_t)width && y < (uint32_t)height)) continue; /* inside, :194 */
                real     pix_f[2]         = { (real)x, (real)y };                  /* :197-200 */
                real     T                = RC(1.0f);
                real     C[3]             = { RC(0.0f), RC(0.0f), RC(0.0f) };
                uint32_t contributor      = 0u;
                uint32_t last_contributor = 0u;
                uint8_t  amb              = 0;
                real     last_depth = RC(0.0f), last_tol = RC(0.0f), sn = RC(0.0f), sq = RC(0.0f), T_unc = RC(0.0f), T_jump = RC(0.0f), fl = RC(0.0f);
                real     last_w     = RC(0.0f); /* T alpha of the previous contributor */
                real     last_Tb    = RC(0.0f); /* T in front of the previous contributor */
                int      have_last  = 0;
/* an ambiguous decision: its class bit (when the flip could move the pixel by more than impact_floor) and its impact */
#define ORC_AMBIGUOUS(bit, impact)                       \
    do {                                                 \
        if ((impact) > impact_floor) amb |= (bit);       \
        fl = fl + (impact);                              \
    } while (0)
                for (uint32_t e = range_start; e < range_end; ++e) {
                    contributor    = contributor + 1u;
                    uint32_t id    = point_list[e];
                    real     dx    = means_2d[2 * (size_t)id + 0] - pix_f[0];
                    real     dy    = means_2d[2 * (size_t)id + 1] - pix_f[1];
                    real     cx    = conic[3 * (size_t)id + 0];
                    real     cy    = conic[3 * (size_t)id + 1];
                    real     cz    = conic[3 * (size_t)id + 2];
                    real     o     = opacity[id];
                    real     power = RC(-0.5f) * (cx * dx * dx + cz * dy * dy) - cy * dx * dy; /* :256 */
                    /* what the splat's own uncertainty does to `power` at this pixel (first order); it widens the windows
                     * of the entry's threshold decisions and feeds the continuous bound */
                    real dpow = RC(0.0f);
                    if (want_sens) {
                        /* the pixel's own evaluation of :256: each product and sum rounds at the size of the TERMS, which
                         * for a needle-shaped splat are far larger than the power they cancel to */
                        dpow = eval_eps * (RC(0.5f) * (R_FABS(cx) * dx * dx + R_FABS(cz) * dy * dy) + R_FABS(cy * dx * dy) +
                                           RC(2.0f)); /* + the exp itself: implementations differ by a few ulp (relative in alpha) */
                        /* the pixel mean ((ndc + 1) S - 1) / 2 is good to about an ulp of itself and of S / 2 -- 1e-4 px
                         * at x = 1500 -- whichever way it is evaluated; the spread over a handful of evaluations below
                         * is a sample of that (often exactly 0), so it gets this floor */
                        dpow = dpow + mean_eps * (R_FABS(cx * dx + cy * dy) * (R_FABS(means_2d[2 * (size_t)id + 0]) + RC(0.5f) * (real)width) +
                                                  R_FABS(cz * dy + cy * dx) * (R_FABS(means_2d[2 * (size_t)id + 1]) + RC(0.5f) * (real)height));
                        real dvar = RC(0.0f);
                        /* drec[id][v] = (d mean.x, d mean.y, d conic.x, d conic.y, d conic.z): the SIGNED distance of
                         * evaluation v's record from this one -- signed, because an ill-conditioned splat's conic terms
                         * are each far larger than the power they sum to, and their errors are as correlated */
                        const real* r = drec + (size_t)id * 5 * (size_t)n_var;
                        for (int v = 0; v < n_var; ++v, r += 5) {
                            real d = RC(-0.5f) * (r[2] * dx * dx + r[4] * dy * dy) - r[3] * dx * dy -
                                     (cx * dx + cy * dy) * r[0] - (cz * dy + cy * dx) * r[1];
                            dvar = r_max(dvar, R_FABS(d));
                        }
                        dpow = dpow + dvar;
                    }
                    const real w_a = ambig_eps + window_factor * dpow; /* relative window of alpha = o exp(power) */
                    if (ambig && R_FABS(power) <= w_a) ORC_AMBIGUOUS(ORC_CLS_THRESHOLD, T * r_min(RC(0.99f), o));
                    if (power > RC(0.0f)) continue;
                    real alpha = r_min(RC(0.99f), o * R_EXP(power));
                    if (ambig && near_rel(alpha, RC(1.0f) / RC(255.0f), w_a)) {
                        ORC_AMBIGUOUS(ORC_CLS_THRESHOLD, T * alpha);
                        T_jump = T_jump + alpha / (RC(1.0f) - alpha); /* blended or not: every later T is that uncertain */
                    }
                    if (!orc_g_smooth && alpha < RC(1.0f) / RC(255.0f)) continue;
                    real test_T = T * (RC(1.0f) - alpha);
                    /* T carries the relative uncertainty of every (1 - alpha_j) in front: alpha_j dpow_j / (1 - alpha_j) */
                    const real t_unc = alpha < RC(0.99f) ? alpha * dpow / (RC(1.0f) - alpha) : RC(0.0f);
                    if (ambig && near_rel(test_T, RC(0.0001f), ambig_eps + window_factor * (T_unc + t_unc) + T_jump))
                        ORC_AMBIGUOUS(ORC_CLS_THRESHOLD, T); /* this entry's T alpha and everything behind it: <= T */
                    if (!orc_g_smooth && test_T < RC(0.0001f)) { /* done = true; loop exits at next iteration, :261-265 */
                        /* ... unless the entry that stops the pixel and the last contributor may trade places: the other order
                         * can blend THIS entry (T_last alpha) and stop on that one instead */
                        if (want_depth && have_last && depth[id] - last_depth <= depth_tol[id] + last_tol)
                            ORC_AMBIGUOUS(ORC_CLS_DEPTH, last_Tb);
                        break;
                    }
                    if (want_depth) {
                        /* swapping two adjacent contributors i, j moves the pixel by T_i alpha_i alpha_j |c_i - c_j| */
                        if (have_last && depth[id] - last_depth <= depth_tol[id] + last_tol)
                            ORC_AMBIGUOUS(ORC_CLS_DEPTH, last_w * alpha);
                        last_w     = T * alpha;
                        last_Tb    = T;
                        last_depth = depth[id];
                        last_tol   = depth_tol[id];
                        have_last  = 1;
                    }
                    if (want_sens) {
                        const real term = T * (alpha < RC(0.99f) ? alpha * dpow : RC(0.0f)) + (dcolor ? T * alpha * dcolor[id] : RC(0.0f));
                        sn    = sn + term;
                        sq    = sq + term * term;
                        T_unc = T_unc + t_unc;
                    }
                    for (int ch = 0; ch < 3; ++ch) C[ch] = C[ch] + T * alpha * color[3 * (size_t)id + ch];
                    T                = test_T;
                    last_contributor = contributor;
                }
                size_t pix_id = (size_t)x + (size_t)width * (size_t)y;
                for (int ch = 0; ch < 3; ++ch) img[pix_id + (size_t)ch * hw] = bg[ch] * T + C[ch]; /* :279-286 */
                if (final_T) final_T[pix_id] = T;
                if (n_contrib) n_contrib[pix_id] = last_contributor;
                if (ambig) ambig[pix_id] = amb;
                if (sens) sens[pix_id] = sn;
                if (sens_rss) sens_rss[pix_id] = R_SQRT(sq);
                if (flip) flip[pix_id] = fl;
#undef ORC_AMBIGUOUS
            }
        }
    }
}

void orc_render_forward(int width, int height, const real bg[3], const uint32_t* ranges,
                        const uint32_t* point_list, const real* means_2d, const real* conic,
                        const real* opacity, const real* color, real* img, real* final_T,
                        uint32_t* n_contrib, uint8_t* ambig, real ambig_eps)
{
    render_forward_impl(width, height, bg, ranges, point_list, means_2d, conic, opacity, color, img, final_T, n_contrib,
                        ambig, ambig_eps, NULL, NULL, 0, NULL, NULL, RC(0.0f), RC(0.0f), RC(0.0f), RC(0.0f), NULL, NULL, NULL);
}

void orc_render_forward_ex(int width, int height, const real bg[3], const uint32_t* ranges,
                           const uint32_t* point_list, const real* means_2d, const real* conic,
                           const real* opacity, const real* color, real* img, real* final_T,
                           uint32_t* n_contrib, uint8_t* cls, real ambig_eps, const real* depth,
                           const real* depth_tol, int n_var, const real* drec, const real* dcolor,
                           real eval_eps, real mean_eps, real window_factor, real impact_floor, real* sens, real* sens_rss, real* flip)
{
    render_forward_impl(width, height, bg, ranges, point_list, means_2d, conic, opacity, color, img, final_T, n_contrib,
                        cls, ambig_eps, depth, depth_tol, n_var, drec, dcolor, eval_eps, mean_eps, window_factor, impact_floor, sens, sens_rss, flip);
}

/* ORC_CLS_RECT: which tiles a splat is listed in is decided by ceil(3 sqrt(lambda)) (shader.cpp:145-148) and by four
 * float -> uint conversions of (pix -+ radius) / 16 (module.cpp:30-35).  Given, per splat, the radii the rounding window
 * allows (r_lo <= radius <= r_hi, from oracle/numerics.py) and the uncertainty of its pixel mean (dmean), a tile is
 * UNCERTAIN for the splat when it lies in the widest rect those allow but not in the narrowest; every pixel of such a
 * tile that the splat would reach with alpha >= (1 - eps) / 255 gets the flag -- whether or not this run listed it there --
 * and alpha added to its flip bound.  (Conservative: the transmittance in front of the splat is ignored.)  Returns the number of splats with an uncertain tile. */
int64_t orc_mark_rect_uncertain(int P, int width, int height, const real* means_pix, const real* conic,
                                const real* opacity, const int32_t* r_lo, const int32_t* r_hi, const real* dmean,
                                real eps, real impact_floor, uint8_t* cls, real* flip)
{
    uint32_t grids[2];
    make_grids(width, height, grids);
    int64_t n_uncertain = 0;
#pragma omp parallel for schedule(dynamic, 4096) reduction(+ : n_uncertain)
    for (int idx = 0; idx < P; ++idx) {
        if (r_hi[idx] <= 0) continue;
        const real p[2]  = { means_pix[2 * (size_t)idx], means_pix[2 * (size_t)idx + 1] };
        const real d[2]  = { dmean[2 * (size_t)idx], dmean[2 * (size_t)idx + 1] };
        const real pm[2] = { p[0] - d[0], p[1] - d[1] }, pp[2] = { p[0] + d[0], p[1] + d[1] };
        uint32_t a[2], b[2], wmin[2], wmax[2], nmin[2], nmax[2];
        get_rect(pm, r_hi[idx], wmin, b, grids); /* widest: lowest min edge ... */
        get_rect(pp, r_hi[idx], a, wmax, grids); /* ... highest max edge */
        if (r_lo[idx] > 0) {
            get_rect(pp, r_lo[idx], nmin, b, grids);
            get_rect(pm, r_lo[idx], a, nmax, grids);
        } else {
            nmin[0] = nmin[1] = nmax[0] = nmax[1] = 0u; /* the splat may not be listed at all */
        }
        if (wmin[0] == nmin[0] && wmin[1] == nmin[1] && wmax[0] == nmax[0] && wmax[1] == nmax[1]) continue;
        n_uncertain += 1;
        const real cx = conic[3 * (size_t)idx], cy = conic[3 * (size_t)idx + 1], cz = conic[3 * (size_t)idx + 2];
        const real o = opacity[idx];
        for (uint32_t j = wmin[1]; j < wmax[1]; ++j)
            for (uint32_t i = wmin[0]; i < wmax[0]; ++i) {
                if (i >= nmin[0] && i < nmax[0] && j >= nmin[1] && j < nmax[1]) continue; /* certain */
                for (uint32_t ly = 0; ly < BLOCK_Y; ++ly)
                    for (uint32_t lx = 0; lx < BLOCK_X; ++lx) {
                        uint32_t x = i * BLOCK_X + lx, y = j * BLOCK_Y + ly;
                        if (!(x < (uint32_t)width && y < (uint32_t)height)) continue;
                        real dx = p[0] - (real)x, dy = p[1] - (real)y;
                        real power = RC(-0.5f) * (cx * dx * dx + cz * dy * dy) - cy * dx * dy;
                        if (power > eps) continue;
                        real alpha = r_min(RC(0.99f), o * R_EXP(r_min(power, RC(0.0f))));
                        if (alpha < (RC(1.0f) - eps) * (RC(1.0f) / RC(255.0f))) continue;
                        uint8_t* c = &cls[(size_t)x + (size_t)width * (size_t)y];
                        if (alpha > impact_floor) {
#pragma omp atomic update
                            *c |= ORC_CLS_RECT;
                        }
                        if (flip) {
                            real* f = &flip[(size_t)x + (size_t)width * (size_t)y];
#pragma omp atomic update
                            *f += alpha;
                        }
                    }
            }
    }
    return n_uncertain;
}

/* gs_tile_splatter/impl.cpp:63-180 */
int64_t orc_tile_splatter_forward(int P, int width, int height, const real bg[3], real* means_2d,
                                  const real* depth, real* covs_2d, const real* color, const real* opacity,
                                  uint32_t* tiles_touched, uint32_t* point_offsets, uint64_t* keys_unsorted,
                                  uint32_t* list_unsorted, uint64_t* keys, uint32_t* list, uint32_t* ranges,
                                  int64_t L_cap, real* img, int32_t* radii, int use_focal, real* final_T,
                                  uint32_t* n_contrib, uint8_t* ambig, real ambig_eps)
{
    uint32_t grids[2];
    make_grids(width, height, grids);
    orc_allocate_tiles(P, width, height, depth, means_2d, covs_2d, tiles_touched, radii, use_focal);
    orc_inclusive_sum(P, tiles_touched, point_offsets);
    int64_t num_rendered = P > 0 ? (int64_t)(int32_t)point_offsets[P - 1] : 0; /* int num_rendered, :106 */
    if (num_rendered <= 0) return 0;                                            /* :109, image untouched */
    if (num_rendered > L_cap) return -1;
    memset(list_unsorted, 0, (size_t)num_rendered * sizeof(uint32_t)); /* :117-118 */
    memset(keys_unsorted, 0, (size_t)num_rendered * sizeof(uint64_t));
    orc_copy_with_keys(P, width, height, means_2d, point_offsets, radii, depth, keys_unsorted, list_unsorted);
    orc_sort_pairs(num_rendered, keys_unsorted, list_unsorted, keys, list);
    memset(ranges, 0, (size_t)grids[0] * grids[1] * 2 * sizeof(uint32_t)); /* :147 */
    orc_get_ranges(num_rendered, keys, ranges);
    orc_render_forward(width, height, bg, ranges, list, means_2d, covs_2d, opacity, color, img, final_T,
                       n_contrib, ambig, ambig_eps);
    return num_rendered;
}

/* app/main.cpp:266-308 with zero-initialised intermediates (the only well-defined case) */
int64_t orc_render(int P, int sh_deg, const real* pos, const real* scale, const real* rotq, const real* sh,
                   const real* opacity, const orc_camera* cam, const real bg[3], real scale_modifier,
                   real* img, int32_t* radii, real* final_T, uint32_t* n_contrib, uint8_t* ambig,
                   real ambig_eps)
{
    int      width = cam->width, height = cam->height;
    uint32_t grids[2];
    make_grids(width, height, grids);
    size_t    G        = (size_t)grids[0] * grids[1];
    real*     color    = (real*)calloc((size_t)P * 3 + 1, sizeof(real));
    real*     means_2d = (real*)calloc((size_t)P * 2 + 1, sizeof(real));
    real*     depth    = (real*)calloc((size_t)P + 1, sizeof(real));
    real*     covs_2d  = (real*)calloc((size_t)P * 3 + 1, sizeof(real));
    uint32_t* tiles    = (uint32_t*)calloc((size_t)P + 1, sizeof(uint32_t));
    uint32_t* offsets  = (uint32_t*)calloc((size_t)P + 1, sizeof(uint32_t));
    uint32_t* ranges   = (uint32_t*)calloc(G * 2, sizeof(uint32_t));
    int32_t*  my_radii = radii ? radii : (int32_t*)calloc((size_t)P + 1, sizeof(int32_t));
    int64_t   result   = -1;
    uint64_t *ku = NULL, *ks = NULL;
    uint32_t *lu = NULL, *ls = NULL;
    if (!color || !means_2d || !depth || !covs_2d || !tiles || !offsets || !ranges || !my_radii) goto done;

    orc_sh_process(P, 3, sh_deg, cam->position, pos, sh, color, NULL);                       /* main.cpp:268 */
    orc_project_gs(P, pos, scale, rotq, scale_modifier, means_2d, depth, covs_2d, cam, 1);   /* main.cpp:269 */
    /* size the pair buffers exactly: run allocate+scan once on copies?  Cheaper: allocate in place,
     * scan, then allocate the pair buffers (the reference pre-allocates 20M, main.cpp:245). */
    orc_allocate_tiles(P, width, height, depth, means_2d, covs_2d, tiles, my_radii, 1);
    if (g_lod_min_radius > 0)
        for (int idx = 0; idx < P; ++idx)
            if (my_radii[idx] < g_lod_min_radius) {
                my_radii[idx] = 0; /* copy_with_keys skips radius <= 0 (shader.cpp:41-42) */
                tiles[idx]    = 0u;
            }
    orc_inclusive_sum(P, tiles, offsets);
    {
        int64_t L = P > 0 ? (int64_t)(int32_t)offsets[P - 1] : 0;
        if (L <= 0) {
            result = 0;
            goto done;
        }
        ku = (uint64_t*)calloc((size_t)L, sizeof(uint64_t));
        ks = (uint64_t*)calloc((size_t)L, sizeof(uint64_t));
        lu = (uint32_t*)calloc((size_t)L, sizeof(uint32_t));
        ls = (uint32_t*)calloc((size_t)L, sizeof(uint32_t));
        if (!ku || !ks || !lu || !ls) goto done;
        orc_copy_with_keys(P, width, height, means_2d, offsets, my_radii, depth, ku, lu);
        orc_sort_pairs(L, ku, lu, ks, ls);
        orc_get_ranges(L, ks, ranges);
        orc_render_forward(width, height, bg, ranges, ls, means_2d, covs_2d, opacity, color, img, final_T,
                           n_contrib, ambig, ambig_eps);
        result = L;
    }
done:
    free(color); free(means_2d); free(depth); free(covs_2d); free(tiles); free(offsets); free(ranges);
    if (!radii) free(my_radii);
    free(ku); free(ks); free(lu); free(ls);
    return result;
}

/* app/main.cpp:323-335 */
void orc_image_to_rgb8(int width, int height, const real* img_chw, uint8_t* rgb)
{
    int w = width, h = height;
    for (int i = 0; i < h; i++) {
        for (int j = 0; j < w; j++) {
            int pixel_idx = (i * w + j) * 3;
            int idx       = (h - i - 1) * w + j;
            for (int ch = 0; ch < 3; ++ch) {
                float v = (float)img_chw[(size_t)ch * h * w + idx] * 255; /* implicit float->uint8 */
                int   q = (int)v;                                          /* truncation toward zero */
                rgb[pixel_idx + ch] = (uint8_t)q;                          /* wraps like the reference's cast */
            }
        }
    }
}
