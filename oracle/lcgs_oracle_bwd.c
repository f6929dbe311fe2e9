/*
 * lcgs_oracle_bwd.c -- CPU oracle for the backward pass.  TEST INFRASTRUCTURE ONLY (see lcgs_oracle.h).
 *
 * The reference has NO backward (README.md:70); its only backward artefacts are the unused per-band dL/dSH
 * helpers in lcgs/include/lcgs/util/sh.hpp:37-40,53-65,87-117,141-165 (their dL_d_dir is a TODO), which the
 * SH part below agrees with (tests/test_oracle_backward.py checks it against oracle/_ref golden vectors).
 * Everything else is the analytic derivative of the forward restated in lcgs_oracle.c (DESIGN.md "Backward"),
 * and is pinned by central finite differences of the f64 build of that same forward.
 *
 * Gradient conventions: w.r.t. the ACTIVATED inputs -- pos[3], scale[3] (post-exp, before
 * scale_modifier), rotq[4] as stored (r,x,y,z; used un-normalised by the forward), sh[(deg+1)^2][3],
 * opacity (post-sigmoid).  Thresholds are treated as constants: the near cull, the alpha < 1/255 skip, the
 * T < 1e-4 stop and power > 0 gate the sums; min(0.99, .) passes no gradient when the cap is active; the
 * colour clamp passes gradient only for 0 < raw < 1; cam_clamp passes none along a saturated axis.
 */
#include "lcgs_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef ORC_IS_DOUBLE
#define R_SQRT sqrt
#define R_EXP exp
#define R_TAN tan
#else
#define R_SQRT sqrtf
#define R_EXP orc_blend_exp_sel /* the forward's exp (lcgs_oracle.c): the same entries pass alpha >= 1/255 */
#define R_TAN tanf
#endif
#define RC(x) ((real)(x))

static inline real b_min(real a, real b) { return a < b ? a : b; }
static inline real b_dot3(const real a[3], const real b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

#define BLOCK_X 16u
#define BLOCK_Y 16u

extern int orc_g_smooth; /* lcgs_oracle.c: FD-validation aid, disables the two hard blend thresholds */

/* ------------------------------------------------------------------ render backward
 * Walks each pixel's tile list from its last contributor back to the front (the forward of
 * gs_tile_splatter/shader.cpp:249-274 in reverse), re-deriving alpha and re-applying the forward's skips. */
void orc_render_backward(int width, int height, const real bg[3], const uint32_t* ranges, const uint32_t* point_list,
                         const real* means_2d, const real* conic, const real* opacity, const real* color,
                         const real* final_T, const uint32_t* n_contrib, const real* dL_dimg, real* dL_dmean2d,
                         real* dL_dconic, real* dL_dopacity, real* dL_dcolor)
{
    const uint32_t gx = ((uint32_t)width + BLOCK_X - 1u) / BLOCK_X;
    const size_t   hw = (size_t)width * (size_t)height;
    /* serial over pixels: the per-splat accumulation order is then fixed (a reproducible oracle) */
    for (int y = 0; y < height; ++y) {
        for (int x = 0; x < width; ++x) {
            const uint32_t tile  = ((uint32_t)y / BLOCK_Y) * gx + (uint32_t)x / BLOCK_X;
            const uint32_t start = ranges[2 * (size_t)tile + 0];
            const size_t   pix   = (size_t)x + (size_t)width * (size_t)y;
            const uint32_t last  = n_contrib[pix]; /* entries [0, last) were examined up to the last contributor */
            if (last == 0u) continue;
            const real T_final = final_T[pix];
            const real dpix[3] = { dL_dimg[pix], dL_dimg[pix + hw], dL_dimg[pix + 2 * hw] };
            const real bg_dot  = bg[0] * dpix[0] + bg[1] * dpix[1] + bg[2] * dpix[2];
            real       T = T_final;
            real       accum[3] = { RC(0.0f), RC(0.0f), RC(0.0f) }, last_color[3] = { RC(0.0f), RC(0.0f), RC(0.0f) };
            real       last_alpha = RC(0.0f);
            for (uint32_t j = last; j-- > 0u;) {
                const uint32_t id = point_list[start + j];
                const real     dx = means_2d[2 * (size_t)id + 0] - (real)x;
                const real     dy = means_2d[2 * (size_t)id + 1] - (real)y;
                const real     ca = conic[3 * (size_t)id + 0], cb = conic[3 * (size_t)id + 1], cc = conic[3 * (size_t)id + 2];
                const real     o  = opacity[id];
                const real     power = RC(-0.5f) * (ca * dx * dx + cc * dy * dy) - cb * dx * dy;
                if (power > RC(0.0f)) continue;
                const real G     = R_EXP(power);
                const real alpha = b_min(RC(0.99f), o * G);
                if (!orc_g_smooth && alpha < RC(1.0f) / RC(255.0f)) continue;
                T = T / (RC(1.0f) - alpha); /* = the forward's T before this splat */
                const real w = alpha * T;
                real       dL_dalpha = RC(0.0f);
                for (int ch = 0; ch < 3; ++ch) {
                    const real c = color[3 * (size_t)id + ch];
                    accum[ch]      = last_alpha * last_color[ch] + (RC(1.0f) - last_alpha) * accum[ch];
                    last_color[ch] = c;
                    dL_dalpha += (c - accum[ch]) * dpix[ch];
                    dL_dcolor[3 * (size_t)id + ch] += w * dpix[ch];
                }
                dL_dalpha *= T;
                last_alpha = alpha;
                dL_dalpha += (-T_final / (RC(1.0f) - alpha)) * bg_dot;
                if (o * G < RC(0.99f)) { /* the 0.99 cap passes no gradient */
                    const real dL_dG = o * dL_dalpha;
                    dL_dopacity[id] += G * dL_dalpha;
                    const real gdx = G * dx, gdy = G * dy;
                    dL_dmean2d[2 * (size_t)id + 0] += dL_dG * (-(gdx * ca + gdy * cb));
                    dL_dmean2d[2 * (size_t)id + 1] += dL_dG * (-(gdy * cc + gdx * cb));
                    dL_dconic[3 * (size_t)id + 0] += RC(-0.5f) * gdx * dx * dL_dG;
                    dL_dconic[3 * (size_t)id + 1] += -gdx * dy * dL_dG;
                    dL_dconic[3 * (size_t)id + 2] += RC(-0.5f) * gdy * dy * dL_dG;
                }
            }
        }
    }
}

/* ------------------------------------------------------------------ preprocess backward */
static const float SH_C0   = 0.28209479177387814f; /* lcgs/include/lcgs/util/sh.hpp:12-28 */
static const float SH_C1   = 0.4886025119029199f;
static const float SH_C2[5] = { 1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                -1.0925484305920792f, 0.5462742152960396f };
static const float SH_C3[7] = { -0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                -0.5900435899266435f };

/* basis_k(dir) and its gradient for k = 0..(deg+1)^2-1, signs as composed by sh_preprocessor.cpp:49-147 */
static void sh_basis(int deg, const real d[3], real basis[16], real dbx[16], real dby[16], real dbz[16])
{
    const real x = d[0], y = d[1], z = d[2];
    for (int k = 0; k < 16; ++k) basis[k] = dbx[k] = dby[k] = dbz[k] = RC(0.0f);
    basis[0] = RC(SH_C0);
    if (deg < 1) return;
    basis[1] = -RC(SH_C1) * y; dby[1] = -RC(SH_C1);
    basis[2] = RC(SH_C1) * z;  dbz[2] = RC(SH_C1);
    basis[3] = -RC(SH_C1) * x; dbx[3] = -RC(SH_C1);
    if (deg < 2) return;
    const real xx = x * x, yy = y * y, zz = z * z;
    basis[4] = RC(SH_C2[0]) * x * y; dbx[4] = RC(SH_C2[0]) * y; dby[4] = RC(SH_C2[0]) * x;
    basis[5] = RC(SH_C2[1]) * y * z; dby[5] = RC(SH_C2[1]) * z; dbz[5] = RC(SH_C2[1]) * y;
    basis[6] = RC(SH_C2[2]) * (RC(2.0f) * zz - xx - yy);
    dbx[6] = RC(SH_C2[2]) * (RC(-2.0f) * x); dby[6] = RC(SH_C2[2]) * (RC(-2.0f) * y); dbz[6] = RC(SH_C2[2]) * (RC(4.0f) * z);
    basis[7] = RC(SH_C2[3]) * z * x; dbx[7] = RC(SH_C2[3]) * z; dbz[7] = RC(SH_C2[3]) * x;
    basis[8] = RC(SH_C2[4]) * (xx - yy); dbx[8] = RC(SH_C2[4]) * RC(2.0f) * x; dby[8] = RC(SH_C2[4]) * RC(-2.0f) * y;
    if (deg < 3) return;
    basis[9]  = RC(SH_C3[0]) * y * (RC(3.0f) * xx - yy);
    dbx[9] = RC(SH_C3[0]) * RC(6.0f) * x * y; dby[9] = RC(SH_C3[0]) * (RC(3.0f) * xx - RC(3.0f) * yy);
    basis[10] = RC(SH_C3[1]) * x * y * z;
    dbx[10] = RC(SH_C3[1]) * y * z; dby[10] = RC(SH_C3[1]) * x * z; dbz[10] = RC(SH_C3[1]) * x * y;
    basis[11] = RC(SH_C3[2]) * y * (RC(4.0f) * zz - xx - yy);
    dbx[11] = RC(SH_C3[2]) * (RC(-2.0f) * x * y); dby[11] = RC(SH_C3[2]) * (RC(4.0f) * zz - xx - RC(3.0f) * yy);
    dbz[11] = RC(SH_C3[2]) * RC(8.0f) * y * z;
    basis[12] = RC(SH_C3[3]) * z * (RC(2.0f) * zz - RC(3.0f) * xx - RC(3.0f) * yy);
    dbx[12] = RC(SH_C3[3]) * (RC(-6.0f) * x * z); dby[12] = RC(SH_C3[3]) * (RC(-6.0f) * y * z);
    dbz[12] = RC(SH_C3[3]) * (RC(6.0f) * zz - RC(3.0f) * xx - RC(3.0f) * yy);
    basis[13] = RC(SH_C3[4]) * x * (RC(4.0f) * zz - xx - yy);
    dbx[13] = RC(SH_C3[4]) * (RC(4.0f) * zz - RC(3.0f) * xx - yy); dby[13] = RC(SH_C3[4]) * (RC(-2.0f) * x * y);
    dbz[13] = RC(SH_C3[4]) * RC(8.0f) * x * z;
    basis[14] = RC(SH_C3[5]) * z * (xx - yy);
    dbx[14] = RC(SH_C3[5]) * RC(2.0f) * x * z; dby[14] = RC(SH_C3[5]) * RC(-2.0f) * y * z; dbz[14] = RC(SH_C3[5]) * (xx - yy);
    basis[15] = RC(SH_C3[6]) * x * (xx - RC(3.0f) * yy);
    dbx[15] = RC(SH_C3[6]) * (RC(3.0f) * xx - RC(3.0f) * yy); dby[15] = RC(SH_C3[6]) * (RC(-6.0f) * x * y);
}

typedef struct bcam {
    real right[3], up[3], front[3], t[3];
    real tanx, tany, fx, fy;
    int  W, H;
} bcam;

static void make_bcam(const orc_camera* cam, bcam* c)
{
    /* gs_projector/impl.cpp:34-42, camera.h:38-51 */
    real fovy = cam->fov / RC(180.0f) * RC(3.1415926536f);
    c->tany   = R_TAN(fovy * RC(0.5f));
    c->tanx   = c->tany * cam->aspect_ratio;
    for (int i = 0; i < 3; ++i) {
        c->right[i] = cam->right[i];
        c->up[i]    = cam->up[i];
        c->front[i] = cam->front[i];
    }
    c->t[0] = -b_dot3(cam->position, cam->right);
    c->t[1] = -b_dot3(cam->position, cam->up);
    c->t[2] = -b_dot3(cam->position, cam->front);
    c->W    = cam->width;
    c->H    = cam->height;
    c->fx   = (real)cam->width / (RC(2.0f) * c->tanx);
    c->fy   = (real)cam->height / (RC(2.0f) * c->tany);
}

void orc_preprocess_backward(int P, int sh_deg, const real* pos, const real* scale, const real* rotq, const real* sh,
                             const orc_camera* cam, real scale_modifier, const int32_t* radii,
                             const real* dL_dmean2d, const real* dL_dconic, const real* dL_dcolor, real* dL_dpos,
                             real* dL_dscale, real* dL_drotq, real* dL_dsh)
{
    bcam c;
    make_bcam(cam, &c);
    const int feat = (sh_deg + 1) * (sh_deg + 1);
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; ++idx) {
        real* gp = &dL_dpos[3 * (size_t)idx];
        real* gs = &dL_dscale[3 * (size_t)idx];
        real* gq = &dL_drotq[4 * (size_t)idx];
        real* gh = &dL_dsh[(size_t)idx * feat * 3];
        gp[0] = gp[1] = gp[2] = gs[0] = gs[1] = gs[2] = gq[0] = gq[1] = gq[2] = gq[3] = RC(0.0f);
        for (int k = 0; k < feat * 3; ++k) gh[k] = RC(0.0f);
        if (radii[idx] <= 0) continue; /* near-culled (or degenerate): the forward wrote nothing for it */
        const real* p = &pos[3 * (size_t)idx];

        /* ---- colour -> SH coefficients and position (through the view direction) */
        {
            real d[3] = { p[0] - cam->position[0], p[1] - cam->position[1], p[2] - cam->position[2] };
            real len2 = b_dot3(d, d), inv = RC(1.0f) / R_SQRT(len2);
            real dir[3] = { d[0] * inv, d[1] * inv, d[2] * inv };
            real basis[16], dbx[16], dby[16], dbz[16];
            sh_basis(sh_deg, dir, basis, dbx, dby, dbz);
            const real* s = &sh[(size_t)idx * feat * 3];
            real        ddir[3] = { RC(0.0f), RC(0.0f), RC(0.0f) };
            for (int ch = 0; ch < 3; ++ch) {
                real raw = RC(0.5f);
                for (int k = 0; k < feat; ++k) raw += basis[k] * s[k * 3 + ch];
                if (!(raw > RC(0.0f) && raw < RC(1.0f))) continue; /* clamp(.,0,1) saturated */
                const real g = dL_dcolor[3 * (size_t)idx + ch];
                for (int k = 0; k < feat; ++k) {
                    gh[k * 3 + ch] = basis[k] * g;
                    ddir[0] += g * s[k * 3 + ch] * dbx[k];
                    ddir[1] += g * s[k * 3 + ch] * dby[k];
                    ddir[2] += g * s[k * 3 + ch] * dbz[k];
                }
            }
            /* dir = d / |d|: dL/dd = (dL/ddir - dir (dir . dL/ddir)) / |d| */
            real dd = b_dot3(dir, ddir);
            for (int i = 0; i < 3; ++i) gp[i] += (ddir[i] - dir[i] * dd) * inv;
        }

        /* ---- geometry: recompute the forward quantities */
        real v[3];
        v[0] = c.right[0] * p[0] + c.right[1] * p[1] + c.right[2] * p[2] + c.t[0];
        v[1] = c.up[0] * p[0] + c.up[1] * p[1] + c.up[2] * p[2] + c.t[1];
        v[2] = c.front[0] * p[0] + c.front[1] * p[1] + c.front[2] * p[2] + c.t[2];
        const real limx = RC(1.3f) * c.tanx, limy = RC(1.3f) * c.tany;
        const real rx = v[0] / v[2], ry = v[1] / v[2];
        const int  clx = (rx < -limx) ? -1 : (rx > limx ? 1 : 0);
        const int  cly = (ry < -limy) ? -1 : (ry > limy ? 1 : 0);
        const real tx = (clx ? (real)clx * limx : rx) * v[2];
        const real ty = (cly ? (real)cly * limy : ry) * v[2];
        const real tz = v[2];
        const real sc[3] = { scale_modifier * scale[3 * (size_t)idx + 0], scale_modifier * scale[3 * (size_t)idx + 1],
                             scale_modifier * scale[3 * (size_t)idx + 2] };
        const real qr = rotq[4 * (size_t)idx + 0], qx = rotq[4 * (size_t)idx + 1], qy = rotq[4 * (size_t)idx + 2],
                   qz = rotq[4 * (size_t)idx + 3];
        /* R (row, col), transform.hpp:196-209 with (x,y,z,w) = (qx,qy,qz,qr) */
        const real x = qx, y = qy, z = qz, w = qr;
        real R[3][3];
        R[0][0] = RC(1.0f) - RC(2.0f) * y * y - RC(2.0f) * z * z; R[0][1] = RC(2.0f) * x * y - RC(2.0f) * z * w; R[0][2] = RC(2.0f) * x * z + RC(2.0f) * y * w;
        R[1][0] = RC(2.0f) * x * y + RC(2.0f) * z * w; R[1][1] = RC(1.0f) - RC(2.0f) * x * x - RC(2.0f) * z * z; R[1][2] = RC(2.0f) * y * z - RC(2.0f) * x * w;
        R[2][0] = RC(2.0f) * x * z - RC(2.0f) * y * w; R[2][1] = RC(2.0f) * y * z + RC(2.0f) * x * w; R[2][2] = RC(1.0f) - RC(2.0f) * x * x - RC(2.0f) * y * y;
        real M[3][3], Sig[3][3];
        for (int r = 0; r < 3; ++r)
            for (int k = 0; k < 3; ++k) M[r][k] = R[r][k] * sc[k];
        for (int r = 0; r < 3; ++r)
            for (int k = 0; k < 3; ++k) Sig[r][k] = M[r][0] * M[k][0] + M[r][1] * M[k][1] + M[r][2] * M[k][2];
        const real j00 = c.fx / tz, j11 = c.fy / tz, j02 = -c.fx * tx / (tz * tz), j12 = -c.fy * ty / (tz * tz);
        real T0[3], T1[3], ST0[3], ST1[3];
        for (int r = 0; r < 3; ++r) {
            T0[r] = c.right[r] * j00 + c.front[r] * j02;
            T1[r] = c.up[r] * j11 + c.front[r] * j12;
        }
        for (int r = 0; r < 3; ++r) {
            ST0[r] = Sig[r][0] * T0[0] + Sig[r][1] * T0[1] + Sig[r][2] * T0[2];
            ST1[r] = Sig[r][0] * T1[0] + Sig[r][1] * T1[1] + Sig[r][2] * T1[2];
        }
        const real a = b_dot3(T0, ST0) + RC(0.3f), b = b_dot3(T1, ST0), cc = b_dot3(T1, ST1) + RC(0.3f);
        const real D = a * cc - b * b + RC(1e-6f);

        /* ---- conic -> filtered cov (a, b, c) */
        const real gA = dL_dconic[3 * (size_t)idx + 0], gB = dL_dconic[3 * (size_t)idx + 1], gC = dL_dconic[3 * (size_t)idx + 2];
        const real iD2 = RC(1.0f) / (D * D);
        const real g00 = (-cc * cc * gA + b * cc * gB + (D - a * cc) * gC) * iD2;
        const real g11 = ((D - a * cc) * gA + a * b * gB - a * a * gC) * iD2;
        const real g01 = (RC(2.0f) * b * cc * gA - (D + RC(2.0f) * b * b) * gB + RC(2.0f) * a * b * gC) * iD2;

        /* ---- cov2d -> Sigma (general matrix G) and T0, T1 */
        real G[3][3], dT0[3], dT1[3];
        for (int r = 0; r < 3; ++r)
            for (int k = 0; k < 3; ++k) G[r][k] = g00 * T0[r] * T0[k] + g01 * T1[r] * T0[k] + g11 * T1[r] * T1[k];
        for (int r = 0; r < 3; ++r) {
            dT0[r] = RC(2.0f) * g00 * ST0[r] + g01 * ST1[r];
            dT1[r] = RC(2.0f) * g11 * ST1[r] + g01 * ST0[r];
        }
        const real dj00 = b_dot3(c.right, dT0), dj02 = b_dot3(c.front, dT0);
        const real dj11 = b_dot3(c.up, dT1), dj12 = b_dot3(c.front, dT1);
        const real itz2 = RC(1.0f) / (tz * tz), itz3 = itz2 / tz;
        const real dtx = dj02 * (-c.fx * itz2);
        const real dty = dj12 * (-c.fy * itz2);
        const real dtz = dj00 * (-c.fx * itz2) + dj11 * (-c.fy * itz2) + dj02 * (RC(2.0f) * c.fx * tx * itz3) +
                         dj12 * (RC(2.0f) * c.fy * ty * itz3);
        real dv[3];
        dv[0] = clx ? RC(0.0f) : dtx;
        dv[1] = cly ? RC(0.0f) : dty;
        dv[2] = dtz + (clx ? dtx * (real)clx * limx : RC(0.0f)) + (cly ? dty * (real)cly * limy : RC(0.0f));

        /* ---- pixel mean -> view-space position (module.cpp:18-20, camera.h:54-72) */
        const real gmx = dL_dmean2d[2 * (size_t)idx + 0], gmy = dL_dmean2d[2 * (size_t)idx + 1];
        const real pw = RC(1.0f) / (v[2] + RC(1e-6f));
        dv[0] += gmx * c.fx * pw;
        dv[1] += gmy * c.fy * pw;
        dv[2] += -(gmx * c.fx * v[0] + gmy * c.fy * v[1]) * pw * pw;
        for (int i = 0; i < 3; ++i) gp[i] += c.right[i] * dv[0] + c.up[i] * dv[1] + c.front[i] * dv[2];

        /* ---- Sigma = M M^T, M = R diag(scale_modifier * s) */
        real dM[3][3];
        for (int r = 0; r < 3; ++r)
            for (int k = 0; k < 3; ++k)
                dM[r][k] = (G[r][0] + G[0][r]) * M[0][k] + (G[r][1] + G[1][r]) * M[1][k] + (G[r][2] + G[2][r]) * M[2][k];
        real dR[3][3];
        for (int k = 0; k < 3; ++k) {
            gs[k] = scale_modifier * (dM[0][k] * R[0][k] + dM[1][k] * R[1][k] + dM[2][k] * R[2][k]);
            for (int r = 0; r < 3; ++r) dR[r][k] = dM[r][k] * sc[k];
        }
        /* R(q) entries are quadratic in (x,y,z,w): */
        const real gx_ = RC(2.0f) * (y * (dR[0][1] + dR[1][0]) + z * (dR[0][2] + dR[2][0]) + w * (dR[2][1] - dR[1][2])) -
                         RC(4.0f) * x * (dR[1][1] + dR[2][2]);
        const real gy_ = RC(2.0f) * (x * (dR[0][1] + dR[1][0]) + z * (dR[1][2] + dR[2][1]) + w * (dR[0][2] - dR[2][0])) -
                         RC(4.0f) * y * (dR[0][0] + dR[2][2]);
        const real gz_ = RC(2.0f) * (x * (dR[0][2] + dR[2][0]) + y * (dR[1][2] + dR[2][1]) + w * (dR[1][0] - dR[0][1])) -
                         RC(4.0f) * z * (dR[0][0] + dR[1][1]);
        const real gw_ = RC(2.0f) * (z * (dR[1][0] - dR[0][1]) + y * (dR[0][2] - dR[2][0]) + x * (dR[2][1] - dR[1][2]));
        gq[0] = gw_; /* stored order (r, x, y, z) */
        gq[1] = gx_;
        gq[2] = gy_;
        gq[3] = gz_;
    }
}

/* forward + backward for one view (forward exactly as orc_render, keeping what the backward needs) */
int64_t orc_render_backward_full(int P, int sh_deg, const real* pos, const real* scale, const real* rotq,
                                 const real* sh, const real* opacity, const orc_camera* cam, const real bg[3],
                                 real scale_modifier, const real* dL_dimg, real* img, real* dL_dpos,
                                 real* dL_dscale, real* dL_drotq, real* dL_dsh, real* dL_dopacity)
{
    const int      W = cam->width, H = cam->height;
    const uint32_t gx = ((uint32_t)W + BLOCK_X - 1u) / BLOCK_X, gy = ((uint32_t)H + BLOCK_Y - 1u) / BLOCK_Y;
    const size_t   G = (size_t)gx * gy, hw = (size_t)W * H;
    const int      feat = (sh_deg + 1) * (sh_deg + 1);
    real*     color    = (real*)calloc((size_t)P * 3 + 1, sizeof(real));
    real*     means_2d = (real*)calloc((size_t)P * 2 + 1, sizeof(real));
    real*     depth    = (real*)calloc((size_t)P + 1, sizeof(real));
    real*     covs_2d  = (real*)calloc((size_t)P * 3 + 1, sizeof(real));
    uint32_t* tiles    = (uint32_t*)calloc((size_t)P + 1, sizeof(uint32_t));
    uint32_t* offsets  = (uint32_t*)calloc((size_t)P + 1, sizeof(uint32_t));
    int32_t*  radii    = (int32_t*)calloc((size_t)P + 1, sizeof(int32_t));
    uint32_t* ranges   = (uint32_t*)calloc(G * 2, sizeof(uint32_t));
    real*     final_T  = (real*)calloc(hw, sizeof(real));
    uint32_t* n_contrib = (uint32_t*)calloc(hw, sizeof(uint32_t));
    real*     g_mean   = (real*)calloc((size_t)P * 2 + 1, sizeof(real));
    real*     g_conic  = (real*)calloc((size_t)P * 3 + 1, sizeof(real));
    real*     g_color  = (real*)calloc((size_t)P * 3 + 1, sizeof(real));
    uint64_t *ku = NULL, *ks = NULL;
    uint32_t *lu = NULL, *ls = NULL;
    int64_t   L = 0;
    memset(dL_dopacity, 0, (size_t)P * sizeof(real));

    orc_sh_process(P, 3, sh_deg, cam->position, pos, sh, color, NULL);
    orc_project_gs(P, pos, scale, rotq, scale_modifier, means_2d, depth, covs_2d, cam, 1);
    orc_allocate_tiles(P, W, H, depth, means_2d, covs_2d, tiles, radii, 1);
    orc_inclusive_sum(P, tiles, offsets);
    L = P > 0 ? (int64_t)(int32_t)offsets[P - 1] : 0;
    if (L > 0) {
        ku = (uint64_t*)calloc((size_t)L, sizeof(uint64_t));
        ks = (uint64_t*)calloc((size_t)L, sizeof(uint64_t));
        lu = (uint32_t*)calloc((size_t)L, sizeof(uint32_t));
        ls = (uint32_t*)calloc((size_t)L, sizeof(uint32_t));
        orc_copy_with_keys(P, W, H, means_2d, offsets, radii, depth, ku, lu);
        orc_sort_pairs(L, ku, lu, ks, ls);
        orc_get_ranges(L, ks, ranges);
        orc_render_forward(W, H, bg, ranges, ls, means_2d, covs_2d, opacity, color, img, final_T, n_contrib, NULL,
                           RC(0.0f));
        orc_render_backward(W, H, bg, ranges, ls, means_2d, covs_2d, opacity, color, final_T, n_contrib, dL_dimg,
                            g_mean, g_conic, dL_dopacity, g_color);
    }
    orc_preprocess_backward(P, sh_deg, pos, scale, rotq, sh, cam, scale_modifier, radii, g_mean, g_conic, g_color,
                            dL_dpos, dL_dscale, dL_drotq, dL_dsh);
    (void)feat;
    free(color); free(means_2d); free(depth); free(covs_2d); free(tiles); free(offsets); free(radii); free(ranges);
    free(final_T); free(n_contrib); free(g_mean); free(g_conic); free(g_color);
    free(ku); free(ks); free(lu); free(ls);
    return L;
}
