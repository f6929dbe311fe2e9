/* Sanitizer driver for the CPU oracle (tests/test_oracle_sanitizers.py builds it together with oracle/*.c under
 * -fsanitize=address,undefined): one forward and one forward+backward of a small pseudo-random scene, a few odd
 * resolutions, including splats that are culled, off screen, huge and degenerate.  Exit code 0 and a silent stderr
 * mean no out-of-bounds access, no use of uninitialised heap, no signed overflow / bad shift / bad float->int cast. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../oracle/lcgs_oracle.h"

static uint64_t s = 0x9E3779B97F4A7C15ull;
static double rnd(void)
{
    s ^= s << 13;
    s ^= s >> 7;
    s ^= s << 17;
    return (double)(s >> 11) / 9007199254740992.0;
}
static double gauss(void) { return (rnd() + rnd() + rnd() + rnd() - 2.0) * 1.7; }

int main(void)
{
    const int P = 1500;
    real *pos = calloc(3 * P, sizeof(real)), *scale = calloc(3 * P, sizeof(real)), *rot = calloc(4 * P, sizeof(real));
    real *sh = calloc(48 * P, sizeof(real)), *op = calloc(P, sizeof(real));
    for (int i = 0; i < P; ++i) {
        for (int c = 0; c < 3; ++c) {
            pos[3 * i + c]   = (real)(0.6 * gauss() + (c == 2 ? 0.5 : 0.0));
            scale[3 * i + c] = (real)(0.02 + 0.05 * rnd());
        }
        real n = 0;
        for (int c = 0; c < 4; ++c) {
            rot[4 * i + c] = (real)gauss();
            n += rot[4 * i + c] * rot[4 * i + c];
        }
        (void)n;
        for (int c = 0; c < 48; ++c) sh[48 * i + c] = (real)(0.3 * gauss() + (c < 3 ? 0.5 : 0.0));
        op[i] = (real)rnd();
    }
    /* the awkward ones */
    scale[0] = scale[1] = scale[2] = 40;                 /* covers everything */
    pos[3] = 100; pos[4] = 100; pos[5] = -100;           /* behind the camera */
    scale[6] = scale[7] = scale[8] = 0;                  /* zero covariance */
    rot[12] = rot[13] = rot[14] = rot[15] = 0;           /* zero quaternion */
    op[4] = 0; op[5] = -1; op[6] = 5;
    const int sizes[3][2] = { { 64, 48 }, { 100, 71 }, { 17, 33 } };
    const real eye[3] = { -3, (real)-0.5, (real)2.3 }, tgt[3] = { 0, 0, (real)0.5 }, up[3] = { 0, 0, 1 }, bg[3] = { (real)0.1, (real)0.2, (real)0.3 };
    for (int k = 0; k < 3; ++k) {
        const int  W = sizes[k][0], H = sizes[k][1];
        orc_camera cam;
        orc_get_lookat_cam(eye, tgt, up, &cam);
        cam.width = W; cam.height = H; cam.aspect_ratio = (real)W / (real)H;
        real*     img = calloc(3 * (size_t)W * H, sizeof(real)), *fT = calloc((size_t)W * H, sizeof(real));
        int32_t*  radii = calloc(P, sizeof(int32_t));
        uint32_t* nc = calloc((size_t)W * H, sizeof(uint32_t));
        uint8_t*  amb = calloc((size_t)W * H, 1);
        const int64_t L = orc_render(P, 3, pos, scale, rot, sh, op, &cam, bg, 1, img, radii, fT, nc, amb, (real)1e-5);
        if (L < 0) return 2;
        real *dL = calloc(3 * (size_t)W * H, sizeof(real)), *gp = calloc(3 * P, sizeof(real)), *gs = calloc(3 * P, sizeof(real));
        real *gr = calloc(4 * P, sizeof(real)), *gsh = calloc(48 * P, sizeof(real)), *go = calloc(P, sizeof(real));
        for (size_t i = 0; i < 3 * (size_t)W * H; ++i) dL[i] = (real)gauss();
        const int64_t L2 = orc_render_backward_full(P, 3, pos, scale, rot, sh, op, &cam, bg, 1, dL, img, gp, gs, gr, gsh, go);
        if (L2 != L) return 3;
        printf("%dx%d num_rendered %lld\n", W, H, (long long)L);
        free(img); free(fT); free(radii); free(nc); free(amb); free(dL); free(gp); free(gs); free(gr); free(gsh); free(go);
    }
    free(pos); free(scale); free(rot); free(sh); free(op);
    return 0;
}
