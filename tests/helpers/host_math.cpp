// tests/helpers/host_math.cpp -- TEST HELPER (not part of the product library).
// Runs the product's per-splat math header (csrc/kernels/gs_math.hpp, __host__ __device__) on the HOST so
// that its operation order can be compared bit-for-bit with the CPU oracle in a container without a GPU.
// The same functions compiled for gfx950 are what the kernels execute.
#include <stdint.h>
#include <string.h>

#include "../../luisacomputegaussiansplatting_amd/csrc/host/camera.cpp"

using namespace lcgs;

extern "C" __attribute__((visibility("default"))) void hm_preprocess(
    int P, const lcgs_camera* cam, int use_focal, float scale_modifier, int deg, const float* pos, const float* scale,
    const float* rotq, const float* sh, float* color, float* means_ndc, float* depth, float* cov, float* means_pix,
    float* conic, int32_t* radii, uint32_t* tiles, uint32_t* rects)
{
    CamParams cp = make_cam_params(*cam);
    const int feat = (deg + 1) * (deg + 1) * 3;
    for (int i = 0; i < P; ++i) {
        float raw[3];
        const float* s = sh + (size_t)i * feat;
        sh_to_color(deg, cp.campos, pos[3 * i], pos[3 * i + 1], pos[3 * i + 2], [&](int k, int c) { return s[k * 3 + c]; }, raw);
        for (int c = 0; c < 3; ++c) color[3 * i + c] = clamp_(raw[c], 0.0f, 1.0f);
        float v[3], ndc[2];
        view_transform(cp, pos[3 * i], pos[3 * i + 1], pos[3 * i + 2], v);
        ndc_from_view(cp, v, ndc);
        radii[i] = 0;
        tiles[i] = 0;
        if (v[2] < 0.2f) continue;
        depth[i]         = v[2];
        means_ndc[2 * i] = ndc[0];
        means_ndc[2 * i + 1] = ndc[1];
        float sc[3] = { scale_modifier * scale[3 * i], scale_modifier * scale[3 * i + 1], scale_modifier * scale[3 * i + 2] };
        float Sig[3][3], t[3], c2[3];
        cov3d_from_scale_rot(sc, rotq[4 * i + 1], rotq[4 * i + 2], rotq[4 * i + 3], rotq[4 * i + 0], Sig);
        cam_clamp(cp, v, t);
        ewa_cov2d(cp, Sig, t, use_focal != 0, c2);
        for (int c = 0; c < 3; ++c) cov[3 * i + c] = c2[c];
        float   con[3];
        int32_t radius;
        conic_and_radius(c2[0], c2[1], c2[2], use_focal != 0, cp.width, cp.height, con, radius);
        for (int c = 0; c < 3; ++c) conic[3 * i + c] = con[c];
        float px = ndc2pix(ndc[0], cp.width), py = ndc2pix(ndc[1], cp.height);
        means_pix[2 * i] = px;
        means_pix[2 * i + 1] = py;
        uint32_t rmin[2], rmax[2];
        get_rect(px, py, radius, cp.grid_x, cp.grid_y, rmin, rmax);
        radii[i] = radius;
        tiles[i] = (rmax[0] - rmin[0]) * (rmax[1] - rmin[1]);
        rects[4 * i] = rmin[0]; rects[4 * i + 1] = rmin[1]; rects[4 * i + 2] = rmax[0]; rects[4 * i + 3] = rmax[1];
    }
}

// the blend's exp as the header defines it (host path of the very function the kernels run)
extern "C" void hm_blend_exp(long long n, const float* x, float* out)
{
    for (long long i = 0; i < n; ++i) out[i] = blend_exp(x[i]);
}
