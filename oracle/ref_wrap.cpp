// oracle/ref_wrap.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Builds the two pieces of the reference that compile from their own sources with no external
// dependency, straight from where they lie under /root/reference (nothing is copied):
//   * lcgs/include/lcgs/util/sh.hpp   (pure templates, no #include) -- the SH band evaluators
//   * app/happly.h                    (std-only, vendored MIT)      -- the PLY parser the app uses
// Output: oracle/_ref/liblcgs_ref.so (git-ignored).  It exists only in the authoring container; it
// is used to (a) validate the CPU restatement in oracle/lcgs_oracle.c and (b) generate the golden
// vectors committed under tests/golden/ (tests/golden/make_golden.py).
//
// The rest of the reference's hot path (gaussian.hpp, transform.hpp, camera.h, every shader)
// includes LuisaCompute headers that are absent here, so it is NOT built (no stand-in headers).
//
// sh.hpp's functions are templates over the float3 type; F3 below is the template argument, not a
// replacement for a missing header.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "lcgs/util/sh.hpp" // -I/root/reference/lcgs/include
#include "happly.h"         // -I/root/reference/app

namespace
{
struct F3 {
    float x, y, z;
};
inline F3 operator*(F3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
inline F3 operator*(float s, F3 a) { return { s * a.x, s * a.y, s * a.z }; }
inline F3 operator+(F3 a, F3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline F3 operator-(F3 a, F3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline F3 ld(const float* p) { return { p[0], p[1], p[2] }; }
} // namespace

extern "C" {

// The per-band sums exactly as lcgs/src/sh_preprocessor.cpp:49-147 composes them:
// result = level0; result = result + level1; result = result + level2; result = result + level3.
void ref_sh_eval_dir(int deg, const float dir[3], const float* shs, float out[3])
{
    F3 d      = ld(dir);
    F3 result = lcgs::compute_color_from_sh_level_0(ld(shs));
    if (deg > 0) {
        result = result + lcgs::compute_color_from_sh_level_1(d, ld(shs + 3), ld(shs + 6), ld(shs + 9));
        if (deg > 1) {
            result = result + lcgs::compute_color_from_sh_level_2(d, ld(shs + 12), ld(shs + 15), ld(shs + 18),
                                                                  ld(shs + 21), ld(shs + 24));
            if (deg > 2) {
                result = result + lcgs::compute_color_from_sh_level_3(d, ld(shs + 27), ld(shs + 30), ld(shs + 33),
                                                                      ld(shs + 36), ld(shs + 39), ld(shs + 42),
                                                                      ld(shs + 45));
            }
        }
    }
    out[0] = result.x;
    out[1] = result.y;
    out[2] = result.z;
}

// dL/dSH per band from the reference's (unused) backward helpers, sh.hpp:37-40,53-65,87-117,141-165.
// out = 16 x 3.  (dL_d_dir is a TODO in the reference and is not produced.)
void ref_sh_backward_coeffs(int deg, const float dir[3], const float dL_dcolor[3], float* out)
{
    F3 d = ld(dir), g = ld(dL_dcolor), o[16], dd{ 0, 0, 0 };
    for (auto& v : o) v = { 0, 0, 0 };
    lcgs::compute_color_from_sh_level_0_backward(g, o[0]);
    if (deg > 0) lcgs::compute_color_from_sh_level_1_backward(g, d, o[1], o[2], o[3], dd);
    if (deg > 1) lcgs::compute_color_from_sh_level_2_backward(g, d, o[4], o[5], o[6], o[7], o[8], dd);
    if (deg > 2) lcgs::compute_color_from_sh_level_3_backward(g, d, o[9], o[10], o[11], o[12], o[13], o[14], o[15], dd);
    for (int k = 0; k < 16; ++k) {
        out[3 * k + 0] = o[k].x;
        out[3 * k + 1] = o[k].y;
        out[3 * k + 2] = o[k].z;
    }
}

// Vertex count of a PLY as happly sees it (app/gaussians.cpp:93), -1 on error.
int64_t ref_ply_vertex_count(const char* path)
{
    try {
        happly::PLYData plyIn(path);
        if (!plyIn.hasElement("vertex")) return -1;
        return (int64_t)plyIn.getElement("vertex").count;
    } catch (...) {
        return -1;
    }
}

// One float column by property name, as app/gaussians.cpp:96-160 reads them.  Returns 0 on success.
int ref_ply_read_column(const char* path, const char* name, float* out, int64_t n)
{
    try {
        happly::PLYData    plyIn(path);
        std::vector<float> v = plyIn.getElement("vertex").getProperty<float>(name);
        if ((int64_t)v.size() != n) return 2;
        std::memcpy(out, v.data(), (size_t)n * sizeof(float));
        return 0;
    } catch (...) {
        return 1;
    }
}

} // extern "C"
