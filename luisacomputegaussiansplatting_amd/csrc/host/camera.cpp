// camera.cpp -- host camera model of the reference (lcgs/include/lcgs/util/camera.h) and the
// per-frame kernel constants derived from it (lcgs/src/gs_projector/impl.cpp:34-42).
#include <math.h>
#include <string.h>

#include "../common.hpp"

namespace
{
inline float dot3(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline void  cross3(const float a[3], const float b[3], float o[3])
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
inline void normalize3(const float v[3], float o[3])
{
    float inv = 1.0f / sqrtf(dot3(v, v));
    o[0]      = v[0] * inv;
    o[1]      = v[1] * inv;
    o[2]      = v[2] * inv;
}
} // namespace

extern "C" {

// get_lookat_cam, camera.h:74-82
void lcgs_get_lookat_cam(const float pos[3], const float target[3], const float world_up[3], lcgs_camera* cam)
{
    float d[3] = { target[0] - pos[0], target[1] - pos[1], target[2] - pos[2] };
    float c[3];
    memcpy(cam->position, pos, 3 * sizeof(float));
    normalize3(d, cam->front);
    cross3(cam->front, world_up, c);
    normalize3(c, cam->right);
    cross3(cam->right, cam->front, c);
    normalize3(c, cam->up);
    cam->fov          = 60.0f; // camera.h:21-24
    cam->aspect_ratio = 1.0f;
    cam->width        = 512;
    cam->height       = 512;
}

// local_to_world_matrix, camera.h:27-36
void lcgs_local_to_world_matrix(const lcgs_camera* cam, float m[16])
{
    for (int r = 0; r < 3; ++r) {
        m[0 * 4 + r] = cam->right[r];
        m[1 * 4 + r] = cam->up[r];
        m[2 * 4 + r] = cam->front[r];
        m[3 * 4 + r] = cam->position[r];
    }
    m[0 * 4 + 3] = m[1 * 4 + 3] = m[2 * 4 + 3] = 0.0f;
    m[3 * 4 + 3]                               = 1.0f;
}

// world_to_local_matrix, camera.h:38-51
void lcgs_world_to_local_matrix(const lcgs_camera* cam, float m[16])
{
    float tx = -dot3(cam->position, cam->right);
    float ty = -dot3(cam->position, cam->up);
    float tz = -dot3(cam->position, cam->front);
    for (int c = 0; c < 3; ++c) {
        m[c * 4 + 0] = cam->right[c];
        m[c * 4 + 1] = cam->up[c];
        m[c * 4 + 2] = cam->front[c];
        m[c * 4 + 3] = 0.0f;
    }
    m[3 * 4 + 0] = tx;
    m[3 * 4 + 1] = ty;
    m[3 * 4 + 2] = tz;
    m[3 * 4 + 3] = 1.0f;
}

// projection_matrix, camera.h:54-72
void lcgs_projection_matrix(float tanfovx, float tanfovy, float znear, float zfar, float m[16])
{
    float zsign   = 1.0f;
    float fx      = 1.0f / tanfovx;
    float fy      = 1.0f / tanfovy;
    float z_range = zfar - znear;
    float a       = zfar / z_range;
    float b       = -zfar * znear / z_range;
    memset(m, 0, 16 * sizeof(float));
    m[0 * 4 + 0] = fx;
    m[1 * 4 + 1] = fy;
    m[2 * 4 + 2] = a * zsign;
    m[2 * 4 + 3] = zsign;
    m[3 * 4 + 2] = b;
}

} // extern "C"

namespace lcgs
{

// gs_projector/impl.cpp:34-42 (fov -> tan, focal) + the entries of the two matrices the kernels use.
CamParams make_cam_params(const lcgs_camera& cam)
{
    CamParams cp;
    float     fovy    = cam.fov / 180.0f * 3.1415926536f;
    float     tanfovy = tanf(fovy * 0.5f);
    float     tanfovx = tanfovy * cam.aspect_ratio;
    float     view[16], proj[16];
    lcgs_world_to_local_matrix(&cam, view);
    lcgs_projection_matrix(tanfovx, tanfovy, 0.1f, 100.0f, proj);
    for (int i = 0; i < 3; ++i) {
        cp.campos[i] = cam.position[i];
        cp.right[i]  = view[i * 4 + 0];
        cp.up[i]     = view[i * 4 + 1];
        cp.front[i]  = view[i * 4 + 2];
    }
    cp.tx       = view[3 * 4 + 0];
    cp.ty       = view[3 * 4 + 1];
    cp.tz       = view[3 * 4 + 2];
    cp.inv_tanx = proj[0 * 4 + 0];
    cp.inv_tany = proj[1 * 4 + 1];
    cp.tanfovx  = tanfovx;
    cp.tanfovy  = tanfovy;
    cp.focalx   = (float)cam.width / (2.0f * tanfovx);
    cp.focaly   = (float)cam.height / (2.0f * tanfovy);
    cp.width    = (uint32_t)cam.width;
    cp.height   = (uint32_t)cam.height;
    cp.grid_x   = (cp.width + kBlockX - 1u) / kBlockX; // gs_tile_splatter/impl.cpp:76-79
    cp.grid_y   = (cp.height + kBlockY - 1u) / kBlockY;
    cp.lod_min_radius = 0;
    cp.list_shift     = 0;
    return cp;
}

} // namespace lcgs
