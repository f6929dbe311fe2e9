// ingest.hip -- the step in front of the hot path: vertex records of a 3DGS PLY, as they lie in
// the file, to the five activated AoS arrays the operators consume.  What read_gs_ply does on the host with 62
// column copies and scalar loops (app/gaussians.cpp:93-168):
//   pos      <- x y z                                            (gaussians.cpp:93-103)
//   feature[j*48 + k*3 + c]: f_dc_c -> k = 0; f_rest_i -> c = i / 15, k = i % 15 + 1   (gaussians.cpp:106-135)
//   opacity  <- sigmoid(raw) = 1 / (1 + exp(-raw))               (gaussians.cpp:15-19,140)
//   scale    <- exp(raw)                                         (gaussians.cpp:21-25,150)
//   rotq     <- (r,x,y,z) / sqrt(x x + y y + z z + r r)          (gaussians.cpp:27-35,154-168)
// One wave per record: lane w < 59 owns wanted column w (in the order pos3 dc3 rest45 opacity scale3 rot4), reads
// its 4 bytes (the wave covers the record's ~248 contiguous bytes), activates and writes.  Same operations in the
// same order as the host path (lcgs_ply_read); only exp() differs (device libm vs host libm, <= 2 ulp).
#include "launch.hpp"

namespace lcgs
{
namespace
{

__global__ void __launch_bounds__(256) k_ply_activate(const unsigned char* __restrict__ raw, int64_t first, int64_t count,
                                                      uint32_t stride, PlyColumns cols, float* __restrict__ pos,
                                                      float* __restrict__ scale, float* __restrict__ rotq,
                                                      float* __restrict__ sh, float* __restrict__ opacity)
{
    const int     lane = threadIdx.x & 63;
    const int64_t r    = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); // record inside this chunk
    if (r >= count) return;
    const int64_t j = first + r; // splat index
    float         v = 0.0f;
    if (lane < 59) v = *reinterpret_cast<const float*>(raw + (size_t)r * stride + cols.offset[lane]);
    // the four quaternion components, for the norm (host order: x x + y y + z z + r r)
    const float qr = __shfl(v, 55, 64), qx = __shfl(v, 56, 64), qy = __shfl(v, 57, 64), qz = __shfl(v, 58, 64);
    if (lane < 3) {
        pos[3 * j + lane] = v;
    } else if (lane < 6) {
        sh[j * 48 + (lane - 3)] = v;
    } else if (lane < 51) {
        const int i = lane - 6, channel = i / 15, k = i % 15 + 1;
        sh[j * 48 + k * 3 + channel] = v;
    } else if (lane == 51) {
        opacity[j] = 1.0f / (1.0f + expf(-v));
    } else if (lane < 55) {
        scale[3 * j + (lane - 52)] = expf(v);
    } else if (lane < 59) {
        const float norm = sqrtf(qx * qx + qy * qy + qz * qz + qr * qr);
        rotq[4 * j + (lane - 55)] = v / norm;
    }
}

} // namespace

void launch_ply_activate(const unsigned char* raw, int64_t first, int64_t count, uint32_t stride, const PlyColumns& cols,
                         float* pos, float* scale, float* rotq, float* sh, float* opacity, hipStream_t stream)
{
    if (count <= 0) return;
    hipLaunchKernelGGL(k_ply_activate, dim3((unsigned)((count + 3) / 4)), dim3(256), 0, stream, raw, first, count, stride,
                       cols, pos, scale, rotq, sh, opacity);
}

} // namespace lcgs
