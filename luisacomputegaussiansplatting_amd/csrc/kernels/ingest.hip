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

// ---------------------------------------------------------------------------------------------------------------
// Spatial order of a scene (lcgs_scene_reorder_spatial).  The frame's per-splat streams -- the 192-byte SH rows the
// record builder gathers, the gradient rows of the backward, the parameter / moment rows of the on-screen-only
// optimiser -- are indexed by splat, and only the splats on screen (39 % on the bicycle stand-in) are touched.  With
// splats in file order those rows are scattered: every DRAM page is opened for a fraction of its bytes.
// Sorted along a Morton curve, the splats of a view come in long runs of consecutive rows.  Three steps: position
// moments (box = mean +- 4 sigma per axis, so that a few far outliers do not flatten the grid), 10-bit-per-axis
// Morton keys, stable sort (pair_sort.hip) -- equal keys, hence coincident splats, keep their file order -- and a
// row gather of the five arrays through the permutation.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kMomentBlocks = 256;

__global__ void __launch_bounds__(256) k_pos_moments(int64_t P, const float* __restrict__ pos, double* __restrict__ partial)
{
    __shared__ double s_acc[4][7];
    double acc[7] = { 0, 0, 0, 0, 0, 0, 0 }; // sum x y z, sum x^2 y^2 z^2, count of finite positions
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < P; i += (int64_t)gridDim.x * 256) {
        const float x = pos[3 * i], y = pos[3 * i + 1], z = pos[3 * i + 2];
        if (!(fabsf(x) < 1e30f && fabsf(y) < 1e30f && fabsf(z) < 1e30f)) continue; // NaN / inf / absurd: not in the box
        acc[0] += x; acc[1] += y; acc[2] += z;
        acc[3] += (double)x * x; acc[4] += (double)y * y; acc[5] += (double)z * z;
        acc[6] += 1.0;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc[k] += __shfl_xor(acc[k], off, 64);
        if (lane == 0) s_acc[wave][k] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < 7) partial[blockIdx.x * 7 + threadIdx.x] = s_acc[0][threadIdx.x] + s_acc[1][threadIdx.x] +
                                                                s_acc[2][threadIdx.x] + s_acc[3][threadIdx.x];
}

__device__ __forceinline__ uint32_t spread3(uint32_t v) // 10 bits -> every third bit
{
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ void __launch_bounds__(256) k_morton_keys(int64_t P, const float* __restrict__ pos, float lox, float loy, float loz,
                                                     float sx, float sy, float sz, uint32_t* __restrict__ keys,
                                                     uint32_t* __restrict__ vals)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    // (NaN compares false everywhere: it lands in cell 0)
    auto cell = [](float p, float lo, float s) {
        const float q = (p - lo) * s;
        return q > 0.0f ? (q < 1023.0f ? (uint32_t)q : 1023u) : 0u;
    };
    keys[i] = spread3(cell(pos[3 * i], lox, sx)) | (spread3(cell(pos[3 * i + 1], loy, sy)) << 1) |
              (spread3(cell(pos[3 * i + 2], loz, sz)) << 2);
    vals[i] = (uint32_t)i;
}

// dst row r = src row perm[r]; rows of `row` floats
__global__ void __launch_bounds__(256) k_gather_rows(int64_t total, int row, const uint32_t* __restrict__ perm,
                                                     const float* __restrict__ src, float* __restrict__ dst)
{
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / row;
        const int     c = (int)(e - r * row);
        dst[e]          = src[(int64_t)perm[r] * row + c];
    }
}
// 16-byte rows / row parts (quaternions; degree-3 SH rows as 12 parts)
__global__ void __launch_bounds__(256) k_gather_rows16(int64_t total, int parts, const uint32_t* __restrict__ perm,
                                                       const float4* __restrict__ src, float4* __restrict__ dst)
{
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / parts;
        const int     c = (int)(e - r * parts);
        dst[e]          = src[(int64_t)perm[r] * parts + c];
    }
}

} // namespace

int    pos_moment_blocks() { return kMomentBlocks; }
void   launch_pos_moments(int64_t P, const float* pos, double* partial, hipStream_t stream)
{
    hipLaunchKernelGGL(k_pos_moments, dim3(kMomentBlocks), dim3(256), 0, stream, P, pos, partial);
}
void launch_morton_keys(int64_t P, const float* pos, const float lo[3], const float cells_per_unit[3], uint32_t* keys,
                        uint32_t* vals, hipStream_t stream)
{
    if (P <= 0) return;
    hipLaunchKernelGGL(k_morton_keys, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, stream, P, pos, lo[0], lo[1], lo[2],
                       cells_per_unit[0], cells_per_unit[1], cells_per_unit[2], keys, vals);
}
void launch_gather_rows(int64_t rows, int row_floats, const uint32_t* perm, const float* src, float* dst, hipStream_t stream)
{
    if (rows <= 0 || row_floats <= 0) return;
    const bool wide = (row_floats % 4 == 0) && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
    const int64_t total = wide ? rows * (row_floats / 4) : rows * row_floats;
    int64_t       blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    if (wide)
        hipLaunchKernelGGL(k_gather_rows16, dim3((unsigned)blocks), dim3(256), 0, stream, total, row_floats / 4, perm,
                           reinterpret_cast<const float4*>(src), reinterpret_cast<float4*>(dst));
    else
        hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)blocks), dim3(256), 0, stream, total, row_floats, perm, src, dst);
}

void launch_ply_activate(const unsigned char* raw, int64_t first, int64_t count, uint32_t stride, const PlyColumns& cols,
                         float* pos, float* scale, float* rotq, float* sh, float* opacity, hipStream_t stream)
{
    if (count <= 0) return;
    hipLaunchKernelGGL(k_ply_activate, dim3((unsigned)((count + 3) / 4)), dim3(256), 0, stream, raw, first, count, stride,
                       cols, pos, scale, rotq, sh, opacity);
}

} // namespace lcgs
