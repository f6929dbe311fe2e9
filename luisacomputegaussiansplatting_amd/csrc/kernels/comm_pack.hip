// comm_pack.hip -- the opt-in f16 transport of the gradient all-reduce (host/comm.cpp): per-attribute |max|, f32 -> f16
// with a power-of-two scale, and back.  The sum over N ranks then crosses xGMI as 2 bytes per value instead of 4.
// f16 keeps 11 significant bits: every value is rounded once on the way in (2^-11 relative) and the reduction rounds
// again per hop, so the sum is good to about sqrt(N) x 5e-4 relative in the norm -- at the 1e-3 gradient bar for a node of
// eight, which is why this is never the default.  Power-of-two scales are exact, and shared by all ranks (the |max| is
// all-reduced first), so the only error is the rounding itself; zeros stay exact zeros.
#include <hip/hip_fp16.h>

#include "launch.hpp"

namespace lcgs
{
namespace
{

// Vector width by alignment: the caller's gradient arrays may start anywhere on a 4-byte boundary (arrays carved from one
// flat buffer), so every kernel comes in 16 / 8 / 4-byte flavours; the staging side is padded to 16 bytes per attribute.
template <int VEC>
struct FloatVec;
template <>
struct FloatVec<4> { using type = float4; };
template <>
struct FloatVec<2> { using type = float2; };
template <>
struct FloatVec<1> { using type = float; };

template <int VEC>
__device__ __forceinline__ void unpack_vec(const typename FloatVec<VEC>::type& v, float (&f)[VEC]);
template <>
__device__ __forceinline__ void unpack_vec<4>(const float4& v, float (&f)[4]) { f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w; }
template <>
__device__ __forceinline__ void unpack_vec<2>(const float2& v, float (&f)[2]) { f[0] = v.x; f[1] = v.y; }
template <>
__device__ __forceinline__ void unpack_vec<1>(const float& v, float (&f)[1]) { f[0] = v; }

// |max| of n floats into out[0] (non-negative floats order like their bit patterns: one atomicMax per workgroup)
template <int VEC>
__global__ void __launch_bounds__(256) k_absmax(const float* __restrict__ x, size_t n, uint32_t* __restrict__ out)
{
    using V = typename FloatVec<VEC>::type;
    __shared__ float s_w[4];
    float            m  = 0.0f;
    const size_t     nv = n / VEC;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (size_t)gridDim.x * 256) {
        float f[VEC];
        unpack_vec<VEC>(reinterpret_cast<const V*>(x)[i], f);
#pragma unroll
        for (int e = 0; e < VEC; ++e) m = fmaxf(m, fabsf(f[e]));
    }
    if (blockIdx.x == 0 && threadIdx.x < n - nv * VEC) m = fmaxf(m, fabsf(x[nv * VEC + threadIdx.x])); // (fewer than VEC)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(s_w[0], s_w[1]), fmaxf(s_w[2], s_w[3]));
        atomicMax(out, __float_as_uint(m)); // (fmaxf drops NaNs: a NaN gradient does not poison the scale, it travels as NaN)
    }
}

// scale[i] = the power of two that maps the largest magnitude any rank holds to at most 16384 / world (the sum of `world`
// values then stays below f16's 65504 with a factor of 4 to spare); inv[i] = 1 / scale[i]
__global__ void k_scales(const float* __restrict__ amax, int world, float* __restrict__ scale, float* __restrict__ inv)
{
    const int i = threadIdx.x;
    if (i >= 5) return;
    const float m = amax[i];
    float       s = 1.0f;
    if (m > 0.0f && m < 3.0e38f) s = exp2f(floorf(log2f(16384.0f / ((float)world * m))));
    if (!(s > 0.0f) || s > 1.0e30f) s = 1.0f; // (degenerate magnitudes: no scaling)
    scale[i] = s;
    inv[i]   = 1.0f / s;
}

template <int VEC>
__global__ void __launch_bounds__(256) k_pack_f16(const float* __restrict__ x, size_t n, const float* __restrict__ scale,
                                                  __half* __restrict__ out)
{
    using V = typename FloatVec<VEC>::type;
    const float  s  = *scale;
    const size_t nv = n / VEC;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (size_t)gridDim.x * 256) {
        float f[VEC];
        unpack_vec<VEC>(reinterpret_cast<const V*>(x)[i], f);
        __half h[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) h[e] = __float2half_rn(f[e] * s);
        if (VEC == 4) reinterpret_cast<uint2*>(out)[i] = *reinterpret_cast<const uint2*>(h);
        else if (VEC == 2) reinterpret_cast<uint32_t*>(out)[i] = *reinterpret_cast<const uint32_t*>(h);
        else out[i] = h[0];
    }
    if (blockIdx.x == 0 && threadIdx.x < n - nv * VEC) out[nv * VEC + threadIdx.x] = __float2half_rn(x[nv * VEC + threadIdx.x] * s);
}

template <int VEC>
__global__ void __launch_bounds__(256) k_unpack_f16(const __half* __restrict__ in, size_t n, const float* __restrict__ inv,
                                                    float* __restrict__ x)
{
    const float  s  = *inv;
    const size_t nv = n / VEC;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (size_t)gridDim.x * 256) {
        __half h[VEC];
        if (VEC == 4) *reinterpret_cast<uint2*>(h) = reinterpret_cast<const uint2*>(in)[i];
        else if (VEC == 2) *reinterpret_cast<uint32_t*>(h) = reinterpret_cast<const uint32_t*>(in)[i];
        else h[0] = in[i];
        float f[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) f[e] = __half2float(h[e]) * s;
        if (VEC == 4) reinterpret_cast<float4*>(x)[i] = make_float4(f[0], f[1], f[2], f[3]);
        else if (VEC == 2) reinterpret_cast<float2*>(x)[i] = make_float2(f[0], f[1]);
        else x[i] = f[0];
    }
    if (blockIdx.x == 0 && threadIdx.x < n - nv * VEC) x[nv * VEC + threadIdx.x] = __half2float(in[nv * VEC + threadIdx.x]) * s;
}

inline int vec_for(const void* p)
{
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    return (a & 15) == 0 ? 4 : ((a & 7) == 0 ? 2 : 1);
}

unsigned grid_for(size_t n)
{
    size_t b = (n + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

} // namespace

#define LCGS_VEC_DISPATCH(VECN, KERNEL, N, ...)                                                                              \
    do {                                                                                                                     \
        const dim3 g(grid_for(((N) + (VECN) - 1) / (VECN)));                                                                 \
        if ((VECN) == 4) hipLaunchKernelGGL((KERNEL<4>), g, dim3(256), 0, stream, __VA_ARGS__);                              \
        else if ((VECN) == 2) hipLaunchKernelGGL((KERNEL<2>), g, dim3(256), 0, stream, __VA_ARGS__);                         \
        else hipLaunchKernelGGL((KERNEL<1>), g, dim3(256), 0, stream, __VA_ARGS__);                                          \
    } while (0)

void launch_absmax(const float* x, size_t n, uint32_t* d_out_bits, hipStream_t stream)
{
    const int v = vec_for(x);
    LCGS_VEC_DISPATCH(v, k_absmax, n, x, n, d_out_bits);
}

void launch_transport_scales(const float* d_absmax5, int world, float* d_scale5, float* d_inv5, hipStream_t stream)
{
    hipLaunchKernelGGL(k_scales, dim3(1), dim3(64), 0, stream, d_absmax5, world, d_scale5, d_inv5);
}

// `out` / `in` (the staging side) must be 16-byte aligned: host/comm.cpp pads every attribute's region
void launch_pack_f16(const float* x, size_t n, const float* d_scale, uint16_t* out, hipStream_t stream)
{
    const int v = vec_for(x);
    LCGS_VEC_DISPATCH(v, k_pack_f16, n, x, n, d_scale, reinterpret_cast<__half*>(out));
}

void launch_unpack_f16(const uint16_t* in, size_t n, const float* d_inv, float* x, hipStream_t stream)
{
    const int v = vec_for(x);
    LCGS_VEC_DISPATCH(v, k_unpack_f16, n, reinterpret_cast<const __half*>(in), n, d_inv, x);
}


// dst[i] = sum over the n source arrays of src[r][i], r ascending (the in-process loopback transport's reductions: N contexts on
// one device stand in for N ranks, host/comm.cpp)
struct SumSources {
    const float* p[16];
};
namespace
{
__global__ void __launch_bounds__(256) k_sum_sources(SumSources src, int n, size_t count, float* dst)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
        float acc = src.p[0][i];
        for (int r = 1; r < n; ++r) acc += src.p[r][i];
        dst[i] = acc;
    }
}
} // namespace
void launch_sum_sources(const float* const* srcs, int n, size_t count, float* dst, hipStream_t stream)
{
    if (count == 0 || n <= 0 || n > 16) return;
    SumSources s{};
    for (int r = 0; r < n; ++r) s.p[r] = srcs[r];
    size_t blocks = (count + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(k_sum_sources, dim3((unsigned)blocks), dim3(256), 0, stream, s, n, count, dst);
}

} // namespace lcgs
