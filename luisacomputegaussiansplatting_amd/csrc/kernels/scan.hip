// scan.hip -- inclusive prefix sum over u32, the primitive the reference borrows from lcpp
// (DeviceScan<>::InclusiveSum, call site lcgs/src/gs_tile_splatter/impl.cpp:104; lcpp itself is not
// part of the reference tree).  u32 wrap-around arithmetic, like the reference's Buffer<uint>.
//
// Shape: reduce-then-scan in three launches (per-block sums -> scan of block sums -> per-block scan
// with carry-in).  HBM-bound: 8 B read + 4 B written per element; wave64 shuffles for the in-block
// scan, one 16-byte load per lane.  The element count may live in device memory (d_n) so that a
// whole frame can be enqueued without a host round trip; blocks past the live range exit at once.
#include "launch.hpp"

namespace lcgs
{
namespace
{

constexpr int kScanThreads = 256;
constexpr int kScanItems   = 4;                          // one uint4 per lane
constexpr int kScanTile    = kScanThreads * kScanItems;  // 1024 elements per block

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t o = __shfl_up(v, off, 64);
        if (lane >= off) v += o;
    }
    return v;
}

// inclusive scan of one value per thread across a 256-thread block; returns the block total in `total`
__device__ __forceinline__ uint32_t block_inclusive_scan(uint32_t v, uint32_t* s_wave /*[4]*/, uint32_t& total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t  inc  = wave_inclusive_scan(v);
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t carry = 0;
#pragma unroll
    for (int w = 0; w < kScanThreads / 64; ++w) {
        uint32_t s = s_wave[w];
        if (w < wave) carry += s;
    }
    total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    __syncthreads();
    return inc + carry;
}

__device__ __forceinline__ void load_tile(const uint32_t* __restrict__ in, int64_t base, int64_t n, uint32_t v[4])
{
    int64_t i = base + (int64_t)threadIdx.x * kScanItems;
    if (i + 3 < n && ((reinterpret_cast<uintptr_t>(in + i) & 15) == 0)) {
        uint4 q = *reinterpret_cast<const uint4*>(in + i);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (i + k < n) ? in[i + k] : 0u;
    }
}

__global__ void __launch_bounds__(kScanThreads) k_scan_reduce(const uint32_t* __restrict__ in, int64_t n_host,
                                                                const uint32_t* __restrict__ d_n,
                                                                uint32_t* __restrict__ block_sums)
{
    __shared__ uint32_t s_wave[4];
    const int64_t n    = d_n ? (int64_t)*d_n : n_host;
    const int64_t base = (int64_t)blockIdx.x * kScanTile;
    if (base >= n) return;
    uint32_t v[4];
    load_tile(in, base, n, v);
    uint32_t sum = v[0] + v[1] + v[2] + v[3];
    uint32_t total;
    block_inclusive_scan(sum, s_wave, total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// exclusive scan of the block sums, in place, by one block (nb <= a few thousand)
__global__ void __launch_bounds__(1024) k_scan_block_sums(uint32_t* __restrict__ block_sums, int64_t n_host,
                                                           const uint32_t* __restrict__ d_n)
{
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry;
    const int64_t n  = d_n ? (int64_t)*d_n : n_host;
    const int64_t nb = (n + kScanTile - 1) / kScanTile;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < nb; base += 1024) {
        int64_t  i   = base + threadIdx.x;
        uint32_t v   = i < nb ? block_sums[i] : 0u;
        uint32_t inc = wave_inclusive_scan(v);
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        uint32_t carry = s_carry;
        for (int w = 0; w < wave; ++w) carry += s_wave[w];
        if (i < nb) block_sums[i] = carry + inc - v; // exclusive
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = carry + inc;
        __syncthreads();
    }
}

__global__ void __launch_bounds__(kScanThreads) k_scan_final(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                               int64_t n_host, const uint32_t* __restrict__ d_n,
                                                               const uint32_t* __restrict__ block_sums)
{
    __shared__ uint32_t s_wave[4];
    const int64_t n    = d_n ? (int64_t)*d_n : n_host;
    const int64_t base = (int64_t)blockIdx.x * kScanTile;
    if (base >= n) return;
    uint32_t v[4];
    load_tile(in, base, n, v);
    v[1] += v[0];
    v[2] += v[1];
    v[3] += v[2];
    uint32_t total;
    uint32_t inc   = block_inclusive_scan(v[3], s_wave, total);
    uint32_t carry = block_sums[blockIdx.x] + (inc - v[3]);
    int64_t  i     = base + (int64_t)threadIdx.x * kScanItems;
    if (i + 3 < n && ((reinterpret_cast<uintptr_t>(out + i) & 15) == 0)) {
        *reinterpret_cast<uint4*>(out + i) = make_uint4(v[0] + carry, v[1] + carry, v[2] + carry, v[3] + carry);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i + k < n) out[i + k] = v[k] + carry;
    }
}

} // namespace

size_t scan_temp_bytes(int64_t n)
{
    int64_t nb = (n + kScanTile - 1) / kScanTile;
    return (size_t)(nb + 1) * sizeof(uint32_t);
}

void launch_inclusive_sum_u32_dyn(const uint32_t* in, uint32_t* out, int64_t n_cap, const uint32_t* d_n, void* temp,
                                  hipStream_t stream)
{
    if (n_cap <= 0) return;
    uint32_t* block_sums = reinterpret_cast<uint32_t*>(temp);
    int64_t   nb         = (n_cap + kScanTile - 1) / kScanTile;
    hipLaunchKernelGGL(k_scan_reduce, dim3((unsigned)nb), dim3(kScanThreads), 0, stream, in, n_cap, d_n, block_sums);
    hipLaunchKernelGGL(k_scan_block_sums, dim3(1), dim3(1024), 0, stream, block_sums, n_cap, d_n);
    hipLaunchKernelGGL(k_scan_final, dim3((unsigned)nb), dim3(kScanThreads), 0, stream, in, out, n_cap, d_n,
                       block_sums);
}

void launch_inclusive_sum_u32(const uint32_t* in, uint32_t* out, int64_t n, void* temp, hipStream_t stream)
{
    launch_inclusive_sum_u32_dyn(in, out, n, nullptr, temp, stream);
}

} // namespace lcgs
