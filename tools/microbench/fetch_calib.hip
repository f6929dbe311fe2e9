// fetch_calib.hip -- calibrates rocprofv3's FETCH_SIZE counter on gfx950 for the ACCESS PATTERNS of this library.
// MI355X_MICROARCH.md: FETCH_SIZE reports exactly half the bytes of a wide (16 B / lane) coalesced streaming read, and
// "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern".  Every kernel
// below requests a known number of bytes from a 1 GiB table (4x the Infinity Cache, so re-reads cannot hide); run under
//     rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -- ./fetch_calib
// and divide the counter (KB per dispatch) by the printed byte counts (tools/microbench/fetch_calib.sh does both).
// Patterns: the streaming reads of the cull pass (4 / 12 / 16 B per lane), and the gathers of the duplication
// (8-byte rects), the renderer (36 of a 48-byte record), the backward (48-byte records) and the record builder
// (192-byte SH rows, 12 lanes x 16 B each).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

constexpr size_t kTableBytes = (size_t)1 << 30;

__device__ __forceinline__ uint64_t mix(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

#define SINK(acc)                                                    \
    if ((acc) == 12345.678f) out[blockIdx.x] = (acc); /* never true */

__global__ void __launch_bounds__(256) k_stream16(const float4* __restrict__ t, size_t n, float* __restrict__ out)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4 v = t[i];
        acc += v.x + v.y + v.z + v.w;
    }
    SINK(acc)
}
__global__ void __launch_bounds__(256) k_stream4(const float* __restrict__ t, size_t n, float* __restrict__ out)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += t[i];
    SINK(acc)
}
// 12-byte rows, three 4-byte loads per lane (the pos / scale arrays)
__global__ void __launch_bounds__(256) k_stream12(const float* __restrict__ t, size_t rows, float* __restrict__ out)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < rows; i += (size_t)gridDim.x * 256)
        acc += t[3 * i] + t[3 * i + 1] + t[3 * i + 2];
    SINK(acc)
}
// random gathers: lane g reads `BYTES` of record (hash(g) mod records), records of `PITCH` bytes
template <int PITCH, int BYTES>
__global__ void __launch_bounds__(256) k_gather(const char* __restrict__ t, size_t records, size_t n, float* __restrict__ out)
{
    float acc = 0.f;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < n; g += (size_t)gridDim.x * 256) {
        const char* p = t + (mix(g) % records) * PITCH;
        if (BYTES == 8) {
            const uint2 v = *reinterpret_cast<const uint2*>(p);
            acc += __uint_as_float(v.x) + __uint_as_float(v.y);
        } else {
#pragma unroll
            for (int b = 0; b + 16 <= BYTES; b += 16) {
                const float4 v = *reinterpret_cast<const float4*>(p + b);
                acc += v.x + v.y + v.z + v.w;
            }
            if (BYTES % 16 == 4) acc += *reinterpret_cast<const float*>(p + BYTES - 4);
        }
    }
    SINK(acc)
}
// 192-byte rows read by 12 consecutive lanes x 16 B (k_build_records' SH fetch): g = row slot * 12 + part
__global__ void __launch_bounds__(256) k_gather_row192(const char* __restrict__ t, size_t records, size_t n_rows,
                                                       float* __restrict__ out)
{
    float acc = 0.f;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < n_rows * 12; g += (size_t)gridDim.x * 256) {
        const size_t slot = g / 12, part = g % 12;
        const float4 v = *reinterpret_cast<const float4*>(t + (mix(slot) % records) * 192 + part * 16);
        acc += v.x + v.y + v.z + v.w;
    }
    SINK(acc)
}
// the same rows in ASCENDING order, 39 % of them (a view's on-screen splats in file order): page-sparse, not random
__global__ void __launch_bounds__(256) k_gather_row192_sparse_ascending(const char* __restrict__ t, size_t records,
                                                                          size_t n_rows, float* __restrict__ out)
{
    float acc = 0.f;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < n_rows * 12; g += (size_t)gridDim.x * 256) {
        const size_t slot = g / 12, part = g % 12;
        const size_t row  = (slot * 100) / 39 + (mix(slot) & 1); // every ~2.56th row, jittered
        const float4 v = *reinterpret_cast<const float4*>(t + (row % records) * 192 + part * 16);
        acc += v.x + v.y + v.z + v.w;
    }
    SINK(acc)
}

template <typename F>
void run(const char* name, double requested_bytes, F launch)
{
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(a, 0);
        launch();
        (void)hipEventRecord(b, 0);
        (void)hipEventSynchronize(b);
        float ms;
        (void)hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    printf("%-36s requested_bytes %.0f  best %.1f us  %.2f TB/s requested\n", name, requested_bytes, best * 1e3,
           requested_bytes / (best * 1e9));
}

int main()
{
    char*  t   = nullptr;
    float* out = nullptr;
    if (hipMalloc(&t, kTableBytes) != hipSuccess || hipMalloc(&out, 1 << 20) != hipSuccess) return 1;
    (void)hipMemset(t, 0, kTableBytes);
    (void)hipDeviceSynchronize();
    const dim3   grid(8192), block(256);
    const size_t n_stream = kTableBytes / 16; // one pass over the whole table
    const size_t N = (size_t)8 << 20;         // gathers per launch
    run("k_stream16", 16.0 * n_stream, [&] { hipLaunchKernelGGL(k_stream16, grid, block, 0, 0, (const float4*)t, n_stream, out); });
    run("k_stream4", 4.0 * (kTableBytes / 4), [&] { hipLaunchKernelGGL(k_stream4, grid, block, 0, 0, (const float*)t, kTableBytes / 4, out); });
    run("k_stream12", 12.0 * (kTableBytes / 12), [&] { hipLaunchKernelGGL(k_stream12, grid, block, 0, 0, (const float*)t, kTableBytes / 12, out); });
    run("k_gather<8,8>", 8.0 * N, [&] { hipLaunchKernelGGL((k_gather<8, 8>), grid, block, 0, 0, t, kTableBytes / 8, N, out); });
    run("k_gather<16,16>", 16.0 * N, [&] { hipLaunchKernelGGL((k_gather<16, 16>), grid, block, 0, 0, t, kTableBytes / 16, N, out); });
    run("k_gather<48,36>", 36.0 * N, [&] { hipLaunchKernelGGL((k_gather<48, 36>), grid, block, 0, 0, t, kTableBytes / 48, N, out); });
    run("k_gather<48,48>", 48.0 * N, [&] { hipLaunchKernelGGL((k_gather<48, 48>), grid, block, 0, 0, t, kTableBytes / 48, N, out); });
    run("k_gather_row192", 192.0 * (N / 4), [&] { hipLaunchKernelGGL(k_gather_row192, grid, block, 0, 0, t, kTableBytes / 192, N / 4, out); });
    run("k_gather_row192_sparse_ascending", 192.0 * (N / 4),
        [&] { hipLaunchKernelGGL(k_gather_row192_sparse_ascending, grid, block, 0, 0, t, kTableBytes / 192, N / 4, out); });
    (void)hipFree(t);
    (void)hipFree(out);
    return 0;
}
