// atomic_rate.hip -- throughput of relaxed agent-scope global atomicAdd (no return) to distinct addresses, the
// pattern a producer kernel would use to leave per-chunk digit counts behind.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_atomics(uint32_t* table, uint32_t table_words, int per_thread, uint32_t stride, uint32_t lane_stride)
{
    // workgroup b touches a window of 512 counters starting at (b * stride) % table_words: ~2.5 workgroups per window
    const uint32_t base = (blockIdx.x * stride) % table_words;
    for (int i = 0; i < per_thread; ++i) {
        // lane_stride 1: thread t -> consecutive words; lane_stride 1200: the [digit][chunk] table of the radix sort
        const uint32_t a = lane_stride == 1u ? (base + threadIdx.x + i * blockDim.x) % table_words
                                             : (threadIdx.x * lane_stride + (blockIdx.x * 2u / 5u + i) % lane_stride);
        atomicAdd(&table[a], 1u);
    }
}

__global__ void k_plain(uint32_t* table, uint32_t table_words, int per_thread, uint32_t stride, uint32_t lane_stride)
{
    const uint32_t base = (blockIdx.x * stride) % table_words;
    for (int i = 0; i < per_thread; ++i) {
        const uint32_t a = lane_stride == 1u ? (base + threadIdx.x + i * blockDim.x) % table_words
                                             : (threadIdx.x * lane_stride + (blockIdx.x * 2u / 5u + i) % lane_stride);
        table[a] = threadIdx.x;
    }
}

int main()
{
    const uint32_t words = 256u * 1200u; // 256 digits x 1200 chunks
    uint32_t*      d;
    (void)hipMalloc(&d, words * 4);
    (void)hipMemset(d, 0, words * 4);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    for (int wg : { 3000 }) {
        for (int per : { 1, 2, 4 }) {
            for (int kind = 0; kind < 4; ++kind) {
                const uint32_t ls = kind < 2 ? 1u : 1200u;
                float best = 1e9f;
                for (int rep = 0; rep < 5; ++rep) {
                    (void)hipEventRecord(a);
                    if ((kind & 1) == 0) hipLaunchKernelGGL(k_atomics, dim3(wg), dim3(256), 0, 0, d, words, per, 205u, ls);
                    else hipLaunchKernelGGL(k_plain, dim3(wg), dim3(256), 0, 0, d, words, per, 205u, ls);
                    (void)hipEventRecord(b);
                    (void)hipEventSynchronize(b);
                    float ms;
                    (void)hipEventElapsedTime(&ms, a, b);
                    if (ms < best) best = ms;
                }
                printf("%s lane-stride %4u wg=%d x 256 thr x %d = %.2f M ops: %.1f us  (%.1f ops/ns)\n", (kind & 1) ? "store " : "atomic", ls, wg, per,
                       wg * 256.0 * per / 1e6, best * 1e3, wg * 256.0 * per / (best * 1e6));
            }
        }
    }
    return 0;
}
