// stream_aos.hip -- how fast can one pass read the cull kernel's inputs (pos[3P] scale[3P] rotq[4P] as separate
// f32 arrays, 12/12/16-byte records per lane) on this part?  Bounds k_cull_compact from below.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int THREADS, int ITEMS, bool TICKET>
__global__ void __launch_bounds__(THREADS) k_read(int P, const float* __restrict__ pos, const float* __restrict__ scale,
                                                  const float* __restrict__ rotq, float* __restrict__ out,
                                                  unsigned* __restrict__ ticket)
{
    __shared__ unsigned s_t;
    unsigned bid = blockIdx.x;
    if (TICKET) {
        if (threadIdx.x == 0) s_t = atomicAdd(ticket, 1u);
        __syncthreads();
        bid = s_t;
    }
    const long base = (long)bid * THREADS * ITEMS;
    float      acc  = 0.f;
    float      v[ITEMS][10];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        long idx = base + (long)k * THREADS + threadIdx.x;
        if (idx >= P) idx = P - 1;
        v[k][0] = pos[3 * idx + 0]; v[k][1] = pos[3 * idx + 1]; v[k][2] = pos[3 * idx + 2];
        v[k][3] = scale[3 * idx + 0]; v[k][4] = scale[3 * idx + 1]; v[k][5] = scale[3 * idx + 2];
        const float4 q = *reinterpret_cast<const float4*>(rotq + 4 * idx);
        v[k][6] = q.x; v[k][7] = q.y; v[k][8] = q.z; v[k][9] = q.w;
    }
#pragma unroll
    for (int k = 0; k < ITEMS; ++k)
#pragma unroll
        for (int j = 0; j < 10; ++j) acc += v[k][j];
    if (acc == 12345.678f) out[bid] = acc; // never true: keeps the loads alive
}

template <int THREADS, int ITEMS, bool TICKET>
void run(const char* name, int P, float* pos, float* scale, float* rotq, float* out, unsigned* ticket)
{
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    const int blocks = (P + THREADS * ITEMS - 1) / (THREADS * ITEMS);
    float     best   = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        (void)hipMemsetAsync(ticket, 0, 4, 0);
        (void)hipEventRecord(a, 0);
        hipLaunchKernelGGL((k_read<THREADS, ITEMS, TICKET>), dim3(blocks), dim3(THREADS), 0, 0, P, pos, scale, rotq, out, ticket);
        (void)hipEventRecord(b, 0);
        (void)hipEventSynchronize(b);
        float ms;
        (void)hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    printf("%-34s %4d blocks  %.1f us  %.2f TB/s\n", name, blocks, best * 1e3, 40.0 * P / (best * 1e9));
}

int main()
{
    const int P = 6131954;
    float *pos, *scale, *rotq, *out;
    unsigned* ticket;
    (void)hipMalloc(&pos, (size_t)P * 12);
    (void)hipMalloc(&scale, (size_t)P * 12);
    (void)hipMalloc(&rotq, (size_t)P * 16);
    (void)hipMalloc(&out, 1 << 20);
    (void)hipMalloc(&ticket, 64);
    (void)hipMemset(pos, 0, (size_t)P * 12);
    (void)hipMemset(scale, 0, (size_t)P * 12);
    (void)hipMemset(rotq, 0, (size_t)P * 16);
    run<512, 4, false>("512 thr x 4, blockIdx", P, pos, scale, rotq, out, ticket);
    run<512, 4, true>("512 thr x 4, ticket", P, pos, scale, rotq, out, ticket);
    run<256, 4, false>("256 thr x 4, blockIdx", P, pos, scale, rotq, out, ticket);
    run<256, 2, false>("256 thr x 2, blockIdx", P, pos, scale, rotq, out, ticket);
    run<256, 1, false>("256 thr x 1, blockIdx", P, pos, scale, rotq, out, ticket);
    run<1024, 2, false>("1024 thr x 2, blockIdx", P, pos, scale, rotq, out, ticket);
    run<256, 8, false>("256 thr x 8, blockIdx", P, pos, scale, rotq, out, ticket);
    return 0;
}
