// cu_mask_probe.hip -- which CUs does a hipExtStreamCreateWithCUMask stream run on?  The mask's bit numbering against the
// eight XCDs of an MI355X is not documented in this image; every workgroup of a small kernel records the XCC it ran on
// (s_getreg_b32 XCC_ID) and its SE / SH / CU (HW_ID), once per mask, and the host prints the distinct placements.
//   hipcc --offload-arch=gfx950 -O3 cu_mask_probe.hip -o cu_mask_probe && ./cu_mask_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>

__global__ void __launch_bounds__(64) k_where(uint32_t* out, uint32_t spin)
{
    uint32_t xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    // keep the workgroup resident for a moment so that the grid spreads over every CU the mask allows
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < spin) {}
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x]     = xcc & 0xF;
        out[2 * blockIdx.x + 1] = hw;
    }
}

static void probe(const char* name, const uint32_t mask[8])
{
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, 8, mask) != hipSuccess) {
        printf("%s: hipExtStreamCreateWithCUMask failed\n", name);
        return;
    }
    const int G = 4096;
    uint32_t* d;
    hipMalloc(&d, G * 8);
    hipLaunchKernelGGL(k_where, dim3(G), dim3(64), 0, s, d, 20000u);
    hipStreamSynchronize(s);
    std::vector<uint32_t> h(2 * G);
    hipMemcpy(h.data(), d, G * 8, hipMemcpyDeviceToHost);
    std::map<uint32_t, std::set<uint32_t>> cus; // xcc -> {se, sh, cu}
    for (int i = 0; i < G; ++i) {
        const uint32_t hw = h[2 * i + 1];
        cus[h[2 * i]].insert(((hw >> 13) & 7) << 8 | ((hw >> 12) & 1) << 4 | ((hw >> 8) & 15));
    }
    int total = 0;
    printf("%-28s", name);
    for (auto& kv : cus) {
        printf(" xcc%u:%zu", kv.first, kv.second.size());
        total += (int)kv.second.size();
    }
    printf("  (= %d CUs)\n", total);
    hipFree(d);
    hipStreamDestroy(s);
}

int main()
{
    uint32_t all[8], low32[8] = { 0xFFFFFFFFu }, stride8[8], spread32[8] = {}, spread64[8] = {};
    for (int i = 0; i < 8; ++i) {
        all[i]     = 0xFFFFFFFFu;
        stride8[i] = 0x01010101u; // bits 0, 8, 16, 24 of every word
    }
    for (int K : { 32, 64 })
        for (int x = 0; x < 8; ++x)
            for (int j = 0; j < K / 8; ++j) (K == 32 ? spread32 : spread64)[x] |= 1u << ((x + j) % 8 + 8 * (j % 4));
    probe("all 256 bits", all);
    probe("bits 0..31 only", low32);
    probe("every 8th bit (32 bits)", stride8);
    probe("library mask K=32", spread32);
    probe("library mask K=64", spread64);
    uint32_t c32[8], c64[8];
    for (int i = 0; i < 8; ++i) {
        c32[i] = ~spread32[i];
        c64[i] = ~spread64[i];
    }
    probe("complement of K=32", c32);
    probe("complement of K=64", c64);
    return 0;
}
