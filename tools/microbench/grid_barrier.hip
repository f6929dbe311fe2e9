// grid_barrier.hip -- what does a grid-wide barrier inside ONE persistent kernel cost on the MI355X, against the launch
// boundary it would replace?  The frame's sort chain is ~22 dependent short launches at a floor of ~7.5 us each
// (DESIGN.md 11); a persistent form pays a barrier per phase instead.  Per phase every thread writes 16 bytes (dirty lines
// in its XCD's L2), the grid synchronises (release -> arrive on a counter -> spin -> acquire, agent scope), and every
// thread reads what a thread of ANOTHER workgroup (another XCD: workgroups are dealt round-robin) wrote and checks it.
//   hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o grid_barrier && ./grid_barrier
// The grid never exceeds what is co-resident (<= 2 workgroups of 256 threads per CU), and every spin has an iteration cap:
// a workgroup that is not scheduled cannot hang the others for ever (the run then reports `timed out`).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ bool grid_sync(uint32_t* counter, uint32_t* flag, uint32_t phase, uint32_t groups)
{
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        // (release / acquire at AGENT scope, named explicitly: one GPU, eight XCDs with an L2 each)
        const uint32_t arrived = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        if (arrived == groups * (phase + 1u)) __hip_atomic_store(flag, phase + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < phase + 1u) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) { // ~ seconds: give up instead of hanging the GPU
                ok = false;
                break;
            }
        }
    }
    __syncthreads();
    return ok;
}

__global__ void __launch_bounds__(256) k_persistent(uint4* buf, uint32_t* counter, uint32_t* flag, uint32_t* errors,
                                                    uint32_t phases, uint32_t bytes_per_thread)
{
    const uint32_t groups = gridDim.x, gid = blockIdx.x * 256u + threadIdx.x, total = groups * 256u;
    const uint32_t vecs = bytes_per_thread / 16u;
    for (uint32_t p = 0; p < phases; ++p) {
        for (uint32_t v = 0; v < vecs; ++v) buf[(size_t)v * total + gid] = make_uint4(p, gid, v, p ^ gid);
        if (!grid_sync(counter, flag, p, groups)) {
            if (threadIdx.x == 0) atomicAdd(errors + 1, 1u);
            return;
        }
        // a thread of the workgroup 5 further on (another XCD) wrote this
        const uint32_t other = (gid + 5u * 256u) % total;
        uint32_t       bad = 0;
        for (uint32_t v = 0; v < vecs; ++v) {
            const uint4 r = buf[(size_t)v * total + other];
            bad |= (r.x != p) | (r.y != other) | (r.z != v);
        }
        if (bad) atomicAdd(errors, 1u);
        // (the next phase overwrites buf: everybody must have read first -- a second barrier would be needed by a real
        //  consumer that reuses its input; here the two alternate between two halves instead)
        buf += (p & 1u) ? -(ptrdiff_t)((size_t)vecs * total) : (ptrdiff_t)((size_t)vecs * total);
    }
}

__global__ void __launch_bounds__(256) k_phase(uint4* buf, uint32_t p, uint32_t bytes_per_thread, uint32_t* errors)
{
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x, total = gridDim.x * 256u, vecs = bytes_per_thread / 16u;
    uint4*         in  = buf + ((p & 1u) ? 0 : (size_t)vecs * total); // what the previous launch wrote
    uint4*         out = buf + ((p & 1u) ? (size_t)vecs * total : 0);
    if (p > 0) {
        const uint32_t other = (gid + 5u * 256u) % total;
        uint32_t       bad = 0;
        for (uint32_t v = 0; v < vecs; ++v) {
            const uint4 r = in[(size_t)v * total + other];
            bad |= (r.x != p - 1u) | (r.y != other);
        }
        if (bad) atomicAdd(errors, 1u);
    }
    for (uint32_t v = 0; v < vecs; ++v) out[(size_t)v * total + gid] = make_uint4(p, gid, v, p ^ gid);
}

int main()
{
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("device: %s, %d CUs\n", prop.name, cus);
    const uint32_t phases = 200;
    uint32_t *     counter, *flag, *errors;
    (void)hipMalloc(&counter, 4);
    (void)hipMalloc(&flag, 4);
    (void)hipMalloc(&errors, 8);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    for (int per_cu : { 1, 2 }) {
        const uint32_t groups = (uint32_t)(cus * per_cu);
        for (uint32_t bytes : { 16u, 64u, 256u, 1024u }) { // per thread and phase: 1 MB .. 134 MB per phase over the grid
            uint4* buf;
            (void)hipMalloc(&buf, (size_t)2 * groups * 256 * bytes);
            float best_p = 1e9f, best_l = 1e9f;
            uint32_t h_err[2] = { 0, 0 };
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipMemset(counter, 0, 4);
                (void)hipMemset(flag, 0, 4);
                (void)hipMemset(errors, 0, 8);
                (void)hipEventRecord(a);
                hipLaunchKernelGGL(k_persistent, dim3(groups), dim3(256), 0, 0, buf, counter, flag, errors, phases, bytes);
                (void)hipEventRecord(b);
                (void)hipEventSynchronize(b);
                float ms;
                (void)hipEventElapsedTime(&ms, a, b);
                if (ms < best_p) best_p = ms;
                (void)hipMemcpy(h_err, errors, 8, hipMemcpyDeviceToHost);
                if (h_err[0] || h_err[1]) break;
                (void)hipMemset(errors, 0, 8);
                (void)hipEventRecord(a);
                for (uint32_t p = 0; p < phases; ++p) hipLaunchKernelGGL(k_phase, dim3(groups), dim3(256), 0, 0, buf, p, bytes, errors);
                (void)hipEventRecord(b);
                (void)hipEventSynchronize(b);
                (void)hipEventElapsedTime(&ms, a, b);
                if (ms < best_l) best_l = ms;
            }
            printf("%4u workgroups x 256 threads, %5u B/thread/phase (%6.1f MB/phase): persistent + grid barrier %6.2f us/phase%s, "
                   "one launch per phase %6.2f us/phase   [visibility errors %u]\n",
                   groups, bytes, groups * 256.0 * bytes / 1e6, best_p * 1e3 / phases, h_err[1] ? " (TIMED OUT)" : "",
                   best_l * 1e3 / phases, h_err[0]);
            (void)hipFree(buf);
        }
    }
    return 0;
}
