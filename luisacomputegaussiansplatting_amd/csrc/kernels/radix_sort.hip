// radix_sort.hip -- stable LSD radix sort of (key, u32 value) pairs, the primitive the reference
// borrows from lcpp (DeviceRadixSort<>::SortPairs<ulong,uint>, call site
// lcgs/src/gs_tile_splatter/impl.cpp:135-143; lcpp itself is not part of the reference tree).
//
// One pass = histogram -> scan of the [digit][block] count matrix -> scatter.  HBM-bound integer
// work: per pass sizeof(key) read (histogram) + sizeof(key)+4 read + sizeof(key)+4 written (scatter).
// CDNA4 specifics:
//   * ranking is done per wave64 with 64-bit ballots ("match any" built from <=8 ballots), rounds of
//     64 consecutive keys in memory order, so equal digits keep their input order (stability);
//   * per-wave digit counters live in LDS; the four waves of a block are combined by one thread per
//     digit; a block owns 4096 consecutive keys (4 waves x 16 rounds);
//   * keys and values are staged through LDS into block-sorted order before the global scatter so
//     that each digit's run leaves the CU as consecutive addresses (>= 64 B segments on average).
// The element count can be read from device memory (d_n) -- blocks past the live range exit at once.
#include "launch.hpp"

namespace lcgs
{

void launch_inclusive_sum_u32_dyn(const uint32_t* in, uint32_t* out, int64_t n_cap, const uint32_t* d_n, void* temp,
                                  hipStream_t stream);

namespace
{

constexpr int kThreads = 256;
constexpr int kWaves   = kThreads / 64;
constexpr int kItems   = 16;
constexpr int kKPB     = kThreads * kItems; // 4096 keys per block
constexpr int kRadix   = 256;

template <typename KeyT>
__device__ __forceinline__ uint32_t digit_of(KeyT k, int shift, uint32_t mask)
{
    return (uint32_t)(k >> shift) & mask;
}

struct DynN {
    int64_t         n_host;
    const uint32_t* d_n;
    __device__ __forceinline__ int64_t get() const { return d_n ? (int64_t)*d_n : n_host; }
};

// counts[d * nb + b] = number of keys of block b whose digit is d
template <typename KeyT>
__global__ void __launch_bounds__(kThreads) k_radix_hist(const KeyT* __restrict__ keys, DynN dn, int shift,
                                                           uint32_t mask, uint32_t* __restrict__ counts)
{
    __shared__ uint32_t s_hist[kRadix];
    const int64_t n  = dn.get();
    const int64_t nb = (n + kKPB - 1) / kKPB;
    if ((int64_t)blockIdx.x >= nb) return;
    s_hist[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kKPB;
#pragma unroll 4
    for (int r = 0; r < kItems; ++r) {
        int64_t i = base + (int64_t)r * kThreads + threadIdx.x;
        if (i < n) atomicAdd(&s_hist[digit_of(keys[i], shift, mask)], 1u);
    }
    __syncthreads();
    counts[(int64_t)threadIdx.x * nb + blockIdx.x] = s_hist[threadIdx.x];
}

template <typename KeyT>
__global__ void __launch_bounds__(kThreads) k_radix_scatter(const KeyT* __restrict__ keys_in,
                                                              const uint32_t* __restrict__ vals_in,
                                                              KeyT* __restrict__ keys_out,
                                                              uint32_t* __restrict__ vals_out, DynN dn, int shift,
                                                              uint32_t mask, int bits,
                                                              const uint32_t* __restrict__ counts_incl)
{
    __shared__ uint32_t s_wave_hist[kWaves][kRadix];
    __shared__ uint32_t s_digit_start[kRadix]; // block-local exclusive start of each digit
    __shared__ uint32_t s_global_delta[kRadix]; // global position of digit d's first key in this block - local start
    __shared__ uint32_t s_scan[kWaves];
    __shared__ KeyT     s_keys[kKPB];
    __shared__ uint32_t s_vals[kKPB];

    const int64_t n  = dn.get();
    const int64_t nb = (n + kKPB - 1) / kKPB;
    if ((int64_t)blockIdx.x >= nb) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t  block_base = (int64_t)blockIdx.x * kKPB;
    const int64_t  wave_base  = block_base + (int64_t)wave * 64 * kItems;
    const uint32_t in_block   = (uint32_t)((n - block_base) < kKPB ? (n - block_base) : kKPB);

#pragma unroll
    for (int w = 0; w < kWaves; ++w) s_wave_hist[w][tid] = 0;
    __syncthreads();

    KeyT     key[kItems];
    uint32_t rank[kItems];
    volatile uint32_t* my_hist = s_wave_hist[wave];

    // ---- phase 1: per-wave stable ranking, 64 keys per round in memory order
#pragma unroll
    for (int r = 0; r < kItems; ++r) {
        const int64_t i     = wave_base + (int64_t)r * 64 + lane;
        const bool    valid = i < n;
        key[r]              = valid ? keys_in[i] : (KeyT)0;
        const uint32_t d    = digit_of(key[r], shift, mask);
        unsigned long long peers = __ballot(valid);
        for (int b = 0; b < bits; ++b) {
            const bool               bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        const unsigned long long lt    = (1ull << lane) - 1ull;
        const uint32_t           below = __popcll(peers & lt);
        const uint32_t           count = __popcll(peers);
        uint32_t prev = 0;
        if (valid) prev = my_hist[d];
        __builtin_amdgcn_wave_barrier();
        if (valid && below == 0) my_hist[d] = prev + count;
        __builtin_amdgcn_wave_barrier();
        rank[r] = prev + below;
    }
    __syncthreads();

    // ---- phase 2: thread d combines the waves for digit d; block-wide exclusive scan over digits
    uint32_t wave_off[kWaves];
    uint32_t total = 0;
    {
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            wave_off[w] = total;
            total += s_wave_hist[w][tid];
        }
        // block exclusive scan of `total` over tid
        uint32_t inc = total;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t o = __shfl_up(inc, off, 64);
            if (lane >= off) inc += o;
        }
        if (lane == 63) s_scan[wave] = inc;
        __syncthreads();
        uint32_t carry = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w)
            if (w < wave) carry += s_scan[w];
        const uint32_t start = carry + inc - total;
        s_digit_start[tid]   = start;
        // counts_incl is the inclusive scan of the digit-major count matrix; exclusive = incl - own count
        const int64_t  ci         = (int64_t)tid * nb + blockIdx.x;
        const uint32_t global_pos = counts_incl[ci] - total;
        s_global_delta[tid]       = global_pos - start;
        // turn the per-wave histograms into per-wave starts inside the block-sorted order
#pragma unroll
        for (int w = 0; w < kWaves; ++w) s_wave_hist[w][tid] = start + wave_off[w];
    }
    __syncthreads();

    // ---- phase 3: stage keys/values into block-sorted order in LDS
#pragma unroll
    for (int r = 0; r < kItems; ++r) {
        const int64_t i = wave_base + (int64_t)r * 64 + lane;
        if (i < n) {
            const uint32_t d   = digit_of(key[r], shift, mask);
            const uint32_t pos = s_wave_hist[wave][d] + rank[r];
            s_keys[pos]        = key[r];
            s_vals[pos]        = vals_in[i];
        }
    }
    __syncthreads();

    // ---- phase 4: coalesced scatter of each digit's run
#pragma unroll 4
    for (uint32_t i = tid; i < in_block; i += kThreads) {
        const KeyT     k   = s_keys[i];
        const uint32_t d   = digit_of(k, shift, mask);
        const uint32_t dst = s_global_delta[d] + i;
        keys_out[dst]      = k;
        vals_out[dst]      = s_vals[i];
    }
}

// number of entries of the count matrix = 256 * nb, written to device memory so the scan can run dynamically
__global__ void k_radix_matrix_len(DynN dn, uint32_t* __restrict__ out_len)
{
    const int64_t n  = dn.get();
    const int64_t nb = (n + kKPB - 1) / kKPB;
    *out_len         = (uint32_t)(nb * kRadix);
}

template <typename KeyT>
__global__ void __launch_bounds__(kThreads) k_copy_pairs(const KeyT* __restrict__ ki, const uint32_t* __restrict__ vi,
                                                           KeyT* __restrict__ ko, uint32_t* __restrict__ vo, DynN dn)
{
    const int64_t n = dn.get();
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        ko[i] = ki[i];
        vo[i] = vi[i];
    }
}

// Runs the passes ping-ponging between (keys_a, vals_a) and (keys_b, vals_b), starting from a.
// Returns 0 if the sorted result ended in a, 1 if in b.
template <typename KeyT>
int sort_pairs_pingpong(KeyT* keys_a, KeyT* keys_b, uint32_t* vals_a, uint32_t* vals_b, int64_t n_cap,
                        const uint32_t* d_n, int begin_bit, int end_bit, void* temp, hipStream_t stream)
{
    if (n_cap <= 0) return 0;
    const int64_t nb_cap   = (n_cap + kKPB - 1) / kKPB;
    uint32_t*     counts   = reinterpret_cast<uint32_t*>(temp);
    uint32_t*     scanned  = counts + nb_cap * kRadix;
    uint32_t*     d_len    = scanned + nb_cap * kRadix;
    void*         scan_tmp = d_len + 4;
    DynN          dn{ n_cap, d_n };

    const int n_pass = (end_bit - begin_bit + 7) / 8;
    if (n_pass <= 0) return 0;
    hipLaunchKernelGGL(k_radix_matrix_len, dim3(1), dim3(1), 0, stream, dn, d_len);

    KeyT*     kb[2] = { keys_a, keys_b };
    uint32_t* vb[2] = { vals_a, vals_b };
    int       src   = 0;
    int       shift = begin_bit;
    for (int p = 0; p < n_pass; ++p) {
        const int      bits = (end_bit - shift) < 8 ? (end_bit - shift) : 8;
        const uint32_t mask = (1u << bits) - 1u;
        const int      dst  = src ^ 1;
        hipLaunchKernelGGL(k_radix_hist<KeyT>, dim3((unsigned)nb_cap), dim3(kThreads), 0, stream, kb[src], dn, shift,
                           mask, counts);
        launch_inclusive_sum_u32_dyn(counts, scanned, nb_cap * kRadix, d_len, scan_tmp, stream);
        hipLaunchKernelGGL(k_radix_scatter<KeyT>, dim3((unsigned)nb_cap), dim3(kThreads), 0, stream, kb[src], vb[src],
                           kb[dst], vb[dst], dn, shift, mask, bits, scanned);
        src = dst;
        shift += bits;
    }
    return src;
}

// Input-preserving form: pass 0 reads (keys_in, vals_in); later passes alternate between (out) and (tmp)
// such that the last pass lands in (out).  keys_in/vals_in must not alias out or tmp.
template <typename KeyT>
void sort_pairs_preserve(const KeyT* keys_in, const uint32_t* vals_in, KeyT* keys_out, uint32_t* vals_out,
                         KeyT* keys_tmp, uint32_t* vals_tmp, int64_t n_cap, const uint32_t* d_n, int begin_bit,
                         int end_bit, void* temp, hipStream_t stream)
{
    if (n_cap <= 0) return;
    const int64_t nb_cap   = (n_cap + kKPB - 1) / kKPB;
    uint32_t*     counts   = reinterpret_cast<uint32_t*>(temp);
    uint32_t*     scanned  = counts + nb_cap * kRadix;
    uint32_t*     d_len    = scanned + nb_cap * kRadix;
    void*         scan_tmp = d_len + 4;
    DynN          dn{ n_cap, d_n };
    const int     n_pass = (end_bit - begin_bit + 7) / 8;
    if (n_pass <= 0) {
        int64_t blocks = (n_cap + kThreads - 1) / kThreads;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(k_copy_pairs<KeyT>, dim3((unsigned)blocks), dim3(kThreads), 0, stream, keys_in, vals_in,
                           keys_out, vals_out, dn);
        return;
    }
    hipLaunchKernelGGL(k_radix_matrix_len, dim3(1), dim3(1), 0, stream, dn, d_len);
    const KeyT*     ksrc  = keys_in;
    const uint32_t* vsrc  = vals_in;
    int             shift = begin_bit;
    for (int p = 0; p < n_pass; ++p) {
        const int      bits   = (end_bit - shift) < 8 ? (end_bit - shift) : 8;
        const uint32_t mask   = (1u << bits) - 1u;
        const bool     to_out = ((n_pass - 1 - p) & 1) == 0;
        KeyT*          kdst   = to_out ? keys_out : keys_tmp;
        uint32_t*      vdst   = to_out ? vals_out : vals_tmp;
        hipLaunchKernelGGL(k_radix_hist<KeyT>, dim3((unsigned)nb_cap), dim3(kThreads), 0, stream, ksrc, dn, shift, mask,
                           counts);
        launch_inclusive_sum_u32_dyn(counts, scanned, nb_cap * kRadix, d_len, scan_tmp, stream);
        hipLaunchKernelGGL(k_radix_scatter<KeyT>, dim3((unsigned)nb_cap), dim3(kThreads), 0, stream, ksrc, vsrc, kdst,
                           vdst, dn, shift, mask, bits, scanned);
        ksrc = kdst;
        vsrc = vdst;
        shift += bits;
    }
}

template <typename KeyT>
void sort_pairs_impl(KeyT* keys_in, KeyT* keys_out, uint32_t* vals_in, uint32_t* vals_out, int64_t n_cap,
                     const uint32_t* d_n, int begin_bit, int end_bit, void* temp, hipStream_t stream)
{
    if (n_cap <= 0) return;
    int where = sort_pairs_pingpong<KeyT>(keys_in, keys_out, vals_in, vals_out, n_cap, d_n, begin_bit, end_bit, temp,
                                          stream);
    if (where == 0) {
        int64_t blocks = (n_cap + kThreads - 1) / kThreads;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(k_copy_pairs<KeyT>, dim3((unsigned)blocks), dim3(kThreads), 0, stream, keys_in, vals_in,
                           keys_out, vals_out, DynN{ n_cap, d_n });
    }
}

} // namespace

size_t sort_temp_bytes(int64_t n)
{
    const int64_t nb = (n + kKPB - 1) / kKPB;
    return (size_t)(2 * nb * kRadix + 8) * sizeof(uint32_t) + scan_temp_bytes(nb * kRadix) + 64;
}

void launch_sort_pairs_u64(uint64_t* keys_in, uint64_t* keys_out, uint32_t* vals_in, uint32_t* vals_out, int64_t n,
                           int begin_bit, int end_bit, void* temp, hipStream_t stream)
{
    sort_pairs_impl<uint64_t>(keys_in, keys_out, vals_in, vals_out, n, nullptr, begin_bit, end_bit, temp, stream);
}

void launch_sort_pairs_u32(uint32_t* keys_in, uint32_t* keys_out, uint32_t* vals_in, uint32_t* vals_out, int64_t n,
                           int begin_bit, int end_bit, void* temp, hipStream_t stream)
{
    sort_pairs_impl<uint32_t>(keys_in, keys_out, vals_in, vals_out, n, nullptr, begin_bit, end_bit, temp, stream);
}

void launch_sort_pairs_u32_dyn(uint32_t* keys_in, uint32_t* keys_out, uint32_t* vals_in, uint32_t* vals_out,
                               const uint32_t* d_n, int64_t n_cap, int begin_bit, int end_bit, void* temp,
                               hipStream_t stream)
{
    sort_pairs_impl<uint32_t>(keys_in, keys_out, vals_in, vals_out, n_cap, d_n, begin_bit, end_bit, temp, stream);
}

void launch_sort_pairs_u64_preserve(const uint64_t* keys_in, const uint32_t* vals_in, uint64_t* keys_out,
                                    uint32_t* vals_out, uint64_t* keys_tmp, uint32_t* vals_tmp, int64_t n, int begin_bit,
                                    int end_bit, void* temp, hipStream_t stream)
{
    sort_pairs_preserve<uint64_t>(keys_in, vals_in, keys_out, vals_out, keys_tmp, vals_tmp, n, nullptr, begin_bit,
                                  end_bit, temp, stream);
}

int launch_sort_pairs_u32_pingpong(uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a, uint32_t* vals_b,
                                   const uint32_t* d_n, int64_t n_cap, int begin_bit, int end_bit, void* temp,
                                   hipStream_t stream)
{
    return sort_pairs_pingpong<uint32_t>(keys_a, keys_b, vals_a, vals_b, n_cap, d_n, begin_bit, end_bit, temp, stream);
}

} // namespace lcgs
