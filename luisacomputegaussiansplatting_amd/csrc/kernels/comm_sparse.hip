// comm_sparse.hip -- device side of the SPARSE gradient exchange (host/comm.cpp: lcgs_adam_step_sparse; SURVEY 8e
// "Collective").  A view touches only its on-screen splats (39 % of the bicycle stand-in), so the rows a rank has to
// hand to a row's owner are a fraction of the dense 236 B x P that a reduce-scatter moves.  Per optimiser step:
//   mark       every dense backward flags its frame's on-screen rows (one byte per splat, k_mark_rows)
//   compact    flags -> ascending list of touched rows (count / scan / scatter: three short launches, no look-back
//              chain -- DESIGN 4 measured what a ticket per chunk costs on this part), and the positions in that list
//              where each owner's shard begins (k_owner_bounds: rows are ascending, shards are contiguous row ranges)
//   pack       the rows of one owner's segment -> one message: [indices][pos][scale][rotq][sh][opacity], attribute by
//              attribute so that both sides stream whole 16-byte words
//   accumulate a received message's rows are ADDED to the owner's dense gradient rows (indices inside a message are
//              unique: plain read-modify-write, no atomics; messages are applied in rank order -> deterministic sums)
// No MFMA, nothing to tile: byte-moving kernels, bound by HBM (gathers of 12-192-byte rows on one side, streams on the
// other).
#include "launch.hpp"

#include "../common.hpp"

namespace lcgs
{
namespace
{

constexpr int kFlagsPerChunk = 4096; // 256 threads x 16 flag bytes

__global__ void __launch_bounds__(256) k_mark_rows(const uint32_t* __restrict__ vis_index, const uint32_t* __restrict__ d_counts,
                                                   uint8_t* __restrict__ flags, uint32_t P)
{
    const uint32_t V = d_counts[0];
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < V; i += gridDim.x * 256u) {
        const uint32_t r = vis_index[i]; // dense id -> splat index (ascending)
        if (r < P) flags[r] = 1;
    }
}

__device__ __forceinline__ uint32_t load_flags16(const uint8_t* __restrict__ flags, uint32_t base, uint32_t P, uint32_t f[4])
{
    // flags is allocated in whole chunks (zero padded), so the 16-byte load is always in bounds
    const uint4 v = *reinterpret_cast<const uint4*>(flags + base);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    uint32_t n = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) n += __popc(f[k] & 0x01010101u);
    (void)P;
    return n;
}

__global__ void __launch_bounds__(256) k_flag_count(const uint8_t* __restrict__ flags, uint32_t P, uint32_t* __restrict__ chunk_count)
{
    __shared__ uint32_t s_w[4];
    uint32_t            f[4];
    uint32_t n = load_flags16(flags, blockIdx.x * (uint32_t)kFlagsPerChunk + threadIdx.x * 16u, P, f);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) chunk_count[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// one workgroup: exclusive scan of the chunk counts (in place) and the total
// (copy_src / copy_dst, nullable: one more word carried along -- a caller that reads *d_total back with another scalar saves a copy launch)
// (host_box, nullable: host-visible pinned words.  Three scalars -- *box_word0, the total, the copied word -- are
//  posted there (word 0 = *box_word0, a scalar of the caller's) followed by `serial`, so that a host thread polling host_box[3] has them without a copy launch and a stream
//  synchronisation, and while the scatter launch behind this one is still running)
__global__ void __launch_bounds__(1024) k_flag_scan(uint32_t* __restrict__ chunk_count, uint32_t chunks, uint32_t* __restrict__ d_total,
                                                    const uint32_t* __restrict__ copy_src, uint32_t* __restrict__ copy_dst,
                                                    const uint32_t* box_word0, volatile uint32_t* host_box, uint32_t serial)
{
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_carry;
    const uint32_t      tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < chunks; base += 1024u) {
        const uint32_t i   = base + tid;
        const uint32_t own = i < chunks ? chunk_count[i] : 0u;
        uint32_t       inc = own;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(inc, off, 64);
            if (lane >= (uint32_t)off) inc += o;
        }
        if (lane == 63u) s_w[wave] = inc;
        __syncthreads();
        uint32_t before = s_carry;
        for (uint32_t w = 0; w < wave; ++w) before += s_w[w];
        if (i < chunks) chunk_count[i] = before + inc - own;
        __syncthreads();
        if (tid == 1023u) s_carry = before + inc;
        __syncthreads();
    }
    if (tid == 0) *d_total = s_carry;
    if (tid == 64u && copy_dst) {
        const uint32_t carried = *copy_src;
        *copy_dst              = carried;
        if (host_box) {
            host_box[0] = *box_word0;
            host_box[1] = s_carry;
            host_box[2] = carried;
            __threadfence_system();
            host_box[3] = serial;
        }
    }
}

__global__ void __launch_bounds__(256) k_flag_scatter(const uint8_t* __restrict__ flags, uint32_t P,
                                                      const uint32_t* __restrict__ chunk_base, uint32_t* __restrict__ rows)
{
    __shared__ uint32_t s_w[4];
    uint32_t            f[4];
    const uint32_t      first = blockIdx.x * (uint32_t)kFlagsPerChunk + threadIdx.x * 16u;
    const uint32_t      own   = load_flags16(flags, first, P, f);
    const uint32_t      lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t            inc = own;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(inc, off, 64);
        if (lane >= (uint32_t)off) inc += o;
    }
    if (lane == 63u) s_w[wave] = inc;
    __syncthreads();
    uint32_t at = chunk_base[blockIdx.x] + inc - own;
    for (uint32_t w = 0; w < wave; ++w) at += s_w[w];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int b = 0; b < 4; ++b)
            if ((f[k] >> (8 * b)) & 1u) rows[at++] = first + 4u * k + b;
}

// bounds[o] = position of the first touched row >= o * shard (o = 0 .. world: shards, then the tail), bounds[world + 1] = total
__global__ void k_owner_bounds(const uint32_t* __restrict__ rows, const uint32_t* __restrict__ d_total, uint32_t shard,
                               uint32_t world, uint32_t* __restrict__ bounds)
{
    const uint32_t o = threadIdx.x;
    if (o > world + 1u) return;
    const uint32_t total = *d_total;
    if (o == world + 1u) {
        bounds[o] = total;
        return;
    }
    const uint64_t key = (uint64_t)o * shard; // (o == world: the tail's first row)
    uint32_t       lo = 0, hi = total;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if ((uint64_t)rows[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    bounds[o] = lo;
}

struct AttrPtrs {
    float *  p0, *p1, *p2, *p3, *p4; // pos, scale, rotq, sh, opacity
    uint32_t feat;                  // floats per SH row
};

// word w of a message of `count` rows: [0, count) indices, then the five attribute blocks
template <bool PACK>
__global__ void __launch_bounds__(256) k_sparse_rows(AttrPtrs a, const uint32_t* __restrict__ rows_or_null, float* __restrict__ msg,
                                                     uint32_t count, uint32_t row_lo, uint32_t row_hi)
{
    const uint64_t  b1 = count, b2 = b1 + 3ull * count, b3 = b2 + 3ull * count, b4 = b3 + 4ull * count,
                   b5 = b4 + (uint64_t)a.feat * count, words = b5 + count;
    const uint32_t* idx = PACK ? rows_or_null : reinterpret_cast<const uint32_t*>(msg);
    for (uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x; w < words; w += (uint64_t)gridDim.x * 256) {
        if (w < b1) {
            if (PACK) reinterpret_cast<uint32_t*>(msg)[w] = idx[w];
            continue;
        }
        // (selects, not an indexed table: a dynamically indexed kernel argument would live in scratch memory)
        float*   base = a.p0;
        uint32_t wd   = 3u;
        uint64_t b0   = b1;
        if (w >= b2) { base = a.p1; b0 = b2; }
        if (w >= b3) { base = a.p2; b0 = b3; wd = 4u; }
        if (w >= b4) { base = a.p3; b0 = b4; wd = a.feat; }
        if (w >= b5) { base = a.p4; b0 = b5; wd = 1u; }
        const uint32_t rel = (uint32_t)(w - b0);
        const uint32_t j = rel / wd, e = rel - j * wd;
        const uint32_t row = idx[j];
        // accumulate: the indices come out of a RECEIVED message -- a row outside the caller's range (a short, corrupt or
        // mismatched-P message) is dropped, never written through
        if (!PACK && !(row >= row_lo && row < row_hi)) continue;
        float* g = base + (size_t)row * wd + e;
        if (PACK) msg[w] = *g;
        else *g = *g + msg[w];
    }
}

unsigned grid_words(uint64_t words)
{
    const uint64_t b = (words + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 32768 ? 32768 : b));
}

} // namespace

size_t sparse_flag_bytes(int64_t P) { return (size_t)div_up64(P, kFlagsPerChunk) * kFlagsPerChunk; }
uint32_t sparse_flag_chunks(int64_t P) { return (uint32_t)div_up64(P, kFlagsPerChunk); }

void launch_mark_rows(const uint32_t* vis_index, const uint32_t* d_counts, uint8_t* flags, int64_t P, int64_t hint_V,
                      hipStream_t stream)
{
    if (P <= 0) return;
    const int64_t n = hint_V > 0 ? hint_V : P;
    hipLaunchKernelGGL(k_mark_rows, dim3(grid_words((uint64_t)n)), dim3(256), 0, stream, vis_index, d_counts, flags, (uint32_t)P);
}

// flags (sparse_flag_bytes(P), zero beyond P) -> rows[0 .. *d_total) ascending; chunk_ws: sparse_flag_chunks(P) x u32
void launch_compact_flags(const uint8_t* flags, int64_t P, uint32_t* chunk_ws, uint32_t* rows, uint32_t* d_total,
                          hipStream_t stream, const uint32_t* copy_src, uint32_t* copy_dst, const uint32_t* box_word0,
                          uint32_t* host_box, uint32_t serial)
{
    const uint32_t chunks = sparse_flag_chunks(P);
    if (chunks == 0) {
        (void)hipMemsetAsync(d_total, 0, 4, stream);
        if (copy_dst) (void)hipMemcpyAsync(copy_dst, copy_src, 4, hipMemcpyDeviceToDevice, stream);
        (void)host_box; // (P == 0 never comes with a host box: the caller reads the scalars back itself)
        return;
    }
    hipLaunchKernelGGL(k_flag_count, dim3(chunks), dim3(256), 0, stream, flags, (uint32_t)P, chunk_ws);
    hipLaunchKernelGGL(k_flag_scan, dim3(1), dim3(1024), 0, stream, chunk_ws, chunks, d_total, copy_src, copy_dst,
                       box_word0, (volatile uint32_t*)host_box, serial);
    hipLaunchKernelGGL(k_flag_scatter, dim3(chunks), dim3(256), 0, stream, flags, (uint32_t)P, chunk_ws, rows);
}

void launch_owner_bounds(const uint32_t* rows, const uint32_t* d_total, int64_t shard, int world, uint32_t* d_bounds,
                         hipStream_t stream)
{
    hipLaunchKernelGGL(k_owner_bounds, dim3(1), dim3(128), 0, stream, rows, d_total, (uint32_t)shard, (uint32_t)world, d_bounds);
}

int64_t sparse_message_words(int64_t count, int sh_degree)
{
    return count * (1 + 3 + 3 + 4 + (int64_t)(sh_degree + 1) * (sh_degree + 1) * 3 + 1);
}

static AttrPtrs attr_ptrs(float* const g[5], int sh_degree)
{
    return { g[0], g[1], g[2], g[3], g[4], (uint32_t)((sh_degree + 1) * (sh_degree + 1) * 3) };
}

void launch_sparse_pack(float* const grads[5], int sh_degree, const uint32_t* rows, int64_t count, float* msg, hipStream_t stream)
{
    if (count <= 0) return;
    hipLaunchKernelGGL((k_sparse_rows<true>), dim3(grid_words((uint64_t)sparse_message_words(count, sh_degree))), dim3(256), 0,
                       stream, attr_ptrs(grads, sh_degree), rows, msg, (uint32_t)count, 0u, 0xFFFFFFFFu);
}

void launch_sparse_accumulate(float* const grads[5], int sh_degree, const float* msg, int64_t count, int64_t row_first,
                              int64_t row_count, hipStream_t stream)
{
    if (count <= 0 || row_count <= 0) return;
    hipLaunchKernelGGL((k_sparse_rows<false>), dim3(grid_words((uint64_t)sparse_message_words(count, sh_degree))), dim3(256), 0,
                       stream, attr_ptrs(grads, sh_degree), (const uint32_t*)nullptr, const_cast<float*>(msg), (uint32_t)count,
                       (uint32_t)row_first, (uint32_t)(row_first + row_count));
}

} // namespace lcgs
