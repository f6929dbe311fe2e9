// comm.cpp -- multi-GPU side of the C ABI: one process per GPU, the scene replicated, every rank renders
// its own view(s); the dense per-splat gradients are summed over the ranks with RCCL over xGMI.  The reference is
// single-device (app/main.cpp:162-163): everything here is new functionality behind the same boundary.
//
// RCCL is bound at run time (dlopen), not at link time: a process that already carries a copy -- torch ships its own
// librccl.so next to its libamdhip64.so -- keeps using that one (two copies of a HIP-facing runtime in one process do
// not end well), a process without one loads the ROCm installation's, and liblcgs_hip.so still loads on a machine
// where RCCL is absent (the lcgs_comm_* calls then fail with a message; nothing else needs it).
//
// Two ways through a training step at N > 1, both exact in f32:
//   lcgs_grads_allreduce     in-place sum of the five dense gradient arrays, issued as splat-range CHUNKS on a
//                            dedicated stream: the dense backward runs its preprocess pass as slices and records an event
//                            behind each (abi_backward.cpp render_backward), so chunk k is on the wire while slices k+1.. are
//                            still being computed.  (SURVEY 8e sketched per-attribute chunks; one kernel writes all five
//                            attributes of a splat, so the chunks are row ranges -- same idea, same bytes.)
//   lcgs_adam_step_sharded   reduce-scatter -> Adam on the rank's own rows -> all-gather of the refreshed activated
//                            arrays.  The wire carries what the all-reduce carries ((N-1)/N S out and in per GPU, twice),
//                            but the optimiser touches P/N rows per GPU instead of P (2.2 ms -> 0.27 ms at N = 8 for the
//                            bicycle stand-in), and moments / raw parameters are only ever needed for the own rows.
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include <rccl/rccl.h> // types and enums only: every entry point is resolved with dlsym

#include "../abi_internal.hpp"

using namespace lcgs;

namespace
{

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*)                                                                = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int)                                         = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t)                                                                   = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t)            = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)                    = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)                          = nullptr;
    ncclResult_t (*GroupStart)()                                                                              = nullptr;
    ncclResult_t (*GroupEnd)()                                                                                = nullptr;
    const char* (*GetErrorString)(ncclResult_t)                                                               = nullptr;
    std::string error; // why loading failed
};

RcclApi& rccl()
{
    static RcclApi       api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
        for (const char* n : names) // a copy the process already carries (torch's) wins
            if (!api.handle) api.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        for (const char* n : names)
            if (!api.handle) api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!api.handle) {
            const char* e = dlerror();
            api.error     = std::string("RCCL is not available (dlopen librccl.so.1: ") + (e ? e : "?") + ")";
            return;
        }
        bool ok = true;
        auto bind = [&](auto& fn, const char* sym) {
            fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(api.handle, sym));
            if (!fn) {
                ok        = false;
                api.error = std::string("librccl lacks ") + sym;
            }
        };
        bind(api.GetUniqueId, "ncclGetUniqueId");
        bind(api.CommInitRank, "ncclCommInitRank");
        bind(api.CommDestroy, "ncclCommDestroy");
        bind(api.AllReduce, "ncclAllReduce");
        bind(api.ReduceScatter, "ncclReduceScatter");
        bind(api.AllGather, "ncclAllGather");
        bind(api.Send, "ncclSend");
        bind(api.Recv, "ncclRecv");
        bind(api.GroupStart, "ncclGroupStart");
        bind(api.GroupEnd, "ncclGroupEnd");
        bind(api.GetErrorString, "ncclGetErrorString");
        if (!ok) api.handle = nullptr;
    });
    return api;
}

lcgs_status rccl_fail(ncclResult_t r, const char* what, int line)
{
    char buf[384];
    snprintf(buf, sizeof(buf), "RCCL error %d (%s) in `%s` at comm.cpp:%d", (int)r,
             rccl().GetErrorString ? rccl().GetErrorString(r) : "?", what, line);
    set_last_error(buf);
    return LCGS_ERR_HIP;
}

#define LCGS_RCCL_CHECK(expr)                                              \
    do {                                                                   \
        ncclResult_t _r = (expr);                                          \
        if (_r != ncclSuccess) return rccl_fail(_r, #expr, __LINE__);      \
    } while (0)

#define LCGS_TRY(expr)                    \
    do {                                  \
        lcgs_status _s = (expr);          \
        if (_s != LCGS_OK) return _s;     \
    } while (0)

lcgs_status need_rccl()
{
    if (rccl().handle) return LCGS_OK;
    set_last_error(rccl().error.empty() ? "RCCL is not available" : rccl().error);
    return LCGS_ERR_NO_DEVICE;
}

struct AttrRows {
    float* ptr[5];
    size_t width[5]; // floats per splat: pos 3, scale 3, rotq 4, sh (deg+1)^2*3, opacity 1
};

AttrRows attr_rows(const lcgs_grads* g, int sh_degree)
{
    const size_t feat = (size_t)(sh_degree + 1) * (sh_degree + 1) * 3;
    return { { g->d_dL_dpos, g->d_dL_dscale, g->d_dL_drotq, g->d_dL_dsh, g->d_dL_dopacity }, { 3, 3, 4, feat, 1 } };
}

AttrRows attr_rows(const lcgs_params* p, int sh_degree)
{
    const size_t feat = (size_t)(sh_degree + 1) * (sh_degree + 1) * 3;
    return { { p->pos, p->scale, p->rotq, p->sh, p->opacity }, { 3, 3, 4, feat, 1 } };
}

} // namespace

// An in-process rendezvous for N communicators on ONE device (lcgs_loopback_*): N contexts, one host thread each, standing in
// for N ranks.  Carries what the ownership step needs -- a small all-gather and grouped sends / receives, as device-to-
// device copies ordered by events -- so that the step's C code path (message layout, offsets, slot state, ordering) runs
// with N > 1 participants on a single GPU, where RCCL refuses a second rank.  Not a transport for production.
struct lcgs_loopback_group {
    int                     world = 0;
    std::mutex              mu;
    std::condition_variable cv;
    int                     arrived = 0;
    uint64_t                generation = 0;
    bool                    failed = false; // a member gave up: everybody leaves the barriers with an error
    std::vector<uint32_t>   table;          // all-gather staging: world x count words
    struct Msg {
        const void* ptr;
        size_t      bytes;
        hipEvent_t  ready; // recorded on the sender's stream behind the data
    };
    std::vector<std::deque<Msg>> box;  // box[dst * world + src]: the sends posted in the open group, in order
    std::vector<hipEvent_t>      done; // per rank: behind the copies of its receives of the last group
    int                          members = 0;
    int                          device  = -1;  // every member's device (one GPU: that is the point)
    std::vector<char>            taken;         // ranks that have a communicator
    // messages nobody received (a member gave up mid-group): their events are not leaked
    void drop_unconsumed()
    {
        for (auto& q : box) {
            for (Msg& m : q)
                if (m.ready) (void)hipEventDestroy(m.ready);
            q.clear();
        }
    }
    // collectives of the open group (all-reduce / reduce-scatter / all-gather): what every rank passed, op by op
    struct CollArgs {
        const float* send;
        float*       recv;
    };
    std::vector<std::vector<CollArgs>> coll;    // coll[op][rank]
    std::vector<float*>                scratch; // per rank: where it leaves its reduced slices (phase 1 of an all-reduce)
    std::vector<hipEvent_t>            ready, reduced; // per rank: inputs complete / phase 1 complete

    bool barrier() // false: the group failed
    {
        std::unique_lock<std::mutex> lock(mu);
        if (failed) return false;
        const uint64_t g = generation;
        if (++arrived == world) {
            arrived = 0;
            ++generation;
            cv.notify_all();
        } else {
            cv.wait(lock, [&] { return generation != g || failed; });
        }
        return !failed;
    }
    void fail()
    {
        std::lock_guard<std::mutex> lock(mu);
        failed = true;
        drop_unconsumed();
        cv.notify_all();
    }
};

struct lcgs_comm {
    lcgs_context* ctx    = nullptr;
    ncclComm_t    comm   = nullptr;
    int           rank   = 0, world = 1;
    hipStream_t   stream = nullptr; // the collectives' own stream: they overlap the compute stream's tail
    hipEvent_t    ev_in = nullptr, ev_out = nullptr;
    // opt-in f16 transport (lcgs_comm_set_transport): staging for the packed gradients and the five scales
    int          transport = LCGS_TRANSPORT_F32;
    DeviceBuffer packed, scales; // 59 P halfs; 5 magnitudes | 5 scales | 5 inverses (floats)
    int          device = 0;     // (kept beyond the context's life: lcgs_comm_destroy selects it)
    // sparse exchange (lcgs_adam_step_sparse): touched-row flags of the current step, their compaction, the messages
    bool         track_rows = false;
    int64_t      flags_P    = 0;     // rows the flag array covers
    DeviceBuffer flags, chunk_ws, rows, bounds, matrix, sendbuf, recvbuf; // bounds: [world + 2] positions + [1] total
    uint32_t*    h_matrix = nullptr; // pinned: world x (world + 2) positions (row r = rank r's owner bounds)
    lcgs_comm_stats stats{};
    // splat-ownership step (lcgs_owner_step_forward / _backward): my rows' records for every view of the step, what I
    // received for my view (owner order), its 2-D gradients, and the 2-D gradients of my rows that came back
    lcgs_loopback_group* loop = nullptr; // set: an in-process communicator (lcgs_comm_create_loopback), comm == NULL
    bool         self_p2p = false;       // test hook LCGS_OWNER_SELF_P2P=1: my own share travels through send / recv too
    DeviceBuffer own_rows, own_recs, in_rows, in_recs, g2d_all, g_in;
    DeviceBuffer loop_scratch; // loopback: this rank's reduced slices of the open group's all-reduces
    struct LoopOp {
        int          kind; // 0 all-reduce (in place), 1 reduce-scatter, 2 all-gather
        const float* send;
        float*       recv;
        size_t       count; // all-reduce: elements; the others: elements per rank
    };
    std::vector<LoopOp> loop_ops; // loopback: the collectives of the open group, executed at its end
    // ... without a read-back (lcgs_owner_step_set_async): message sizes come from the PREVIOUS step's all-gathered counts
    // (x 1.25 + 1024: every rank derives the same table), the true counts stay on the device, a clipped message or a
    // truncated frame raises a flag that is max-reduced over the ranks and read by lcgs_owner_step_finish -- the redo is
    // everybody's or nobody's
    bool         owner_async = false, force_sync_once = false;
    struct {
        bool                  have = false;
        int                   world = 0;
        int64_t               P = 0;
        std::vector<uint32_t> table; // [o * N + v]: rows of owner o on view v's screen, last step
    } prev;
    uint32_t*    h_next = nullptr;   // pinned: the table of the step in flight [N x N] + the reduced flag [1]
    DeviceBuffer flag_dev;           // u32: bit 0 a message was clipped, bit 1 a frame's pairs were truncated (any rank)
    hipEvent_t   ev_checked = nullptr; // behind the flag's reduction and the copies to h_next
    struct {
        bool     valid = false;
        bool     async = false;                 // the step in flight used padded messages (sizes below)
        int64_t  cap_in[LCGS_MAX_RANKS]{};      // rows owner o's message to me holds (>= its true count, else clipped)
        int64_t  cap_out[LCGS_MAX_RANKS]{};     // rows my message to view v holds
        int64_t  n_all = 0;                     // rows on my view's screen (all owners)
        int64_t  in_off[LCGS_MAX_RANKS + 1]{};  // owner o's rows start here in in_rows / in_recs / g2d_all
        uint32_t out[LCGS_MAX_RANKS]{};         // my rows on view v's screen (= table[me][v])
        std::vector<std::pair<void*, std::pair<size_t, int>>> recvs; // loopback: receives of the open group
    } own;
};

namespace lcgs
{
void comm_forget_context(lcgs_comm* c)
{
    if (!c) return;
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    c->ctx = nullptr;
}
} // namespace lcgs

namespace
{
// ---------------------------------------------------------------------------------------------------------------------
// The transport of one communicator: RCCL over xGMI, or the in-process loopback (N contexts on one device, one host thread
// each).  Every collective and point-to-point call of this file goes through it, so the code above this seam -- chunking,
// shard and message arithmetic, stream ordering -- is the same whichever carries the bytes.  Calls between group_begin and
// group_end form one group (RCCL: ncclGroupStart / End; loopback: recorded, executed by group_end, which every member of
// the group reaches with the same sequence of calls).  Streams: everything is enqueued on the communicator's stream.
// ---------------------------------------------------------------------------------------------------------------------
struct Wire {
    lcgs_comm* c;

    lcgs_status loop_failed()
    {
        set_last_error("loopback: another member of the group failed");
        return LCGS_ERR_STATE;
    }
    lcgs_status group_begin()
    {
        if (!c->loop) LCGS_RCCL_CHECK(rccl().GroupStart());
        else {
            c->own.recvs.clear();
            c->loop_ops.clear();
        }
        return LCGS_OK;
    }
    // (RCCL: an error inside an open group closes it before it is reported)
    lcgs_status rccl_call(ncclResult_t r, const char* what, int line)
    {
        if (r == ncclSuccess) return LCGS_OK;
        (void)rccl().GroupEnd();
        return rccl_fail(r, what, line);
    }
    lcgs_status allreduce_sum(float* p, size_t count) // in place
    {
        if (!c->loop) return rccl_call(rccl().AllReduce(p, p, count, ncclFloat32, ncclSum, c->comm, c->stream), "ncclAllReduce", __LINE__);
        c->loop_ops.push_back({ 0, p, p, count });
        return LCGS_OK;
    }
    lcgs_status reduce_scatter_sum(const float* send, float* recv, size_t count_per_rank)
    {
        if (!c->loop)
            return rccl_call(rccl().ReduceScatter(send, recv, count_per_rank, ncclFloat32, ncclSum, c->comm, c->stream),
                             "ncclReduceScatter", __LINE__);
        c->loop_ops.push_back({ 1, send, recv, count_per_rank });
        return LCGS_OK;
    }
    lcgs_status allgather(const float* send, float* recv, size_t count_per_rank)
    {
        if (!c->loop)
            return rccl_call(rccl().AllGather(send, recv, count_per_rank, ncclFloat32, c->comm, c->stream), "ncclAllGather", __LINE__);
        c->loop_ops.push_back({ 2, send, recv, count_per_rank });
        return LCGS_OK;
    }
    lcgs_status send(const void* d_buf, size_t bytes, int peer)
    {
        if (!c->loop) return rccl_call(rccl().Send(d_buf, bytes, ncclUint8, peer, c->comm, c->stream), "ncclSend", __LINE__);
        lcgs_loopback_group::Msg m{ d_buf, bytes, nullptr };
        LCGS_HIP_CHECK(hipEventCreateWithFlags(&m.ready, hipEventDisableTiming));
        LCGS_HIP_CHECK(hipEventRecord(m.ready, c->stream));
        std::lock_guard<std::mutex> lock(c->loop->mu);
        c->loop->box[(size_t)peer * c->loop->world + c->rank].push_back(m);
        return LCGS_OK;
    }
    lcgs_status recv(void* d_buf, size_t bytes, int peer)
    {
        if (!c->loop) return rccl_call(rccl().Recv(d_buf, bytes, ncclUint8, peer, c->comm, c->stream), "ncclRecv", __LINE__);
        c->own.recvs.push_back({ d_buf, { bytes, peer } });
        return LCGS_OK;
    }
    // a small all-gather of `count` words per rank, outside any group (message sizes: the host reads the result back)
    lcgs_status allgather_u32(const uint32_t* d_send, uint32_t* d_recv, size_t count)
    {
        if (!c->loop) {
            LCGS_RCCL_CHECK(rccl().AllGather(d_send, d_recv, count, ncclUint32, c->comm, c->stream));
            return LCGS_OK;
        }
        lcgs_loopback_group* g = c->loop;
        std::vector<uint32_t> mine(count);
        LCGS_HIP_CHECK(hipMemcpyAsync(mine.data(), d_send, count * 4, hipMemcpyDeviceToHost, c->stream));
        LCGS_HIP_CHECK(hipStreamSynchronize(c->stream));
        {
            std::lock_guard<std::mutex> lock(g->mu);
            if (g->table.size() < (size_t)g->world * count) g->table.resize((size_t)g->world * count);
            std::copy(mine.begin(), mine.end(), g->table.begin() + (size_t)c->rank * count);
        }
        if (!g->barrier()) return loop_failed();
        std::vector<uint32_t> all;
        {
            std::lock_guard<std::mutex> lock(g->mu);
            all.assign(g->table.begin(), g->table.begin() + (size_t)g->world * count);
        }
        LCGS_HIP_CHECK(hipMemcpy(d_recv, all.data(), all.size() * 4, hipMemcpyHostToDevice));
        if (!g->barrier()) return loop_failed(); // nobody overwrites the table before everybody has read it
        return LCGS_OK;
    }
    // a few words max-reduced in place, outside any group (the ownership step's redo flag)
    lcgs_status allreduce_max_u32(uint32_t* d_buf, size_t count)
    {
        if (!c->loop) {
            LCGS_RCCL_CHECK(rccl().AllReduce(d_buf, d_buf, count, ncclUint32, ncclMax, c->comm, c->stream));
            return LCGS_OK;
        }
        LCGS_REQUIRE(count <= 8, "loopback: the flag reduction carries a few words");
        const int N = c->loop->world;
        DeviceBuffer all;
        LCGS_TRY(all.ensure((size_t)N * count * 4));
        lcgs_status s = allgather_u32(d_buf, all.as<uint32_t>(), count); // (the loopback's rendezvous is host-side anyway)
        uint32_t    h[8 * LCGS_MAX_RANKS], m[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        if (s == LCGS_OK && hipMemcpy(h, all.ptr, (size_t)N * count * 4, hipMemcpyDeviceToHost) != hipSuccess) s = LCGS_ERR_HIP;
        all.release();
        LCGS_TRY(s);
        for (int r = 0; r < N; ++r)
            for (size_t k = 0; k < count; ++k) m[k] = std::max(m[k], h[(size_t)r * count + k]);
        LCGS_HIP_CHECK(hipMemcpyAsync(d_buf, m, count * 4, hipMemcpyHostToDevice, c->stream));
        LCGS_HIP_CHECK(hipStreamSynchronize(c->stream)); // (m lives on this stack)
        return LCGS_OK;
    }
    lcgs_status group_end()
    {
        if (!c->loop) {
            LCGS_RCCL_CHECK(rccl().GroupEnd());
            return LCGS_OK;
        }
        lcgs_loopback_group* g  = c->loop;
        const int            N  = g->world, me = c->rank;
        hipStream_t          st = c->stream;
        // ---- what this rank's collectives of the group need as scratch: its slice of every all-reduce
        size_t scratch_elems = 0;
        for (const auto& op : c->loop_ops)
            if (op.kind == 0) scratch_elems += (op.count + N - 1) / N;
        LCGS_TRY(c->loop_scratch.ensure(scratch_elems * 4 + 16));
        {
            std::lock_guard<std::mutex> lock(g->mu);
            if (g->coll.size() < c->loop_ops.size()) g->coll.resize(c->loop_ops.size());
            for (size_t k = 0; k < c->loop_ops.size(); ++k) {
                g->coll[k].resize(N);
                g->coll[k][me] = { c->loop_ops[k].send, c->loop_ops[k].recv };
            }
            g->scratch[me] = c->loop_scratch.as<float>();
        }
        LCGS_HIP_CHECK(hipEventRecord(g->ready[me], st)); // my inputs (and my posted sends) are complete behind this
        if (!g->barrier()) return loop_failed();           // every rank has posted its sends and its collectives' buffers
        for (int r = 0; r < N; ++r)
            if (r != me) LCGS_HIP_CHECK(hipStreamWaitEvent(st, g->ready[r], 0));
        // ---- point-to-point: copy what was sent to me
        std::vector<hipEvent_t> consumed;
        for (auto& r : c->own.recvs) {
            lcgs_loopback_group::Msg m{};
            {
                std::lock_guard<std::mutex> lock(g->mu);
                auto& q = g->box[(size_t)me * N + r.second.second];
                if (q.empty() || q.front().bytes != r.second.first) {
                    g->failed = true;
                    g->cv.notify_all();
                    set_last_error("loopback: a receive has no matching send of the same size (ranks disagree on the message table)");
                    return LCGS_ERR_STATE;
                }
                m = q.front();
                q.pop_front();
            }
            LCGS_HIP_CHECK(hipStreamWaitEvent(st, m.ready, 0));
            if (m.bytes) LCGS_HIP_CHECK(hipMemcpyAsync(r.first, m.ptr, m.bytes, hipMemcpyDeviceToDevice, st));
            consumed.push_back(m.ready);
        }
        // ---- collectives, phase 1: reductions that only READ the other ranks' buffers (sums in rank order)
        std::vector<lcgs_loopback_group::CollArgs> args; // (a private copy: the shared table is re-used by the next group)
        size_t                                      at = 0;
        for (size_t k = 0; k < c->loop_ops.size(); ++k) {
            const auto& op = c->loop_ops[k];
            {
                std::lock_guard<std::mutex> lock(g->mu);
                args = g->coll[k];
            }
            const float* srcs[16];
            if (op.kind == 0) { // all-reduce: I reduce slice `me` of everybody's array into my scratch
                const size_t per = (op.count + N - 1) / N, lo = std::min(op.count, per * (size_t)me),
                             n = std::min(op.count, lo + per) - lo;
                for (int r = 0; r < N; ++r) srcs[r] = args[r].send + lo;
                launch_sum_sources(srcs, N, n, c->loop_scratch.as<float>() + at, st);
                at += per;
            } else if (op.kind == 1) { // reduce-scatter: my slice of everybody's send array, straight into my recv
                for (int r = 0; r < N; ++r) srcs[r] = args[r].send + op.count * (size_t)me;
                launch_sum_sources(srcs, N, op.count, op.recv, st);
            }
        }
        LCGS_HIP_CHECK(hipGetLastError());
        LCGS_HIP_CHECK(hipEventRecord(g->reduced[me], st));
        if (!g->barrier()) return loop_failed(); // every rank's phase 1 is enqueued
        for (int r = 0; r < N; ++r)
            if (r != me) LCGS_HIP_CHECK(hipStreamWaitEvent(st, g->reduced[r], 0));
        // ---- phase 2: gathers that WRITE my own arrays from what the others left (their scratch slices / send arrays)
        std::vector<float*> scr;
        {
            std::lock_guard<std::mutex> lock(g->mu);
            scr = g->scratch;
        }
        at = 0;
        for (size_t k = 0; k < c->loop_ops.size(); ++k) {
            const auto& op = c->loop_ops[k];
            {
                std::lock_guard<std::mutex> lock(g->mu);
                args = g->coll[k];
            }
            if (op.kind == 0) {
                const size_t per = (op.count + N - 1) / N;
                for (int r = 0; r < N; ++r) {
                    const size_t lo = std::min(op.count, per * (size_t)r), n = std::min(op.count, lo + per) - lo;
                    if (n) LCGS_HIP_CHECK(hipMemcpyAsync(op.recv + lo, scr[r] + at, n * 4, hipMemcpyDeviceToDevice, st));
                }
                at += per;
            } else if (op.kind == 2) {
                for (int r = 0; r < N; ++r)
                    if (op.count && args[r].send != op.recv + op.count * (size_t)r) // (in place: my own slice is where it belongs)
                        LCGS_HIP_CHECK(hipMemcpyAsync(op.recv + op.count * (size_t)r, args[r].send, op.count * 4,
                                                      hipMemcpyDeviceToDevice, st));
            }
        }
        LCGS_HIP_CHECK(hipEventRecord(g->done[me], st));
        if (!g->barrier()) return loop_failed(); // every copy out of my buffers / scratch has been enqueued
        for (int p = 0; p < N; ++p)               // ... and has run before I touch them again
            if (p != me) LCGS_HIP_CHECK(hipStreamWaitEvent(st, g->done[p], 0));
        for (hipEvent_t e : consumed) (void)hipEventDestroy(e);
        if (!g->barrier()) return loop_failed(); // (the shared events are not re-recorded before everybody has waited on them)
        return LCGS_OK;
    }
};

// a member of a loopback group that leaves a step early (any error) releases the others from their barriers
struct LoopGuard {
    lcgs_comm* c;
    bool       ok = false;
    ~LoopGuard()
    {
        if (!ok && c && c->loop) c->loop->fail();
    }
};

size_t row_bytes(int sh_degree) { return (size_t)(3 + 3 + 4 + (sh_degree + 1) * (sh_degree + 1) * 3 + 1) * 4; }

// lcgs_adam_step on the rank's own rows [first, first + count) and on the tail rows every rank keeps (fewer than N)
lcgs_status adam_own_rows(lcgs_context* ctx, lcgs_comm* c, int64_t P, int sh_degree, const lcgs_adam_config* cfg,
                          const AttrRows& g, const lcgs_params* raw, const lcgs_params* m, const lcgs_params* v,
                          const lcgs_params* activated)
{
    int64_t first = 0, count = 0;
    lcgs_comm_shard_rows(P, c->world, c->rank, &first, &count);
    const int64_t tail0 = count * c->world, tail = P - tail0;
    auto sub = [&](const lcgs_params* p, int64_t row) {
        const AttrRows a = attr_rows(p, sh_degree);
        lcgs_params    o;
        o.pos = a.ptr[0] + (size_t)row * a.width[0]; o.scale = a.ptr[1] + (size_t)row * a.width[1];
        o.rotq = a.ptr[2] + (size_t)row * a.width[2]; o.sh = a.ptr[3] + (size_t)row * a.width[3];
        o.opacity = a.ptr[4] + (size_t)row * a.width[4];
        return o;
    };
    auto step_rows = [&](int64_t row, int64_t rows) -> lcgs_status {
        if (rows <= 0) return LCGS_OK;
        lcgs_grads gg = { g.ptr[0] + (size_t)row * g.width[0], g.ptr[1] + (size_t)row * g.width[1],
                          g.ptr[2] + (size_t)row * g.width[2], g.ptr[3] + (size_t)row * g.width[3],
                          g.ptr[4] + (size_t)row * g.width[4] };
        const lcgs_params r_ = sub(raw, row), m_ = sub(m, row), v_ = sub(v, row), a_ = sub(activated, row);
        return lcgs_adam_step(ctx, (int)rows, sh_degree, cfg, &gg, &r_, &m_, &v_, &a_);
    };
    LCGS_TRY(step_rows(first, count));
    return step_rows(tail0, tail);
}

// all-gather of the refreshed ACTIVATED rows (what every rank's renderer reads).  Raw parameters and moments stay
// authoritative on their owner only (plus the tail everywhere).
lcgs_status allgather_activated(lcgs_context* ctx, lcgs_comm* c, int64_t P, int sh_degree, const AttrRows& act)
{
    int64_t first = 0, count = 0;
    lcgs_comm_shard_rows(P, c->world, c->rank, &first, &count);
    // the other ranks' rows land in these arrays: whatever a context derived from them (the cull pass's 16-byte rows) is
    // stale from here on -- also on a rank whose own shard is empty and whose lcgs_adam_step therefore wrote nothing
    abi::scene_arrays_written(ctx, act.ptr[0], act.ptr[1], act.ptr[2]);
    LCGS_HIP_CHECK(hipEventRecord(c->ev_in, ctx->stream));
    LCGS_HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_in, 0));
    if (count > 0) {
        Wire wire{ c };
        LCGS_TRY(wire.group_begin());
        for (int i = 0; i < 5; ++i)
            LCGS_TRY(wire.allgather(act.ptr[i] + (size_t)first * act.width[i], act.ptr[i], (size_t)count * act.width[i]));
        LCGS_TRY(wire.group_end());
        c->stats.collective_groups += 1;
        const int64_t b = (int64_t)((uint64_t)(c->world - 1) * (uint64_t)count * row_bytes(sh_degree));
        c->stats.bytes_sent += b;
        c->stats.bytes_received += b;
    }
    LCGS_HIP_CHECK(hipEventRecord(c->ev_out, c->stream));
    LCGS_HIP_CHECK(hipStreamWaitEvent(ctx->stream, c->ev_out, 0));
    return LCGS_OK;
}
} // namespace

extern "C" {

lcgs_status lcgs_comm_unique_id(lcgs_comm_id* out)
{
    LCGS_REQUIRE(out != nullptr, "out is NULL");
    static_assert(sizeof(lcgs_comm_id) == sizeof(ncclUniqueId), "lcgs_comm_id must hold an ncclUniqueId");
    LCGS_TRY(need_rccl());
    ncclUniqueId id;
    LCGS_RCCL_CHECK(rccl().GetUniqueId(&id));
    memcpy(out->bytes, id.internal, sizeof(id.internal));
    return LCGS_OK;
}

void lcgs_comm_shard_rows(int64_t num_gaussians, int world_size, int rank, int64_t* first, int64_t* count)
{
    // equal shards of floor(P / N) rows; the P mod N rows behind them ("the tail") belong to every rank
    const int64_t c = world_size > 0 ? num_gaussians / world_size : num_gaussians;
    if (first) *first = c * rank;
    if (count) *count = c;
}

lcgs_status lcgs_comm_create(lcgs_context* ctx, const lcgs_comm_id* id, int rank, int world_size, lcgs_comm** out)
{
    LCGS_REQUIRE(ctx && id && out, "NULL argument");
    *out = nullptr;
    LCGS_REQUIRE(world_size >= 1 && rank >= 0 && rank < world_size, "rank / world_size out of range");
    LCGS_REQUIRE(world_size <= LCGS_MAX_RANKS, "world_size above LCGS_MAX_RANKS");
    LCGS_REQUIRE(ctx->comm == nullptr, "the context already has a communicator attached");
    LCGS_TRY(need_rccl());
    LCGS_HIP_CHECK(hipSetDevice(ctx->device));
    lcgs_comm* c = new (std::nothrow) lcgs_comm();
    if (!c) return LCGS_ERR_OUT_OF_MEMORY;
    c->ctx    = ctx;
    c->device = ctx->device;
    c->rank   = rank;
    c->world = world_size;
    // highest dispatch priority: the collective's workgroups should get their slots as soon as a chunk is ready, not queue
    // behind the backward's remaining slices (the auxiliary stream of the context has the LOWEST, for the opposite reason)
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi); // hi = numerically smallest = highest priority
    hipError_t e = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi);
    if (e != hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming);
    if (e != hipSuccess) {
        (void)lcgs_comm_destroy(c);
        LCGS_HIP_CHECK(e);
    }
    ncclUniqueId nid;
    memcpy(nid.internal, id->bytes, sizeof(nid.internal));
    ncclResult_t r = rccl().CommInitRank(&c->comm, world_size, nid, rank);
    if (r != ncclSuccess) {
        c->comm = nullptr;
        (void)lcgs_comm_destroy(c);
        return rccl_fail(r, "ncclCommInitRank", __LINE__);
    }
    ctx->comm = c;
    if (const char* s = getenv("LCGS_OWNER_SELF_P2P")) c->self_p2p = s[0] == '1'; // test hook (see lcgs_owner_step_forward)
    // chunked all-reduce: the dense backward slices its preprocess pass (tuning hook LCGS_GRAD_SLICES; 1 = one chunk)
    int slices = 4;
    if (const char* s = getenv("LCGS_GRAD_SLICES")) slices = atoi(s);
    ctx->grad_slices = std::min(std::max(slices, 1), kMaxGradSlices);
    *out = c;
    return LCGS_OK;
}

lcgs_status lcgs_comm_destroy(lcgs_comm* c)
{
    if (!c) return LCGS_OK;
    (void)hipSetDevice(c->device); // (also after the context has gone: comm_forget_context)
    if (c->ctx) {
        if (c->ctx->comm == c) {
            c->ctx->comm        = nullptr;
            c->ctx->grad_slices = 1;
        }
    }
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm && rccl().CommDestroy) (void)rccl().CommDestroy(c->comm);
    if (c->loop) {
        std::lock_guard<std::mutex> lock(c->loop->mu);
        --c->loop->members;
        if (c->rank >= 0 && (size_t)c->rank < c->loop->taken.size()) c->loop->taken[(size_t)c->rank] = 0;
    }
    c->packed.release();
    c->scales.release();
    for (DeviceBuffer* b : { &c->flags, &c->chunk_ws, &c->rows, &c->bounds, &c->matrix, &c->sendbuf, &c->recvbuf, &c->own_rows,
                             &c->own_recs, &c->in_rows, &c->in_recs, &c->g2d_all, &c->g_in, &c->loop_scratch })
        b->release();
    if (c->h_matrix) (void)hipHostFree(c->h_matrix);
    if (c->h_next) (void)hipHostFree(c->h_next);
    c->flag_dev.release();
    if (c->ev_checked) (void)hipEventDestroy(c->ev_checked);
    if (c->ev_in) (void)hipEventDestroy(c->ev_in);
    if (c->ev_out) (void)hipEventDestroy(c->ev_out);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return LCGS_OK;
}

lcgs_status lcgs_comm_set_transport(lcgs_comm* c, int transport)
{
    LCGS_REQUIRE(c != nullptr, "comm is NULL");
    LCGS_REQUIRE(transport == LCGS_TRANSPORT_F32 || transport == LCGS_TRANSPORT_F16, "unknown transport");
    c->transport = transport;
    return LCGS_OK;
}

lcgs_status lcgs_comm_info(const lcgs_comm* c, int* rank, int* world_size)
{
    LCGS_REQUIRE(c != nullptr, "comm is NULL");
    if (rank) *rank = c->rank;
    if (world_size) *world_size = c->world;
    return LCGS_OK;
}

lcgs_status lcgs_grads_allreduce(lcgs_context* ctx, lcgs_comm* c, int num_gaussians, int sh_degree,
                                 const lcgs_grads* grads)
{
    LCGS_REQUIRE(ctx && c && grads, "NULL argument");
    LCGS_REQUIRE(c->ctx == ctx, "the communicator belongs to another (or a destroyed) context");
    LCGS_REQUIRE(c->loop == nullptr || c->transport == LCGS_TRANSPORT_F32, "the in-process (loopback) transport moves f32 only");
    LCGS_REQUIRE(num_gaussians >= 0 && sh_degree >= 0 && sh_degree <= 3, "bad num_gaussians / sh_degree");
    LCGS_REQUIRE(grads->d_dL_dpos && grads->d_dL_dscale && grads->d_dL_drotq && grads->d_dL_dsh && grads->d_dL_dopacity,
                 "NULL gradient buffer");
    if (num_gaussians == 0) return LCGS_OK;
    LCGS_HIP_CHECK(hipSetDevice(ctx->device));
    const AttrRows a = attr_rows(grads, sh_degree);
    const int64_t  P = num_gaussians;
    c->stats = lcgs_comm_stats{}; // "what the LAST collective call moved": reset on every path
    if (c->transport == LCGS_TRANSPORT_F16) {
        // Opt-in: the sum crosses the wire as f16 with one power-of-two scale per attribute, agreed by all ranks (the
        // magnitudes are max-reduced first).  One chunk behind the backward's tail: the scales need every row.
        size_t total = 0, start[5]; // (halfs; every attribute's region starts 16-byte aligned: vector stores on that side)
        for (int i = 0; i < 5; ++i) {
            start[i] = total;
            total += ((size_t)P * a.width[i] + 7) & ~(size_t)7;
        }
        const void* had = c->packed.ptr;
        LCGS_TRY(c->packed.ensure(total * 2));
        if (c->packed.ptr != had) LCGS_HIP_CHECK(hipMemsetAsync(c->packed.ptr, 0, total * 2, c->stream)); // the padding is summed too
        LCGS_TRY(c->scales.ensure(16 * sizeof(float)));
        float*    amax  = c->scales.as<float>();
        float*    scale = amax + 5;
        float*    inv   = amax + 10;
        uint16_t* pk    = c->packed.as<uint16_t>();
        LCGS_HIP_CHECK(hipEventRecord(c->ev_in, ctx->stream));
        LCGS_HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_in, 0));
        ctx->slices_recorded = 0;
        LCGS_HIP_CHECK(hipMemsetAsync(amax, 0, 5 * sizeof(float), c->stream));
        for (int i = 0; i < 5; ++i) launch_absmax(a.ptr[i], (size_t)P * a.width[i], reinterpret_cast<uint32_t*>(amax + i), c->stream);
        LCGS_RCCL_CHECK(rccl().AllReduce(amax, amax, 5, ncclFloat32, ncclMax, c->comm, c->stream));
        launch_transport_scales(amax, c->world, scale, inv, c->stream);
        for (int i = 0; i < 5; ++i) launch_pack_f16(a.ptr[i], (size_t)P * a.width[i], scale + i, pk + start[i], c->stream);
        LCGS_RCCL_CHECK(rccl().AllReduce(pk, pk, total, ncclFloat16, ncclSum, c->comm, c->stream));
        for (int i = 0; i < 5; ++i) launch_unpack_f16(pk + start[i], (size_t)P * a.width[i], inv + i, a.ptr[i], c->stream);
        LCGS_HIP_CHECK(hipGetLastError());
        LCGS_HIP_CHECK(hipEventRecord(c->ev_out, c->stream));
        LCGS_HIP_CHECK(hipStreamWaitEvent(ctx->stream, c->ev_out, 0));
        c->stats.collective_groups = 2; // the magnitudes, the packed sum
        c->stats.bytes_sent = c->stats.bytes_received =
            (int64_t)(2 * (uint64_t)(c->world - 1) * ((uint64_t)total * 2 + 5 * 4) / (uint64_t)c->world);
        return LCGS_OK;
    }
    // The NUMBER and the row ranges of the chunks come from values every rank shares (P, the slice count set when the
    // communicator was created) -- never from what this rank happened to do before the call: a rank without a view in
    // the last round of a batch, or with an empty frame, has run no backward and must still issue the very same
    // sequence of collectives as its peers (RCCL: mismatched counts are undefined behaviour).  Only what a chunk WAITS
    // for is local: the event of the backward slice that produced its rows when those events belong to these arrays,
    // else the tail of the context's stream.
    const int  K         = (ctx->grad_slices > 1 && P >= 4096) ? ctx->grad_slices : 1; // (render_backward's own rule)
    const bool by_slice  = K > 1 && ctx->slices_recorded == K && ctx->slices_of == (const void*)grads->d_dL_dpos &&
                          ctx->P == num_gaussians;
    LCGS_HIP_CHECK(hipEventRecord(c->ev_in, ctx->stream));
    if (!by_slice) LCGS_HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_in, 0));
    Wire      wire{ c };
    LoopGuard guard{ c };
    for (int k = 0; k < K; ++k) {
        if (by_slice) LCGS_HIP_CHECK(hipStreamWaitEvent(c->stream, ctx->ev_slice[k], 0));
        // the last chunk also waits for whatever was enqueued on the context's stream behind the backward
        if (by_slice && k == K - 1) LCGS_HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_in, 0));
        const int64_t r0 = (int64_t)(((uint64_t)P * (uint64_t)k) / (uint64_t)K);       // (k_slice_bounds' split)
        const int64_t r1 = (int64_t)(((uint64_t)P * (uint64_t)(k + 1)) / (uint64_t)K);
        if (r1 <= r0) continue;
        LCGS_TRY(wire.group_begin());
        for (int i = 0; i < 5; ++i) LCGS_TRY(wire.allreduce_sum(a.ptr[i] + (size_t)r0 * a.width[i], (size_t)(r1 - r0) * a.width[i]));
        LCGS_TRY(wire.group_end());
        c->stats.collective_groups += 1;
    }
    {
        size_t row = 0;
        for (int i = 0; i < 5; ++i) row += a.width[i] * 4;
        c->stats.bytes_sent = c->stats.bytes_received = (int64_t)(2 * (uint64_t)(c->world - 1) * (uint64_t)P * row / (uint64_t)c->world);
    }
    ctx->slices_recorded = 0; // consumed
    // whatever the caller enqueues next on the context's stream (the optimiser) sees the sums
    LCGS_HIP_CHECK(hipEventRecord(c->ev_out, c->stream));
    LCGS_HIP_CHECK(hipStreamWaitEvent(ctx->stream, c->ev_out, 0));
    guard.ok = true;
    return LCGS_OK;
}

lcgs_status lcgs_adam_step_sharded(lcgs_context* ctx, lcgs_comm* c, int num_gaussians, int sh_degree,
                                   const lcgs_adam_config* cfg, const lcgs_grads* grads, const lcgs_params* raw,
                                   const lcgs_params* m, const lcgs_params* v, const lcgs_params* activated)
{
    LCGS_REQUIRE(ctx && c && cfg && grads && raw && m && v && activated, "NULL argument");
    LCGS_REQUIRE(c->ctx == ctx, "the communicator belongs to another context");
    LCGS_REQUIRE(c->loop == nullptr || c->transport == LCGS_TRANSPORT_F32, "the in-process (loopback) transport moves f32 only");
    LCGS_REQUIRE(cfg->visible_only == 0, "the sharded step is dense (per-splat rows): visible_only must be 0");
    LCGS_REQUIRE(num_gaussians >= 0 && sh_degree >= 0 && sh_degree <= 3, "bad num_gaussians / sh_degree");
    if (num_gaussians == 0) return LCGS_OK;
    LCGS_HIP_CHECK(hipSetDevice(ctx->device));
    const int64_t P = num_gaussians, N = c->world;
    int64_t       first = 0, count = 0;
    lcgs_comm_shard_rows(P, c->world, c->rank, &first, &count);
    const int64_t  tail0 = count * N, tail = P - tail0; // rows every rank keeps (fewer than N)
    const AttrRows g = attr_rows(grads, sh_degree), act = attr_rows(activated, sh_degree);
    for (int i = 0; i < 5; ++i) LCGS_REQUIRE(g.ptr[i] && act.ptr[i], "NULL device pointer");

    // ---- 1. reduce-scatter: rank r ends up with the summed gradient rows [r c, (r + 1) c); the tail is all-reduced
    LCGS_HIP_CHECK(hipEventRecord(c->ev_in, ctx->stream));
    LCGS_HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_in, 0));
    ctx->slices_recorded = 0;
    Wire      wire{ c };
    LoopGuard guard{ c };
    LCGS_TRY(wire.group_begin());
    for (int i = 0; i < 5; ++i) {
        if (count > 0)
            LCGS_TRY(wire.reduce_scatter_sum(g.ptr[i], g.ptr[i] + (size_t)first * g.width[i], (size_t)count * g.width[i]));
        if (tail > 0) LCGS_TRY(wire.allreduce_sum(g.ptr[i] + (size_t)tail0 * g.width[i], (size_t)tail * g.width[i]));
    }
    LCGS_TRY(wire.group_end());
    LCGS_HIP_CHECK(hipEventRecord(c->ev_out, c->stream));
    LCGS_HIP_CHECK(hipStreamWaitEvent(ctx->stream, c->ev_out, 0));

    c->stats                   = lcgs_comm_stats{};
    c->stats.collective_groups = 1;
    c->stats.bytes_sent = c->stats.bytes_received = (int64_t)((uint64_t)(N - 1) * (uint64_t)count * row_bytes(sh_degree));

    // ---- 2. Adam on the own rows (and on the tail, identically on every rank); 3. all-gather of the ACTIVATED rows
    LCGS_TRY(adam_own_rows(ctx, c, P, sh_degree, cfg, g, raw, m, v, activated));
    LCGS_TRY(allgather_activated(ctx, c, P, sh_degree, act));
    guard.ok = true;
    return LCGS_OK;
}


// ------------------------------------------------------------------------------------------------------------------
// Sparse gradient exchange (round 3).  Dense rows stay the layout; what crosses xGMI in the REDUCE half of the step is
// only what a rank's views touched: rank r hands owner o the touched rows of o's shard (indices + 59 floats each),
// the owner adds them to its own rows in rank order, runs Adam on its shard and the refreshed activated rows are
// all-gathered as in the sharded step.  Exact in f32 up to the order of the sum.
// ------------------------------------------------------------------------------------------------------------------
lcgs_status lcgs_comm_track_touched_rows(lcgs_comm* c, int enable)
{
    LCGS_REQUIRE(c != nullptr, "comm is NULL");
    c->track_rows = enable != 0;
    c->flags_P    = 0; // the next marking backward starts from a cleared array
    return LCGS_OK;
}

lcgs_status lcgs_comm_get_stats(const lcgs_comm* c, lcgs_comm_stats* out)
{
    LCGS_REQUIRE(c && out, "NULL argument");
    *out = c->stats;
    return LCGS_OK;
}

int64_t lcgs_sparse_message_words(int64_t count, int sh_degree) { return sparse_message_words(count, sh_degree); }

lcgs_status lcgs_sparse_pack(lcgs_context* ctx, int sh_degree, const lcgs_grads* grads, const uint32_t* d_rows, int64_t count,
                             float* d_msg)
{
    LCGS_REQUIRE(ctx && grads && (count == 0 || (d_rows && d_msg)) && count >= 0 && sh_degree >= 0 && sh_degree <= 3,
                 "bad argument");
    LCGS_HIP_CHECK(hipSetDevice(ctx->device));
    float* g[5] = { grads->d_dL_dpos, grads->d_dL_dscale, grads->d_dL_drotq, grads->d_dL_dsh, grads->d_dL_dopacity };
    launch_sparse_pack(g, sh_degree, d_rows, count, d_msg, ctx->stream);
    LCGS_HIP_CHECK(hipGetLastError());
    return LCGS_OK;
}

lcgs_status lcgs_sparse_accumulate(lcgs_context* ctx, int sh_degree, const lcgs_grads* grads, const float* d_msg, int64_t count,
                                   int64_t row_first, int64_t row_count)
{
    LCGS_REQUIRE(ctx && grads && (count == 0 || d_msg) && count >= 0 && sh_degree >= 0 && sh_degree <= 3, "bad argument");
    LCGS_REQUIRE(row_first >= 0 && row_count >= 0 && row_first + row_count <= ((int64_t)1 << 30), "bad row range");
    LCGS_HIP_CHECK(hipSetDevice(ctx->device));
    float* g[5] = { grads->d_dL_dpos, grads->d_dL_dscale, grads->d_dL_drotq, grads->d_dL_dsh, grads->d_dL_dopacity };
    launch_sparse_accumulate(g, sh_degree, d_msg, count, row_first, row_count, ctx->stream);
    LCGS_HIP_CHECK(hipGetLastError());
    return LCGS_OK;
}

} // extern "C"

namespace
{
// flags -> ascending rows + owner bounds, on the context's stream; the flags are consumed (cleared) behind it
lcgs_status compact_touched(lcgs_context* ctx, lcgs_comm* c, int64_t P, int world)
{
    LCGS_REQUIRE(c->track_rows, "lcgs_comm_track_touched_rows(comm, 1) must be set before the step's backward passes");
    const size_t fb = sparse_flag_bytes(P);
    if (c->flags_P != P) { // no backward has marked anything for this scene since tracking began: an empty set
        LCGS_TRY(c->flags.ensure(fb));
        LCGS_HIP_CHECK(hipMemsetAsync(c->flags.ptr, 0, fb, ctx->stream));
        c->flags_P = P;
    }
    LCGS_TRY(c->chunk_ws.ensure((size_t)sparse_flag_chunks(P) * 4 + 4));
    LCGS_TRY(c->rows.ensure((size_t)P * 4 + 4));
    LCGS_TRY(c->bounds.ensure((size_t)(LCGS_MAX_RANKS + 3) * 4));
    uint32_t* bounds = c->bounds.as<uint32_t>();
    uint32_t* total  = bounds + LCGS_MAX_RANKS + 2;
    launch_compact_flags(c->flags.as<uint8_t>(), P, c->chunk_ws.as<uint32_t>(), c->rows.as<uint32_t>(), total, ctx->stream);
    int64_t first = 0, shard = 0;
    lcgs_comm_shard_rows(P, world, 0, &first, &shard);
    // (P < N: shards are empty, every row is a tail row -- the bounds all sit at 0 and the tail starts there)
    launch_owner_bounds(c->rows.as<uint32_t>(), total, shard, world, bounds, ctx->stream);
    LCGS_HIP_CHECK(hipMemsetAsync(c->flags.ptr, 0, fb, ctx->stream)); // consumed: the next step starts empty
    LCGS_HIP_CHECK(hipGetLastError());
    return LCGS_OK;
}
} // namespace

namespace lcgs
{
// (abi_backward.cpp render_backward) flag the rows of the frame a dense backward has just differentiated
lcgs_status comm_mark_touched(lcgs_comm* c, const uint32_t* vis_index, const uint32_t* d_counts, int64_t P, int64_t hint_V,
                              bool accumulate, hipStream_t stream)
{
    if (!c || !c->track_rows || P <= 0) return LCGS_OK;
    const size_t fb = sparse_flag_bytes(P);
    if (c->flags_P != P || !accumulate) {
        LCGS_TRY(c->flags.ensure(fb));
        LCGS_HIP_CHECK(hipMemsetAsync(c->flags.ptr, 0, fb, stream));
        c->flags_P = P;
    }
    launch_mark_rows(vis_index, d_counts, c->flags.as<uint8_t>(), P, hint_V, stream);
    return LCGS_OK;
}
} // namespace lcgs

extern "C" {

lcgs_status lcgs_sparse_touched_rows(lcgs_context* ctx, lcgs_comm* c, int num_gaussians, int world_size, lcgs_sparse_rows* out)
{
    LCGS_REQUIRE(ctx && c && out, "NULL argument");
    LCGS_REQUIRE(c->ctx == ctx, "the communicator belongs to another context");
    LCGS_REQUIRE(num_gaussians >= 0 && world_size >= 1 && world_size <= LCGS_MAX_RANKS, "bad num_gaussians / world_size");
    LCGS_HIP_CHECK(hipSetDevice(ctx->device));
    memset(out, 0, sizeof(*out));
    if (num_gaussians == 0) return LCGS_OK;
    LCGS_TRY(compact_touched(ctx, c, num_gaussians, world_size));
    if (!c->h_matrix) LCGS_HIP_CHECK(hipHostMalloc((void**)&c->h_matrix, (size_t)LCGS_MAX_RANKS * (LCGS_MAX_RANKS + 2) * 4, 0));
    LCGS_HIP_CHECK(hipMemcpyAsync(c->h_matrix, c->bounds.ptr, (size_t)(world_size + 2) * 4, hipMemcpyDeviceToHost, ctx->stream));
    LCGS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    out->d_rows = c->rows.as<uint32_t>();
    for (int o = 0; o <= world_size + 1; ++o) out->owner_first[o] = c->h_matrix[o];
    out->num_rows = out->owner_first[world_size + 1];
    return LCGS_OK;
}

lcgs_status lcgs_adam_step_sparse(lcgs_context* ctx, lcgs_comm* c, int num_gaussians, int sh_degree,
                                  const lcgs_adam_config* cfg, const lcgs_grads* grads, const lcgs_params* raw,
                                  const lcgs_params* m, const lcgs_params* v, const lcgs_params* activated)
{
    LCGS_REQUIRE(ctx && c && cfg && grads && raw && m && v && activated, "NULL argument");
    LCGS_REQUIRE(c->ctx == ctx, "the communicator belongs to another context");
    LCGS_REQUIRE(c->loop == nullptr || c->transport == LCGS_TRANSPORT_F32, "the in-process (loopback) transport moves f32 only");
    LCGS_REQUIRE(cfg->visible_only == 0, "the sparse step keeps dense-Adam semantics (every row decays): visible_only must be 0");
    LCGS_REQUIRE(num_gaussians >= 0 && sh_degree >= 0 && sh_degree <= 3, "bad num_gaussians / sh_degree");
    if (num_gaussians == 0) return LCGS_OK;
    LCGS_HIP_CHECK(hipSetDevice(ctx->device));
    const int64_t  P = num_gaussians;
    const int      N = c->world, me = c->rank;
    int64_t        first = 0, count = 0;
    lcgs_comm_shard_rows(P, N, me, &first, &count);
    const int64_t  tail0 = count * N, tail = P - tail0;
    const AttrRows g = attr_rows(grads, sh_degree), act = attr_rows(activated, sh_degree);
    for (int i = 0; i < 5; ++i) LCGS_REQUIRE(g.ptr[i] && act.ptr[i], "NULL device pointer");
    ctx->slices_recorded = 0;
    c->stats             = lcgs_comm_stats{};

    // ---- 1. this rank's touched rows, ascending, and where each owner's shard begins in that list
    LCGS_TRY(compact_touched(ctx, c, P, N));
    const int W = N + 2; // positions per rank: N shard starts, the tail's start, the total
    LCGS_TRY(c->matrix.ensure((size_t)N * W * 4));
    if (!c->h_matrix) LCGS_HIP_CHECK(hipHostMalloc((void**)&c->h_matrix, (size_t)LCGS_MAX_RANKS * (LCGS_MAX_RANKS + 2) * 4, 0));
    LCGS_HIP_CHECK(hipEventRecord(c->ev_in, ctx->stream));
    LCGS_HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_in, 0));
    // ---- 2. everybody learns everybody's counts (message sizes are host arguments of send / recv): one small
    //         all-gather + read-back, the step's only host synchronisation
    Wire      wire{ c };
    LoopGuard guard{ c };
    LCGS_TRY(wire.allgather_u32(c->bounds.as<uint32_t>(), c->matrix.as<uint32_t>(), (size_t)W));
    LCGS_HIP_CHECK(hipMemcpyAsync(c->h_matrix, c->matrix.ptr, (size_t)N * W * 4, hipMemcpyDeviceToHost, c->stream));
    LCGS_HIP_CHECK(hipStreamSynchronize(c->stream));
    auto rows_of = [&](int src, int owner) -> int64_t { // rows rank `src` holds for owner's shard
        return (int64_t)c->h_matrix[src * W + owner + 1] - (int64_t)c->h_matrix[src * W + owner];
    };
    c->stats.touched_rows = (int64_t)c->h_matrix[me * W + N + 1];
    int64_t send_words = 0, recv_words = 0, send_off[LCGS_MAX_RANKS], recv_off[LCGS_MAX_RANKS];
    for (int o = 0; o < N; ++o) {
        send_off[o] = send_words;
        recv_off[o] = recv_words;
        if (o == me) continue;
        send_words += sparse_message_words(rows_of(me, o), sh_degree);
        recv_words += sparse_message_words(rows_of(o, me), sh_degree);
    }
    LCGS_TRY(c->sendbuf.ensure((size_t)send_words * 4 + 16));
    LCGS_TRY(c->recvbuf.ensure((size_t)recv_words * 4 + 16));

    // ---- 3. pack one message per peer (context's stream), exchange (communicator's stream); the tail rows -- fewer than
    //         N, kept by everyone -- are all-reduced densely behind it
    float* gp[5] = { g.ptr[0], g.ptr[1], g.ptr[2], g.ptr[3], g.ptr[4] };
    for (int o = 0; o < N; ++o)
        if (o != me)
            launch_sparse_pack(gp, sh_degree, c->rows.as<uint32_t>() + c->h_matrix[me * W + o], rows_of(me, o),
                               c->sendbuf.as<float>() + send_off[o], ctx->stream);
    LCGS_HIP_CHECK(hipGetLastError());
    LCGS_HIP_CHECK(hipEventRecord(c->ev_in, ctx->stream));
    LCGS_HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_in, 0));
    LCGS_TRY(wire.group_begin());
    for (int o = 0; o < N; ++o) {
        if (o == me) continue;
        const int64_t sw = sparse_message_words(rows_of(me, o), sh_degree), rw = sparse_message_words(rows_of(o, me), sh_degree);
        if (sw > 0) LCGS_TRY(wire.send(c->sendbuf.as<float>() + send_off[o], (size_t)sw * 4, o));
        if (rw > 0) LCGS_TRY(wire.recv(c->recvbuf.as<float>() + recv_off[o], (size_t)rw * 4, o));
    }
    LCGS_TRY(wire.group_end());
    if (tail > 0) { // (its own group: point-to-point and collective calls are not mixed in one)
        LCGS_TRY(wire.group_begin());
        for (int i = 0; i < 5; ++i) LCGS_TRY(wire.allreduce_sum(g.ptr[i] + (size_t)tail0 * g.width[i], (size_t)tail * g.width[i]));
        LCGS_TRY(wire.group_end());
    }
    c->stats.collective_groups = 2 + (tail > 0 ? 1 : 0); // the counts, the messages, the tail
    c->stats.bytes_sent        = send_words * 4 + (int64_t)(N - 1) * W * 4;
    c->stats.bytes_received    = recv_words * 4 + (int64_t)(N - 1) * W * 4;
    LCGS_HIP_CHECK(hipEventRecord(c->ev_out, c->stream));
    LCGS_HIP_CHECK(hipStreamWaitEvent(ctx->stream, c->ev_out, 0));

    // ---- 4. the owner adds what it received, message by message in rank order (a fixed order: reproducible sums)
    for (int o = 0; o < N; ++o)
        if (o != me) // (only rows of the own shard are accepted, whatever the message says)
            launch_sparse_accumulate(gp, sh_degree, c->recvbuf.as<float>() + recv_off[o], rows_of(o, me), first, count, ctx->stream);
    LCGS_HIP_CHECK(hipGetLastError());

    // ---- 5. Adam on the own rows (+ the tail), 6. all-gather of the refreshed ACTIVATED rows: as in the sharded step
    LCGS_TRY(adam_own_rows(ctx, c, P, sh_degree, cfg, g, raw, m, v, activated));
    LCGS_TRY(allgather_activated(ctx, c, P, sh_degree, act));
    guard.ok = true;
    return LCGS_OK;
}

} // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// The splat-ownership step with its transport (the device halves are abi_owner.cpp's).  What travels: per
// view v, from every owner o to rank v, the rows of o's range that reach v's screen -- [row index u32] + [48-byte packed
// record] -- and back, from rank v to every owner, the 48-byte 2-D gradient row of each of them.  Sizes are agreed through
// ONE small all-gather (the N counts of every owner) and one read-back: the step's only host synchronisation besides the
// view's own pair-buffer check.  Point-to-point over RCCL (ncclSend / ncclRecv in one group per direction); the same code
// runs over the in-process loopback with N contexts on one device.
// ---------------------------------------------------------------------------------------------------------------------
namespace
{
// bytes of one message row
constexpr size_t kRecBytes = LCGS_OWNER_RECORD_FLOATS * 4, kG2dBytes = LCGS_OWNER_GRAD_FLOATS * 4;
} // namespace

extern "C" {

void lcgs_comm_owner_rows(int64_t num_gaussians, int world_size, int rank, int64_t* first, int64_t* count)
{
    // equal contiguous shards of floor(P / N) rows, the P mod N tail with the last rank (multi_gpu.owner_range)
    const int64_t c = world_size > 0 ? num_gaussians / world_size : num_gaussians;
    if (first) *first = c * rank;
    if (count) *count = rank < world_size - 1 ? c : num_gaussians - c * rank;
}

lcgs_status lcgs_loopback_group_create(int world_size, lcgs_loopback_group** out)
{
    LCGS_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    LCGS_REQUIRE(world_size >= 1 && world_size <= LCGS_MAX_OWNER_VIEWS, "world_size out of range");
    lcgs_loopback_group* g = new (std::nothrow) lcgs_loopback_group();
    if (!g) return LCGS_ERR_OUT_OF_MEMORY;
    g->world = world_size;
    g->box.resize((size_t)world_size * world_size);
    g->done.assign(world_size, nullptr);
    g->ready.assign(world_size, nullptr);
    g->reduced.assign(world_size, nullptr);
    g->scratch.assign(world_size, nullptr);
    g->taken.assign(world_size, 0);
    *out = g;
    return LCGS_OK;
}

lcgs_status lcgs_loopback_group_destroy(lcgs_loopback_group* g)
{
    if (!g) return LCGS_OK;
    LCGS_REQUIRE(g->members == 0, "communicators of this group are still alive");
    g->drop_unconsumed();
    for (auto* evs : { &g->done, &g->ready, &g->reduced })
        for (hipEvent_t e : *evs)
            if (e) (void)hipEventDestroy(e);
    delete g;
    return LCGS_OK;
}

lcgs_status lcgs_comm_create_loopback(lcgs_context* ctx, lcgs_loopback_group* group, int rank, lcgs_comm** out)
{
    LCGS_REQUIRE(ctx && group && out, "NULL argument");
    *out = nullptr;
    LCGS_REQUIRE(rank >= 0 && rank < group->world, "rank out of range");
    LCGS_REQUIRE(ctx->comm == nullptr, "the context already has a communicator attached");
    {
        std::lock_guard<std::mutex> lock(group->mu);
        LCGS_REQUIRE(!group->taken[(size_t)rank], "this rank of the loopback group already has a communicator");
        LCGS_REQUIRE(group->device < 0 || group->device == ctx->device, "the members of a loopback group share ONE device");
        group->taken[(size_t)rank] = 1;
        group->device              = ctx->device;
    }
    LCGS_HIP_CHECK(hipSetDevice(ctx->device));
    lcgs_comm* c = new (std::nothrow) lcgs_comm();
    if (!c) return LCGS_ERR_OUT_OF_MEMORY;
    c->ctx = ctx, c->device = ctx->device, c->rank = rank, c->world = group->world, c->loop = group;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming);
    {
        std::lock_guard<std::mutex> lock(group->mu);
        for (auto* evs : { &group->done, &group->ready, &group->reduced })
            if (e == hipSuccess && !(*evs)[rank]) e = hipEventCreateWithFlags(&(*evs)[rank], hipEventDisableTiming);
        if (e == hipSuccess) ++group->members;
    }
    if (e != hipSuccess) {
        {
            std::lock_guard<std::mutex> lock(group->mu);
            group->taken[(size_t)rank] = 0;
        }
        c->loop = nullptr;
        (void)lcgs_comm_destroy(c);
        LCGS_HIP_CHECK(e);
    }
    ctx->comm = c;
    int slices = 4; // (as lcgs_comm_create: the dense backward slices its preprocess pass for the chunked all-reduce)
    if (const char* sl = getenv("LCGS_GRAD_SLICES")) slices = atoi(sl);
    ctx->grad_slices = std::min(std::max(slices, 1), kMaxGradSlices);
    *out             = c;
    return LCGS_OK;
}

static_assert(LCGS_MAX_OWNER_VIEWS <= lcgs::kMaxOwnerSegs, "OwnerSegs holds one segment per view slot");
// rows a padded message from an owner of `owner_count` rows holds when it carried n rows in the last step
static int64_t padded_rows(int64_t n, int64_t owner_count) { return std::min(owner_count, n + n / 4 + 1024); }

lcgs_status lcgs_owner_step_set_async(lcgs_comm* c, int enable)
{
    LCGS_REQUIRE(c != nullptr, "comm is NULL");
    c->owner_async = enable != 0;
    return LCGS_OK;
}

lcgs_status lcgs_owner_step_forward(lcgs_context* ctx, lcgs_comm* c, const lcgs_camera* cameras, const float bg_color[3],
                                    float scale_modifier, float* d_img)
{
    LCGS_REQUIRE(ctx && c && cameras && bg_color && d_img, "NULL argument");
    LCGS_REQUIRE(c->ctx == ctx, "the communicator belongs to another (or a destroyed) context");
    LCGS_REQUIRE(c->world <= LCGS_MAX_OWNER_VIEWS, "world_size above LCGS_MAX_OWNER_VIEWS (one view slot per rank)");
    LCGS_REQUIRE(ctx->pos != nullptr && ctx->P > 0, "no scene bound");
    LCGS_HIP_CHECK(hipSetDevice(ctx->device));
    const int N = c->world, me = c->rank;
    int64_t   first = 0, count = 0;
    lcgs_comm_owner_rows(ctx->P, N, me, &first, &count);
    c->own.valid = false;
    c->own.async = false;
    c->stats     = lcgs_comm_stats{};
    Wire      wire{ c };
    LoopGuard guard{ c };

    // ---- 1. my rows, every view of the step (view v = rank v's): N asynchronous projections, side by side
    LCGS_TRY(c->own_rows.ensure((size_t)N * (size_t)count * 4 + 16));
    LCGS_TRY(c->own_recs.ensure((size_t)N * (size_t)count * kRecBytes + 16));
    {
        uint32_t* rows_v[LCGS_MAX_OWNER_VIEWS];
        float*    recs_v[LCGS_MAX_OWNER_VIEWS];
        for (int v = 0; v < N; ++v) {
            rows_v[v] = c->own_rows.as<uint32_t>() + (size_t)v * count;
            recs_v[v] = c->own_recs.as<float>() + (size_t)v * count * LCGS_OWNER_RECORD_FLOATS;
        }
        // (the N pipelines side by side on the context's lanes, joined on its stream: abi_owner.cpp)
        LCGS_TRY(lcgs_owner_project_views(ctx, 0, N, cameras, scale_modifier, (int)first, (int)count, /*keep_state=*/1, rows_v, recs_v));
    }
    // ---- 2. everybody learns everybody's counts: table[o][v] = rows of owner o on view v's screen
    LCGS_TRY(c->bounds.ensure((size_t)(N + 2) * 4));
    LCGS_TRY(c->matrix.ensure((size_t)N * N * 4));
    if (!c->h_matrix) LCGS_HIP_CHECK(hipHostMalloc((void**)&c->h_matrix, (size_t)LCGS_MAX_RANKS * (LCGS_MAX_RANKS + 2) * 4, 0));
    LCGS_HIP_CHECK(hipMemsetAsync(c->bounds.ptr, 0, (size_t)N * 4, ctx->stream));
    for (int v = 0; v < N; ++v)
        if (ctx->owner[v].valid && ctx->owner[v].row_count > 0)
            LCGS_HIP_CHECK(hipMemcpyAsync(c->bounds.as<uint32_t>() + v, ctx->owner[v].counts.ptr, 4, hipMemcpyDeviceToDevice, ctx->stream));

    // The step WITHOUT a read-back (lcgs_owner_step_set_async): possible once a previous step's table is known, for the
    // same scene and world, and while the padded segments of my view fit the workspace the scene sizes
    bool    async = c->owner_async && !c->force_sync_once && c->prev.have && c->prev.world == N && c->prev.P == ctx->P;
    int64_t cap_total = 0;
    if (async) {
        for (int o = 0; o < N; ++o) {
            int64_t of = 0, oc = 0;
            lcgs_comm_owner_rows(ctx->P, N, o, &of, &oc);
            c->own.cap_in[o]  = padded_rows(c->prev.table[(size_t)o * N + me], oc);
            c->own.in_off[o]  = cap_total;
            cap_total += c->own.cap_in[o];
            c->own.cap_out[o] = padded_rows(c->prev.table[(size_t)me * N + o], count);
        }
        c->own.in_off[N] = cap_total;
        // EVERY rank must take the same branch (the branches size their messages differently): the test runs over every
        // view's padded segments, computed from the table all ranks share -- not over this rank's own column alone
        for (int v = 0; v < N && async; ++v) {
            int64_t total_v = 0;
            for (int o = 0; o < N; ++o) {
                int64_t of = 0, oc = 0;
                lcgs_comm_owner_rows(ctx->P, N, o, &of, &oc);
                total_v += padded_rows(c->prev.table[(size_t)o * N + v], oc);
            }
            if (total_v > ctx->P || total_v >= (int64_t)0x7FFFFFFF) async = false;
        }
    }
    c->force_sync_once = false;

    if (async) {
        if (!c->h_next) LCGS_HIP_CHECK(hipHostMalloc((void**)&c->h_next, ((size_t)LCGS_MAX_RANKS * LCGS_MAX_RANKS + 4) * 4, 0));
        if (!c->ev_checked) LCGS_HIP_CHECK(hipEventCreateWithFlags(&c->ev_checked, hipEventDisableTiming));
        LCGS_TRY(c->flag_dev.ensure(16));
        LCGS_HIP_CHECK(hipMemsetAsync(c->flag_dev.ptr, 0, 16, ctx->stream));
        LCGS_HIP_CHECK(hipEventRecord(c->ev_in, ctx->stream));
        LCGS_HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_in, 0));
        LCGS_TRY(wire.allgather_u32(c->bounds.as<uint32_t>(), c->matrix.as<uint32_t>(), (size_t)N));
        // (for lcgs_owner_step_finish and the next step's sizes: nobody waits for this copy here)
        LCGS_HIP_CHECK(hipMemcpyAsync(c->h_next, c->matrix.ptr, (size_t)N * N * 4, hipMemcpyDeviceToHost, c->stream));
        // ---- 3'. padded messages: sizes from the last step's table, true counts on the device
        const bool alias = N == 1 && !c->self_p2p; // one rank: my view reads my own projection where it lies
        if (!alias) {
            LCGS_TRY(c->in_rows.ensure((size_t)cap_total * 4 + 16));
            LCGS_TRY(c->in_recs.ensure((size_t)cap_total * kRecBytes + 16));
        }
        int64_t sent = 0, received = 0;
        if (!alias && !c->self_p2p && c->own.cap_in[me] > 0) { // my own share stays on the device (cap_in[me] == cap_out[me])
            LCGS_HIP_CHECK(hipMemcpyAsync(c->in_rows.as<uint32_t>() + c->own.in_off[me], c->own_rows.as<uint32_t>() + (size_t)me * count,
                                          (size_t)c->own.cap_in[me] * 4, hipMemcpyDeviceToDevice, c->stream));
            LCGS_HIP_CHECK(hipMemcpyAsync(c->in_recs.as<float>() + (size_t)c->own.in_off[me] * LCGS_OWNER_RECORD_FLOATS,
                                          c->own_recs.as<float>() + (size_t)me * count * LCGS_OWNER_RECORD_FLOATS,
                                          (size_t)c->own.cap_in[me] * kRecBytes, hipMemcpyDeviceToDevice, c->stream));
        }
        LCGS_TRY(wire.group_begin());
        for (int o = 0; o < N && !alias; ++o) {
            const int64_t   n_out = c->own.cap_out[o], n_in = c->own.cap_in[o];
            const uint32_t* rows_out = c->own_rows.as<uint32_t>() + (size_t)o * count;
            const float*    recs_out = c->own_recs.as<float>() + (size_t)o * count * LCGS_OWNER_RECORD_FLOATS;
            uint32_t*       rows_in  = c->in_rows.as<uint32_t>() + c->own.in_off[o];
            float*          recs_in  = c->in_recs.as<float>() + (size_t)c->own.in_off[o] * LCGS_OWNER_RECORD_FLOATS;
            if (o == me && !c->self_p2p) continue;
            if (n_out > 0) {
                LCGS_TRY(wire.send(rows_out, (size_t)n_out * 4, o));
                LCGS_TRY(wire.send(recs_out, (size_t)n_out * kRecBytes, o));
                sent += n_out * (int64_t)(4 + kRecBytes);
            }
            if (n_in > 0) {
                LCGS_TRY(wire.recv(rows_in, (size_t)n_in * 4, o));
                LCGS_TRY(wire.recv(recs_in, (size_t)n_in * kRecBytes, o));
                received += n_in * (int64_t)(4 + kRecBytes);
            }
        }
        LCGS_TRY(wire.group_end());
        LCGS_HIP_CHECK(hipEventRecord(c->ev_out, c->stream));
        LCGS_HIP_CHECK(hipStreamWaitEvent(ctx->stream, c->ev_out, 0));
        c->stats.bytes_sent        = sent + (int64_t)(N - 1) * N * 4;
        c->stats.bytes_received    = received + (int64_t)(N - 1) * N * 4;
        c->stats.touched_rows      = cap_total; // (the capacity: the row count itself is on the device)
        c->stats.collective_groups = 3;         // the counts, the records, the flag
        c->own.n_all               = cap_total;
        for (int v = 0; v < N; ++v) {
            c->own.out[v] = (uint32_t)c->own.cap_out[v];
            if (ctx->owner[v].valid && ctx->owner[v].row_count > 0) ctx->owner[v].num = (int)c->own.cap_out[v]; // (a launch bound)
        }
        // ---- 4'. my view from everybody's rows: no read-back, the verdicts go into the flag word
        abi::OwnerAsyncFrame af;
        af.segs.n = (uint32_t)N;
        for (int o = 0; o <= N; ++o) af.segs.off[o] = (uint32_t)c->own.in_off[o];
        af.table    = c->matrix.as<uint32_t>();
        af.view     = (uint32_t)me;
        af.overflow = c->flag_dev.as<uint32_t>();
        const uint32_t* rows_v = alias ? c->own_rows.as<uint32_t>() : c->in_rows.as<uint32_t>();
        const float*    recs_v = alias ? c->own_recs.as<float>() : c->in_recs.as<float>();
        if (cap_total > 0)
            LCGS_TRY(abi::owner_render_frame(ctx, &cameras[me], bg_color, (int)cap_total, rows_v, recs_v, d_img, /*keep_state=*/1, &af));
        // ---- 5'. one verdict for everybody: the flag, max-reduced; with the table it reaches pinned memory behind ev_checked
        LCGS_HIP_CHECK(hipEventRecord(c->ev_in, ctx->stream));
        LCGS_HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_in, 0));
        LCGS_TRY(wire.allreduce_max_u32(c->flag_dev.as<uint32_t>(), 1));
        LCGS_HIP_CHECK(hipMemcpyAsync(c->h_next + (size_t)N * N, c->flag_dev.ptr, 4, hipMemcpyDeviceToHost, c->stream));
        LCGS_HIP_CHECK(hipEventRecord(c->ev_checked, c->stream));
        c->own.valid = true;
        c->own.async = true;
        guard.ok     = true;
        return LCGS_OK;
    }

    LCGS_HIP_CHECK(hipEventRecord(c->ev_in, ctx->stream));
    LCGS_HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_in, 0));
    LCGS_TRY(wire.allgather_u32(c->bounds.as<uint32_t>(), c->matrix.as<uint32_t>(), (size_t)N));
    LCGS_HIP_CHECK(hipMemcpyAsync(c->h_matrix, c->matrix.ptr, (size_t)N * N * 4, hipMemcpyDeviceToHost, c->stream));
    LCGS_HIP_CHECK(hipStreamSynchronize(c->stream)); // the step's one host synchronisation for message sizes
    auto table = [&](int o, int v) -> int64_t { return (int64_t)c->h_matrix[o * N + v]; };
    int64_t n_all = 0;
    for (int o = 0; o < N; ++o) {
        int64_t of = 0, oc = 0;
        lcgs_comm_owner_rows(ctx->P, N, o, &of, &oc);
        for (int v = 0; v < N; ++v) LCGS_REQUIRE(table(o, v) <= oc, "an owner reports more on-screen rows than it owns");
        c->own.in_off[o] = n_all;
        n_all += table(o, me);
    }
    c->own.in_off[N] = n_all;
    for (int v = 0; v < N; ++v) {
        c->own.out[v] = (uint32_t)table(me, v);
        if (ctx->owner[v].valid && ctx->owner[v].row_count > 0) ctx->owner[v].num = (int)table(me, v);
    }
    // (what the next step sizes its padded messages from, if it runs without a read-back)
    c->prev.table.assign(c->h_matrix, c->h_matrix + (size_t)N * N);
    c->prev.have = true, c->prev.world = N, c->prev.P = ctx->P;
    // ---- 3. the records travel: mine to every view's rank, every owner's to me (owner order = ascending rows)
    LCGS_TRY(c->in_rows.ensure((size_t)n_all * 4 + 16));
    LCGS_TRY(c->in_recs.ensure((size_t)n_all * kRecBytes + 16));
    int64_t sent = 0, received = 0;
    // (my own share stays on the device: copied in front of the group, so that nothing but RCCL calls sits inside it)
    if (!c->self_p2p && table(me, me) > 0) {
        LCGS_HIP_CHECK(hipMemcpyAsync(c->in_rows.as<uint32_t>() + c->own.in_off[me], c->own_rows.as<uint32_t>() + (size_t)me * count,
                                      (size_t)table(me, me) * 4, hipMemcpyDeviceToDevice, c->stream));
        LCGS_HIP_CHECK(hipMemcpyAsync(c->in_recs.as<float>() + (size_t)c->own.in_off[me] * LCGS_OWNER_RECORD_FLOATS,
                                      c->own_recs.as<float>() + (size_t)me * count * LCGS_OWNER_RECORD_FLOATS,
                                      (size_t)table(me, me) * kRecBytes, hipMemcpyDeviceToDevice, c->stream));
    }
    LCGS_TRY(wire.group_begin());
    for (int o = 0; o < N; ++o) {
        const int64_t n_out = table(me, o), n_in = table(o, me);
        const uint32_t* rows_out = c->own_rows.as<uint32_t>() + (size_t)o * count;
        const float*    recs_out = c->own_recs.as<float>() + (size_t)o * count * LCGS_OWNER_RECORD_FLOATS;
        uint32_t*       rows_in  = c->in_rows.as<uint32_t>() + c->own.in_off[o];
        float*          recs_in  = c->in_recs.as<float>() + (size_t)c->own.in_off[o] * LCGS_OWNER_RECORD_FLOATS;
        if (o == me && !c->self_p2p) continue;
        if (n_out > 0) {
            LCGS_TRY(wire.send(rows_out, (size_t)n_out * 4, o));
            LCGS_TRY(wire.send(recs_out, (size_t)n_out * kRecBytes, o));
            sent += n_out * (int64_t)(4 + kRecBytes);
        }
        if (n_in > 0) {
            LCGS_TRY(wire.recv(rows_in, (size_t)n_in * 4, o));
            LCGS_TRY(wire.recv(recs_in, (size_t)n_in * kRecBytes, o));
            received += n_in * (int64_t)(4 + kRecBytes);
        }
    }
    LCGS_TRY(wire.group_end());
    LCGS_HIP_CHECK(hipEventRecord(c->ev_out, c->stream));
    LCGS_HIP_CHECK(hipStreamWaitEvent(ctx->stream, c->ev_out, 0));
    c->stats.bytes_sent        = sent + (int64_t)(N - 1) * N * 4;
    c->stats.bytes_received    = received + (int64_t)(N - 1) * N * 4;
    c->stats.touched_rows      = n_all; // rows on this rank's screen
    c->stats.collective_groups = 2;     // the counts, the records
    c->own.n_all               = n_all;
    // ---- 4. my view from everybody's rows
    LCGS_TRY(lcgs_owner_render(ctx, &cameras[me], bg_color, (int)n_all, c->in_rows.as<uint32_t>(), c->in_recs.as<float>(), d_img,
                               /*keep_state=*/1));
    c->own.valid = true;
    guard.ok     = true;
    return LCGS_OK;
}

lcgs_status lcgs_owner_step_backward(lcgs_context* ctx, lcgs_comm* c, const float* d_dL_dimg, const lcgs_grads* grads)
{
    LCGS_REQUIRE(ctx && c && d_dL_dimg && grads, "NULL argument");
    LCGS_REQUIRE(c->ctx == ctx, "the communicator belongs to another (or a destroyed) context");
    if (!c->own.valid) {
        set_last_error("lcgs_owner_step_backward needs a preceding lcgs_owner_step_forward");
        return LCGS_ERR_STATE;
    }
    LCGS_HIP_CHECK(hipSetDevice(ctx->device));
    const int     N = c->world, me = c->rank;
    const int64_t n_all = c->own.n_all; // (a step without read-back: the padded segments' total capacity)
    const bool    alias = c->own.async && N == 1 && !c->self_p2p;
    c->own.valid        = false;
    Wire      wire{ c };
    LoopGuard guard{ c };
    // ---- 1. my view's 2-D gradients, one 48-byte row per received row (owner order; padded segments keep their positions)
    int64_t first = 0, count = 0;
    lcgs_comm_owner_rows(ctx->P, N, me, &first, &count);
    // (a step without read-back: the per-splat kernel of step 3 walks a view's TRUE row count, which a clipped message falls
    // short of -- the step is then repeated, but until the verdict is read nothing may be read out of bounds: every view's
    // rows get room for my whole range)
    LCGS_TRY(c->g2d_all.ensure((size_t)std::max(n_all, alias ? count : (int64_t)0) * kG2dBytes + 16));
    // (the own rows of the dense gradient arrays are cleared as a side job of the render-backward: view 0 then ADDS like the rest)
    const size_t feat = (size_t)(ctx->sh_deg + 1) * (ctx->sh_deg + 1) * 3;
    const bool   filled = n_all > 0 && count > 0 && (size_t)count * feat < ((size_t)1 << 32) && grads->d_dL_dpos && grads->d_dL_dscale &&
                        grads->d_dL_drotq && grads->d_dL_dsh && grads->d_dL_dopacity;
    DenseFill fill;
    if (filled) {
        fill.b0 = grads->d_dL_dpos + 3 * (size_t)first, fill.b1 = grads->d_dL_dscale + 3 * (size_t)first;
        fill.b2 = grads->d_dL_drotq + 4 * (size_t)first, fill.b3 = grads->d_dL_dsh + feat * (size_t)first;
        fill.b4 = grads->d_dL_dopacity + (size_t)first;
        const size_t n[5] = { (size_t)count * 3, (size_t)count * 3, (size_t)count * 4, (size_t)count * feat, (size_t)count };
        for (int a = 0; a < 5; ++a) fill.n[a] = (uint32_t)n[a];
    }
    if (n_all > 0) LCGS_TRY(abi::owner_render_backward_into(ctx, d_dL_dimg, c->g2d_all.as<float>(), filled ? &fill : nullptr));
    // ---- 2. every owner gets its rows' share back; I get my rows' share of every view
    int64_t gin_off[LCGS_MAX_RANKS + 1], total_in = 0;
    for (int v = 0; v < N; ++v) {
        gin_off[v] = total_in;
        total_in += c->own.async ? count : (int64_t)c->own.out[v];
    }
    if (!alias) LCGS_TRY(c->g_in.ensure((size_t)total_in * kG2dBytes + 16));
    const float* g_in = alias ? c->g2d_all.as<float>() : c->g_in.as<float>();
    LCGS_HIP_CHECK(hipEventRecord(c->ev_in, ctx->stream));
    LCGS_HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_in, 0));
    int64_t sent = 0, received = 0;
    if (!alias && !c->self_p2p && c->own.out[me] > 0) // (my own share: a device copy in front of the group)
        LCGS_HIP_CHECK(hipMemcpyAsync(c->g_in.as<float>() + (size_t)gin_off[me] * LCGS_OWNER_GRAD_FLOATS,
                                      c->g2d_all.as<float>() + (size_t)c->own.in_off[me] * LCGS_OWNER_GRAD_FLOATS,
                                      (size_t)c->own.out[me] * kG2dBytes, hipMemcpyDeviceToDevice, c->stream));
    LCGS_TRY(wire.group_begin());
    for (int o = 0; o < N && !alias; ++o) {
        const int64_t n_out = c->own.in_off[o + 1] - c->own.in_off[o]; // owner o's rows on my screen: their gradients go back
        const int64_t n_in  = c->own.out[o];                           // my rows on view o's screen: their gradients come in
        const float*  out   = c->g2d_all.as<float>() + (size_t)c->own.in_off[o] * LCGS_OWNER_GRAD_FLOATS;
        float*        in    = c->g_in.as<float>() + (size_t)gin_off[o] * LCGS_OWNER_GRAD_FLOATS;
        if (o == me && !c->self_p2p) continue;
        if (n_out > 0) {
            LCGS_TRY(wire.send(out, (size_t)n_out * kG2dBytes, o));
            sent += n_out * (int64_t)kG2dBytes;
        }
        if (n_in > 0) {
            LCGS_TRY(wire.recv(in, (size_t)n_in * kG2dBytes, o));
            received += n_in * (int64_t)kG2dBytes;
        }
    }
    LCGS_TRY(wire.group_end());
    LCGS_HIP_CHECK(hipEventRecord(c->ev_out, c->stream));
    LCGS_HIP_CHECK(hipStreamWaitEvent(ctx->stream, c->ev_out, 0));
    c->stats.bytes_sent += sent;
    c->stats.bytes_received += received;
    c->stats.collective_groups += 1;
    // ---- 3. my rows: the 2-D gradients of every view -> parameter gradients, summed in view order
    for (int v = 0; v < N; ++v)
        LCGS_TRY(abi::owner_backward_rows(ctx, v, g_in + (size_t)gin_off[v] * LCGS_OWNER_GRAD_FLOATS, grads, v > 0 ? 1 : (filled ? 2 : 0)));
    guard.ok = true;
    return LCGS_OK;
}

lcgs_status lcgs_owner_step_finish(lcgs_context* ctx, lcgs_comm* c, int* redo)
{
    LCGS_REQUIRE(ctx && c && redo, "NULL argument");
    LCGS_REQUIRE(c->ctx == ctx, "the communicator belongs to another (or a destroyed) context");
    *redo = 0;
    if (!c->own.async) return LCGS_OK; // a step that read its sizes back has nothing left to report
    LCGS_HIP_CHECK(hipSetDevice(ctx->device));
    c->own.async = false;
    // waits for the FORWARD half of the step at most (the flag's reduction sits behind every rank's frame, in front of the
    // backward's messages on the communicator's stream): by now the device is normally far into the backward
    LCGS_HIP_CHECK(hipEventSynchronize(c->ev_checked));
    const int N = c->world, me = c->rank;
    c->prev.table.assign(c->h_next, c->h_next + (size_t)N * N);
    c->prev.have = true, c->prev.world = N, c->prev.P = ctx->P;
    abi::owner_frame_settle(ctx);
    int64_t rows = 0;
    for (int o = 0; o < N; ++o) rows += std::min<int64_t>(c->h_next[(size_t)o * N + me], c->own.cap_in[o]);
    c->stats.touched_rows = rows; // rows on this rank's screen
    if (c->h_next[(size_t)N * N] != 0u) { // somebody's message was clipped, or somebody's frame truncated: everybody redoes
        *redo              = 1;
        c->force_sync_once = true; // (the redo reads its sizes back: exact, and the frame grows its own buffers)
    }
    return LCGS_OK;
}

} // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// lcgs_comm_selftest: what the first N > 1 run on hardware should say about ITSELF before anything is timed.  Every rank of
// the communicator calls it (a collective).  Three phases, each run by a worker thread and watched against timeout_s from
// the calling thread (an RCCL call that never returns, a kernel that never finishes: the phase is reported, not waited for):
//   1. a 1 KB all-reduce (256 floats, rank r contributes r + 1: every element must come back as N (N + 1) / 2);
//   2. point-to-point inside ONE group: to every peer (to itself at N = 1) a zero-byte and a one-byte message, and the same
//      back -- the two message shapes the ownership step's tables can produce at their edge;
//   3. one ownership step on a 10 000-splat scene the call generates (the context's own binding is put back afterwards):
//      with and without read-back, the rank's image against its fused frame of the same scene (bit for bit), its own rows'
//      gradients against the sum of the N views' ordinary backward passes.
// ---------------------------------------------------------------------------------------------------------------------
#include <atomic>
#include <chrono>
#include <cmath>
#include <memory>
#include <thread>

namespace
{
struct SelftestShared {
    std::atomic<int>    phase{ 0 }; // 1..3 while running, 4 when done
    std::atomic<double> phase_start{ 0.0 };
    std::atomic<bool>   done{ false };
    lcgs_status                status = LCGS_OK;
    lcgs_comm_selftest_report  rep{};
    std::string                error;
};
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct DevArr {
    void* p = nullptr;
    ~DevArr()
    {
        if (p) (void)hipFree(p);
    }
    lcgs_status alloc(size_t bytes)
    {
        LCGS_HIP_CHECK(hipMalloc(&p, std::max<size_t>(bytes, 16)));
        return LCGS_OK;
    }
    template <class T>
    T* as() const { return static_cast<T*>(p); }
};

lcgs_status selftest_allreduce(lcgs_comm* c, lcgs_comm_selftest_report* rep)
{
    const int N = c->world;
    DevArr    buf;
    LCGS_TRY(buf.alloc(1024));
    std::vector<float> h(256, (float)(c->rank + 1));
    LCGS_HIP_CHECK(hipMemcpyAsync(buf.p, h.data(), 1024, hipMemcpyHostToDevice, c->stream));
    Wire         wire{ c };
    const double t0 = now_s();
    LCGS_TRY(wire.group_begin());
    LCGS_TRY(wire.allreduce_sum(buf.as<float>(), 256));
    LCGS_TRY(wire.group_end());
    LCGS_HIP_CHECK(hipMemcpyAsync(h.data(), buf.p, 1024, hipMemcpyDeviceToHost, c->stream));
    LCGS_HIP_CHECK(hipStreamSynchronize(c->stream));
    rep->allreduce_ms = (now_s() - t0) * 1e3;
    const float want = 0.5f * (float)N * (float)(N + 1);
    rep->allreduce_ok = 1;
    for (float x : h)
        if (x != want) rep->allreduce_ok = 0;
    return LCGS_OK;
}

lcgs_status selftest_p2p(lcgs_comm* c, lcgs_comm_selftest_report* rep)
{
    const int N = c->world, me = c->rank;
    DevArr    out, in;
    LCGS_TRY(out.alloc(16));
    LCGS_TRY(in.alloc((size_t)N + 16));
    const unsigned char mine = (unsigned char)(me + 1);
    LCGS_HIP_CHECK(hipMemcpyAsync(out.p, &mine, 1, hipMemcpyHostToDevice, c->stream));
    LCGS_HIP_CHECK(hipMemsetAsync(in.p, 0, (size_t)N + 16, c->stream));
    Wire         wire{ c };
    const double t0 = now_s();
    LCGS_TRY(wire.group_begin());
    for (int p = 0; p < N; ++p) {
        if (N > 1 && p == me) continue; // (one rank: the messages go to itself)
        LCGS_TRY(wire.send(out.p, 0, p));
        LCGS_TRY(wire.send(out.p, 1, p));
        LCGS_TRY(wire.recv(in.as<unsigned char>() + p, 0, p));
        LCGS_TRY(wire.recv(in.as<unsigned char>() + p, 1, p));
    }
    LCGS_TRY(wire.group_end());
    std::vector<unsigned char> h((size_t)N);
    LCGS_HIP_CHECK(hipMemcpyAsync(h.data(), in.p, (size_t)N, hipMemcpyDeviceToHost, c->stream));
    LCGS_HIP_CHECK(hipStreamSynchronize(c->stream));
    rep->p2p_ms = (now_s() - t0) * 1e3;
    rep->p2p_ok = 1;
    for (int p = 0; p < N; ++p)
        if ((N == 1 || p != me) && h[(size_t)p] != (unsigned char)(p + 1)) rep->p2p_ok = 0;
    return LCGS_OK;
}

lcgs_status selftest_owner_step(lcgs_context* ctx, lcgs_comm* c, lcgs_comm_selftest_report* rep)
{
    const int N = c->world, me = c->rank;
    rep->owner_step_ok = -1; // not run
    if (N > LCGS_MAX_OWNER_VIEWS) return LCGS_OK;
    constexpr int P = 10000, W = 320, H = 240, feat = 48;
    // ---- the scene (the same on every rank: a counter-based generator) and the N views
    std::vector<float> h_pos((size_t)P * 3), h_sh((size_t)P * feat), h_op((size_t)P), h_scale((size_t)P * 3), h_rotq((size_t)P * 4);
    LCGS_TRY(lcgs_synth_scene(0, 4242u, 0, P, h_pos.data(), h_sh.data(), h_op.data(), h_scale.data(), h_rotq.data()));
    const size_t widths[5] = { 3, 3, 4, (size_t)feat, 1 };
    const float* host[5]   = { h_pos.data(), h_scale.data(), h_rotq.data(), h_sh.data(), h_op.data() };
    DevArr       act[5], g_ref[5], g_own[5], img_ref, img_own, dL;
    for (int a = 0; a < 5; ++a) {
        LCGS_TRY(act[a].alloc((size_t)P * widths[a] * 4));
        LCGS_TRY(g_ref[a].alloc((size_t)P * widths[a] * 4));
        LCGS_TRY(g_own[a].alloc((size_t)P * widths[a] * 4));
        LCGS_HIP_CHECK(hipMemcpy(act[a].p, host[a], (size_t)P * widths[a] * 4, hipMemcpyHostToDevice));
    }
    const size_t img_bytes = (size_t)3 * W * H * 4;
    LCGS_TRY(img_ref.alloc(img_bytes));
    LCGS_TRY(img_own.alloc(img_bytes));
    LCGS_TRY(dL.alloc(img_bytes));
    {
        std::vector<float> h((size_t)3 * W * H);
        uint32_t           x = 12345u + (uint32_t)me * 977u; // (every rank differentiates its own view with its own loss gradient)
        for (float& v : h) {
            x = x * 1664525u + 1013904223u;
            v = ((float)(x >> 8) / 16777216.0f) - 0.5f;
        }
        LCGS_HIP_CHECK(hipMemcpy(dL.p, h.data(), img_bytes, hipMemcpyHostToDevice));
    }
    std::vector<lcgs_camera> cams((size_t)N);
    for (int v = 0; v < N; ++v) {
        const float a = 0.35f * (float)v, pos[3] = { -3.0f * std::cos(a), -0.5f + 3.0f * std::sin(a), 2.3f }, tgt[3] = { 0, 0, 0.5f },
                    up[3] = { 0, 0, 1 };
        lcgs_get_lookat_cam(pos, tgt, up, &cams[(size_t)v]);
        cams[(size_t)v].width = W, cams[(size_t)v].height = H, cams[(size_t)v].aspect_ratio = (float)W / (float)H;
    }
    // ---- the context's own binding is put back whatever happens below
    struct Binding {
        lcgs_context* ctx;
        int           P = 0, deg = 3;
        const float * pos = nullptr, *scale = nullptr, *rotq = nullptr, *sh = nullptr, *op = nullptr;
        ~Binding()
        {
            (void)lcgs_synchronize(ctx);
            (void)lcgs_scene_bind(ctx, pos ? P : 0, deg, pos, scale, rotq, sh, op);
        }
    } keep{ ctx };
    LCGS_TRY(lcgs_scene_pointers(ctx, &keep.P, &keep.deg, &keep.pos, &keep.scale, &keep.rotq, &keep.sh, &keep.op));
    LCGS_TRY(lcgs_scene_bind(ctx, P, 3, act[0].as<float>(), act[1].as<float>(), act[2].as<float>(), act[3].as<float>(), act[4].as<float>()));
    const float bg[3] = { 0.1f, 0.2f, 0.3f };
    // ---- what the step must reproduce: my view's fused frame; my rows' gradients = the sum of every view's ordinary backward
    // (every view differentiated with ITS rank's loss gradient: regenerate those)
    lcgs_grads gr = { g_ref[0].as<float>(), g_ref[1].as<float>(), g_ref[2].as<float>(), g_ref[3].as<float>(), g_ref[4].as<float>() };
    DevArr     dLv, scratch;
    LCGS_TRY(dLv.alloc(img_bytes));
    LCGS_TRY(scratch.alloc(img_bytes));
    for (int v = 0; v < N; ++v) {
        std::vector<float> h((size_t)3 * W * H);
        uint32_t           x = 12345u + (uint32_t)v * 977u;
        for (float& q : h) {
            x = x * 1664525u + 1013904223u;
            q = ((float)(x >> 8) / 16777216.0f) - 0.5f;
        }
        LCGS_HIP_CHECK(hipMemcpy(dLv.p, h.data(), img_bytes, hipMemcpyHostToDevice));
        int n = 0;
        LCGS_TRY(lcgs_render_forward(ctx, &cams[(size_t)v], bg, 1.0f, v == me ? img_ref.as<float>() : scratch.as<float>(), nullptr, 1, &n));
        if (n > 0) LCGS_TRY(v == 0 ? lcgs_render_backward(ctx, dLv.as<float>(), &gr) : lcgs_render_backward_accumulate(ctx, dLv.as<float>(), &gr));
        else if (v == 0)
            for (int a = 0; a < 5; ++a) LCGS_HIP_CHECK(hipMemsetAsync(g_ref[a].p, 0, (size_t)P * widths[a] * 4, ctx->stream));
        LCGS_TRY(lcgs_synchronize(ctx)); // (dLv is rewritten by a blocking copy at the top of the loop: the backward must be through)
    }
    // ---- the step: once reading its sizes back, then twice without (the second of those is sized by the first's table)
    lcgs_grads   go = { g_own[0].as<float>(), g_own[1].as<float>(), g_own[2].as<float>(), g_own[3].as<float>(), g_own[4].as<float>() };
    int64_t      first = 0, count = 0;
    lcgs_comm_owner_rows(P, N, me, &first, &count);
    const bool   was_async = c->owner_async;
    const double t0 = now_s();
    double       worst = 0.0;
    bool         image_ok = true;
    lcgs_status  st = LCGS_OK;
    for (int round = 0; round < 3 && st == LCGS_OK; ++round) {
        c->owner_async = round > 0;
        LCGS_HIP_CHECK(hipMemsetAsync(img_own.p, 0, img_bytes, ctx->stream));
        for (int attempt = 0; attempt < 3 && st == LCGS_OK; ++attempt) {
            st = lcgs_owner_step_forward(ctx, c, cams.data(), bg, 1.0f, img_own.as<float>());
            if (st == LCGS_OK) st = lcgs_owner_step_backward(ctx, c, dL.as<float>(), &go);
            int redo = 0;
            if (st == LCGS_OK) st = lcgs_owner_step_finish(ctx, c, &redo);
            if (!redo) break;
        }
        if (st != LCGS_OK) break;
        if (lcgs_synchronize(ctx) != LCGS_OK) st = LCGS_ERR_HIP;
        std::vector<float> a((size_t)3 * W * H), b((size_t)3 * W * H);
        LCGS_HIP_CHECK(hipMemcpy(a.data(), img_own.p, img_bytes, hipMemcpyDeviceToHost));
        LCGS_HIP_CHECK(hipMemcpy(b.data(), img_ref.p, img_bytes, hipMemcpyDeviceToHost));
        if (memcmp(a.data(), b.data(), img_bytes) != 0) image_ok = false;
        for (int k = 0; k < 5; ++k) {
            const size_t       n = (size_t)count * widths[k];
            std::vector<float> x(n), y(n);
            if (n == 0) continue;
            LCGS_HIP_CHECK(hipMemcpy(x.data(), g_own[k].as<float>() + (size_t)first * widths[k], n * 4, hipMemcpyDeviceToHost));
            LCGS_HIP_CHECK(hipMemcpy(y.data(), g_ref[k].as<float>() + (size_t)first * widths[k], n * 4, hipMemcpyDeviceToHost));
            double num = 0.0, den = 0.0;
            for (size_t i = 0; i < n; ++i) {
                num += ((double)x[i] - y[i]) * ((double)x[i] - y[i]);
                den += (double)y[i] * y[i];
            }
            worst = std::max(worst, std::sqrt(num / std::max(den, 1e-30)));
            if (getenv("LCGS_SELFTEST_DEBUG")) fprintf(stderr, "[selftest] rank %d round %d attr %d: err %.3e (|ref| %.3e), image_ok %d\n", me, round, k, std::sqrt(num / std::max(den, 1e-30)), std::sqrt(den), (int)image_ok);
        }
    }
    c->owner_async = was_async;
    c->prev.have   = false; // (the table belongs to the scratch scene)
    LCGS_TRY(st);
    rep->owner_step_ms      = (now_s() - t0) * 1e3;
    rep->owner_max_grad_err = worst;
    rep->owner_step_ok      = (image_ok && worst <= 1e-4) ? 1 : 0;
    return LCGS_OK;
}
} // namespace

extern "C" {

lcgs_status lcgs_comm_selftest(lcgs_context* ctx, lcgs_comm* c, double timeout_s, lcgs_comm_selftest_report* out)
{
    LCGS_REQUIRE(ctx && c && out, "NULL argument");
    LCGS_REQUIRE(c->ctx == ctx, "the communicator belongs to another (or a destroyed) context");
    LCGS_REQUIRE(timeout_s > 0.0, "timeout_s must be positive");
    *out            = lcgs_comm_selftest_report{};
    out->world_size = c->world;
    out->rank       = c->rank;
    auto sh = std::make_shared<SelftestShared>(); // (outlives this call if a phase never returns: the worker is then detached)
    sh->rep = *out;
    std::thread worker([sh, ctx, c] {
        lcgs_status s = hipSetDevice(ctx->device) == hipSuccess ? LCGS_OK : LCGS_ERR_HIP;
        auto        enter = [&](int p) {
            sh->phase_start.store(now_s());
            sh->phase.store(p);
        };
        if (s == LCGS_OK) {
            enter(1);
            s = selftest_allreduce(c, &sh->rep);
        }
        if (s == LCGS_OK) {
            enter(2);
            s = selftest_p2p(c, &sh->rep);
        }
        if (s == LCGS_OK) {
            enter(3);
            s = selftest_owner_step(ctx, c, &sh->rep);
        }
        if (s != LCGS_OK) sh->error = lcgs_last_error(); // (thread-local in the worker: carried over)
        sh->status = s;
        sh->phase.store(4);
        sh->done.store(true);
    });
    sh->phase_start.store(now_s());
    while (!sh->done.load()) {
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
        if (now_s() - sh->phase_start.load() > timeout_s && !sh->done.load()) {
            const int p = sh->phase.load();
            worker.detach(); // it may never return; the process is expected to report and exit
            *out           = sh->rep;
            out->timed_out = p > 0 ? p : 1;
            snprintf(out->message, sizeof(out->message), "phase %d (%s) did not finish within %.1f s", out->timed_out,
                     out->timed_out == 1 ? "all-reduce" : (out->timed_out == 2 ? "point-to-point" : "ownership step"), timeout_s);
            set_last_error(out->message);
            return LCGS_ERR_STATE;
        }
    }
    worker.join();
    *out = sh->rep;
    if (sh->status != LCGS_OK) {
        snprintf(out->message, sizeof(out->message), "%s", sh->error.c_str());
        set_last_error(sh->error);
        return sh->status;
    }
    const bool ok = out->allreduce_ok == 1 && out->p2p_ok == 1 && out->owner_step_ok != 0;
    snprintf(out->message, sizeof(out->message), ok ? "ok" : "a phase gave wrong results (see the *_ok fields)");
    if (!ok) {
        set_last_error("lcgs_comm_selftest: a phase gave wrong results");
        return LCGS_ERR_STATE;
    }
    return LCGS_OK;
}

} // extern "C"
