// lcgs.hpp -- C++ host-side mirror of the reference's operator API on top of the C ABI (include/lcgs_hip.h).
// Same class names, method names, argument meaning and proxy structs as the reference's headers, so a caller
// written against lcgs/include/lcgs/{sh_preprocessor,gs_projector,gs_tile_splatter,proxy}.h maps one to one:
//   luisa::compute::Device / Stream      -> lcgs::Device (one lcgs_context = one GPU + one HIP stream)
//   luisa::compute::BufferView<T>        -> lcgs::BufferView<T> (non-owning device pointer + element count)
//   CommandList                          -> not needed: every call enqueues on the Device's stream
// Errors: the reference is noexcept + LUISA_ERROR(abort); here a failing call throws lcgs::Error carrying the
// lcgs_status and lcgs_last_error() text.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "lcgs_hip.h"

namespace lcgs
{

struct Error : std::runtime_error {
    lcgs_status status;
    Error(lcgs_status s, const std::string& what) : std::runtime_error(what), status(s) {}
};

inline void check(lcgs_status s)
{
    if (s != LCGS_OK) throw Error(s, std::string("lcgs status ") + std::to_string((int)s) + ": " + lcgs_last_error());
}

template <typename T>
struct BufferView {
    T*     ptr  = nullptr;
    size_t size = 0;
    BufferView() = default;
    BufferView(T* p, size_t n) : ptr(p), size(n) {}
    BufferView subview(size_t offset, size_t n) const { return BufferView(ptr + offset, n); }
};

// owning device buffer (Device::create_buffer<T>, app/main.cpp:180-186)
template <typename T>
class Buffer
{
public:
    Buffer() = default;
    explicit Buffer(size_t n) : m_size(n)
    {
        if (hipMalloc(reinterpret_cast<void**>(&m_ptr), (n ? n : 1) * sizeof(T)) != hipSuccess)
            throw Error(LCGS_ERR_OUT_OF_MEMORY, "hipMalloc failed");
    }
    Buffer(const Buffer&)            = delete;
    Buffer& operator=(const Buffer&) = delete;
    Buffer(Buffer&& o) noexcept : m_ptr(o.m_ptr), m_size(o.m_size) { o.m_ptr = nullptr; }
    Buffer& operator=(Buffer&& o) noexcept
    {
        if (this != &o) {
            if (m_ptr) (void)hipFree(m_ptr);
            m_ptr    = o.m_ptr;
            m_size   = o.m_size;
            o.m_ptr  = nullptr;
            o.m_size = 0;
        }
        return *this;
    }
    ~Buffer()
    {
        if (m_ptr) (void)hipFree(m_ptr);
    }
    BufferView<T> view() const { return BufferView<T>(m_ptr, m_size); }
    operator BufferView<T>() const { return view(); }
    T*     data() const { return m_ptr; }
    size_t size() const { return m_size; }

private:
    T*     m_ptr  = nullptr;
    size_t m_size = 0;
};

// lcgs/include/lcgs/util/camera.h:15-25 -- the POD is shared with the C ABI
using Camera = lcgs_camera;
inline Camera get_lookat_cam(const float pos[3], const float target[3], const float world_up[3])
{
    Camera c;
    lcgs_get_lookat_cam(pos, target, world_up, &c);
    return c;
}

class Device
{
public:
    explicit Device(int device_id = 0, hipStream_t stream = nullptr) { check(lcgs_create(device_id, stream, &m_ctx)); }
    Device(const Device&)            = delete;
    Device& operator=(const Device&) = delete;
    ~Device() { lcgs_destroy(m_ctx); }
    lcgs_context* ctx() const { return m_ctx; }
    void          synchronize() { check(lcgs_synchronize(m_ctx)); }
    template <typename T>
    Buffer<T> create_buffer(size_t n) { return Buffer<T>(n); }

private:
    lcgs_context* m_ctx = nullptr;
};

// ---- proxies: lcgs/include/lcgs/sh_preprocessor.h:16-20, gs_projector.h:16-28, proxy.h:43-71 ----
struct GPUPointsProxy {
    int               N      = 0;
    int               stride = 3;
    BufferView<float> pos;
};
struct GSProjectorInputProxy {
    int               num_gaussians;
    BufferView<float> pos, scale, rotq;
    float             scale_modifier;
};
struct GSProjectorOutputProxy {
    BufferView<float> means_2d, covs_2d, depth;
};
struct GSTileSplatterInputProxy {
    int               num_gaussians;
    float             bg_color[3];
    BufferView<float> means_2d, depth_features, conic, color_features, opacity_features;
};
struct GSTileSplatterAccelProxy {
    BufferView<uint32_t> tiles_touched, point_offsets;
    BufferView<uint64_t> point_list_keys_unsorted;
    BufferView<uint32_t> point_list_unsorted;
    BufferView<uint64_t> point_list_keys;
    BufferView<uint32_t> point_list, ranges;
};
struct GSSplatForwardOutputProxy {
    int               height, width;
    BufferView<float> target_img; // written planar CHW (gs_tile_splatter/shader.cpp:279-286)
    BufferView<int>   radii;
};

// lcgs::SHProcessor (sh_preprocessor.h:22-57)
class SHProcessor
{
public:
    void create(Device& device) noexcept { m_dev = &device; }
    void process(GPUPointsProxy proxy, Camera& camera, BufferView<float> sh, BufferView<float> color, int channel = 3,
                 int level = 3)
    {
        check(lcgs_sh_process(m_dev->ctx(), proxy.N, proxy.pos.ptr, &camera, sh.ptr, color.ptr, level, channel));
    }

private:
    Device* m_dev = nullptr;
};

// lcgs::GSProjector (gs_projector.h:30-87)
class GSProjector
{
public:
    void create(Device& device) noexcept { m_dev = &device; }
    void forward(GSProjectorInputProxy input, GSProjectorOutputProxy output, Camera& cam, bool use_focal = true)
    {
        check(lcgs_project_forward(m_dev->ctx(), input.num_gaussians, input.pos.ptr, input.scale.ptr, input.rotq.ptr,
                                   input.scale_modifier, output.means_2d.ptr, output.covs_2d.ptr, output.depth.ptr, &cam,
                                   use_focal ? 1 : 0));
    }

private:
    Device* m_dev = nullptr;
};

// lcgs::GSTileSplatter (gs_tile_splatter.h:19-106).  The scan / radix-sort / filler objects the reference
// borrows (set_device_scan etc.) are owned by the context here.
class GSTileSplatter
{
public:
    int      num_rendered = 0;            // gs_tile_splatter.h:23
    uint32_t m_blocks[2]  = { 16u, 16u }; // module.h:17
    void     create(Device& device) noexcept { m_dev = &device; }
    int forward(GSTileSplatterAccelProxy accel, GSTileSplatterInputProxy input, GSSplatForwardOutputProxy output,
                bool use_focal = true)
    {
        lcgs_tile_accel a{ accel.tiles_touched.ptr, accel.point_offsets.ptr, accel.point_list_keys_unsorted.ptr,
                           accel.point_list_unsorted.ptr, accel.point_list_keys.ptr, accel.point_list.ptr,
                           accel.ranges.ptr, (int64_t)accel.point_list.size };
        lcgs_tile_input i{ input.num_gaussians, { input.bg_color[0], input.bg_color[1], input.bg_color[2] },
                           input.means_2d.ptr, input.depth_features.ptr, input.conic.ptr, input.color_features.ptr,
                           input.opacity_features.ptr };
        lcgs_tile_output o{ output.height, output.width, output.target_img.ptr, output.radii.ptr, nullptr, nullptr };
        check(lcgs_tile_splat_forward(m_dev->ctx(), &a, &i, &o, use_focal ? 1 : 0, &num_rendered));
        return num_rendered;
    }

private:
    Device* m_dev = nullptr;
};

// ---- beyond the reference's three operators: the fused frame and what surrounds it (no counterpart classes in the
// reference; thin wrappers over the C ABI so that C++ callers do not have to drop to it) ----
class Scene
{
public:
    explicit Scene(Device& device) : m_dev(&device) {}
    // read_gs_ply + upload with the de-interleave / activations on the device (app/gaussians.cpp:75-171,
    // app/main.cpp:180-186,216-223)
    int load_ply(const std::string& path)
    {
        int n = 0;
        check(lcgs_scene_load_ply(m_dev->ctx(), path.c_str(), &n));
        return n;
    }
    void bind(int num_gaussians, BufferView<float> pos, BufferView<float> scale, BufferView<float> rotq,
              BufferView<float> sh, BufferView<float> opacity, int sh_degree = 3)
    {
        check(lcgs_scene_bind(m_dev->ctx(), num_gaussians, sh_degree, pos.ptr, scale.ptr, rotq.ptr, sh.ptr, opacity.ptr));
    }
    // one frame (app/main.cpp:266-308 in one submission); returns the reference's num_rendered
    int render(const Camera& cam, BufferView<float> img, const float bg[3], float scale_modifier = 1.0f,
               bool keep_state = false)
    {
        int n = 0;
        check(lcgs_render_forward(m_dev->ctx(), &cam, bg, scale_modifier, img.ptr, nullptr, keep_state ? 1 : 0, &n));
        return n;
    }
    // a batch of views, two frames in flight; enqueues only
    void render_batch(const std::vector<Camera>& cams, const std::vector<float*>& imgs, const float bg[3],
                      float scale_modifier = 1.0f)
    {
        if (cams.size() != imgs.size()) throw Error(LCGS_ERR_INVALID_ARG, "render_batch: cams/imgs size mismatch");
        check(lcgs_render_forward_batch(m_dev->ctx(), (int)cams.size(), cams.data(), bg, scale_modifier, imgs.data()));
    }
    void backward(BufferView<float> dL_dimg, const lcgs_grads& grads)
    {
        check(lcgs_render_backward(m_dev->ctx(), dL_dimg.ptr, &grads));
    }
    // a further view of a multi-view step: its gradients are added to the arrays
    void backward_accumulate(BufferView<float> dL_dimg, const lcgs_grads& grads)
    {
        check(lcgs_render_backward_accumulate(m_dev->ctx(), dL_dimg.ptr, &grads));
    }
    // the views of one optimiser step: forward -> L2 loss against targets[j] -> backward, gradients summed into `grads`,
    // d_losses[j] = view j's loss; a view's forward runs beside the previous view's backward (lcgs_fit_views)
    void fit_views(const std::vector<Camera>& cams, const std::vector<const float*>& targets, const lcgs_grads& grads,
                   float* d_losses, const float bg[3], float scale_modifier = 1.0f)
    {
        if (cams.size() != targets.size()) throw Error(LCGS_ERR_INVALID_ARG, "fit_views: cams/targets size mismatch");
        check(lcgs_fit_views(m_dev->ctx(), (int)cams.size(), cams.data(), bg, scale_modifier, targets.data(), &grads, d_losses));
    }
    // the same gradients as compact rows (row r = the frame's r-th on-screen splat, see visible_rows): single-GPU steps
    void backward_compact(BufferView<float> dL_dimg, const lcgs_grads& grads)
    {
        check(lcgs_render_backward_compact(m_dev->ctx(), dL_dimg.ptr, &grads));
    }
    // splat index of every compact row and the address of the device-side row count (valid until the next frame)
    void visible_rows(const uint32_t** d_rows, const uint32_t** d_count)
    {
        check(lcgs_visible_rows(m_dev->ctx(), d_rows, d_count));
    }
    // Morton order at ingest: the context renders from its own re-ordered copy; d_perm[r] = old index of new splat r
    void reorder_spatial(uint32_t* d_perm = nullptr) { check(lcgs_scene_reorder_spatial(m_dev->ctx(), d_perm)); }

private:
    Device* m_dev = nullptr;
};

// The RCCL communicator of a view-parallel job (one process per GPU): gradient sums over the ranks.
// `id` is made by one rank (Comm::unique_id) and carried to the others by the host (lcgs-app: pipes from the launcher).
class Comm
{
public:
    static lcgs_comm_id unique_id()
    {
        lcgs_comm_id id;
        check(lcgs_comm_unique_id(&id));
        return id;
    }
    Comm(Device& device, const lcgs_comm_id& id, int rank, int world_size) : m_dev(&device)
    {
        check(lcgs_comm_create(device.ctx(), &id, rank, world_size, &m_comm));
    }
    Comm(const Comm&)            = delete;
    Comm& operator=(const Comm&) = delete;
    ~Comm() { lcgs_comm_destroy(m_comm); }
    // in-place sum over the ranks of the dense gradients of lcgs_render_backward (chunked behind its slices)
    void allreduce(int num_gaussians, const lcgs_grads& grads, int sh_degree = 3)
    {
        check(lcgs_grads_allreduce(m_dev->ctx(), m_comm, num_gaussians, sh_degree, &grads));
    }
    // reduce-scatter -> Adam on the own rows -> all-gather of the activated arrays
    void adam_step_sharded(int num_gaussians, const lcgs_adam_config& cfg, const lcgs_grads& grads, const lcgs_params& raw,
                           const lcgs_params& m, const lcgs_params& v, const lcgs_params& activated, int sh_degree = 3)
    {
        check(lcgs_adam_step_sharded(m_dev->ctx(), m_comm, num_gaussians, sh_degree, &cfg, &grads, &raw, &m, &v, &activated));
    }
    // the same step with a sparse reduce half: only the rows this rank's views touched travel to their owners
    // (track_touched_rows(true) on every rank before the step's backward passes)
    void track_touched_rows(bool enable) { check(lcgs_comm_track_touched_rows(m_comm, enable ? 1 : 0)); }
    void adam_step_sparse(int num_gaussians, const lcgs_adam_config& cfg, const lcgs_grads& grads, const lcgs_params& raw,
                          const lcgs_params& m, const lcgs_params& v, const lcgs_params& activated, int sh_degree = 3)
    {
        check(lcgs_adam_step_sparse(m_dev->ctx(), m_comm, num_gaussians, sh_degree, &cfg, &grads, &raw, &m, &v, &activated));
    }
    // Splat ownership: rank r owns the rows owner_rows(P, N, r) and renders view r of cameras[N].  forward:
    // own rows projected for every view, records exchanged over RCCL point-to-point, this rank's view rendered into d_img
    // (the whole scene's fused frame, bit for bit); backward: the view's 2-D gradients back to the owners, parameter
    // gradients at the own rows of `grads` (apply lcgs_adam_step to those rows).  Every rank calls both, in this order.
    static void owner_rows(int64_t num_gaussians, int world_size, int rank, int64_t& first, int64_t& count)
    {
        lcgs_comm_owner_rows(num_gaussians, world_size, rank, &first, &count);
    }
    void owner_step_forward(const lcgs_camera* cameras, const float bg_color[3], float* d_img, float scale_modifier = 1.0f)
    {
        check(lcgs_owner_step_forward(m_dev->ctx(), m_comm, cameras, bg_color, scale_modifier, d_img));
    }
    void owner_step_backward(const float* d_dL_dimg, const lcgs_grads& grads)
    {
        check(lcgs_owner_step_backward(m_dev->ctx(), m_comm, d_dL_dimg, &grads));
    }
    // The step without a host read-back (lcgs_owner_step_set_async): message sizes from the previous step's counts.  Every
    // such step is closed with owner_step_finish(); true = some rank's step was short, EVERY rank repeats forward + backward.
    void owner_step_set_async(bool enable = true) { check(lcgs_owner_step_set_async(m_comm, enable ? 1 : 0)); }
    bool owner_step_finish()
    {
        int redo = 0;
        check(lcgs_owner_step_finish(m_dev->ctx(), m_comm, &redo));
        return redo != 0;
    }
    // forward + backward + finish, repeated while the step was short; returns the repetitions (0 normally)
    int owner_step(const lcgs_camera* cameras, const float bg_color[3], float* d_img, const float* d_dL_dimg,
                   const lcgs_grads& grads, float scale_modifier = 1.0f)
    {
        for (int attempt = 0; attempt < 3; ++attempt) {
            owner_step_forward(cameras, bg_color, d_img, scale_modifier);
            owner_step_backward(d_dL_dimg, grads);
            if (!owner_step_finish()) return attempt;
        }
        throw Error(LCGS_ERR_STATE, "the ownership step did not settle after two repetitions");
    }
    // lcgs_comm_selftest (a collective): 1 KB all-reduce, zero- and one-byte messages to every peer, an ownership step on a
    // scratch scene -- each against timeout_s.  Throws when a phase failed or never returned (the report says which).
    lcgs_comm_selftest_report selftest(double timeout_s = 30.0, bool throw_on_failure = true)
    {
        lcgs_comm_selftest_report rep;
        const lcgs_status         s = lcgs_comm_selftest(m_dev->ctx(), m_comm, timeout_s, &rep);
        if (s != LCGS_OK && throw_on_failure) throw Error(s, std::string("communicator self-test: ") + rep.message);
        return rep;
    }
    lcgs_comm_stats stats() const
    {
        lcgs_comm_stats st;
        check(lcgs_comm_get_stats(m_comm, &st));
        return st;
    }
    lcgs_comm* handle() const { return m_comm; }

private:
    Device*    m_dev  = nullptr;
    lcgs_comm* m_comm = nullptr;
};

} // namespace lcgs
