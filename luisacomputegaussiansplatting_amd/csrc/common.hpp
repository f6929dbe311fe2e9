// common.hpp -- host-side plumbing shared by the C-ABI translation units.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/lcgs_hip.h"
#include "kernels/gs_math.hpp"

namespace lcgs
{

void        set_last_error(const std::string& msg);
lcgs_status hip_fail(hipError_t e, const char* what, const char* file, int line);

#define LCGS_HIP_CHECK(expr)                                                       \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess) return ::lcgs::hip_fail(_e, #expr, __FILE__, __LINE__); \
    } while (0)

#define LCGS_REQUIRE(cond, msg)                                  \
    do {                                                         \
        if (!(cond)) {                                           \
            ::lcgs::set_last_error(std::string("invalid argument: ") + (msg)); \
            return LCGS_ERR_INVALID_ARG;                         \
        }                                                        \
    } while (0)

// A device allocation that only ever grows (geometric growth, like the reference's
// ensure_*_temp_buffer, lcgs/src/gs_tile_splatter/impl.cpp:31-61).
struct DeviceBuffer {
    void*  ptr   = nullptr;
    size_t bytes = 0;
    lcgs_status ensure(size_t need);
    void        release();
    template <typename T>
    T* as() const { return reinterpret_cast<T*>(ptr); }
};

// host camera -> kernel constants (lcgs/src/gs_projector/impl.cpp:34-42, util/camera.h:38-72)
CamParams make_cam_params(const lcgs_camera& cam);

// stream of an (opaque) context, for translation units that only see the forward declaration
hipStream_t context_stream(lcgs_context* ctx);
int         context_device(lcgs_context* ctx);

// what the device ingest path needs to know about a PLY file (host/ply.cpp)
struct PlyProbe {
    int64_t  num_vertices = 0;
    size_t   stride = 0, payload_offset = 0;
    uint32_t column_offset[59] = {};
    bool     device_ok = false;
};
lcgs_status ply_probe(const char* path, PlyProbe* out);

inline uint32_t div_up(uint32_t a, uint32_t b) { return (a + b - 1u) / b; }
inline int64_t  div_up64(int64_t a, int64_t b) { return (a + b - 1) / b; }

} // namespace lcgs
