// stream_access.hpp -- streaming (non-temporal) global accesses for data a kernel touches once and nothing reads again soon:
// the L2 / MALL keep what IS re-read (count tables, sorted keys, records) instead.  Measured per use, same-box A/B
// (profiles/r04_nt_accesses_ab.txt): kept only where it paid.
#pragma once
#include <hip/hip_runtime.h>

namespace lcgs
{
typedef float v4f_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_stream(const float4* p)
{
    const v4f_t t = __builtin_nontemporal_load(reinterpret_cast<const v4f_t*>(p));
    return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void st_stream(float4* p, const float4& x)
{
    v4f_t t = { x.x, x.y, x.z, x.w };
    __builtin_nontemporal_store(t, reinterpret_cast<v4f_t*>(p));
}
__device__ __forceinline__ float ld_stream(const float* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void  st_stream(float* p, float x) { __builtin_nontemporal_store(x, p); }
} // namespace lcgs
