// global_ptr.h -- plane pointers that come out of a table in device memory, as pointers into the GLOBAL address space.
//
// The planes of a frame are reached through a table of pointers (FrameBuildPtrs).  As plain pointers the compiler has to treat them as
// FLAT addresses that any store of the kernel may have changed: it reloads the pointer before every access and waits for that scalar
// load, which on this hardware waits for every flat access issued before it (flat instructions count on both counters) -- one memory
// access in flight per lane.  A pointer read once into a local and cast to address space 1 becomes global_load / global_store with
// independent counters, and all loads of a tile are in flight together.
#pragma once
#include <hip/hip_runtime.h>

namespace dvo_hip {

template <typename T> using Global = __attribute__((address_space(1))) T*;
template <typename T> __device__ __forceinline__ Global<T> global_ptr(T* p) { return (Global<T>)p; }

// HIP's float2 / float4 are classes whose copy operations do not take address-space-qualified operands: vectors move through the
// compiler's native vector types
typedef float GlobalF32x2 __attribute__((ext_vector_type(2)));
typedef float GlobalF32x4 __attribute__((ext_vector_type(4)));
typedef unsigned GlobalU32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gstore(Global<float2> p, float2 v) { *(Global<GlobalF32x2>)p = GlobalF32x2{v.x, v.y}; }
__device__ __forceinline__ void gstore(Global<float4> p, float4 v) { *(Global<GlobalF32x4>)p = GlobalF32x4{v.x, v.y, v.z, v.w}; }
// two adjacent float2 elements (16-byte aligned) in one store
__device__ __forceinline__ void gstore_pair(Global<float2> p, float4 v) { *(Global<GlobalF32x4>)p = GlobalF32x4{v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ float2 gload(Global<const float2> p) {
  const GlobalF32x2 v = *(Global<const GlobalF32x2>)p;
  return make_float2(v.x, v.y);
}
__device__ __forceinline__ float4 gload(Global<const float4> p) {
  const GlobalF32x4 v = *(Global<const GlobalF32x4>)p;
  return make_float4(v.x, v.y, v.z, v.w);
}

}  // namespace dvo_hip
