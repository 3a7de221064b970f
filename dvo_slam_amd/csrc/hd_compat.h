// hd_compat.h -- lets the per-pixel math and the Gauss-Newton state machine (pixel_math.h,
// solver_logic.h, se3_device.h) be compiled either by hipcc for gfx950 (the product) or by a plain
// host C++ compiler inside tests/ (logic emulation of the device code without a GPU; test-only,
// never linked into libdvo_hip.so).
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DVO_HD __host__ __device__ __forceinline__
#if defined(__HIP_DEVICE_COMPILE__)
#define DVO_FENCE_BLOCK() __threadfence_block()
#else
#define DVO_FENCE_BLOCK() ((void)0)
#endif
#else
#include <math.h>
#define DVO_HD inline
#define DVO_FENCE_BLOCK() ((void)0)
#ifndef __host__
#define __host__
#endif
#ifndef __device__
#define __device__
#endif
struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
#endif
