// hbm_stream.hip -- what this MI355X sustains on streaming kernels (the yardstick next to the 8 TB/s spec peak in DESIGN.md
// section 5): read-only sum, copy, and the sweep kernel's own mix (40 B read + 8 B written per element), 16 B per lane,
// grid-stride, buffers far larger than the 256 MB Infinity Cache.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ a, size_t n, float* out) {
  float s = 0.0f;
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
    const float4 v = a[i];
    s += v.x + v.y + v.z + v.w;
  }
  if (s == 123.456f) *out = s;
}

__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) b[i] = a[i];
}

// per element: 16 B + 16 B + 8 B read, 8 B written (the reduce sweep's stream mix, without its gather)
__global__ __launch_bounds__(256) void k_mix(const float4* __restrict__ r, const float4* __restrict__ a, const float2* __restrict__ b,
                                             float2* __restrict__ o, size_t n) {
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
    const float4 x = r[i], y = a[i];
    const float2 z = b[i];
    o[i] = make_float2(x.x + y.y + z.x, x.w + y.z + z.y);
  }
}

template <typename F>
static double time_ms(F f, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  f();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main() {
  const size_t n = size_t(128) * 640 * 480;          // the sweep's 39.3 M elements
  const size_t big = size_t(1) << 27;                // 2 GiB of float4
  float4 *a, *b;
  float2 *c, *o;
  float* out;
  if (hipMalloc(&a, big * 16) != hipSuccess || hipMalloc(&b, big * 16) != hipSuccess || hipMalloc(&c, n * 8) != hipSuccess ||
      hipMalloc(&o, n * 8) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) {
    std::printf("allocation failed\n");
    return 1;
  }
  hipMemset(a, 0, big * 16);
  hipMemset(b, 0, big * 16);
  hipMemset(c, 0, n * 8);
  for (int blocks : {2048, 8192, 32768}) {
    const double r = time_ms([&] { k_read<<<blocks, 256>>>(a, big, out); }, 10);
    const double cp = time_ms([&] { k_copy<<<blocks, 256>>>(a, b, big); }, 10);
    const double mx = time_ms([&] { k_mix<<<blocks, 256>>>(a, b, c, o, n); }, 20);
    std::printf("blocks %6d: read %.0f GB/s, copy %.0f GB/s (read + write), sweep mix %.0f GB/s (48 B per element, %.3f ms for %zu elements)\n",
                blocks, big * 16 / r / 1e6, 2.0 * big * 16 / cp / 1e6, n * 48.0 / mx / 1e6, mx, n);
  }
  // the same mix on footprints that fit the 256 MB Infinity Cache: what a sub-batch of pairs swept repeatedly would see
  for (int pairs : {2, 4, 8, 16, 32, 64}) {
    const size_t m = size_t(pairs) * 640 * 480;
    const double mx = time_ms([&] { k_mix<<<8192, 256>>>(a, b, c, o, m); }, 40);
    std::printf("sweep mix on %2d pairs (%4.0f MB footprint, repeated): %.0f GB/s, %.4f ms\n", pairs, m * 48.0 / 1e6, m * 48.0 / mx / 1e6, mx);
  }
  return 0;
}
