// scratch_cost.hip -- what a kernel's private segment costs at launch: the same trivial kernel with 0 .. 1056 bytes of scratch per lane, grids of
// 512 .. 24576 workgroups of 256 threads, 50 back-to-back launches each (HIP events).  Question behind it (round 6): the sweep with the
// solver step in its tail needs 752 B of scratch per lane for the step's serial float64 lane; is that what made it slower?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int WORDS>
__global__ __launch_bounds__(256) void k(float* out, const int* idx, int n) {
  if constexpr (WORDS > 0) {
    volatile float a[WORDS];
    for (int i = 0; i < WORDS; ++i) a[i] = float(i);
    float s = 0.0f;
    if (blockIdx.x == 0x7fffffff) for (int i = 0; i < n; ++i) s += a[idx[i] % WORDS];   // (dynamic index: the array stays in scratch; never taken)
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = s + a[idx[0] % WORDS];
  } else {
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = float(n);
  }
}
template <int WORDS>
static void run(float* out, int* idx, int grid) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) k<WORDS><<<grid, 256>>>(out, idx, 1);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 50; ++i) k<WORDS><<<grid, 256>>>(out, idx, 1);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::printf("scratch %4d B/lane  grid %6d: %7.2f us per launch\n", WORDS * 4, grid, ms * 1000.0f / 50);
}
int main() {
  float* out; int* idx;
  hipMalloc(&out, 4); hipMalloc(&idx, 4); hipMemset(idx, 0, 4);
  for (int grid : {512, 2048, 8192, 24576}) {
    run<0>(out, idx, grid); run<16>(out, idx, grid); run<64>(out, idx, grid); run<88>(out, idx, grid); run<188>(out, idx, grid); run<264>(out, idx, grid);
  }
  return 0;
}
