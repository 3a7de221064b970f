// valu_rate.hip -- micro-benchmark: issue cost of v_fma_f32 vs v_pk_fma_f32 vs v_mul+v_add on gfx950 (wave64).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float __attribute__((ext_vector_type(2))) f2;

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  float x[16];
  f2 y[8];
  for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
  for (int i = 0; i < 8; ++i) y[i] = f2{x[2 * i], x[2 * i + 1]};
  const f2 a2 = {a, a}, b2 = {b, b};
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[i]) : "v"(a2), "v"(b2));
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y[i]) : "v"(a2));
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += x[i];
  for (int i = 0; i < 8; ++i) s += y[i].x + y[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int per_iter_instr, int flops_per_instr_lane) {
  float* out;
  const int blocks = 256 * 8, iters = 4000;
  hipMalloc(&out, blocks * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(out, 10, 1.0001f, 0.5f);
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(out, iters, 1.0001f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double waves = blocks * 4.0, instr = waves * iters * per_iter_instr;
  const double per_simd = instr / 1024.0;                 // 256 CUs x 4 SIMDs
  printf("%-14s %8.3f ms  %.2f ns per wave-instr per SIMD  (= %.2f cycles at 2.4 GHz)  %.1f TFLOP/s\n", name, ms, ms * 1e6 / per_simd,
         ms * 1e6 / per_simd * 2.4, instr * 64 * flops_per_instr_lane / (ms * 1e-3) / 1e12);
  hipFree(out);
}

int main() {
  run<0>("v_fma_f32", 16, 2);
  run<1>("v_pk_fma_f32", 8, 4);
  run<2>("v_mul_f32", 16, 1);
  run<3>("v_pk_mul_f32", 8, 2);
  return 0;
}
