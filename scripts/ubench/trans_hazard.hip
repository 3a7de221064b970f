// trans_hazard.hip -- reproducer (round 4): on gfx950 a vector instruction that reads the result of a transcendental instruction
// (v_rcp_f32 here) needs a wait state behind it.  hipcc inserts it for instructions it generates; it does not look into inline
// assembly.  Kernel 0 multiplies the fresh reciprocal inside inline assembly (what dvo_slam_amd/csrc/gram_f16.h::mul_legacy did
// with the reciprocal of the depth whenever the scheduler put the two next to each other) -- the product is made with the register's OLD
// content; kernel 1 puts `s_nop 1` between the two (gram_f16.h::rcp_for_inline_asm); kernel 2 is the plain C++ product.
//   hipcc --offload-arch=gfx950 -O3 -o trans_hazard trans_hazard.hip && ./trans_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void k(const float* in, float* out) {
  const float x = in[threadIdx.x], a = in[64 + threadIdx.x];
  float y;
  if (MODE == 0) {
    float r;                                              // one asm block: nothing can be scheduled between the two instructions
    asm volatile("v_mov_b32 %1, 0x42280000\n\tv_rcp_f32 %1, %2\n\tv_mul_legacy_f32 %0, %3, %1" : "=v"(y), "=&v"(r) : "v"(x), "v"(a));
  } else if (MODE == 1) {
    float r;
    asm volatile("v_mov_b32 %1, 0x42280000\n\tv_rcp_f32 %1, %2\n\ts_nop 1\n\tv_mul_legacy_f32 %0, %3, %1" : "=v"(y), "=&v"(r) : "v"(x), "v"(a));
  } else {
    y = a * __builtin_amdgcn_rcpf(x);
  }
  out[threadIdx.x] = y;
}

int main() {
  std::vector<float> h(128);
  for (int i = 0; i < 64; ++i) { h[i] = 1.0f + 0.25f * i; h[64 + i] = 3.0f; }
  float *d_in, *d_out;
  hipMalloc(&d_in, 512); hipMalloc(&d_out, 256);
  hipMemcpy(d_in, h.data(), 512, hipMemcpyHostToDevice);
  int bad[3] = {0, 0, 0}, first_bad = -1;
  float bad_value = 0;
  for (int mode = 0; mode < 3; ++mode) {
    if (mode == 0) k<0><<<1, 64>>>(d_in, d_out);
    if (mode == 1) k<1><<<1, 64>>>(d_in, d_out);
    if (mode == 2) k<2><<<1, 64>>>(d_in, d_out);
    std::vector<float> o(64);
    hipMemcpy(o.data(), d_out, 256, hipMemcpyDeviceToHost);
    for (int i = 0; i < 64; ++i) {
      const bool off = !(o[i] > 0.999f * 3.0f / h[i] && o[i] < 1.001f * 3.0f / h[i]);
      bad[mode] += off;
      if (off && mode == 0 && first_bad < 0) { first_bad = i; bad_value = o[i]; }
    }
  }
  printf("a * rcp(x), 64 lanes, a = 3: lanes off by more than 1e-3 -- asm back to back %d (first: lane %d = %g, expected %g; 3 x the register's old value 42 = 126), "
         "asm with s_nop 1 %d, compiler-generated %d\n", bad[0], first_bad, bad_value, first_bad >= 0 ? 3.0f / h[first_bad] : 0.0f, bad[1], bad[2]);
  return bad[1] || bad[2];
}
