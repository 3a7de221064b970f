// issue_rate.hip -- micro-benchmark (round 4): issue cost per wavefront instruction of the vector instructions the window sweep is made
// of, on gfx950 (wave64, 5 and 8 wavefronts per SIMD resident).  Sixteen independent chains per lane, 4000 iterations.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float __attribute__((ext_vector_type(2))) f2;
typedef unsigned __attribute__((ext_vector_type(2))) u2;

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  float x[16];
  for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i + 1.0f;
  const unsigned long long mask = __builtin_amdgcn_ballot_w64(threadIdx.x & 1);
  const float sa = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a)));
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
      if (MODE == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
      if (MODE == 2) asm volatile("v_mul_legacy_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
      if (MODE == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
      if (MODE == 4) asm volatile("v_rsq_f32 %0, %0" : "+v"(x[i]));
      if (MODE == 5) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
      if (MODE == 6) asm volatile("v_fma_mix_f32 %0, %0, 1.0, -%1 op_sel_hi:[0,0,1]" : "+v"(x[i]) : "v"(a));
      if (MODE == 7) asm volatile("v_fract_f32 %0, %0" : "+v"(x[i]));
      if (MODE == 8) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(x[i]));
      if (MODE == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a));
      if (MODE == 10) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x[i]));
      if (MODE == 11) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x[i]));
      if (MODE == 12) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(x[i]) : "v"(a));
      if (MODE == 13) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(x[i]), "v"(a) : "vcc");
      if (MODE == 14) asm volatile("v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x[i]));
      if (MODE == 15) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "s"(mask));
      if (MODE == 16) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
      if (MODE == 17) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
      if (MODE == 18) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x[i]) : "v"(a));
      if (MODE == 19) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
      if (MODE == 24) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
      if (MODE == 25) asm volatile("v_floor_f32 %0, %0" : "+v"(x[i]));
      if (MODE == 26) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x[i]) : "v"(a));
      if (MODE == 27) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
      if (MODE == 28) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "s"(sa), "v"(b));
      if (MODE == 29) asm volatile("v_cmp_le_u32 vcc, %0, %1" : : "v"(x[i]), "v"(a) : "vcc");
      if (MODE == 30) asm volatile("v_mov_b32 %0, %1" : "+v"(x[i]) : "v"(a));
      if (MODE == 31) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "s"(sa));
    }
    if (MODE == 20) {
#pragma unroll
      for (int i = 0; i < 16; i += 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*reinterpret_cast<f2*>(&x[i])) : "v"(f2{a, b}));
    }
    if (MODE == 21) {
#pragma unroll
      for (int i = 0; i < 16; i += 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*reinterpret_cast<f2*>(&x[i])) : "v"(f2{a, a}), "v"(f2{b, b}));
    }
    if (MODE == 22) {
#pragma unroll
      for (int i = 0; i < 16; i += 2) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x[i]), "+v"(x[i + 1]));
    }
    if (MODE == 23) {
#pragma unroll
      for (int i = 0; i < 16; i += 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<f2*>(&x[i])) : "v"(f2{a, a}));
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int per_iter_instr) {
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
  printf("%-24s %8.3f ms  %.3f ns per wave-instr per SIMD  (= %.2f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
  hipFree(out);
}

int main() {
  run<0>("v_fma_f32", 16);
  run<1>("v_mul_f32", 16);
  run<2>("v_mul_legacy_f32", 16);
  run<3>("v_rcp_f32", 16);
  run<4>("v_rsq_f32", 16);
  run<5>("v_cvt_pk_f16_f32", 16);
  run<6>("v_fma_mix_f32", 16);
  run<7>("v_fract_f32", 16);
  run<8>("v_cvt_i32_f32", 16);
  run<9>("v_cndmask_b32", 16);
  run<10>("v_mov_b32_dpp row_shr", 16);
  run<11>("v_mov_b32_dpp wave_shr", 16);
  run<12>("v_pk_min_i16", 16);
  run<13>("v_cmp_gt_f32", 16);
  run<14>("v_max_i32_dpp", 16);
  run<15>("v_cndmask_b32_e64 sgpr", 16);
  run<16>("v_and_b32", 16);
  run<17>("v_add_u32", 16);
  run<18>("v_lshl_add_u32", 16);
  run<19>("v_sub_f32", 16);
  run<24>("v_min_f32", 16);
  run<25>("v_floor_f32", 16);
  run<26>("v_mad_u32_u24", 16);
  run<27>("v_fmac_f32", 16);
  run<28>("v_fma_f32 sgpr operand", 16);
  run<29>("v_cmp_le_u32", 16);
  run<30>("v_mov_b32", 16);
  run<31>("v_mul_f32 sgpr operand", 16);
  run<20>("v_pk_add_f32", 8);
  run<21>("v_pk_fma_f32", 8);
  run<23>("v_pk_mul_f32", 8);
  run<22>("v_permlane32_swap_b32", 8);
  return 0;
}
