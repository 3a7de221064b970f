// gram_f16.hip -- check of the f16 hi/lo Gram accumulation used by the sweep kernel (align_mfma.hip) in isolation:
// 64 pixel vectors of 16 float components -> G = sum_p v_p v_p^T through v_mfma_f32_16x16x32_f16 with each component split as
// v = hi + lo (two f16), G ~ H H^T + H L^T + L H^T; operands fetched with the LDS transpose read ds_read_b64_tr_b16 from a
// pixel-major [64][16] f16 image (row stride 48 B).  Prints the largest error relative to the largest entry against float64.
//   hipcc --offload-arch=gfx950 -O3 -o gram_f16 gram_f16.hip && ./gram_f16
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 __attribute__((ext_vector_type(8))) h8;
typedef __fp16 __attribute__((__vector_size__(4 * sizeof(__fp16)))) fp4v;
typedef float __attribute__((ext_vector_type(4))) f4;
constexpr int kRow = 24;   // halfs per pixel row (16 used + 8 pad: 48 B, conflict-free 16-B stores)

__device__ inline h8 read_operand(const _Float16* img, int lane, int chunk32) {
  const int i = lane & 15, g = lane >> 4;
  const _Float16* p0 = img + (chunk32 * 32 + 8 * g + (i >> 2)) * kRow + (i & 3) * 4;
  const fp4v a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp4v*)p0);
  const fp4v b = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp4v*)(p0 + 4 * kRow));
  return h8{(_Float16)a[0], (_Float16)a[1], (_Float16)a[2], (_Float16)a[3], (_Float16)b[0], (_Float16)b[1], (_Float16)b[2], (_Float16)b[3]};
}

__global__ void k_gram(const float* in, float* out_hh, float* out_full) {
  __shared__ __attribute__((aligned(16))) _Float16 H[64 * kRow], L[64 * kRow];
  const int lane = threadIdx.x;
  for (int c = 0; c < 16; ++c) {
    const float v = in[lane * 16 + c];
    const _Float16 hi = (_Float16)v;
    H[lane * kRow + c] = hi;
    L[lane * kRow + c] = (_Float16)(v - (float)hi);
  }
  __syncthreads();
  f4 hh = {0, 0, 0, 0}, hl = {0, 0, 0, 0};
  for (int chunk = 0; chunk < 2; ++chunk) {
    const h8 h = read_operand(H, lane, chunk), l = read_operand(L, lane, chunk);
    hh = __builtin_amdgcn_mfma_f32_16x16x32_f16(h, h, hh, 0, 0, 0);
    hl = __builtin_amdgcn_mfma_f32_16x16x32_f16(h, l, hl, 0, 0, 0);   // S = H L^T ; G = HH + S + S^T
  }
  const int i = lane & 15, g = lane >> 4;
  for (int r = 0; r < 4; ++r) {
    out_hh[(g * 4 + r) * 16 + i] = hh[r];
    out_full[(g * 4 + r) * 16 + i] = hl[r];
  }
}

// round 4: the layout of align_fast.hip's STORE 2 -- 32-byte rows with 16 bytes of padding behind every eighth, lane group g takes
// pixels 4 g .. 4 g + 3 and 16 + 4 g .. 19 + 4 g of a block of 32
__device__ inline h8 read_operand2(const char* base, int lane, int block) {
  const int i = lane & 15, g = lane >> 4;
  const char* rd = base + g * 128 + (g >> 1) * 16 + (i >> 2) * 32 + (i & 3) * 8 + block * 1088;
  const fp4v a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp4v*)rd);
  const fp4v b = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp4v*)(rd + 544));
  return h8{(_Float16)a[0], (_Float16)a[1], (_Float16)a[2], (_Float16)a[3], (_Float16)b[0], (_Float16)b[1], (_Float16)b[2], (_Float16)b[3]};
}

__global__ void k_gram2(const float* in, float* out_hh, float* out_full) {
  __shared__ __attribute__((aligned(16))) char H[2304], L[2304];
  const int lane = threadIdx.x;
  _Float16* hrow = reinterpret_cast<_Float16*>(H + lane * 32 + (lane >> 3) * 16);
  _Float16* lrow = reinterpret_cast<_Float16*>(L + lane * 32 + (lane >> 3) * 16);
  for (int c = 0; c < 16; ++c) {
    const float v = in[lane * 16 + c];
    const _Float16 hi = (_Float16)v;
    hrow[c] = hi;
    lrow[c] = (_Float16)(v - (float)hi);
  }
  __syncthreads();
  f4 hh = {0, 0, 0, 0}, hl = {0, 0, 0, 0};
  for (int block = 0; block < 2; ++block) {
    const h8 h = read_operand2(H, lane, block), l = read_operand2(L, lane, block);
    hh = __builtin_amdgcn_mfma_f32_16x16x32_f16(h, h, hh, 0, 0, 0);
    hl = __builtin_amdgcn_mfma_f32_16x16x32_f16(h, l, hl, 0, 0, 0);
  }
  const int i = lane & 15, g = lane >> 4;
  for (int r = 0; r < 4; ++r) {
    out_hh[(g * 4 + r) * 16 + i] = hh[r];
    out_full[(g * 4 + r) * 16 + i] = hl[r];
  }
}

int main() {
  std::vector<float> v(64 * 16);
  srand(7);
  for (int p = 0; p < 64; ++p)
    for (int c = 0; c < 16; ++c) {
      const float mag = c < 12 ? (c % 3 == 0 ? 300.0f : 3.0f) : 0.05f * 256.0f;   // Jacobian-like and (scaled) residual-like magnitudes
      v[p * 16 + c] = mag * (float(rand()) / RAND_MAX - 0.47f) * (c == 7 ? 1e-3f : 1.0f);
    }
  float *d_in, *d_hh, *d_s;
  hipMalloc(&d_in, v.size() * 4);
  hipMalloc(&d_hh, 256 * 4);
  hipMalloc(&d_s, 256 * 4);
  hipMemcpy(d_in, v.data(), v.size() * 4, hipMemcpyHostToDevice);
  k_gram<<<1, 64>>>(d_in, d_hh, d_s);
  std::vector<float> hh(256), s(256);
  hipMemcpy(hh.data(), d_hh, 1024, hipMemcpyDeviceToHost);
  hipMemcpy(s.data(), d_s, 1024, hipMemcpyDeviceToHost);
  double worst = 0, worst_rel_entry = 0, gmax = 0;
  for (int a = 0; a < 16; ++a)
    for (int b = 0; b < 16; ++b) {
      double ref = 0;
      for (int p = 0; p < 64; ++p) ref += double(v[p * 16 + a]) * double(v[p * 16 + b]);
      const double got = double(hh[a * 16 + b]) + double(s[a * 16 + b]) + double(s[b * 16 + a]);
      gmax = fmax(gmax, fabs(ref));
      worst = fmax(worst, fabs(got - ref));
      if (fabs(ref) > 0) worst_rel_entry = fmax(worst_rel_entry, fabs(got - ref) / fabs(ref));
    }
  printf("gram_f16: max |G - G64| / max|G| = %.3e ; worst per-entry relative error = %.3e (f32 accumulation of 64 terms: ~1e-7)\n", worst / gmax, worst_rel_entry);
  k_gram2<<<1, 64>>>(d_in, d_hh, d_s);
  hipMemcpy(hh.data(), d_hh, 1024, hipMemcpyDeviceToHost);
  hipMemcpy(s.data(), d_s, 1024, hipMemcpyDeviceToHost);
  double worst2 = 0;
  for (int a = 0; a < 16; ++a)
    for (int b = 0; b < 16; ++b) {
      double ref = 0;
      for (int p = 0; p < 64; ++p) ref += double(v[p * 16 + a]) * double(v[p * 16 + b]);
      worst2 = fmax(worst2, fabs(double(hh[a * 16 + b]) + double(s[a * 16 + b]) + double(s[b * 16 + a]) - ref));
    }
  printf("gram_f16, 32-byte rows (align_fast.hip STORE 2): max |G - G64| / max|G| = %.3e\n", worst2 / gmax);
  return worst / gmax < 1e-6 && worst2 / gmax < 1e-6 ? 0 : 1;
}
