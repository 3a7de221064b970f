// div_check.hip -- the sweep's short division (pixel_math.h::divide2_correctly_rounded: one refined reciprocal, two remainder
// corrections, no range scaling / fix-up) against the compiler's IEEE division, bit for bit, on 2^32 operand pairs of the domain the
// sweep uses it on: divisor = a depth-like value (random mantissa, |d| in [2^-6, 2^9), both signs), numerator random mantissa with
// |n| in [2^-20, 2^22) -- plus the patterns that are hard for a Newton / remainder scheme: divisor mantissas of all ones, quotients
// landing next to a rounding boundary (n = RN(q d) +- 1 ulp for a random q).
//   hipcc --offload-arch=gfx950 -O3 -I../../dvo_slam_amd/csrc -o _build/div_check div_check.hip && _build/div_check
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "pixel_math.h"

__device__ inline uint32_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return uint32_t(x);
}

__global__ void k_check(unsigned long long* bad, unsigned long long* bad_in_range, float* first_bad, int rounds) {
  const uint64_t tid = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  unsigned long long nb = 0, nr = 0;
  for (int it = 0; it < rounds; ++it) {
    const uint64_t id = tid * uint64_t(rounds) + it;
    const uint32_t a = mix(id * 2 + 1), b = mix(id * 2 + 2), c = mix(id * 2 + 0x9e3779b97f4a7c15ull);
    // divisor: exponent 2^-6 .. 2^8, random mantissa (every 16th: all ones in the top k bits), random sign
    uint32_t dm = b & 0x7fffff;
    if ((c & 15) == 0) dm |= ~((1u << (c >> 27)) - 1) & 0x7fffff;
    const uint32_t db = ((121 + (b >> 23) % 15) << 23) | dm | (c & 0x80000000u);
    float d = __builtin_bit_cast(float, db);
    float n;
    if (c & 16) {   // a numerator one ulp around q * d for a random q in [0, 1024): quotients next to rounding boundaries
      const float q = float(a & 0xffffff) * (1.0f / 16384.0f);
      const float p = q * d;
      n = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, p) + ((a >> 24) % 3) - 1);
    } else {
      n = __builtin_bit_cast(float, ((107 + (a >> 23) % 42) << 23) | (a & 0x7fffff) | ((c << 1) & 0x80000000u));
    }
    float u, v;
    dvo_hip::divide2_correctly_rounded(n, -n, d, u, v);
    const float ru = n / d, rv = -n / d;
    const bool same = __builtin_bit_cast(uint32_t, u) == __builtin_bit_cast(uint32_t, ru) && __builtin_bit_cast(uint32_t, v) == __builtin_bit_cast(uint32_t, rv);
    if (!same) {
      ++nb;
      // (a zero of the other sign, or subnormal operands / quotients, is not a difference the sweep can see: u >= 0, floor(u) and
      // u - floor(u) come out the same)
      const bool tiny = fabsf(ru) < 1e-30f && fabsf(u) < 1e-30f && fabsf(rv) < 1e-30f && fabsf(v) < 1e-30f;
      if (fabsf(ru) < 32768.0f && !tiny) {
        if (nr == 0 && atomicAdd(bad_in_range + 1, 1ull) == 0) { first_bad[0] = n; first_bad[1] = d; first_bad[2] = u; first_bad[3] = ru; }
        ++nr;
      }
    }
  }
  if (nb) atomicAdd(bad, nb);
  if (nr) atomicAdd(bad_in_range, nr);
}

int main() {
  unsigned long long *d_bad, h[3] = {0, 0, 0};
  float *d_first, f[4] = {0, 0, 0, 0};
  hipMalloc(&d_bad, 24);
  hipMalloc(&d_first, 16);
  hipMemset(d_bad, 0, 24);
  const int blocks = 256 * 64, threads = 256, rounds = 1024;   // 2^32 pairs
  k_check<<<blocks, threads>>>(d_bad, d_bad + 1, d_first, rounds);
  hipDeviceSynchronize();
  hipMemcpy(h, d_bad, 24, hipMemcpyDeviceToHost);
  hipMemcpy(f, d_first, 16, hipMemcpyDeviceToHost);
  printf("div_check: %llu operand pairs, %llu differ from the IEEE division (signed zeros / subnormals), %llu of them with 1e-30 < |quotient| < 32768", (unsigned long long)blocks * threads * rounds, h[0], h[1]);
  if (h[1]) printf("  (first: %.9g / %.9g -> %.9g vs %.9g)", f[0], f[1], f[2], f[3]);
  printf("\n");
  return h[1] == 0 ? 0 : 1;
}
