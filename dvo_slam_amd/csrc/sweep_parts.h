// sweep_parts.h -- pieces shared by the two kernels that sweep a level on the matrix cores: k_residual_reduce_mfma (align_mfma.hip, one
// launch per Gauss-Newton step) and k_match_resident (align_resident.hip, one launch per match).
#pragma once

#include "align_common.h"

namespace dvo_hip {

typedef float __attribute__((ext_vector_type(4))) f32x4;
typedef float __attribute__((ext_vector_type(2))) f32x2;

// The per-wavefront LDS slab in which a row's 64 per-pixel 16-vectors  sqrt(w) * [J0(6), J1(6), r0, r1, 0, 0]  are transposed into
// matrix operands: component quad q of pixel p at  q * kQuadStride + p * 4 ; the operand of pixel group g is one conflict-free
// ds_read_b32 at a constant offset.
constexpr int kQuadStride = 264;                 // floats per component quad: 64 pixels x 4 + 8 skew (bank-conflict-free reads)
constexpr int kSlabFloats = 4 * kQuadStride;     // per-wavefront LDS slab (4224 B)

// The eight bilinear taps through buffer loads: the plane is a raw buffer resource held in scalar registers, a tap's address is
// ONE 32-bit vector offset (tap (u0, v0)) plus a scalar row offset plus an immediate -- instead of a 64-bit vector address per
// tap -- and the two depth-gradient taps of a row, adjacent in memory, travel as one 16-byte load.  Reads past the plane's end
// return zero.
struct TapPlanes {
  __amdgpu_buffer_rsrc_t A, B;
  int rowA, rowB;                                   // bytes per image row
  __device__ __forceinline__ void fetch(int base, PixelTaps& t) const {
    const int oa = base * 16, ob = base * 8;
    const f32x4 a00 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(A, oa, 0, 0));
    const f32x4 a10 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(A, oa + 16, 0, 0));
    const f32x4 a01 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(A, oa, rowA, 0));
    const f32x4 a11 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(A, oa + 16, rowA, 0));
    const f32x2 b00 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(B, ob, 0, 0));
    const f32x2 b10 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(B, ob + 8, 0, 0));
    const f32x2 b01 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(B, ob, rowB, 0));
    const f32x2 b11 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(B, ob + 8, rowB, 0));
    t.A00 = make_float4(a00.x, a00.y, a00.z, a00.w); t.A10 = make_float4(a10.x, a10.y, a10.z, a10.w);
    t.A01 = make_float4(a01.x, a01.y, a01.z, a01.w); t.A11 = make_float4(a11.x, a11.y, a11.z, a11.w);
    t.B00 = make_float2(b00.x, b00.y); t.B10 = make_float2(b10.x, b10.y);
    t.B01 = make_float2(b01.x, b01.y); t.B11 = make_float2(b11.x, b11.y);
  }
};

// Which entries of the 16x16 Gram matrix of the per-pixel vectors (components 0..5 = J0, 6..11 = J1, 12 = r0, 13 = r1) make up
// accumulator k (kAccN < k < kNumAcc) of the canonical partial row (device_types.h): entry e1 = row * 16 + column, plus e2 for the
// symmetrised blocks (-1: none).
__device__ __forceinline__ void gram_entries_of_accumulator(int k, int& e1, int& e2) {
  auto E = [](int r, int c) { return r * 16 + c; };
  e2 = -1;
  if (k == kAccS) e1 = E(12, 12);
  else if (k == kAccS + 1) e1 = E(12, 13);
  else if (k == kAccS + 2) e1 = E(13, 13);
  else if (k < kAccB00) {
    const int blockId = (k - kAccJ00) / 21;                  // 0: J0J0, 1: J1J1, 2: J0J1 symmetrised
    int o = (k - kAccJ00) % 21, i = 0;
    while (o >= 6 - i) { o -= 6 - i; ++i; }                  // upper-triangular row-major index -> (i, j)
    const int j = i + o;
    if (blockId == 0) e1 = E(i, j);
    else if (blockId == 1) e1 = E(6 + i, 6 + j);
    else { e1 = E(i, 6 + j); e2 = E(j, 6 + i); }
  } else if (k < kAccB01) e1 = E(k - kAccB00, 12);
  else if (k < kAccB11) { e1 = E(k - kAccB01, 13); e2 = E(6 + (k - kAccB01), 12); }
  else e1 = E(6 + (k - kAccB11), 13);
}

// The same map as a table built at compile time (entry k: e1 | e2 << 8, e2 = 0xff for none): one load per thread at the top of a
// kernel instead of the index arithmetic in its epilogue.
struct GramEntryTable {
  unsigned short e[kNumAcc];
};
constexpr GramEntryTable make_gram_entry_table() {
  GramEntryTable t{};
  auto E = [](int r, int c) { return r * 16 + c; };
  for (int k = 0; k < kNumAcc; ++k) {
    int e1 = 0, e2 = 0xff;
    if (k == kAccN) e1 = 0;
    else if (k == kAccS) e1 = E(12, 12);
    else if (k == kAccS + 1) e1 = E(12, 13);
    else if (k == kAccS + 2) e1 = E(13, 13);
    else if (k < kAccB00) {
      const int blockId = (k - kAccJ00) / 21;
      int o = (k - kAccJ00) % 21, i = 0;
      while (o >= 6 - i) { o -= 6 - i; ++i; }
      const int j = i + o;
      if (blockId == 0) e1 = E(i, j);
      else if (blockId == 1) e1 = E(6 + i, 6 + j);
      else { e1 = E(i, 6 + j); e2 = E(j, 6 + i); }
    } else if (k < kAccB01) e1 = E(k - kAccB00, 12);
    else if (k < kAccB11) { e1 = E(k - kAccB01, 13); e2 = E(6 + (k - kAccB01), 12); }
    else e1 = E(6 + (k - kAccB11), 13);
    t.e[k] = (unsigned short)(e1 | e2 << 8);
  }
  return t;
}
__device__ __constant__ const GramEntryTable kGramEntryTable = make_gram_entry_table();

}  // namespace dvo_hip
