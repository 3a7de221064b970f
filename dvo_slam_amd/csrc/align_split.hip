// align_split.hip -- the fused residual / Jacobian / reduce sweep with the Gram accumulators SPLIT across the
// four wavefronts of a workgroup (schedule variant 3 of k_residual_reduce; same arithmetic, same outputs).
//
// Why: the one-wave-owns-everything schedule needs 85 float accumulators per lane, which pins the kernel at
// 160-190 VGPRs = 2-3 wavefronts per SIMD, and the kernel is bound by its own memory latency (measured: 610 us
// per 128-pair finest-level launch at 3 waves/SIMD, 770 us at 2).  Here every wavefront still warps, samples and
// forms the residual + 2x6 Jacobian rows of ITS pixel rows (dense_tracking_impl.cpp:148-281,
// dense_tracking.cpp:448-476), but publishes the 15 per-pixel terms {r0, r1, w, J0[6], J1[6]} through LDS; then
// each wavefront accumulates only its quarter of the normal-equation sums (least_squares.cpp:58-64) over the
// pixels of all four wavefronts:
//   wave 0   sum w J0_i J0_j (21)              sum w J0_k r0 (6)
//   wave 1   sum w J1_i J1_j (21)              sum w J1_k r1 (6)
//   wave 2   sum w J0_i J1_j, i = 0..2 (18)    sum w J0_k r1 (6)    n, sum w r r^T (4)
//   wave 3   sum w J0_i J1_j, i = 3..5 (18)    sum w J1_k r0 (6)
// -> at most 28 accumulators per lane, a quarter of the DPP reduction work per wavefront, no cross-wave
// summation at all (each sum has exactly one owner), one s_barrier per pixel row (LDS double-buffered).
// The workgroup finally folds the 106 split sums into the canonical 85-entry partial row of device_types.h.
#include "align_common.h"

namespace dvo_hip {

constexpr int kSplitAcc = 30;            // 28 used, padded to a multiple of 5 for the DPP stages
constexpr int kSplitTotal = 4 * kSplitAcc;

// accumulate one published pixel into this wavefront's share
template <int WAVE>
__device__ __forceinline__ void accumulate_share(float* acc, const float4 q0, const float4 q1, const float4 q2, const float4 q3) {
  const float r0 = q0.x, r1 = q0.y, w = q0.z;
  const float J0[6] = {q0.w, q1.x, q1.y, q1.z, q1.w, q2.x};
  const float J1[6] = {q2.y, q2.z, q2.w, q3.x, q3.y, q3.z};
  if constexpr (WAVE == 0) {
    float wJ0[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) wJ0[i] = w * J0[i];
    int o = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = i; j < 6; ++j) acc[o++] += wJ0[i] * J0[j];
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[21 + i] += wJ0[i] * r0;
  } else if constexpr (WAVE == 1) {
    float wJ1[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) wJ1[i] = w * J1[i];
    int o = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = i; j < 6; ++j) acc[o++] += wJ1[i] * J1[j];
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[21 + i] += wJ1[i] * r1;
  } else if constexpr (WAVE == 2) {
    float wJ0[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) wJ0[i] = w * J0[i];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) acc[i * 6 + j] += wJ0[i] * J1[j];
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[18 + i] += wJ0[i] * r1;
    const float wr0 = w * r0;
    acc[24] += (w > 0.0f) ? 1.0f : 0.0f;           // published w is 0 for pixels without a constraint
    acc[25] += wr0 * r0;
    acc[26] += wr0 * r1;
    acc[27] += (w * r1) * r1;
  } else {
    float wJ0[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) wJ0[i] = w * J0[3 + i];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) acc[i * 6 + j] += wJ0[i] * J1[j];
    const float wr0 = w * r0;
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[18 + i] += wr0 * J1[i];
  }
}

template <int RPW, bool FINEST, bool PIPE>
__global__ __launch_bounds__(kBlock) void k_residual_reduce_split(
    const LevelGeom g, const PairPtrs* __restrict__ pairs, const PairState* __restrict__ states, int n_pairs,
    float* __restrict__ partials, float2* __restrict__ scratch, int blocks_per_xcd) {
  // XCD-aware (pair, tile) -> workgroup mapping, see k_residual_reduce
  const int tiles = g.tiles_x * g.tiles_y;
  const int total = tiles * n_pairs;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int item = xcd * blocks_per_xcd + slot;
  if (item >= total) return;
  const int pair = item / tiles, tile = item - pair * tiles;
  const PairState& st = states[pair];
  if (!st.active) return;
  const PairPtrs pp = pairs[pair];
  const GlobalLoad4 refR{(GlobalVec4)pp.refR}, curA{(GlobalVec4)pp.curA};
  const GlobalLoad2 curB{(GlobalVec2)pp.curB};

  float KT[12], Pp[4];
#pragma unroll
  for (int i = 0; i < 12; ++i) KT[i] = st.KT[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) Pp[i] = st.P_prev[i];
  const bool first = st.first != 0;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int u_r = (tile % g.tiles_x) * kTileW + lane;
  // wavefront w owns rows w, w+4, w+8, ... of the tile: the four rows of one round are adjacent, so their taps
  // share cache lines
  const int row0 = (tile / g.tiles_x) * (kWavesPerBlock * RPW) + wave;
  const size_t pix_base = size_t(pair) * size_t(g.w) * g.h;
  const float nanv = __builtin_nanf("");
  const bool col_ok = u_r < g.w;

  __shared__ float4 terms[2][kWavesPerBlock][4][64];   // [buffer][producer wave][quad][lane], 32 KiB
  __shared__ float fin[kSplitTotal];

  float acc[kSplitAcc];
#pragma unroll
  for (int i = 0; i < kSplitAcc; ++i) acc[i] = 0.0f;

  auto load_ref = [&](int k) -> float4 {
    const int v_r = row0 + kWavesPerBlock * k;
    if (k < RPW && col_ok && v_r < g.h) return refR[v_r * g.w + u_r];
    return make_float4(nanv, 0.0f, 0.0f, 0.0f);
  };
  auto row_of = [&](int k) { return min(row0 + kWavesPerBlock * k, g.h - 1); };

  // stage 3 of a row: blend, tests, residual store, weight, Jacobian rows -> LDS
  auto publish_row = [&](int k, const float4 ref, const PixelProj& p, const PixelTaps& t) {
    const int v_r = row0 + kWavesPerBlock * k;
    PixelTerms o;
    const bool in_image = col_ok && v_r < g.h;
    const bool valid = in_image && p.ok && pixel_finish(g, ref, p, t, o);
    if (in_image) scratch[pix_base + size_t(v_r) * g.w + u_r] = valid ? make_float2(o.r0, o.r1) : make_float2(nanv, nanv);
    float4 q0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), q1 = q0, q2 = q0, q3 = q0;
    if (valid) {
      // t-distribution weight with the PREVIOUS pass' precision (Q11); first pass on a level: w = 1
      const float w = first ? 1.0f : tdist_weight(o.r0, o.r1, Pp);
      float J0[6], J1[6];
      jacobian_rows(o, J0, J1);
      q0 = make_float4(o.r0, o.r1, w, J0[0]);
      q1 = make_float4(J0[1], J0[2], J0[3], J0[4]);
      q2 = make_float4(J0[5], J1[0], J1[1], J1[2]);
      q3 = make_float4(J1[3], J1[4], J1[5], 0.0f);
    }
    float4(*buf)[4][64] = terms[k & 1];
    buf[wave][0][lane] = q0;
    buf[wave][1][lane] = q1;
    buf[wave][2][lane] = q2;
    buf[wave][3][lane] = q3;
  };
  auto consume_round = [&](int k) {
    const float4(*buf)[4][64] = terms[k & 1];
#pragma unroll 1
    for (int pw = 0; pw < kWavesPerBlock; ++pw) {
      const float4 q0 = buf[pw][0][lane], q1 = buf[pw][1][lane], q2 = buf[pw][2][lane], q3 = buf[pw][3][lane];
      switch (wave) {   // wave-uniform
        case 0: accumulate_share<0>(acc, q0, q1, q2, q3); break;
        case 1: accumulate_share<1>(acc, q0, q1, q2, q3); break;
        case 2: accumulate_share<2>(acc, q0, q1, q2, q3); break;
        default: accumulate_share<3>(acc, q0, q1, q2, q3); break;
      }
    }
  };

  if constexpr (PIPE) {
    float4 ref_cur = load_ref(0);
    float4 ref_next = load_ref(1);
    PixelProj p_cur = pixel_project(g, KT, ref_cur, col_ok ? u_r : 0, row_of(0));
    PixelTaps t_cur;
    if (p_cur.ok) pixel_fetch(g, curA, curB, p_cur, t_cur);
#pragma unroll 1
    for (int k = 0; k < RPW; ++k) {
      const float4 ref_next2 = load_ref(k + 2);
      const PixelProj p_next = pixel_project(g, KT, ref_next, col_ok ? u_r : 0, row_of(k + 1));
      PixelTaps t_next;
      if (p_next.ok) pixel_fetch(g, curA, curB, p_next, t_next);   // next row's gathers fly during publish + consume
      publish_row(k, ref_cur, p_cur, t_cur);
      __syncthreads();
      consume_round(k);
      ref_cur = ref_next; ref_next = ref_next2;
      p_cur = p_next; t_cur = t_next;
    }
  } else {
#pragma unroll 1
    for (int k = 0; k < RPW; ++k) {
      const float4 ref = load_ref(k);
      const PixelProj p = pixel_project(g, KT, ref, col_ok ? u_r : 0, row_of(k));
      PixelTaps t;
      if (p.ok) pixel_fetch(g, curA, curB, p, t);
      publish_row(k, ref, p, t);
      __syncthreads();
      consume_round(k);
    }
  }

  // every sum has one owning wavefront: DPP-reduce the share, lane 63 hands it to the workgroup
  wave_sum_all_to_lane63<kSplitAcc>(acc);
  if (lane == 63) {
#pragma unroll
    for (int i = 0; i < kSplitAcc; ++i) fin[wave * kSplitAcc + i] = acc[i];
  }
  __syncthreads();
  // fold into the canonical partial row (device_types.h): J01 is stored symmetrised, B01 = J0 r1 + J1 r0
  const int k = threadIdx.x;
  if (k < kNumAcc) {
    const float* s0 = fin;
    const float* s1 = fin + kSplitAcc;
    const float* s2 = fin + 2 * kSplitAcc;
    const float* s3 = fin + 3 * kSplitAcc;
    auto j01 = [&](int i, int j) { return i < 3 ? s2[i * 6 + j] : s3[(i - 3) * 6 + j]; };   // sum w J0_i J1_j
    float v;
    if (k == kAccN) v = s2[24];
    else if (k < kAccJ00) v = s2[25 + (k - kAccS)];
    else if (k < kAccJ11) v = s0[k - kAccJ00];
    else if (k < kAccJ01) v = s1[k - kAccJ11];
    else if (k < kAccB00) {
      int o = k - kAccJ01, i = 0;
      while (o >= 6 - i) { o -= 6 - i; ++i; }   // upper-triangular row-major index -> (i, j)
      const int j = i + o;
      v = j01(i, j) + j01(j, i);
    } else if (k < kAccB01) v = s0[21 + (k - kAccB00)];
    else if (k < kAccB11) v = s2[18 + (k - kAccB01)] + s3[18 + (k - kAccB01)];
    else v = s1[21 + (k - kAccB11)];
    partials[(size_t(pair) * tiles + tile) * kAccStride + k] = v;
  }
}

template <int RPW, bool PIPE>
static void launch_split(hipStream_t s, bool finest, const LevelGeom& g, const PairPtrs* pairs, const PairState* states,
                         int n_pairs, float* partials, float2* scratch) {
  const int total = g.tiles_x * g.tiles_y * n_pairs;
  const int per_xcd = (total + 7) / 8;
  if (finest)
    k_residual_reduce_split<RPW, true, PIPE><<<dim3(per_xcd * 8), dim3(kBlock), 0, s>>>(g, pairs, states, n_pairs, partials, scratch, per_xcd);
  else
    k_residual_reduce_split<RPW, false, PIPE><<<dim3(per_xcd * 8), dim3(kBlock), 0, s>>>(g, pairs, states, n_pairs, partials, scratch, per_xcd);
}

template <bool PIPE>
static void launch_split_p(hipStream_t s, int rows_per_wave, bool finest, const LevelGeom& g, const PairPtrs* pairs,
                           const PairState* states, int n_pairs, float* partials, float2* scratch) {
  switch (rows_per_wave) {
    case 1: launch_split<1, PIPE>(s, finest, g, pairs, states, n_pairs, partials, scratch); break;
    case 2: launch_split<2, PIPE>(s, finest, g, pairs, states, n_pairs, partials, scratch); break;
    case 4: launch_split<4, PIPE>(s, finest, g, pairs, states, n_pairs, partials, scratch); break;
    case 16: launch_split<16, PIPE>(s, finest, g, pairs, states, n_pairs, partials, scratch); break;
    default: launch_split<8, PIPE>(s, finest, g, pairs, states, n_pairs, partials, scratch); break;
  }
}

void launch_residual_reduce_split(hipStream_t s, bool pipelined, int rows_per_wave, bool finest, const LevelGeom& g,
                                  const PairPtrs* pairs, const PairState* states, int n_pairs, float* partials, float2* scratch) {
  if (pipelined) launch_split_p<true>(s, rows_per_wave, finest, g, pairs, states, n_pairs, partials, scratch);
  else launch_split_p<false>(s, rows_per_wave, finest, g, pairs, states, n_pairs, partials, scratch);
}

}  // namespace dvo_hip
