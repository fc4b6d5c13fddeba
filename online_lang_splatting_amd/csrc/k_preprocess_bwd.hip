// k_preprocess_bwd.hip — per-Gaussian backward: instance-row reduction + the analytic chain
// from (mean2D, conic, opacity, colour, depth) back to (mean3D, cov3D, scale, quaternion, SH, pose).
//
// Replaces the atomicAdd accumulation of CR/backward.cu:1176-1198 together with
// computeCov2DCUDA (:150-346), preprocessCUDA / language_preprocessCUDA (:418-539, 541-682),
// computeCov3D (:350-413) and computeColorFromSH (:21-145).  Two kernels:
//   row_reduce_big_kernel — sums the partial-gradient rows of the Gaussians listed in more than
//     OLSR_MID_FOOTPRINT tiles.  Rows are compacted in emission (= depth) order, so a Gaussian's rows are
//     ONE dense run [rowbase[u0], rowbase[u0+n]); a near splat owns thousands.  The forward's emission
//     left two work lists behind; a persistent grid gives every listed Gaussian a whole wave — 64 lanes
//     stride over the rows, then one multi-value butterfly.  Fixed summation order: bit-reproducible.
//   preprocess_bwd_kernel — the analytic chain, one lane per Gaussian in INDEX order so that
//     every per-Gaussian input and output is a coalesced access.  Gaussians with few tiles (most) sum
//     their own rows inline, in ascending order; dL_dmean2D / dL_dconic / dL_dcolor / dL_ddepth stay
//     in registers and the four reference stages are fused.  With a gradient bucket the row of the
//     flat all-reduce buffer is written (or added) by the same kernel, staged through LDS.
// Every output row is written exactly once (zeros for culled Gaussians), so no memset of the
// gradient tensors is needed (the reference zero-fills 12 tensors per call,
// DGR/rasterize_points.cu:386-398).
//
// dL_dtau is additionally reduced on the device, deterministically: fixed-order block
// partials, then a single-wave pass (replaces the torch.sum of
// DGR/diff_gaussian_rasterization/__init__.py:383).
#include <algorithm>

#include "olsr_device.h"
#include "olsr_kernels.h"

namespace olsr {

#ifndef OLSR_PB_THREADS
#define OLSR_PB_THREADS 128
#endif
#ifndef OLSR_PB_SPLIT
#define OLSR_PB_SPLIT 0  // 1: row sums for every Gaussian, the chain only for those with rows (three kernels: built and measured
                         // late in round 4 - same headline, the isolated stage 63 -> 84 us; profiles/r4_experiments.json); 0: one kernel
#endif
constexpr int PB_THREADS = OLSR_PB_THREADS;
int tau_partial_blocks(int P) { return (P + PB_THREADS - 1) / PB_THREADS; }

constexpr int next_pow2_(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}
// static-index multi-value wave butterfly (see k_render_bwd.hip)
template <int H, int M, int N>
__device__ __forceinline__ void wave_reduce_rec(float (&v)[N], int lane) {
  if constexpr (H >= 1) {
    const bool upper = (lane & M) != 0;
#pragma unroll
    for (int i = 0; i < H; ++i) {
      const float send = upper ? v[i] : v[i + H];
      const float keep = upper ? v[i + H] : v[i];
      v[i] = keep + __shfl_xor(send, M);
    }
    wave_reduce_rec<H / 2, M / 2, N>(v, lane);
  } else if constexpr (M >= 1) {
    v[0] += __shfl_xor(v[0], M);
    wave_reduce_rec<0, M / 2, N>(v, lane);
  }
}

constexpr int RR_THREADS = 256;

#ifndef OLSR_RR_BLOCKS
#define OLSR_RR_BLOCKS 8192  // (65 k listed Gaussians at config 3: 1024 / 2048 / 4096 / 8192 blocks: 30 / 29 / 24 / 23 us)
#endif
constexpr int RR_BIG_BLOCKS = OLSR_RR_BLOCKS;  // persistent grid of the wave-per-Gaussian kernel (8 waves/SIMD)

template <int F, int N>
__device__ __forceinline__ void add_row(const float* __restrict__ rows, u32 row, float (&acc)[N]) {
  static_assert(N >= 10 + F, "accumulator array too small");
  constexpr int ROW = grad_row(F);
  constexpr int NVAL = 10 + F;
  const float4* p = reinterpret_cast<const float4*>(rows + (size_t)row * ROW);
#pragma unroll
  for (int v4 = 0; v4 < (NVAL + 3) / 4; ++v4) {
    const float4 x = p[v4];
    if (4 * v4 + 0 < NVAL) acc[4 * v4 + 0] += x.x;
    if (4 * v4 + 1 < NVAL) acc[4 * v4 + 1] += x.y;
    if (4 * v4 + 2 < NVAL) acc[4 * v4 + 2] += x.z;
    if (4 * v4 + 3 < NVAL) acc[4 * v4 + 3] += x.w;
  }
}

// Row sums of the Gaussians listed in more than OLSR_MID_FOOTPRINT tiles (the forward's emission built the two
// lists: {Gaussian, first instance, #instances}).  One wave per Gaussian: its rows are the dense run
// [rowbase[u0], rowbase[u0 + n]) — ascending (tile, wave) order; 64 lanes stride over them, then one multi-value
// butterfly.  Persistent grid; the next item's descriptor is fetched while the current one is reduced.  Every
// other Gaussian is summed inline by preprocess_bwd_kernel (at most 4 * OLSR_MID_FOOTPRINT rows, typically a few).
template <int F>
__global__ __launch_bounds__(RR_THREADS) void row_reduce_big_kernel(const float* __restrict__ rows,
                                                                    float* __restrict__ gacc,
                                                                    const uint4* __restrict__ big_list, int P,
                                                                    const u32* __restrict__ rowbase,
                                                                    const int32_t* __restrict__ counters, int rows_stamp) {
  constexpr int ROW = grad_row(F);
  constexpr int NVAL = 10 + F;
  constexpr int NP = next_pow2_(NVAL);
  constexpr int G_LANES = 64 / NP;
  const int lane = threadIdx.x & 63;
  const int wave = (int)(blockIdx.x * (RR_THREADS / 64) + (threadIdx.x >> 6));
  const int nwaves = (int)(gridDim.x * (RR_THREADS / 64));
  const int nbig = counters[5], count = nbig + counters[4];  // front list, then the medium list from the back
  if (wave >= count || frame_unusable(counters, rows_stamp)) return;
  auto item_at = [&](int i) { return big_list[i < nbig ? i : P - 1 - (i - nbig)]; };
  // Most listed Gaussians lie behind the saturation depth of every tile they cover and have NO row (config 3: 65 k listed,
  // a few thousand with rows): such an item ends after its look-up — no butterfly, no gacc row (preprocess_bwd_kernel looks
  // the count up itself and discards what it reads there).  The look-ups are pipelined two items deep by hand: an iteration
  // issues the descriptor load of the item after next and the two rowbase loads of the next one, whose descriptor arrived an
  // iteration ago, and works on an item whose everything has arrived — one dependent trip per item.  (Measured in round 4,
  // profiles/r4_experiments.json: the same early exit WITHOUT the explicit pipeline won 4 us at config 3 and lost 60 us at
  // config 5, where a wave walks twelve items and the compiler had sunk the prefetch behind the exit; 64 look-ups per wave
  // at once with the items that have rows then served in series lost 100 us: those cluster at the front of the list.)
  uint4 cur = item_at(wave);
  const int i1_ = wave + nwaves;
  uint4 nx = item_at(i1_ < count ? i1_ : wave);
  u32 cur_first = rowbase[cur.y], cur_end = rowbase[cur.y + cur.z];
  for (int item = wave; item < count; item += nwaves) {
    const int i2_ = item + 2 * nwaves;
    const uint4 nn = item_at(i2_ < count ? i2_ : item);                          // descriptor of the item after next
    const u32 nx_first = rowbase[nx.y], nx_end = rowbase[nx.y + nx.z];          // the next item's run of rows
    const u32 idx = cur.x, first = cur_first, nrows = cur_end - cur_first;
    if (nrows != 0u) {  // (wave-uniform)
      float acc[NP];
#pragma unroll
      for (int v = 0; v < NP; ++v) acc[v] = 0.f;
      for (u32 t = (u32)lane; t < nrows; t += 64) add_row<F, NP>(rows, first + t, acc);
      wave_reduce_rec<NP / 2, 32, NP>(acc, lane);
      float v = __shfl(acc[0], (lane * G_LANES) & 63);
      if (lane >= NVAL) v = 0.f;
      if (lane < ROW) gacc[(size_t)idx * ROW + lane] = v;
    }
    cur = nx;
    cur_first = nx_first;
    cur_end = nx_end;
    nx = nn;
  }
}

// -skew(v) column i (CR/math.h:27-31 negated)
__device__ __forceinline__ f3 nskew_col(const f3& v, int i) {
  if (i == 0) return {-0.f, -v.z, v.y};
  if (i == 1) return {v.z, -0.f, -v.x};
  return {-v.y, v.x, -0.f};
}
__device__ __forceinline__ float dot3(const f3& a, const f3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 cross3(const f3& a, const f3& b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// dnormvdv, CR/auxiliary.h:107-117
__device__ __forceinline__ f3 dnormvdv(f3 v, f3 dv) {
  const float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
  const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
  f3 o;
  o.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
  o.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
  o.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
  return o;
}

// computeColorFromSH backward, CR/backward.cu:21-145.  Writes dL_dsh rows (all M coefficients,
// zeros above the active degree), returns dL_dmean contribution.
// `dL_dsh` is the Gaussian's own row of 3M floats; with `add` the values are added to it (fused
// accumulation into the gradient bucket).
__device__ __forceinline__ f3 sh_backward(int idx, int deg, int M, const f3& pos, const float* campos,
                                          const float* __restrict__ shs, const uint8_t* __restrict__ clamped,
                                          const float* dL_dcolor3, float* __restrict__ dL_dsh, bool add) {
  const f3 dir_orig = {pos.x - campos[0], pos.y - campos[1], pos.z - campos[2]};
  const float len = sqrtf(dir_orig.x * dir_orig.x + dir_orig.y * dir_orig.y + dir_orig.z * dir_orig.z);
  const f3 dir = {dir_orig.x / len, dir_orig.y / len, dir_orig.z / len};
  const float* sh = shs + (size_t)idx * M * 3;
  float dL_dRGB[3];
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) dL_dRGB[ch] = dL_dcolor3[ch] * (clamped[3 * (size_t)idx + ch] ? 0 : 1);
  float dRGBdx[3] = {0, 0, 0}, dRGBdy[3] = {0, 0, 0}, dRGBdz[3] = {0, 0, 0};
  const float x = dir.x, y = dir.y, z = dir.z;
  auto S = [&](int k, int ch) { return sh[3 * k + ch]; };
  auto setsh = [&](int k, float w) {
    if (dL_dsh == nullptr) return;  // pose-only backward: only the dL_dmean contribution below is wanted
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float v = w * dL_dRGB[ch];
      dL_dsh[3 * k + ch] = add ? dL_dsh[3 * k + ch] + v : v;
    }
  };
  setsh(0, SH_C0);
  int written = 1;
  if (deg > 0) {
    setsh(1, -SH_C1 * y);
    setsh(2, SH_C1 * z);
    setsh(3, -SH_C1 * x);
    written = 4;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      dRGBdx[ch] = -SH_C1 * S(3, ch);
      dRGBdy[ch] = -SH_C1 * S(1, ch);
      dRGBdz[ch] = SH_C1 * S(2, ch);
    }
    if (deg > 1) {
      const float xx = x * x, yy = y * y, zz = z * z;
      const float xy = x * y, yz = y * z, xz = x * z;
      setsh(4, SH_C2[0] * xy);
      setsh(5, SH_C2[1] * yz);
      setsh(6, SH_C2[2] * (2.f * zz - xx - yy));
      setsh(7, SH_C2[3] * xz);
      setsh(8, SH_C2[4] * (xx - yy));
      written = 9;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        dRGBdx[ch] += SH_C2[0] * y * S(4, ch) + SH_C2[2] * 2.f * -x * S(6, ch) + SH_C2[3] * z * S(7, ch) +
                      SH_C2[4] * 2.f * x * S(8, ch);
        dRGBdy[ch] += SH_C2[0] * x * S(4, ch) + SH_C2[1] * z * S(5, ch) + SH_C2[2] * 2.f * -y * S(6, ch) +
                      SH_C2[4] * 2.f * -y * S(8, ch);
        dRGBdz[ch] += SH_C2[1] * y * S(5, ch) + SH_C2[2] * 2.f * 2.f * z * S(6, ch) + SH_C2[3] * x * S(7, ch);
      }
      if (deg > 2) {
        setsh(9, SH_C3[0] * y * (3.f * xx - yy));
        setsh(10, SH_C3[1] * xy * z);
        setsh(11, SH_C3[2] * y * (4.f * zz - xx - yy));
        setsh(12, SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy));
        setsh(13, SH_C3[4] * x * (4.f * zz - xx - yy));
        setsh(14, SH_C3[5] * z * (xx - yy));
        setsh(15, SH_C3[6] * x * (xx - 3.f * yy));
        written = 16;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
          dRGBdx[ch] += (SH_C3[0] * S(9, ch) * 3.f * 2.f * xy + SH_C3[1] * S(10, ch) * yz +
                         SH_C3[2] * S(11, ch) * -2.f * xy + SH_C3[3] * S(12, ch) * -3.f * 2.f * xz +
                         SH_C3[4] * S(13, ch) * (-3.f * xx + 4.f * zz - yy) + SH_C3[5] * S(14, ch) * 2.f * xz +
                         SH_C3[6] * S(15, ch) * 3.f * (xx - yy));
          dRGBdy[ch] += (SH_C3[0] * S(9, ch) * 3.f * (xx - yy) + SH_C3[1] * S(10, ch) * xz +
                         SH_C3[2] * S(11, ch) * (-3.f * yy + 4.f * zz - xx) + SH_C3[3] * S(12, ch) * -3.f * 2.f * yz +
                         SH_C3[4] * S(13, ch) * -2.f * xy + SH_C3[5] * S(14, ch) * -2.f * yz +
                         SH_C3[6] * S(15, ch) * -3.f * 2.f * xy);
          dRGBdz[ch] += (SH_C3[1] * S(10, ch) * xy + SH_C3[2] * S(11, ch) * 4.f * 2.f * yz +
                         SH_C3[3] * S(12, ch) * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * S(13, ch) * 4.f * 2.f * xz +
                         SH_C3[5] * S(14, ch) * (xx - yy));
        }
      }
    }
  }
  if (!add && dL_dsh != nullptr)
    for (int k = written; k < M; ++k) {
      dL_dsh[3 * k + 0] = 0.f;
      dL_dsh[3 * k + 1] = 0.f;
      dL_dsh[3 * k + 2] = 0.f;
    }
  const f3 dL_ddir = {dRGBdx[0] * dL_dRGB[0] + dRGBdx[1] * dL_dRGB[1] + dRGBdx[2] * dL_dRGB[2],
                      dRGBdy[0] * dL_dRGB[0] + dRGBdy[1] * dL_dRGB[1] + dRGBdy[2] * dL_dRGB[2],
                      dRGBdz[0] * dL_dRGB[0] + dRGBdz[1] * dL_dRGB[1] + dRGBdz[2] * dL_dRGB[2]};
  return dnormvdv(dir_orig, dL_ddir);
}

// Sigma = Rq diag(s)^2 Rq^T, backward: what computeCov3D's backward computes (CR/backward.cu:350-413), in ITS operation
// order (rounds 1-4 had the matrix form dL/dN = 2 N D, dL/ds_i = sum_k dL/dN(i, k) Rq(k, i), ...: same algebra, other rounding).
// The reference differentiates the unnormalised-quaternion rotation formula and no normalisation (its dnormvdv call is
// commented out, :412).
__device__ __forceinline__ void cov3d_backward(const float* scale, float mod, const float* rot, const float* d,
                                               float* ds, float* dq) {
  // operation for operation what the reference's glm expressions evaluate (CR/backward.cu:350-413; m3 is glm's
  // column-major layout and mul() its accumulation order, olsr_device.h): M = S R, dL/dM = (2 M) dL/dSigma, the scale
  // gradient from the rows of R^T and dL/dM^T, then dL/dM^T scaled by s and contracted with the quaternion derivative
  const float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
  const m3 R = {{{1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
                 {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
                 {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}}};
  m3 S = {{{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}}};
  const f3 sv = {mod * scale[0], mod * scale[1], mod * scale[2]};
  S.c[0][0] = sv.x;
  S.c[1][1] = sv.y;
  S.c[2][2] = sv.z;
  const m3 Mm = mul(S, R);
  const m3 dL_dSigma = {{{d[0], 0.5f * d[1], 0.5f * d[2]}, {0.5f * d[1], d[3], 0.5f * d[4]}, {0.5f * d[2], 0.5f * d[4], d[5]}}};
  m3 M2;
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) M2.c[c][rr] = Mm.c[c][rr] * 2.0f;
  const m3 dL_dM = mul(M2, dL_dSigma);
  const m3 Rt = transpose(R);
  m3 dL_dMt = transpose(dL_dM);
  ds[0] = Rt.c[0][0] * dL_dMt.c[0][0] + Rt.c[0][1] * dL_dMt.c[0][1] + Rt.c[0][2] * dL_dMt.c[0][2];
  ds[1] = Rt.c[1][0] * dL_dMt.c[1][0] + Rt.c[1][1] * dL_dMt.c[1][1] + Rt.c[1][2] * dL_dMt.c[1][2];
  ds[2] = Rt.c[2][0] * dL_dMt.c[2][0] + Rt.c[2][1] * dL_dMt.c[2][1] + Rt.c[2][2] * dL_dMt.c[2][2];
  // (gradient with respect to s_i = mod * scale_i: like the reference, :393-396, the factor `mod` of the chain to
  //  the raw scale is NOT applied — visible only to callers that render with scale_modifier != 1, i.e. the GUI)
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    dL_dMt.c[0][k] *= sv.x;
    dL_dMt.c[1][k] *= sv.y;
    dL_dMt.c[2][k] *= sv.z;
  }
#define OLSR_MT(c_, r_) dL_dMt.c[c_][r_]
  dq[0] = 2 * z * (OLSR_MT(0, 1) - OLSR_MT(1, 0)) + 2 * y * (OLSR_MT(2, 0) - OLSR_MT(0, 2)) + 2 * x * (OLSR_MT(1, 2) - OLSR_MT(2, 1));
  dq[1] = 2 * y * (OLSR_MT(1, 0) + OLSR_MT(0, 1)) + 2 * z * (OLSR_MT(2, 0) + OLSR_MT(0, 2)) + 2 * r * (OLSR_MT(1, 2) - OLSR_MT(2, 1)) -
          4 * x * (OLSR_MT(2, 2) + OLSR_MT(1, 1));
  dq[2] = 2 * x * (OLSR_MT(1, 0) + OLSR_MT(0, 1)) + 2 * r * (OLSR_MT(2, 0) - OLSR_MT(0, 2)) + 2 * z * (OLSR_MT(1, 2) + OLSR_MT(2, 1)) -
          4 * y * (OLSR_MT(2, 2) + OLSR_MT(0, 0));
  dq[3] = 2 * r * (OLSR_MT(0, 1) - OLSR_MT(1, 0)) + 2 * x * (OLSR_MT(2, 0) + OLSR_MT(0, 2)) + 2 * y * (OLSR_MT(1, 2) + OLSR_MT(2, 1)) -
          4 * z * (OLSR_MT(1, 1) + OLSR_MT(0, 0));
#undef OLSR_MT
}

// The analytic chain of ONE Gaussian from its composite-level gradient sums acc[0..9] = {mean2D.x, mean2D.y, conic.x, conic.y,
// conic.w, (opacity), colour r g b, depth} to dL_dmean3D, dL_dcov3D, dL_dscale, dL_drotation, the SH rows and the pose
// (computeCov2DCUDA, preprocessCUDA / language_preprocessCUDA, computeCov3D, computeColorFromSH backward: CR/backward.cu:21-145,
// 150-346, 350-413, 418-682).  dmean / dcov / dscale / drot must come in zeroed; tau is accumulated into; sh_row is the
// Gaussian's row of 3 M SH gradients (NULL: not wanted), written or — sh_add — added to.  Used by the one-kernel form and by
// the compacted chain kernel below.
template <int N>
__device__ __forceinline__ void pb_chain(u32 idx, const float (&acc)[N], int D, int M, const float* __restrict__ means3D,
                                         const float* __restrict__ shs, const uint8_t* __restrict__ clamped,
                                         const float* __restrict__ scales, const float* __restrict__ rotations,
                                         float scale_modifier, const float* __restrict__ cov3Ds,
                                         const float* __restrict__ view, const float* __restrict__ proj,
                                         const float* __restrict__ proj_raw, const float* __restrict__ campos, float h_x,
                                         float h_y, float tan_fovx, float tan_fovy, int act, float* sh_row, bool sh_add,
                                         float (&dmean)[3], float (&dcov)[6], float (&dscale)[3], float (&drot)[4],
                                         float (&tau)[6], bool& sh_written) {
  static_assert(N >= 10, "acc holds the ten composite-level sums");
  // Since late round 4 every expression below is written in the association of the reference's source (which the oracle
  // restates line by line): this translation unit is built without contraction, so on identical inputs the chain's outputs
  // are the oracle's bits.  Rounds 1-4 derived the chain in matrix form (M = J R, S = -adj G adj / det^2, dL/dV = M^T S M,
  // dL/dM = 2 S M V, theta as cross products): the same algebra, 30 % fewer operations — and wherever two large terms cancel
  // (dL/dM for a strongly anisotropic covariance, the entries of -adj G adj) a different association is a different rounding,
  // which a 1e-4 comparison notices (scripts/probe/one_stress_scene.py 2585 90000 base: six elements 3e-4 off on identical
  // inputs).  The kernel is bound by its loads and stores, not by this arithmetic.
  const f3 mean = {means3D[3 * (size_t)idx], means3D[3 * (size_t)idx + 1], means3D[3 * (size_t)idx + 2]};
  // ---- computeCov2DCUDA, backward (CR/backward.cu:150-346) --------------------------------------------------------------
  {
    float cov3D[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) cov3D[i] = cov3Ds[6 * (size_t)idx + i];
    Cov2D ci;
    cov2d_common(mean, h_x, h_y, tan_fovx, tan_fovy, cov3D, view, ci);
    const f3 t = ci.t;
    const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
    const float x_grad_mul = (ci.txtz < -limx || ci.txtz > limx) ? 0.f : 1.f;  // a clamped coordinate passes no gradient
    const float y_grad_mul = (ci.tytz < -limy || ci.tytz > limy) ? 0.f : 1.f;
    const m3& J = ci.J;
    const m3& Wm = ci.Wm;
    const m3& T = ci.T;
    const m3& Vrk = ci.Vrk;
    const float a = ci.cov.c[0][0] + 0.3f, b = ci.cov.c[0][1], c = ci.cov.c[1][1] + 0.3f;
    const float denom = a * c - b * b;
    float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
    const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    const float gx = acc[2], gy = acc[3], gw = acc[4];  // dL_dconic .x .y .w
    if (denom2inv != 0.f) {  // (det^2 overflowed otherwise: the reference leaves these gradients zero)
      dL_da = denom2inv * (-c * c * gx + 2 * b * c * gy + (denom - a * c) * gw);
      dL_dc = denom2inv * (-a * a * gw + 2 * a * b * gy + (denom - a * c) * gx);
      dL_db = denom2inv * 2 * (b * c * gx - (denom + 2 * b * b) * gy + a * b * gw);
      dcov[0] = (T.c[0][0] * T.c[0][0] * dL_da + T.c[0][0] * T.c[1][0] * dL_db + T.c[1][0] * T.c[1][0] * dL_dc);
      dcov[3] = (T.c[0][1] * T.c[0][1] * dL_da + T.c[0][1] * T.c[1][1] * dL_db + T.c[1][1] * T.c[1][1] * dL_dc);
      dcov[5] = (T.c[0][2] * T.c[0][2] * dL_da + T.c[0][2] * T.c[1][2] * dL_db + T.c[1][2] * T.c[1][2] * dL_dc);
      dcov[1] = 2 * T.c[0][0] * T.c[0][1] * dL_da + (T.c[0][0] * T.c[1][1] + T.c[0][1] * T.c[1][0]) * dL_db +
                2 * T.c[1][0] * T.c[1][1] * dL_dc;
      dcov[2] = 2 * T.c[0][0] * T.c[0][2] * dL_da + (T.c[0][0] * T.c[1][2] + T.c[0][2] * T.c[1][0]) * dL_db +
                2 * T.c[1][0] * T.c[1][2] * dL_dc;
      dcov[4] = 2 * T.c[0][2] * T.c[0][1] * dL_da + (T.c[0][1] * T.c[1][2] + T.c[0][2] * T.c[1][1]) * dL_db +
                2 * T.c[1][1] * T.c[1][2] * dL_dc;
    }
    // gradients with respect to the upper 2x3 part of T (:246-257)
    const float dL_dT00 = 2 * (T.c[0][0] * Vrk.c[0][0] + T.c[0][1] * Vrk.c[0][1] + T.c[0][2] * Vrk.c[0][2]) * dL_da +
                          (T.c[1][0] * Vrk.c[0][0] + T.c[1][1] * Vrk.c[0][1] + T.c[1][2] * Vrk.c[0][2]) * dL_db;
    const float dL_dT01 = 2 * (T.c[0][0] * Vrk.c[1][0] + T.c[0][1] * Vrk.c[1][1] + T.c[0][2] * Vrk.c[1][2]) * dL_da +
                          (T.c[1][0] * Vrk.c[1][0] + T.c[1][1] * Vrk.c[1][1] + T.c[1][2] * Vrk.c[1][2]) * dL_db;
    const float dL_dT02 = 2 * (T.c[0][0] * Vrk.c[2][0] + T.c[0][1] * Vrk.c[2][1] + T.c[0][2] * Vrk.c[2][2]) * dL_da +
                          (T.c[1][0] * Vrk.c[2][0] + T.c[1][1] * Vrk.c[2][1] + T.c[1][2] * Vrk.c[2][2]) * dL_db;
    const float dL_dT10 = 2 * (T.c[1][0] * Vrk.c[0][0] + T.c[1][1] * Vrk.c[0][1] + T.c[1][2] * Vrk.c[0][2]) * dL_dc +
                          (T.c[0][0] * Vrk.c[0][0] + T.c[0][1] * Vrk.c[0][1] + T.c[0][2] * Vrk.c[0][2]) * dL_db;
    const float dL_dT11 = 2 * (T.c[1][0] * Vrk.c[1][0] + T.c[1][1] * Vrk.c[1][1] + T.c[1][2] * Vrk.c[1][2]) * dL_dc +
                          (T.c[0][0] * Vrk.c[1][0] + T.c[0][1] * Vrk.c[1][1] + T.c[0][2] * Vrk.c[1][2]) * dL_db;
    const float dL_dT12 = 2 * (T.c[1][0] * Vrk.c[2][0] + T.c[1][1] * Vrk.c[2][1] + T.c[1][2] * Vrk.c[2][2]) * dL_dc +
                          (T.c[0][0] * Vrk.c[2][0] + T.c[0][1] * Vrk.c[2][1] + T.c[0][2] * Vrk.c[2][2]) * dL_db;
    const float dL_dJ00 = Wm.c[0][0] * dL_dT00 + Wm.c[0][1] * dL_dT01 + Wm.c[0][2] * dL_dT02;
    const float dL_dJ02 = Wm.c[2][0] * dL_dT00 + Wm.c[2][1] * dL_dT01 + Wm.c[2][2] * dL_dT02;
    const float dL_dJ11 = Wm.c[1][0] * dL_dT10 + Wm.c[1][1] * dL_dT11 + Wm.c[1][2] * dL_dT12;
    const float dL_dJ12 = Wm.c[2][0] * dL_dT10 + Wm.c[2][1] * dL_dT11 + Wm.c[2][2] * dL_dT12;
    const float tz = 1.f / t.z;
    const float tz2 = tz * tz;
    const float tz3 = tz2 * tz;
    const float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
    const float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
    const float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t.x) * tz3 * dL_dJ02 +
                         (2 * h_y * t.y) * tz3 * dL_dJ12;
    // pose, part 1 (:273-288): dp_C/drho = I, dp_C/dtheta = -skew(t)
    {
      const f3 rho_cols[3] = {{1.f, 0.f, 0.f}, {0.f, 1.f, 0.f}, {0.f, 0.f, 1.f}};
      const f3 th_cols[3] = {nskew_col(t, 0), nskew_col(t, 1), nskew_col(t, 2)};
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        tau[i] += dL_dtx * rho_cols[i].x + dL_dty * rho_cols[i].y + dL_dtz * rho_cols[i].z;
        tau[i + 3] += dL_dtx * th_cols[i].x + dL_dty * th_cols[i].y + dL_dtz * th_cols[i].z;
      }
    }
    const f3 dm = transformVec4x3Transpose({dL_dtx, dL_dty, dL_dtz}, view);  // (assigned: the reference's first writer, :292-297)
    dmean[0] = dm.x;
    dmean[1] = dm.y;
    dmean[2] = dm.z;
    // pose, part 2 (:299-345): dL/dW = J^T dL/dT, contracted with the columns of -skew of the columns of R
    const float dL_dW00 = J.c[0][0] * dL_dT00;
    const float dL_dW01 = J.c[0][0] * dL_dT01;
    const float dL_dW02 = J.c[0][0] * dL_dT02;
    const float dL_dW10 = J.c[1][1] * dL_dT10;
    const float dL_dW11 = J.c[1][1] * dL_dT11;
    const float dL_dW12 = J.c[1][1] * dL_dT12;
    const float dL_dW20 = J.c[0][2] * dL_dT00 + J.c[1][2] * dL_dT10;
    const float dL_dW21 = J.c[0][2] * dL_dT01 + J.c[1][2] * dL_dT11;
    const float dL_dW22 = J.c[0][2] * dL_dT02 + J.c[1][2] * dL_dT12;
    const f3 c1 = {view[0], view[1], view[2]}, c2 = {view[4], view[5], view[6]}, c3 = {view[8], view[9], view[10]};
    const f3 dW1 = {dL_dW00, dL_dW10, dL_dW20}, dW2 = {dL_dW01, dL_dW11, dL_dW21}, dW3 = {dL_dW02, dL_dW12, dL_dW22};
#pragma unroll
    for (int i = 0; i < 3; ++i)
      tau[3 + i] += dot3(dW1, nskew_col(c1, i)) + dot3(dW2, nskew_col(c2, i)) + dot3(dW3, nskew_col(c3, i));
  }
  // ---- preprocessCUDA / language_preprocessCUDA, backward (CR/backward.cu:418-682): the projected mean and the depth ----
  {
    const f3 m = mean;
    const f4 m_hom = transformPoint4x4(m, proj);
    const float m_w = 1.0f / (m_hom.w + 0.0000001f);
    const float g2x = acc[0], g2y = acc[1];
    const float mul1 = (proj[0] * m.x + proj[4] * m.y + proj[8] * m.z + proj[12]) * m_w * m_w;
    const float mul2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * m_w * m_w;
    dmean[0] += (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
    dmean[1] += (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
    dmean[2] += (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;
    const float alpha = 1.0f * m_w;
    const float beta = -m_hom.x * m_w * m_w;
    const float gamma = -m_hom.y * m_w * m_w;
    const float pa = proj_raw[0], pb = proj_raw[5], pe = proj_raw[11];
    // p_C = R m + t through the reference's SE3 (CR/math.h:322-324): the rotation first, then the translation
    const f3 c0 = {view[0], view[1], view[2]}, c1 = {view[4], view[5], view[6]}, c2 = {view[8], view[9], view[10]};
    const f3 tt = {view[12], view[13], view[14]};
    const f3 Rm = {c0.x * m.x + c1.x * m.y + c2.x * m.z, c0.y * m.x + c1.y * m.y + c2.y * m.z,
                   c0.z * m.x + c1.z * m.y + c2.z * m.z};
    const f3 p_C = {Rm.x + tt.x, Rm.y + tt.y, Rm.z + tt.z};
    const f3 th_cols[3] = {nskew_col(p_C, 0), nskew_col(p_C, 1), nskew_col(p_C, 2)};
    const f3 I_cols[3] = {{1.f, 0.f, 0.f}, {0.f, 1.f, 0.f}, {0.f, 0.f, 1.f}};
    const f3 d1 = {alpha * pa, 0.f, beta * pe};
    const f3 d2 = {0.f, alpha * pb, gamma * pe};
    float dmx[6], dmy[6];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      dmx[i] = dot3(I_cols[i], d1);
      dmy[i] = dot3(I_cols[i], d2);
      dmx[i + 3] = dot3(th_cols[i], d1);
      dmy[i + 3] = dot3(th_cols[i], d2);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) tau[i] += g2x * dmx[i] + g2y * dmy[i];
    const float dL_dpCz = acc[9];
    dmean[0] += dL_dpCz * view[2];
    dmean[1] += dL_dpCz * view[6];
    dmean[2] += dL_dpCz * view[10];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      tau[i] += dL_dpCz * I_cols[i].z;
      tau[i + 3] += dL_dpCz * th_cols[i].z;
    }
      if (shs) {
        const float dcol[3] = {acc[6], acc[7], acc[8]};
        // one evaluation: into the caller's dL_dsh row if there is one (copied to the bucket below),
        // else straight into the bucket row
        const f3 dm_sh = sh_backward((int)idx, D, M, mean, campos, shs, clamped, dcol, sh_row, sh_add);
        sh_written = true;
        dmean[0] += dm_sh.x;
        dmean[1] += dm_sh.y;
        dmean[2] += dm_sh.z;
        tau[0] += -dm_sh.x;
        tau[1] += -dm_sh.y;
        tau[2] += -dm_sh.z;
      }
      if (scales) {
        // the activated scale / rotation are recomputed from the raw parameters (OLSR_ACT_*), and the
        // gradients chained back through exp / normalize
        float sc3[3], q4[4];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float v = scales[3 * (size_t)idx + k];
          sc3[k] = (act & OLSR_ACT_SCALE_EXP) ? expf(v) : v;
        }
        if (act & OLSR_ACT_ROTATION_NORMALIZE) {
          act_normalize4(rotations + 4 * (size_t)idx, q4);
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) q4[k] = rotations[4 * (size_t)idx + k];
        }
        cov3d_backward(sc3, scale_modifier, q4, dcov, dscale, drot);
        if (act & OLSR_ACT_SCALE_EXP) {
#pragma unroll
          for (int k = 0; k < 3; ++k) dscale[k] *= sc3[k];  // d exp(x) / dx = exp(x)
        }
        if (act & OLSR_ACT_ROTATION_NORMALIZE) {
          float graw[4];
          act_normalize4_backward(rotations + 4 * (size_t)idx, drot, graw);
#pragma unroll
          for (int k = 0; k < 4; ++k) drot[k] = graw[k];
        }
      }
    }
  
}

// SPLIT (OLSR_PB_SPLIT=1, an experiment that is OFF: see the macro): the chain does not run here.  Saturation leaves ~2 % of a view's Gaussians with a
// partial-gradient row (config 3: 9 471 of 500 000) and every other Gaussian's chain is a product with zeros, yet in index
// order nearly every wave holds one of the 2 % and issued the chain's ~800 instructions for all 64 lanes.  With SPLIT this
// kernel sums the rows, writes the composite-level gradients, the statistics, the bucket's opacity / language columns and zeros
// everywhere else, parks the ten sums of a Gaussian WITH rows in its gacc row and lists it (per block, ascending);
// pb_compact_kernel concatenates the blocks' lists and pb_chain_kernel runs the chain for the listed Gaussians only, a lane
// each, overwriting (or adding to) what this kernel left.  Same operations on the same values: the results are the one-kernel
// form's, bit for bit, up to the sign of zeros; dL_dtau's device-side sum is taken over the listed Gaussians in ascending order.
template <int F, bool SPLIT>
__global__ __launch_bounds__(PB_THREADS) void preprocess_bwd_kernel(
    int P, int D, int M, const float* __restrict__ gacc, const u32* __restrict__ tiles_touched,
    const u32* __restrict__ inst_start, const u32* __restrict__ rowbase, const float* __restrict__ rows,
    const int32_t* __restrict__ counters, const float* __restrict__ means3D, const int32_t* __restrict__ radii, const float* __restrict__ shs,
    const uint8_t* __restrict__ clamped, const float* __restrict__ scales, const float* __restrict__ rotations,
    float scale_modifier, const float* __restrict__ cov3Ds, const float* __restrict__ view,
    const float* __restrict__ proj, const float* __restrict__ proj_raw, const float* __restrict__ campos, float h_x,
    float h_y, float tan_fovx, float tan_fovy, float* __restrict__ dL_dmeans2D, float* __restrict__ dL_dconic,
    float* __restrict__ dL_dopacity, float* __restrict__ dL_dcolors, float* __restrict__ dL_dlanguage,
    float* __restrict__ dL_ddepths, float* __restrict__ dL_dmeans3D, float* __restrict__ dL_dcov3D,
    float* __restrict__ dL_dsh, float* __restrict__ dL_dscales, float* __restrict__ dL_drotations,
    float* __restrict__ dL_dtau, float* __restrict__ tau_partials, float* __restrict__ bucket_flat,
    float* __restrict__ bucket_densify, int32_t* __restrict__ bucket_max_radii, int bucket_assign, int act,
    const float* __restrict__ opacities_raw, int F_out, u64* __restrict__ bucket_row_mask, float* gacc_park,
    u32* __restrict__ act_list, u32* __restrict__ act_count, const uint8_t* __restrict__ blended, int rows_stamp) {
  // F: language channels of the partial-gradient rows; F_out: the scene's (width of dL_dlanguage and of the bucket's
  // language columns).  F == 0 < F_out: the backward ran without a language cotangent, those gradients are zero.
  constexpr int ROW = grad_row(F);
  constexpr int NVAL = 10 + F;
  extern __shared__ float s_bucket[];  // [PB_THREADS][width] when a gradient bucket is given
  const int width = 11 + 3 * M + F_out;  // floats per Gaussian in the gradient bucket
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  float tau[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  bool has_rows = false;  // this Gaussian has at least one partial-gradient row (else every gradient of it is an exact zero)
  if (r < P) {
    const u32 idx = (u32)r;
    const bool vis = radii[idx] > 0;
    float acc[NVAL];
#pragma unroll
    for (int v = 0; v < NVAL; ++v) acc[v] = 0.f;
    // (a Gaussian listed in no tile, or one no pixel blended — 98 % of the visible ones behind the saturation depth —, has no
    //  rows: all zeros, and the dependent look-ups below — emission index, two ends of its run of rows — are not made for it)
    const u32 ntiles_g = (vis && blended[idx] != 0) ? tiles_touched[idx] : 0u;
    if (ntiles_g > OLSR_MID_FOOTPRINT) {
      if (!frame_unusable(counters, rows_stamp)) {  // summed by row_reduce_big_kernel
        // (looked up since the chain below is skipped without rows: a large footprint behind the saturation depth has none)
        const u32 u0 = inst_start[idx];
        has_rows = rowbase[u0 + ntiles_g] != rowbase[u0];
        const float4* row = reinterpret_cast<const float4*>(gacc + (size_t)idx * ROW);
        // row_reduce_big_kernel writes no row for a Gaussian without any: the row is read regardless — its loads go out
        // beside the look-up's instead of behind it (a fourth dependent trip cost config 5 60 us) — and discarded then
#pragma unroll
        for (int v4 = 0; v4 < (NVAL + 3) / 4; ++v4) {
          const float4 x = row[v4];
          if (4 * v4 + 0 < NVAL) acc[4 * v4 + 0] = has_rows ? x.x : 0.f;
          if (4 * v4 + 1 < NVAL) acc[4 * v4 + 1] = has_rows ? x.y : 0.f;
          if (4 * v4 + 2 < NVAL) acc[4 * v4 + 2] = has_rows ? x.z : 0.f;
          if (4 * v4 + 3 < NVAL) acc[4 * v4 + 3] = has_rows ? x.w : 0.f;
        }
      }
    } else if (ntiles_g > 0 && !frame_unusable(counters, rows_stamp)) {
      // the Gaussian's partial-gradient rows are one dense run (emission order): sum them here, in ascending
      // (tile, wave) order — no intermediate per-Gaussian buffer
      const u32 u0 = inst_start[idx];
      const u32 first = rowbase[u0], nrows = rowbase[u0 + ntiles_g] - first;
      for (u32 t = 0; t < nrows; ++t) add_row<F, NVAL>(rows, first + t, acc);
      has_rows = nrows > 0;
    }
    // bucket row of this Gaussian, staged in LDS: a lane's 116-byte row would be 29 scattered 4-byte stores,
    // the block's rows together are one contiguous span that is written (or added) coalesced at the end
    if (act & OLSR_ACT_OPACITY_SIGMOID) {  // d sigmoid(x) / dx = o (1 - o)
      const float o = act_sigmoid(opacities_raw[idx]);
      acc[5] *= o * (1.0f - o);
    }
    float* brow = bucket_flat ? s_bucket + (size_t)threadIdx.x * width : nullptr;
    constexpr bool badd = false;  // LDS rows are assigned; assign / add is applied by the block's copy-out
    // what the composite's atomics produced in the reference
    if (dL_dmeans2D) {
      dL_dmeans2D[3 * (size_t)idx + 0] = acc[0];
      dL_dmeans2D[3 * (size_t)idx + 1] = acc[1];
      dL_dmeans2D[3 * (size_t)idx + 2] = 0.f;
    }
    if (brow) {
      // densification statistics (gaussian_model.py:965-969): per-view norm of the screen-space gradient
      const int rad = radii[idx];
      const float nrm = vis ? sqrtf(acc[0] * acc[0] + acc[1] * acc[1]) : 0.f;
      const float cnt = vis ? 1.f : 0.f;
      float2* dz = reinterpret_cast<float2*>(bucket_densify) + idx;
      if (bucket_assign) {
        *dz = make_float2(nrm, cnt);
        bucket_max_radii[idx] = rad;
      } else {
        const float2 o = *dz;
        *dz = make_float2(o.x + nrm, o.y + cnt);
        bucket_max_radii[idx] = max(bucket_max_radii[idx], rad);
      }
    }
    if (dL_dconic) {
      dL_dconic[4 * (size_t)idx + 0] = acc[2];
      dL_dconic[4 * (size_t)idx + 1] = acc[3];
      dL_dconic[4 * (size_t)idx + 2] = 0.f;
      dL_dconic[4 * (size_t)idx + 3] = acc[4];
    }
    if (dL_dopacity) dL_dopacity[idx] = acc[5];
    if (dL_dcolors) {
      dL_dcolors[3 * (size_t)idx + 0] = acc[6];
      dL_dcolors[3 * (size_t)idx + 1] = acc[7];
      dL_dcolors[3 * (size_t)idx + 2] = acc[8];
    }
    if (dL_ddepths) dL_ddepths[idx] = acc[9];
    if constexpr (F > 0) {
      if (dL_dlanguage) {
#pragma unroll
        for (int ch = 0; ch < F; ++ch) dL_dlanguage[(size_t)idx * F + ch] = acc[10 + ch];
      }
    } else if (dL_dlanguage) {
      for (int ch = 0; ch < F_out; ++ch) dL_dlanguage[(size_t)idx * F_out + ch] = 0.f;
    }
    if (brow) {
      float* o = brow + 3 + 3 * M;  // [opacity | scale 3 | rotation 4 | language F_out]
      o[0] = badd ? o[0] + acc[5] : acc[5];
      if constexpr (F > 0) {
#pragma unroll
        for (int ch = 0; ch < F; ++ch) o[8 + ch] = badd ? o[8 + ch] + acc[10 + ch] : acc[10 + ch];
      } else if (!badd) {
        for (int ch = 0; ch < F_out; ++ch) o[8 + ch] = 0.f;
      }
    }

    float dmean[3] = {0.f, 0.f, 0.f};
    float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float dscale[3] = {0.f, 0.f, 0.f};
    float drot[4] = {0.f, 0.f, 0.f, 0.f};
    bool sh_written = false;
    // A Gaussian without a partial-gradient row has acc == 0 and every gradient below is a product with it: an exact zero (for
    // finite parameters), which is what the defaults above and the zero fill of dL_dsh below leave.  Saturation leaves 98 % of
    // a view's visible Gaussians without a row (config 3), so a wave whose 64 Gaussians are all such skips the chain's ~800
    // instructions (three waves in ten do; the branch is uniform for them, the other lanes ride along as before).
    if constexpr (SPLIT) {
      if (vis && has_rows) {  // park the sums for pb_chain_kernel (a large footprint's row already holds them: same values)
        float4* park = reinterpret_cast<float4*>(gacc_park + (size_t)idx * ROW);
        park[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        park[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
        park[2] = make_float4(acc[8], acc[9], 0.f, 0.f);
      }
    } else if (vis && has_rows) {
      float* sh_row = dL_dsh ? dL_dsh + (size_t)idx * M * 3 : (brow ? brow + 3 : nullptr);
      pb_chain(idx, acc, D, M, means3D, shs, clamped, scales, rotations, scale_modifier, cov3Ds, view, proj, proj_raw, campos,
               h_x, h_y, tan_fovx, tan_fovy, act, sh_row, !dL_dsh && badd, dmean, dcov, dscale, drot, tau, sh_written);
    }
    if (M > 0 && !sh_written) {
      if (dL_dsh) {
        float* o = dL_dsh + (size_t)idx * M * 3;
        for (int k = 0; k < 3 * M; ++k) o[k] = 0.f;
      }
      if (brow && !badd)
        for (int k = 0; k < 3 * M; ++k) brow[3 + k] = 0.f;
    } else if (M > 0 && brow && dL_dsh) {
      const float* o = dL_dsh + (size_t)idx * M * 3;  // just written by this thread
      for (int k = 0; k < 3 * M; ++k) brow[3 + k] = badd ? brow[3 + k] + o[k] : o[k];
    }
    if (dL_dmeans3D) {
#pragma unroll
      for (int i = 0; i < 3; ++i) dL_dmeans3D[3 * (size_t)idx + i] = dmean[i];
    }
    if (dL_dcov3D) {
#pragma unroll
      for (int i = 0; i < 6; ++i) dL_dcov3D[6 * (size_t)idx + i] = dcov[i];
    }
    if (dL_dscales) {
#pragma unroll
      for (int i = 0; i < 3; ++i) dL_dscales[3 * (size_t)idx + i] = dscale[i];
    }
    if (dL_drotations) {
#pragma unroll
      for (int i = 0; i < 4; ++i) dL_drotations[4 * (size_t)idx + i] = drot[i];
    }
    if (dL_dtau) {
#pragma unroll
      for (int i = 0; i < 6; ++i) dL_dtau[6 * (size_t)idx + i] = tau[i];
    }
    if (brow) {
#pragma unroll
      for (int i = 0; i < 3; ++i) brow[i] = badd ? brow[i] + dmean[i] : dmean[i];
      float* o = brow + 4 + 3 * M;
#pragma unroll
      for (int i = 0; i < 3; ++i) o[i] = badd ? o[i] + dscale[i] : dscale[i];
#pragma unroll
      for (int i = 0; i < 4; ++i) o[3 + i] = badd ? o[3 + i] + drot[i] : drot[i];
    }
  }

  if (bucket_flat) {
    __syncthreads();
    const int g0 = blockIdx.x * PB_THREADS;
    const int count = min(PB_THREADS, P - g0) * width;
    float* out = bucket_flat + (size_t)g0 * width;
    // row mask (olsr_grad_bucket.row_mask): bit = the row may be non-zero.  One 64-bit word per wave of this block.
    u64* mask_word = bucket_row_mask ? bucket_row_mask + ((size_t)g0 >> 6) + (threadIdx.x >> 6) : nullptr;
    static_assert(PB_THREADS % 64 == 0, "a wave of the block owns one word of the row mask");
    const bool wave_in_range = g0 + (int)(threadIdx.x & ~63u) < P;  // (wave-uniform: the word exists)
    if (bucket_assign && mask_word != nullptr) {
      // Overwrite with a mask: rows that have a gradient now are written, rows that may hold an earlier one are zeroed,
      // the others are zero already and stay untouched (config 3: 98 % of them — 150 MB of stores per view).  A lane per row
      // (uncoalesced, but few); blocks where more than a quarter of the rows need a store keep the coalesced pass.
      __shared__ u32 s_nstore_blk;
      if (threadIdx.x == 0) s_nstore_blk = 0;
      __syncthreads();
      const u64 now = ballot(has_rows);
      const u64 before = wave_in_range ? *mask_word : 0ull;
      const u64 store = now | before;
      if (lane_id() == 0 && store != 0ull) atomicAdd(&s_nstore_blk, (u32)__popcll(store));
      __syncthreads();
      if (4u * s_nstore_blk <= (u32)PB_THREADS) {
        if (((store >> lane_id()) & 1ull) != 0ull && r < P) {
          float* o = out + (size_t)threadIdx.x * width;
          const float* b = s_bucket + (size_t)threadIdx.x * width;  // (all zeros for a Gaussian without rows)
          for (int k = 0; k < width; ++k) o[k] = b[k];
        }
      } else {
        for (int e = threadIdx.x; e < count; e += PB_THREADS) out[e] = s_bucket[e];
      }
      if (lane_id() == 0 && wave_in_range) *mask_word = now;
    } else if (bucket_assign) {
      for (int e = threadIdx.x; e < count; e += PB_THREADS) out[e] = s_bucket[e];
    } else {
      if (mask_word != nullptr && wave_in_range) {
        const u64 now = ballot(has_rows);
        if (lane_id() == 0 && now != 0ull) *mask_word |= now;
      }
      // Adding a later view of the step.  Saturation leaves most Gaussians of a view without a single gradient row
      // (config 3: 98 % of the visible ones), and adding their zero rows would read and write the whole bucket for
      // nothing: where at most a quarter of the block's Gaussians have rows, each of those adds its own row (a lane per
      // row: uncoalesced, but few); denser blocks keep the coalesced pass over all rows.
      __shared__ u32 s_nrows_blk;
      if (threadIdx.x == 0) s_nrows_blk = 0;
      __syncthreads();
      const u64 hm = ballot(has_rows);
      if (lane_id() == 0 && hm != 0ull) atomicAdd(&s_nrows_blk, (u32)__popcll(hm));
      __syncthreads();
      if (4u * s_nrows_blk <= (u32)PB_THREADS) {
        if (has_rows) {
          float* o = out + (size_t)threadIdx.x * width;
          const float* b = s_bucket + (size_t)threadIdx.x * width;
          for (int k = 0; k < width; ++k) o[k] += b[k];
        }
      } else {
        for (int e = threadIdx.x; e < count; e += PB_THREADS) out[e] += s_bucket[e];
      }
    }
  }

  if constexpr (SPLIT) {
    // this block's Gaussians with rows, ascending: act_list[block * PB_THREADS + 0 .. count)
    __shared__ u32 s_listed[PB_THREADS / 64];
    const u64 lm = ballot(has_rows);
    const int lane = lane_id(), w = threadIdx.x >> 6;
    if (lane == 0) s_listed[w] = (u32)__popcll(lm);
    __syncthreads();
    u32 before = 0, total = 0;
#pragma unroll
    for (int k = 0; k < PB_THREADS / 64; ++k) {
      before += (k < w) ? s_listed[k] : 0u;
      total += s_listed[k];
    }
    if (has_rows) act_list[(size_t)blockIdx.x * PB_THREADS + before + (u32)__popcll(lm & ((1ull << lane) - 1ull))] = (u32)r;
    if (threadIdx.x == 0) act_count[blockIdx.x] = total;
    return;  // (dL_dtau's partial sums come from pb_chain_kernel)
  }
  // deterministic block partial of tau (fixed butterfly order, then waves in order)
  if (tau_partials) {
    __shared__ float wsum[PB_THREADS / 64][6];
    const int lane = lane_id(), w = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      float v = tau[i];
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
      if (lane == 0) wsum[w][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
      float v = 0.f;
      for (int k = 0; k < PB_THREADS / 64; ++k) v += wsum[k][threadIdx.x];
      tau_partials[(size_t)blockIdx.x * 6 + threadIdx.x] = v;
    }
  }
}

// The blocks' lists concatenated, in block order (= ascending Gaussian index): compact[0 .. *total).  One block: every thread
// sums the counts of its contiguous share of the blocks, one block-wide scan, then copies its blocks' entries.
constexpr int PC_THREADS = 1024;
__global__ __launch_bounds__(PC_THREADS) void pb_compact_kernel(int nb, const u32* __restrict__ act_count,
                                                                const u32* __restrict__ act_list, u32* __restrict__ compact,
                                                                int32_t* __restrict__ total_out) {
  __shared__ u32 s_w[PC_THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int per = (nb + PC_THREADS - 1) / PC_THREADS;
  const int b0 = min(tid * per, nb), b1 = min(b0 + per, nb);
  u32 mine = 0;
  for (int b = b0; b < b1; ++b) mine += act_count[b];
  u32 incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const u32 v = __shfl_up(incl, o);
    if (lane >= o) incl += v;
  }
  if (lane == 63) s_w[w] = incl;
  __syncthreads();
  u32 off = incl - mine, total = 0;
#pragma unroll
  for (int k = 0; k < PC_THREADS / 64; ++k) {
    off += (k < w) ? s_w[k] : 0u;
    total += s_w[k];
  }
  for (int b = b0; b < b1; ++b) {
    const u32 c = act_count[b];
    for (u32 i = 0; i < c; ++i) compact[off + i] = act_list[(size_t)b * PB_THREADS + i];
    off += c;
  }
  if (tid == 0) *total_out = (int32_t)total;
}

// The chain for the listed Gaussians, a lane each (pb_chain), over what the SPLIT form of preprocess_bwd_kernel left: per-Gaussian
// outputs are overwritten, the bucket's chain columns written (assign) or added to.  Persistent grid; block partials of dL_dtau.
constexpr int CH_THREADS = 128;
__global__ __launch_bounds__(CH_THREADS) void pb_chain_kernel(
    const int32_t* __restrict__ total_p, const u32* __restrict__ compact, const float* __restrict__ gacc, int ROW, int D, int M,
    const float* __restrict__ means3D, const float* __restrict__ shs, const uint8_t* __restrict__ clamped,
    const float* __restrict__ scales, const float* __restrict__ rotations, float scale_modifier,
    const float* __restrict__ cov3Ds, const float* __restrict__ view, const float* __restrict__ proj,
    const float* __restrict__ proj_raw, const float* __restrict__ campos, float h_x, float h_y, float tan_fovx,
    float tan_fovy, int act, float* __restrict__ dL_dmeans3D, float* __restrict__ dL_dcov3D, float* __restrict__ dL_dsh,
    float* __restrict__ dL_dscales, float* __restrict__ dL_drotations, float* __restrict__ dL_dtau,
    float* __restrict__ tau_partials, float* __restrict__ bucket_flat, int bucket_assign, int width) {
  const int total = *total_p;
  float tau[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int base = blockIdx.x * CH_THREADS; base < total; base += gridDim.x * CH_THREADS) {
    const int pos = base + threadIdx.x;
    if (pos >= total) continue;
    const u32 idx = compact[pos];
    float acc[12];
    {
      const float4* row = reinterpret_cast<const float4*>(gacc + (size_t)idx * ROW);
      const float4 x0 = row[0], x1 = row[1], x2 = row[2];
      acc[0] = x0.x; acc[1] = x0.y; acc[2] = x0.z; acc[3] = x0.w;
      acc[4] = x1.x; acc[5] = x1.y; acc[6] = x1.z; acc[7] = x1.w;
      acc[8] = x2.x; acc[9] = x2.y; acc[10] = 0.f; acc[11] = 0.f;
    }
    float dmean[3] = {0.f, 0.f, 0.f};
    float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float dscale[3] = {0.f, 0.f, 0.f};
    float drot[4] = {0.f, 0.f, 0.f, 0.f};
    float gtau[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bool sh_written = false;
    float* brow = bucket_flat ? bucket_flat + (size_t)idx * width : nullptr;
    const bool badd = bucket_flat != nullptr && bucket_assign == 0;
    float* sh_row = dL_dsh ? dL_dsh + (size_t)idx * M * 3 : (brow ? brow + 3 : nullptr);
    pb_chain(idx, acc, D, M, means3D, shs, clamped, scales, rotations, scale_modifier, cov3Ds, view, proj, proj_raw, campos, h_x,
             h_y, tan_fovx, tan_fovy, act, sh_row, !dL_dsh && badd, dmean, dcov, dscale, drot, gtau, sh_written);
    if (M > 0 && sh_written && brow && dL_dsh) {
      const float* o = dL_dsh + (size_t)idx * M * 3;  // just written by this thread
      for (int k = 0; k < 3 * M; ++k) brow[3 + k] = badd ? brow[3 + k] + o[k] : o[k];
    }
    if (dL_dmeans3D) {
#pragma unroll
      for (int i = 0; i < 3; ++i) dL_dmeans3D[3 * (size_t)idx + i] = dmean[i];
    }
    if (dL_dcov3D) {
#pragma unroll
      for (int i = 0; i < 6; ++i) dL_dcov3D[6 * (size_t)idx + i] = dcov[i];
    }
    if (dL_dscales) {
#pragma unroll
      for (int i = 0; i < 3; ++i) dL_dscales[3 * (size_t)idx + i] = dscale[i];
    }
    if (dL_drotations) {
#pragma unroll
      for (int i = 0; i < 4; ++i) dL_drotations[4 * (size_t)idx + i] = drot[i];
    }
    if (dL_dtau) {
#pragma unroll
      for (int i = 0; i < 6; ++i) dL_dtau[6 * (size_t)idx + i] = gtau[i];
    }
    if (brow) {
#pragma unroll
      for (int i = 0; i < 3; ++i) brow[i] = badd ? brow[i] + dmean[i] : dmean[i];
      float* o = brow + 4 + 3 * M;
#pragma unroll
      for (int i = 0; i < 3; ++i) o[i] = badd ? o[i] + dscale[i] : dscale[i];
#pragma unroll
      for (int i = 0; i < 4; ++i) o[3 + i] = badd ? o[3 + i] + drot[i] : drot[i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) tau[i] += gtau[i];
  }
  if (tau_partials) {  // deterministic block partial (fixed butterfly order, then waves in order); every block writes one
    __shared__ float wsum[CH_THREADS / 64][6];
    const int lane = lane_id(), w = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      float v = tau[i];
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
      if (lane == 0) wsum[w][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
      float v = 0.f;
      for (int k = 0; k < CH_THREADS / 64; ++k) v += wsum[k][threadIdx.x];
      tau_partials[(size_t)blockIdx.x * 6 + threadIdx.x] = v;
    }
  }
}

// one block: TAU_T threads stride over the block partials (six consecutive floats each), fixed-order reduction.  Four waves
// since round 5 (sixteen before): with several frames in flight a 1024-thread block needs sixteen free wave slots on ONE CU,
// which another frame's composite kernel leaves only in its tail — a kernel trace showed this 5 us kernel, the last of a
// frame's backward, waiting 88 us for a CU.
constexpr int TAU_T = 256;
__global__ __launch_bounds__(TAU_T) void tau_final_kernel(const float* __restrict__ partials, int nb,
                                                         float* __restrict__ out, const int32_t* __restrict__ counters,
                                                         int32_t* __restrict__ status_dev, int32_t* sticky,
                                                         int status_rows) {
  __shared__ float red[TAU_T / 64][6];
  // rows compacted by the forward's last launch (olsr_scene.backward_row_capacity): what launch_row_compaction would have
  // told the caller — {live rows, row / instance overflow}; a cut-off miss (counters[9], folded into counters[7]) is
  // reported as 3 below
  // (status_rows = the stamp the rows must carry, olsr_device.h: rows_stamp_of; a forward that did not compact them for this
  //  scratch leaves the backward without rows: zero gradients, reported as an overflow of zero rows)
  if (threadIdx.x == 0 && status_rows != 0 && status_dev != nullptr) {
    const bool stale = counters[11] != status_rows;
    status_dev[0] = stale ? 0 : counters[6];
    status_dev[1] = stale ? 1 : ((counters[7] != 0 && counters[9] == 0) ? 1 : 0);
  }
  // the backward's last kernel: a synchronisation error of this frame (olsr_state.h, counters[8]) reaches the caller here
  if (threadIdx.x == 0 && counters[8] != 0) {
    if (status_dev != nullptr) status_dev[1] = 2;
    if (sticky != nullptr) __hip_atomic_store(sticky, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  } else if (threadIdx.x == 0 && counters[9] != 0 && status_dev != nullptr) {
    status_dev[1] = 3;  // the forward reported a depth cut-off miss: every gradient of this call is zero (olsr_device.h)
  }
  if (out == nullptr) return;
  float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int b = threadIdx.x; b < nb; b += TAU_T) {
    const float2* p = reinterpret_cast<const float2*>(partials + (size_t)b * 6);  // 24-byte records: 8-byte aligned
    const float2 a0 = p[0], a1 = p[1], a2 = p[2];
    acc[0] += a0.x; acc[1] += a0.y; acc[2] += a1.x; acc[3] += a1.y; acc[4] += a2.x; acc[5] += a2.y;
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    float v = acc[c];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][c] = v;
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    float v = 0.f;
    for (int w = 0; w < TAU_T / 64; ++w) v += red[w][threadIdx.x];
    out[threadIdx.x] = v;
  }
}

template <int F>
static void launch_pb_t(const olsr_scene& s, const FrameDims& d, const GeometryState& g, const BinningState& b,
                        const float* rows, const int32_t* radii, const GradOut& o, float* tau_partials,
                        hipStream_t st) {
  const int nb = tau_partial_blocks(s.P);
  const float* cov3D_ptr = s.cov3D_precomp ? s.cov3D_precomp : g.cov3D;
  // (persistent grid: at most one wave per 4 Gaussians — the lists never hold more than a fraction of them)
  const int rr_blocks = std::min(RR_BIG_BLOCKS, std::max(256, s.P / 16));
  row_reduce_big_kernel<F><<<rr_blocks, RR_THREADS, 0, st>>>(rows, g.gacc, g.big_list, s.P, b.rowbase, g.counters,
                                                             rows_stamp_of(s.backward_row_capacity));
  const int F_out = s.F;
  const size_t bucket_lds = o.bucket_flat ? sizeof(float) * PB_THREADS * (size_t)(11 + 3 * s.M + F_out) : 0;
#define OLSR_PB_ARGS                                                                                                      \
  s.P, s.D, s.M, g.gacc, g.tiles_touched, g.inst_start, b.rowbase, rows, g.counters, s.means3D, radii, s.shs, g.clamped,  \
      s.scales, s.rotations, s.scale_modifier, cov3D_ptr, s.viewmatrix, s.projmatrix, s.projmatrix_raw, s.cam_pos,        \
      d.focal_x, d.focal_y, s.tan_fovx, s.tan_fovy, o.dL_dmeans2D, o.dL_dconic, o.dL_dopacity, o.dL_dcolors,              \
      o.dL_dlanguage, o.dL_ddepths, o.dL_dmeans3D, o.dL_dcov3D, o.dL_dsh, o.dL_dscales, o.dL_drotations, o.dL_dtau
  int n_partials = nb;
#if OLSR_PB_SPLIT
  {
    // scratch: the depth sort's key / value buffers are dead once the forward's emission has run (olsr_state.h)
    u32 *act_list = g.key_a, *act_count = g.key_b, *compact = g.val_b;
    int32_t* total = &g.counters[10];
    preprocess_bwd_kernel<F, true><<<nb, PB_THREADS, bucket_lds, st>>>(
        OLSR_PB_ARGS, nullptr, o.bucket_flat, o.bucket_densify, o.bucket_max_radii, o.bucket_assign, s.activations,
        s.opacities, F_out, o.bucket_row_mask, g.gacc, act_list, act_count, g.blended, rows_stamp_of(s.backward_row_capacity));
    pb_compact_kernel<<<1, PC_THREADS, 0, st>>>(nb, act_count, act_list, compact, total);
    n_partials = std::min(nb, 256);
    pb_chain_kernel<<<n_partials, CH_THREADS, 0, st>>>(
        total, compact, g.gacc, grad_row(F), s.D, s.M, s.means3D, s.shs, g.clamped, s.scales, s.rotations, s.scale_modifier,
        cov3D_ptr, s.viewmatrix, s.projmatrix, s.projmatrix_raw, s.cam_pos, d.focal_x, d.focal_y, s.tan_fovx, s.tan_fovy,
        s.activations, o.dL_dmeans3D, o.dL_dcov3D, o.dL_dsh, o.dL_dscales, o.dL_drotations, o.dL_dtau,
        o.dL_dtau_sum ? tau_partials : nullptr, o.bucket_flat, o.bucket_assign, 11 + 3 * s.M + F_out);
  }
#else
  preprocess_bwd_kernel<F, false><<<nb, PB_THREADS, bucket_lds, st>>>(
      OLSR_PB_ARGS, o.dL_dtau_sum ? tau_partials : nullptr, o.bucket_flat, o.bucket_densify, o.bucket_max_radii,
      o.bucket_assign, s.activations, s.opacities, F_out, o.bucket_row_mask, nullptr, nullptr, nullptr, g.blended,
      rows_stamp_of(s.backward_row_capacity));
#endif
#undef OLSR_PB_ARGS
  if (o.dL_dtau_sum)
    tau_final_kernel<<<1, TAU_T, 0, st>>>(tau_partials, n_partials, o.dL_dtau_sum, g.counters, o.status_dev, o.sticky_error,
                                          o.status_rows ? rows_stamp_of(s.backward_row_capacity) : 0);
  else if (o.status_dev || o.sticky_error)
    tau_final_kernel<<<1, 64, 0, st>>>(tau_partials, 0, nullptr, g.counters, o.status_dev, o.sticky_error,
                                       o.status_rows ? rows_stamp_of(s.backward_row_capacity) : 0);
}

void launch_preprocess_backward(const olsr_scene& s, int F_rows, const FrameDims& d, const GeometryState& g,
                                const BinningState& b, const float* rows, const int32_t* radii, const GradOut& o,
                                float* tau_partials, hipStream_t st) {
  if (s.P <= 0) {
    if (o.dL_dtau_sum) (void)hipMemsetAsync(o.dL_dtau_sum, 0, 6 * sizeof(float), st);
    return;
  }
  switch (F_rows) {
    case 0: launch_pb_t<0>(s, d, g, b, rows, radii, o, tau_partials, st); break;
    case 3: launch_pb_t<3>(s, d, g, b, rows, radii, o, tau_partials, st); break;
    case 15: launch_pb_t<15>(s, d, g, b, rows, radii, o, tau_partials, st); break;
    case 16: launch_pb_t<16>(s, d, g, b, rows, radii, o, tau_partials, st); break;
    case 32: launch_pb_t<32>(s, d, g, b, rows, radii, o, tau_partials, st); break;
    default: break;
  }
}

}  // namespace olsr
