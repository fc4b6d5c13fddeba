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
#include "olsr_device.h"
#include "olsr_kernels.h"

namespace olsr {

constexpr int PB_THREADS = 128;
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
constexpr int RR_BIG_BLOCKS = 2048;  // persistent grid of the wave-per-Gaussian kernel (8 waves/SIMD)

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
                                                                    const int32_t* __restrict__ counters) {
  constexpr int ROW = grad_row(F);
  constexpr int NVAL = 10 + F;
  constexpr int NP = next_pow2_(NVAL);
  constexpr int G_LANES = 64 / NP;
  const int lane = threadIdx.x & 63;
  const int wave = (int)(blockIdx.x * (RR_THREADS / 64) + (threadIdx.x >> 6));
  const int nwaves = (int)(gridDim.x * (RR_THREADS / 64));
  const int nbig = counters[5], count = nbig + counters[4];  // front list, then the medium list from the back
  if (wave >= count || counters[7] != 0) return;
  auto item_at = [&](int i) { return big_list[i < nbig ? i : P - 1 - (i - nbig)]; };
  uint4 cur = item_at(wave);
  for (int item = wave; item < count; item += nwaves) {
    const int nxt = item + nwaves;
    const uint4 next = item_at(nxt < count ? nxt : item);
    const u32 idx = cur.x, first = rowbase[cur.y], nrows = rowbase[cur.y + cur.z] - first;
    float acc[NP];
#pragma unroll
    for (int v = 0; v < NP; ++v) acc[v] = 0.f;
    for (u32 t = (u32)lane; t < nrows; t += 64) add_row<F, NP>(rows, first + t, acc);
    wave_reduce_rec<NP / 2, 32, NP>(acc, lane);
    float v = __shfl(acc[0], (lane * G_LANES) & 63);
    if (lane >= NVAL) v = 0.f;
    if (lane < ROW) gacc[(size_t)idx * ROW + lane] = v;
    cur = next;
  }
}

// -skew(v) column i (CR/math.h:27-31 negated)
__device__ __forceinline__ f3 nskew_col(const f3& v, int i) {
  if (i == 0) return {-0.f, -v.z, v.y};
  if (i == 1) return {v.z, -0.f, -v.x};
  return {-v.y, v.x, -0.f};
}
__device__ __forceinline__ float dot3(const f3& a, const f3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

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

// computeCov3D backward, CR/backward.cu:350-413
__device__ __forceinline__ void cov3d_backward(const float* scale, float mod, const float* rot, const float* d,
                                               float* ds, float* dq) {
  const float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
  const m3 R = {{{1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
                 {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
                 {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}}};
  m3 S = {{{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}}};
  const f3 sv = {mod * scale[0], mod * scale[1], mod * scale[2]};
  S.c[0][0] = sv.x;
  S.c[1][1] = sv.y;
  S.c[2][2] = sv.z;
  const m3 M = mul(S, R);
  const m3 dL_dSigma = {
      {{d[0], 0.5f * d[1], 0.5f * d[2]}, {0.5f * d[1], d[3], 0.5f * d[4]}, {0.5f * d[2], 0.5f * d[4], d[5]}}};
  m3 M2;
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) M2.c[c][rr] = M.c[c][rr] * 2.0f;
  const m3 dL_dM = mul(M2, dL_dSigma);
  const m3 Rt = transpose(R);
  m3 dL_dMt = transpose(dL_dM);
  ds[0] = Rt.c[0][0] * dL_dMt.c[0][0] + Rt.c[0][1] * dL_dMt.c[0][1] + Rt.c[0][2] * dL_dMt.c[0][2];
  ds[1] = Rt.c[1][0] * dL_dMt.c[1][0] + Rt.c[1][1] * dL_dMt.c[1][1] + Rt.c[1][2] * dL_dMt.c[1][2];
  ds[2] = Rt.c[2][0] * dL_dMt.c[2][0] + Rt.c[2][1] * dL_dMt.c[2][1] + Rt.c[2][2] * dL_dMt.c[2][2];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    dL_dMt.c[0][k] *= sv.x;
    dL_dMt.c[1][k] *= sv.y;
    dL_dMt.c[2][k] *= sv.z;
  }
#define MT(ci_, ri_) dL_dMt.c[ci_][ri_]
  dq[0] = 2 * z * (MT(0, 1) - MT(1, 0)) + 2 * y * (MT(2, 0) - MT(0, 2)) + 2 * x * (MT(1, 2) - MT(2, 1));
  dq[1] = 2 * y * (MT(1, 0) + MT(0, 1)) + 2 * z * (MT(2, 0) + MT(0, 2)) + 2 * r * (MT(1, 2) - MT(2, 1)) -
          4 * x * (MT(2, 2) + MT(1, 1));
  dq[2] = 2 * x * (MT(1, 0) + MT(0, 1)) + 2 * r * (MT(2, 0) - MT(0, 2)) + 2 * z * (MT(1, 2) + MT(2, 1)) -
          4 * y * (MT(2, 2) + MT(0, 0));
  dq[3] = 2 * r * (MT(0, 1) - MT(1, 0)) + 2 * x * (MT(2, 0) + MT(0, 2)) + 2 * y * (MT(1, 2) + MT(2, 1)) -
          4 * z * (MT(1, 1) + MT(0, 0));
#undef MT
}

template <int F>
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
    const float* __restrict__ opacities_raw) {
  constexpr int ROW = grad_row(F);
  constexpr int NVAL = 10 + F;
  extern __shared__ float s_bucket[];  // [PB_THREADS][width] when a gradient bucket is given
  const int width = 11 + 3 * M + F;    // floats per Gaussian in the gradient bucket
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  float tau[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (r < P) {
    const u32 idx = (u32)r;
    const bool vis = radii[idx] > 0;
    float acc[NVAL];
#pragma unroll
    for (int v = 0; v < NVAL; ++v) acc[v] = 0.f;
    const u32 ntiles_g = vis ? tiles_touched[idx] : 0u;  // (a Gaussian listed in no tile has no rows: all zeros)
    if (ntiles_g > OLSR_MID_FOOTPRINT) {
      if (counters[7] == 0) {  // summed by row_reduce_big_kernel
        const float4* row = reinterpret_cast<const float4*>(gacc + (size_t)idx * ROW);
#pragma unroll
        for (int v4 = 0; v4 < (NVAL + 3) / 4; ++v4) {
          const float4 x = row[v4];
          if (4 * v4 + 0 < NVAL) acc[4 * v4 + 0] = x.x;
          if (4 * v4 + 1 < NVAL) acc[4 * v4 + 1] = x.y;
          if (4 * v4 + 2 < NVAL) acc[4 * v4 + 2] = x.z;
          if (4 * v4 + 3 < NVAL) acc[4 * v4 + 3] = x.w;
        }
      }
    } else if (ntiles_g > 0 && counters[7] == 0) {
      // the Gaussian's partial-gradient rows are one dense run (emission order): sum them here, in ascending
      // (tile, wave) order — no intermediate per-Gaussian buffer
      const u32 u0 = inst_start[idx];
      const u32 first = rowbase[u0], nrows = rowbase[u0 + ntiles_g] - first;
      for (u32 t = 0; t < nrows; ++t) add_row<F, NVAL>(rows, first + t, acc);
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
    }
    if (brow) {
      float* o = brow + 3 + 3 * M;  // [opacity | scale 3 | rotation 4 | language F]
      o[0] = badd ? o[0] + acc[5] : acc[5];
      if constexpr (F > 0) {
#pragma unroll
        for (int ch = 0; ch < F; ++ch) o[8 + ch] = badd ? o[8 + ch] + acc[10 + ch] : acc[10 + ch];
      }
    }

    float dmean[3] = {0.f, 0.f, 0.f};
    float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float dscale[3] = {0.f, 0.f, 0.f};
    float drot[4] = {0.f, 0.f, 0.f, 0.f};
    bool sh_written = false;
    if (vis) {
      const f3 mean = {means3D[3 * (size_t)idx], means3D[3 * (size_t)idx + 1], means3D[3 * (size_t)idx + 2]};
      // ---- computeCov2DCUDA, CR/backward.cu:150-346
      {
        float cov3D[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) cov3D[i] = cov3Ds[6 * (size_t)idx + i];
        const f3 dL_dconic3 = {acc[2], acc[3], acc[4]};
        Cov2D ci;
        cov2d_common(mean, h_x, h_y, tan_fovx, tan_fovy, cov3D, view, ci);
        const f3 t = ci.t;
        const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
        const float x_grad_mul = ci.txtz < -limx || ci.txtz > limx ? 0 : 1;
        const float y_grad_mul = ci.tytz < -limy || ci.tytz > limy ? 0 : 1;
        const m3& J = ci.J;
        const m3& Wm = ci.Wm;
        const m3& T = ci.T;
        const m3& Vrk = ci.Vrk;
        const float a = ci.cov.c[0][0] + 0.3f;
        const float b = ci.cov.c[0][1];
        const float c = ci.cov.c[1][1] + 0.3f;
        const float denom = a * c - b * b;
        float dL_da = 0, dL_db = 0, dL_dc = 0;
        const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        if (denom2inv != 0) {
          dL_da = denom2inv * (-c * c * dL_dconic3.x + 2 * b * c * dL_dconic3.y + (denom - a * c) * dL_dconic3.z);
          dL_dc = denom2inv * (-a * a * dL_dconic3.z + 2 * a * b * dL_dconic3.y + (denom - a * c) * dL_dconic3.x);
          dL_db = denom2inv * 2 * (b * c * dL_dconic3.x - (denom + 2 * b * b) * dL_dconic3.y + a * b * dL_dconic3.z);
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
        const f3 rho_cols[3] = {{1.f, 0.f, 0.f}, {0.f, 1.f, 0.f}, {0.f, 0.f, 1.f}};
        float dL_dt[6];
#pragma unroll
        for (int i = 0; i < 3; i++) {
          const f3 c_theta = nskew_col(t, i);
          dL_dt[i] = dL_dtx * rho_cols[i].x + dL_dty * rho_cols[i].y + dL_dtz * rho_cols[i].z;
          dL_dt[i + 3] = dL_dtx * c_theta.x + dL_dty * c_theta.y + dL_dtz * c_theta.z;
        }
#pragma unroll
        for (int i = 0; i < 6; i++) tau[i] += dL_dt[i];
        const f3 dm = transformVec4x3Transpose({dL_dtx, dL_dty, dL_dtz}, view);
        dmean[0] = dm.x;
        dmean[1] = dm.y;
        dmean[2] = dm.z;
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
      // ---- preprocessCUDA / language_preprocessCUDA backward, CR/backward.cu:569-681
      {
        const f3 m = mean;
        const f4 m_hom = transformPoint4x4(m, proj);
        const float m_w = 1.0f / (m_hom.w + 0.0000001f);
        const float g2x = acc[0], g2y = acc[1];
        const float mul1 = (proj[0] * m.x + proj[4] * m.y + proj[8] * m.z + proj[12]) * m_w * m_w;
        const float mul2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * m_w * m_w;
        f3 dL_dmean;
        dL_dmean.x = (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
        dL_dmean.y = (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
        dL_dmean.z = (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;
        dmean[0] += dL_dmean.x;
        dmean[1] += dL_dmean.y;
        dmean[2] += dL_dmean.z;
        const float alpha = 1.0f * m_w;
        const float beta = -m_hom.x * m_w * m_w;
        const float gamma = -m_hom.y * m_w * m_w;
        const float a = proj_raw[0];
        const float b = proj_raw[5];
        const float e = proj_raw[11];
        const f3 c0 = {view[0], view[1], view[2]}, c1 = {view[4], view[5], view[6]}, c2 = {view[8], view[9], view[10]};
        const f3 tt = {view[12], view[13], view[14]};
        const f3 Rm = {c0.x * m.x + c1.x * m.y + c2.x * m.z, c0.y * m.x + c1.y * m.y + c2.y * m.z,
                       c0.z * m.x + c1.z * m.y + c2.z * m.z};
        const f3 p_C = {Rm.x + tt.x, Rm.y + tt.y, Rm.z + tt.z};
        const f3 d1 = {alpha * a, 0.f, beta * e};
        const f3 d2 = {0.f, alpha * b, gamma * e};
        const f3 I_cols[3] = {{1.f, 0.f, 0.f}, {0.f, 1.f, 0.f}, {0.f, 0.f, 1.f}};
        float dmx[6], dmy[6];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const f3 th = nskew_col(p_C, i);
          dmx[i] = I_cols[i].x * d1.x + I_cols[i].y * d1.y + I_cols[i].z * d1.z;
          dmy[i] = I_cols[i].x * d2.x + I_cols[i].y * d2.y + I_cols[i].z * d2.z;
          dmx[3 + i] = th.x * d1.x + th.y * d1.y + th.z * d1.z;
          dmy[3 + i] = th.x * d2.x + th.y * d2.y + th.z * d2.z;
        }
        float dL_dt[6];
#pragma unroll
        for (int i = 0; i < 6; i++) dL_dt[i] = g2x * dmx[i] + g2y * dmy[i];
#pragma unroll
        for (int i = 0; i < 6; i++) tau[i] += dL_dt[i];
        const float dL_dpCz = acc[9];
        dmean[0] += dL_dpCz * view[2];
        dmean[1] += dL_dpCz * view[6];
        dmean[2] += dL_dpCz * view[10];
#pragma unroll
        for (int i = 0; i < 3; i++) {
          const f3 th = nskew_col(p_C, i);
          tau[i] += dL_dpCz * I_cols[i].z;
          tau[i + 3] += dL_dpCz * th.z;
        }
        if (shs) {
          const float dcol[3] = {acc[6], acc[7], acc[8]};
          // one evaluation: into the caller's dL_dsh row if there is one (copied to the bucket below),
          // else straight into the bucket row
          float* sh_row = dL_dsh ? dL_dsh + (size_t)idx * M * 3 : (brow ? brow + 3 : nullptr);
          const f3 dm_sh = sh_backward((int)idx, D, M, mean, campos, shs, clamped, dcol, sh_row, !dL_dsh && badd);
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
    if (bucket_assign)
      for (int e = threadIdx.x; e < count; e += PB_THREADS) out[e] = s_bucket[e];
    else
      for (int e = threadIdx.x; e < count; e += PB_THREADS) out[e] += s_bucket[e];
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

// one block: 1024 threads stride over the block partials (six consecutive floats each), fixed-order reduction
__global__ __launch_bounds__(1024) void tau_final_kernel(const float* __restrict__ partials, int nb,
                                                         float* __restrict__ out) {
  __shared__ float red[16][6];
  float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int b = threadIdx.x; b < nb; b += 1024) {
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
    for (int w = 0; w < 16; ++w) v += red[w][threadIdx.x];
    out[threadIdx.x] = v;
  }
}

template <int F>
static void launch_pb_t(const olsr_scene& s, const FrameDims& d, const GeometryState& g, const BinningState& b,
                        const float* rows, const int32_t* radii, const GradOut& o, float* tau_partials,
                        hipStream_t st) {
  const int nb = tau_partial_blocks(s.P);
  const float* cov3D_ptr = s.cov3D_precomp ? s.cov3D_precomp : g.cov3D;
  row_reduce_big_kernel<F><<<RR_BIG_BLOCKS, RR_THREADS, 0, st>>>(rows, g.gacc, g.big_list, s.P, b.rowbase, g.counters);
  const size_t bucket_lds = o.bucket_flat ? sizeof(float) * PB_THREADS * (size_t)(11 + 3 * s.M + s.F) : 0;
  preprocess_bwd_kernel<F><<<nb, PB_THREADS, bucket_lds, st>>>(
      s.P, s.D, s.M, g.gacc, g.tiles_touched, g.inst_start, b.rowbase, rows, g.counters, s.means3D, radii, s.shs,
      g.clamped,
      s.scales, s.rotations, s.scale_modifier, cov3D_ptr, s.viewmatrix, s.projmatrix, s.projmatrix_raw, s.cam_pos,
      d.focal_x, d.focal_y, s.tan_fovx, s.tan_fovy, o.dL_dmeans2D, o.dL_dconic, o.dL_dopacity, o.dL_dcolors,
      o.dL_dlanguage, o.dL_ddepths, o.dL_dmeans3D, o.dL_dcov3D, o.dL_dsh, o.dL_dscales, o.dL_drotations, o.dL_dtau,
      o.dL_dtau_sum ? tau_partials : nullptr, o.bucket_flat, o.bucket_densify, o.bucket_max_radii, o.bucket_assign,
      s.activations, s.opacities);
  if (o.dL_dtau_sum) tau_final_kernel<<<1, 1024, 0, st>>>(tau_partials, nb, o.dL_dtau_sum);
}

void launch_preprocess_backward(const olsr_scene& s, const FrameDims& d, const GeometryState& g,
                                const BinningState& b, const float* rows, const int32_t* radii, const GradOut& o,
                                float* tau_partials, hipStream_t st) {
  if (s.P <= 0) {
    if (o.dL_dtau_sum) (void)hipMemsetAsync(o.dL_dtau_sum, 0, 6 * sizeof(float), st);
    return;
  }
  switch (s.F) {
    case 0: launch_pb_t<0>(s, d, g, b, rows, radii, o, tau_partials, st); break;
    case 3: launch_pb_t<3>(s, d, g, b, rows, radii, o, tau_partials, st); break;
    case 15: launch_pb_t<15>(s, d, g, b, rows, radii, o, tau_partials, st); break;
    case 16: launch_pb_t<16>(s, d, g, b, rows, radii, o, tau_partials, st); break;
    case 32: launch_pb_t<32>(s, d, g, b, rows, radii, o, tau_partials, st); break;
    default: break;
  }
}

}  // namespace olsr
