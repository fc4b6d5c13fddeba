// oracle.cpp — CPU restatement of the reference rasterizer.  TEST INFRASTRUCTURE ONLY.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
// library; the product (online_lang_splatting_amd) never does.
//
// PARITY UNPINNED by the reference's own tests: rpng/online_lang_splatting ships no
// tests, golden vectors or fixtures for the rasterizer and no CPU path, and its CUDA
// sources cannot be built in this image (no nvcc / cub / cooperative_groups; SURVEY.md
// §8(c)).  What IS pinned: SH evaluation, projection-matrix and SE(3) conventions
// against fixtures generated from the reference's importable Python helpers
// (tests/golden/), and the `exact` backward against PyTorch autograd on an independent
// dense formulation (tests/test_oracle_autograd.py).
//
// Every function cites the reference lines it restates:
//   CR  = /root/reference/submodules/diff-gaussian-rasterization/cuda_rasterizer
//   DGR = /root/reference/submodules/diff-gaussian-rasterization
//
// Numerics: strict IEEE fp32, built with -ffp-contract=off (no implicit FMA), in the
// reference's source operation order; the one FMA nvcc's default contraction makes on the
// forward's hot accumulation is written explicitly (see forward()).  glm matrix products follow glm's left-to-right
// accumulation.  `exp` is the one libm call on a decision path; the CUDA libm bits are
// unobtainable here, so the oracle pins exp to a fully specified fp32 routine
// (oracle_expf below, Cephes-style, <= 1 ulp typical) that the HIP kernels restate
// operation for operation — thresholds (alpha < 1/255, T < 1e-4, T > 0.5) therefore
// resolve identically on both sides.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <omp.h>

#include "../include/olsr.h"

namespace {

// ---------------------------------------------------------------- small vector helpers
struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };
struct m3 { float c[3][3]; };  // glm layout: c[col][row]

// glm::mat3 operator* (column-major): result[c][r] = sum_k a[k][r] * b[c][k], k ascending
static m3 mul(const m3& a, const m3& b) {
  m3 o;
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r)
      o.c[c][r] = a.c[0][r] * b.c[c][0] + a.c[1][r] * b.c[c][1] + a.c[2][r] * b.c[c][2];
  return o;
}
static m3 transpose(const m3& a) {
  m3 o;
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) o.c[c][r] = a.c[r][c];
  return o;
}

// CR/auxiliary.h:58-97
static f3 transformPoint4x3(const f3& p, const float* m) {
  return {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
          m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]};
}
static f4 transformPoint4x4(const f3& p, const float* m) {
  return {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
          m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14], m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]};
}
static f3 transformVec4x3Transpose(const f3& p, const float* m) {
  return {m[0] * p.x + m[1] * p.y + m[2] * p.z, m[4] * p.x + m[5] * p.y + m[6] * p.z,
          m[8] * p.x + m[9] * p.y + m[10] * p.z};
}

// CR/auxiliary.h:107-117
static f3 dnormvdv(f3 v, f3 dv) {
  float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
  float invsum32 = 1.0f / std::sqrt(sum2 * sum2 * sum2);
  f3 o;
  o.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
  o.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
  o.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
  return o;
}

// CR/auxiliary.h:22-39
const float SH_C0 = 0.28209479177387814f;
const float SH_C1 = 0.4886025119029199f;
const float SH_C2[] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                       -1.0925484305920792f, 0.5462742152960396f};
const float SH_C3[] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                       -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

// The pinned exp (see header).  Valid for x <= 88; x is clamped below at -87 (the result,
// ~1.6e-38, times any opacity <= 1 is far under the 1/255 alpha floor).
static inline float bits2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline uint32_t f2bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static float oracle_expf(float x) {
#ifdef ORACLE_LIBM_EXP
  // sensitivity variant (scripts/cuda_sensitivity.py): the host libm instead of the pinned routine
  return std::exp(x);
#endif
  x = (x < -87.0f) ? -87.0f : x;
  x = (x > 88.0f) ? 88.0f : x;
  float n = std::nearbyintf(x * 1.44269504088896341f);  // round-half-even
  float r = __builtin_fmaf(n, -0.693359375f, x);
  r = __builtin_fmaf(n, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = __builtin_fmaf(p, r, 1.3981999507e-3f);
  p = __builtin_fmaf(p, r, 8.3334519073e-3f);
  p = __builtin_fmaf(p, r, 4.1665795894e-2f);
  p = __builtin_fmaf(p, r, 1.6666665459e-1f);
  p = __builtin_fmaf(p, r, 5.0000001201e-1f);
  float e = __builtin_fmaf(p, r * r, r) + 1.0f;
  int ni = (int)n;
  return e * bits2f((uint32_t)(ni + 127) << 23);
}

// float -> int conversion with the GPU's saturating semantics (cvt.rzi.s32.f32 on the
// reference's hardware, v_cvt_i32_f32 on gfx950): truncate toward zero, clamp, NaN -> 0.
static inline int f2i_sat(float v) {
  if (v != v) return 0;
  if (v >= 2147483648.0f) return 2147483647;
  if (v <= -2147483648.0f) return (int)0x80000000;
  return (int)v;
}

// CR/auxiliary.h:41-44 — evaluated in double (the literals are double), then narrowed.
static inline float ndc2Pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

// CR/auxiliary.h:46-56
static void getRect(f2 p, int max_radius, int tile, int gx, int gy, int& x0, int& y0, int& x1, int& y1) {
  x0 = std::min(gx, std::max(0, f2i_sat((p.x - max_radius) / tile)));
  y0 = std::min(gy, std::max(0, f2i_sat((p.y - max_radius) / tile)));
  x1 = std::min(gx, std::max(0, f2i_sat((p.x + max_radius + tile - 1) / tile)));
  y1 = std::min(gy, std::max(0, f2i_sat((p.y + max_radius + tile - 1) / tile)));
}

// CR/rasterizer_impl.cu:35-50
static uint32_t getHigherMsb(uint32_t n) {
  uint32_t msb = sizeof(n) * 4;
  uint32_t step = msb;
  while (step > 1) {
    step /= 2;
    if (n >> msb) msb += step; else msb -= step;
  }
  if (n >> msb) msb++;
  return msb;
}

struct State {
  int P = 0, F = 0, W = 0, H = 0, tile = 15, gx = 0, gy = 0, R = 0;
  std::vector<float> depths, means2D, cov3D, conic_opacity, rgb;
  std::vector<uint8_t> clamped;
  std::vector<int> radii;
  std::vector<uint32_t> tiles_touched, point_offsets;
  std::vector<uint64_t> keys;
  std::vector<uint32_t> point_list;
  std::vector<uint32_t> ranges;  // [tiles][2]
  std::vector<float> final_T;
  std::vector<uint32_t> n_contrib;
  // backward internals (kept for inspection)
  std::vector<float> dL_dconic, dL_ddepths;
  // optional (oracle_set_record): per sorted list position, which thread ranks of its tile blended it (256 bits)
  std::vector<uint32_t> contrib_mask;
};

// ------------------------------------------------------------------ forward: preprocess
// CR/forward.cu:23-74
static f3 computeColorFromSH(int idx, int deg, int max_coeffs, const float* means, const float* campos,
                             const float* shs, uint8_t* clamped) {
  f3 pos = {means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]};
  f3 dir = {pos.x - campos[0], pos.y - campos[1], pos.z - campos[2]};
  float len = std::sqrt(dir.x * dir.x + dir.y * dir.y + dir.z * dir.z);
  dir = {dir.x / len, dir.y / len, dir.z / len};
  const float* sh = shs + (size_t)idx * max_coeffs * 3;
  auto S = [&](int k, int ch) { return sh[3 * k + ch]; };
  float res[3];
  for (int ch = 0; ch < 3; ++ch) {
    float result = SH_C0 * S(0, ch);
    if (deg > 0) {
      float x = dir.x, y = dir.y, z = dir.z;
      result = result - SH_C1 * y * S(1, ch) + SH_C1 * z * S(2, ch) - SH_C1 * x * S(3, ch);
      if (deg > 1) {
        float xx = x * x, yy = y * y, zz = z * z;
        float xy = x * y, yz = y * z, xz = x * z;
        result = result + SH_C2[0] * xy * S(4, ch) + SH_C2[1] * yz * S(5, ch) +
                 SH_C2[2] * (2.0f * zz - xx - yy) * S(6, ch) + SH_C2[3] * xz * S(7, ch) +
                 SH_C2[4] * (xx - yy) * S(8, ch);
        if (deg > 2) {
          result = result + SH_C3[0] * y * (3.0f * xx - yy) * S(9, ch) + SH_C3[1] * xy * z * S(10, ch) +
                   SH_C3[2] * y * (4.0f * zz - xx - yy) * S(11, ch) +
                   SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * S(12, ch) +
                   SH_C3[4] * x * (4.0f * zz - xx - yy) * S(13, ch) + SH_C3[5] * z * (xx - yy) * S(14, ch) +
                   SH_C3[6] * x * (xx - 3.0f * yy) * S(15, ch);
        }
      }
    }
    result += 0.5f;
    clamped[3 * idx + ch] = (result < 0);
    res[ch] = std::max(result, 0.0f);
  }
  return {res[0], res[1], res[2]};
}

// CR/forward.cu:121-155
static void computeCov3D(const float* scale, float mod, const float* rot, float* cov3D) {
  m3 S = {{{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}};
  S.c[0][0] = mod * scale[0];
  S.c[1][1] = mod * scale[1];
  S.c[2][2] = mod * scale[2];
  float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
  m3 R = {{{1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
           {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
           {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}}};
  m3 M = mul(S, R);
  m3 Sigma = mul(transpose(M), M);
  cov3D[0] = Sigma.c[0][0];
  cov3D[1] = Sigma.c[0][1];
  cov3D[2] = Sigma.c[0][2];
  cov3D[3] = Sigma.c[1][1];
  cov3D[4] = Sigma.c[1][2];
  cov3D[5] = Sigma.c[2][2];
}

struct Cov2DIntermediates {
  f3 t;  // clamped view-space mean
  float txtz, tytz;
  m3 J, Wm, T, Vrk, cov2D;
};
// CR/forward.cu:77-116 (shared with the recomputation in CR/backward.cu:171-206)
static void cov2D_common(const f3& mean, float focal_x, float focal_y, float tan_fovx, float tan_fovy,
                         const float* cov3D, const float* view, Cov2DIntermediates& o) {
  f3 t = transformPoint4x3(mean, view);
  const float limx = 1.3f * tan_fovx;
  const float limy = 1.3f * tan_fovy;
  o.txtz = t.x / t.z;
  o.tytz = t.y / t.z;
  t.x = std::min(limx, std::max(-limx, o.txtz)) * t.z;
  t.y = std::min(limy, std::max(-limy, o.tytz)) * t.z;
  o.t = t;
  o.J = {{{focal_x / t.z, 0.0f, -(focal_x * t.x) / (t.z * t.z)},
          {0.0f, focal_y / t.z, -(focal_y * t.y) / (t.z * t.z)},
          {0, 0, 0}}};
  o.Wm = {{{view[0], view[4], view[8]}, {view[1], view[5], view[9]}, {view[2], view[6], view[10]}}};
  o.T = mul(o.Wm, o.J);
  o.Vrk = {{{cov3D[0], cov3D[1], cov3D[2]}, {cov3D[1], cov3D[3], cov3D[4]}, {cov3D[2], cov3D[4], cov3D[5]}}};
  o.cov2D = mul(mul(transpose(o.T), transpose(o.Vrk)), o.T);
}

// CR/forward.cu:262-371 (languagePreprocessCUDA) == :158-259 (preprocessCUDA)
static void preprocess(const olsr_scene& s, State& st, int* radii_out) {
  const int P = s.P;
  const float focal_y = s.height / (2.0f * s.tan_fovy);  // CR/rasterizer_impl.cu:394-395
  const float focal_x = s.width / (2.0f * s.tan_fovx);
#pragma omp parallel for schedule(static)
  for (int idx = 0; idx < P; ++idx) {
    radii_out[idx] = 0;
    st.tiles_touched[idx] = 0;
    // in_frustum, CR/auxiliary.h:139-164
    f3 p_orig = {s.means3D[3 * idx], s.means3D[3 * idx + 1], s.means3D[3 * idx + 2]};
    f3 p_view = transformPoint4x3(p_orig, s.viewmatrix);
    if (p_view.z <= 0.2f) continue;  // (prefiltered => printf + trap in the reference)
    f4 p_hom = transformPoint4x4(p_orig, s.projmatrix);
    float p_w = 1.0f / (p_hom.w + 0.0000001f);
    f3 p_proj = {p_hom.x * p_w, p_hom.y * p_w, p_hom.z * p_w};
    const float* cov3D;
    if (s.cov3D_precomp != nullptr) {
      cov3D = s.cov3D_precomp + (size_t)idx * 6;
    } else {
      computeCov3D(s.scales + 3 * (size_t)idx, s.scale_modifier, s.rotations + 4 * (size_t)idx,
                   st.cov3D.data() + (size_t)idx * 6);
      cov3D = st.cov3D.data() + (size_t)idx * 6;
    }
    Cov2DIntermediates ci;
    cov2D_common(p_orig, focal_x, focal_y, s.tan_fovx, s.tan_fovy, cov3D, s.viewmatrix, ci);
    f3 cov = {ci.cov2D.c[0][0] + 0.3f, ci.cov2D.c[0][1], ci.cov2D.c[1][1] + 0.3f};
    float det = (cov.x * cov.z - cov.y * cov.y);
    if (det == 0.0f) continue;
    float det_inv = 1.f / det;
    f3 conic = {cov.z * det_inv, -cov.y * det_inv, cov.x * det_inv};
    float mid = 0.5f * (cov.x + cov.z);
    float lambda1 = mid + std::sqrt(std::max(0.1f, mid * mid - det));
    float lambda2 = mid - std::sqrt(std::max(0.1f, mid * mid - det));
    float my_radius = std::ceil(3.f * std::sqrt(std::max(lambda1, lambda2)));
    f2 point_image = {ndc2Pix(p_proj.x, s.width), ndc2Pix(p_proj.y, s.height)};
    int x0, y0, x1, y1;
    getRect(point_image, f2i_sat(my_radius), s.tile, st.gx, st.gy, x0, y0, x1, y1);
    if ((x1 - x0) * (y1 - y0) == 0) {
      // OLSR_FLAG_SIGNED_EMPTY_RADII (include/olsr.h): the radius DGR-D's preprocess would still report when the
      // Gaussian's other covariance set covers a tile (DGR-D/cuda_rasterizer/forward.cu:391-431)
      if (s.flags & OLSR_FLAG_SIGNED_EMPTY_RADII) radii_out[idx] = -f2i_sat(my_radius);
      continue;
    }
    if (s.colors_precomp == nullptr) {
      f3 c = computeColorFromSH(idx, s.D, s.M, s.means3D, s.cam_pos, s.shs, st.clamped.data());
      st.rgb[3 * (size_t)idx + 0] = c.x;
      st.rgb[3 * (size_t)idx + 1] = c.y;
      st.rgb[3 * (size_t)idx + 2] = c.z;
    }
    st.depths[idx] = p_view.z;
    radii_out[idx] = f2i_sat(my_radius);
    st.means2D[2 * (size_t)idx] = point_image.x;
    st.means2D[2 * (size_t)idx + 1] = point_image.y;
    st.conic_opacity[4 * (size_t)idx + 0] = conic.x;
    st.conic_opacity[4 * (size_t)idx + 1] = conic.y;
    st.conic_opacity[4 * (size_t)idx + 2] = conic.z;
    st.conic_opacity[4 * (size_t)idx + 3] = s.opacities[idx];
    st.tiles_touched[idx] = (uint32_t)((y1 - y0) * (x1 - x0));
  }
}

// ------------------------------------------------------------------ forward: binning
// CR/rasterizer_impl.cu:451-493, 70-138
static void bin_and_sort(const olsr_scene& s, State& st, const int* radii) {
  const int P = s.P;
  uint32_t run = 0;
  for (int i = 0; i < P; ++i) {  // cub::DeviceScan::InclusiveSum
    run += st.tiles_touched[i];
    st.point_offsets[i] = run;
  }
  st.R = (int)run;
  const size_t R = run;
  std::vector<uint64_t> keys_unsorted(R);
  std::vector<uint32_t> vals_unsorted(R);
#pragma omp parallel for schedule(static)
  for (int idx = 0; idx < P; ++idx) {  // duplicateWithKeys
    if (radii[idx] > 0) {
      uint32_t off = (idx == 0) ? 0 : st.point_offsets[idx - 1];
      int x0, y0, x1, y1;
      f2 p = {st.means2D[2 * (size_t)idx], st.means2D[2 * (size_t)idx + 1]};
      getRect(p, radii[idx], s.tile, st.gx, st.gy, x0, y0, x1, y1);
      for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++) {
          uint64_t key = (uint64_t)(y * st.gx + x);
          key <<= 32;
          key |= f2bits(st.depths[idx]);
          keys_unsorted[off] = key;
          vals_unsorted[off] = (uint32_t)idx;
          off++;
        }
    }
  }
  // cub::DeviceRadixSort::SortPairs on bits [0, 32 + getHigherMsb(tiles)): a stable sort.
  const int bit = (int)getHigherMsb((uint32_t)(st.gx * st.gy));
  const uint64_t mask = (32 + bit >= 64) ? ~0ull : ((1ull << (32 + bit)) - 1);
  std::vector<uint32_t> order(R);
  const size_t ntiles = (size_t)st.gx * (size_t)st.gy;
  if (bit < 32 && ((uint64_t)(ntiles ? ntiles - 1 : 0) >> bit) == 0) {
    // Every tile id fits below the mask (always, by getHigherMsb's definition): the stable sort by (tile, depth bits)
    // is a stable partition by tile followed by an independent stable sort of every tile's run by depth bits — the
    // same permutation, bit for bit, with the parallelism of a host CPU (the baseline spent most of a
    // frame in one thread's std::stable_sort).  Partition: per-chunk tile histograms -> exclusive prefix in (tile,
    // chunk) order -> scatter (chunks are contiguous index ranges in ascending order, so equal tiles keep their order).
    const int nthr = std::max(1, omp_get_max_threads());
    const size_t chunk = (R + (size_t)nthr - 1) / (size_t)nthr;
    std::vector<uint32_t> hist((size_t)nthr * ntiles, 0u);
#pragma omp parallel for schedule(static, 1)
    for (int t = 0; t < nthr; ++t) {
      uint32_t* h = &hist[(size_t)t * ntiles];
      const size_t b = std::min(R, (size_t)t * chunk), e = std::min(R, b + chunk);
      for (size_t i = b; i < e; ++i) h[keys_unsorted[i] >> 32]++;
    }
    std::vector<uint32_t> tile_start(ntiles + 1, 0u);
    {
      uint32_t run2 = 0;
      for (size_t tl = 0; tl < ntiles; ++tl) {
        tile_start[tl] = run2;
        for (int t = 0; t < nthr; ++t) {
          const uint32_t c = hist[(size_t)t * ntiles + tl];
          hist[(size_t)t * ntiles + tl] = run2;
          run2 += c;
        }
      }
      tile_start[ntiles] = run2;
    }
#pragma omp parallel for schedule(static, 1)
    for (int t = 0; t < nthr; ++t) {
      uint32_t* h = &hist[(size_t)t * ntiles];
      const size_t b = std::min(R, (size_t)t * chunk), e = std::min(R, b + chunk);
      for (size_t i = b; i < e; ++i) order[h[keys_unsorted[i] >> 32]++] = (uint32_t)i;
    }
#pragma omp parallel for schedule(dynamic, 8)
    for (long long tl = 0; tl < (long long)ntiles; ++tl)
      std::stable_sort(order.begin() + tile_start[tl], order.begin() + tile_start[tl + 1], [&](uint32_t a, uint32_t b) {
        return (uint32_t)keys_unsorted[a] < (uint32_t)keys_unsorted[b];
      });
  } else {
    for (size_t i = 0; i < R; ++i) order[i] = (uint32_t)i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
      return (keys_unsorted[a] & mask) < (keys_unsorted[b] & mask);
    });
  }
  st.keys.resize(R);
  st.point_list.resize(R);
#pragma omp parallel for schedule(static)
  for (long long i = 0; i < (long long)R; ++i) {
    st.keys[i] = keys_unsorted[order[i]];
    st.point_list[i] = vals_unsorted[order[i]];
  }
  // cudaMemset + identifyTileRanges (every write below is to a distinct element: the loop order does not matter)
  std::fill(st.ranges.begin(), st.ranges.end(), 0u);
#pragma omp parallel for schedule(static)
  for (long long idx = 0; idx < (long long)R; ++idx) {
    uint32_t currtile = (uint32_t)(st.keys[idx] >> 32);
    if (idx == 0)
      st.ranges[2 * currtile] = 0;
    else {
      uint32_t prevtile = (uint32_t)(st.keys[idx - 1] >> 32);
      if (currtile != prevtile) {
        st.ranges[2 * prevtile + 1] = (uint32_t)idx;
        st.ranges[2 * currtile] = (uint32_t)idx;
      }
    }
    if ((size_t)idx == R - 1) st.ranges[2 * currtile + 1] = (uint32_t)R;
  }
}

// ------------------------------------------------------------------ forward: composite
// CR/forward.cu:377-513 (language_renderCUDA) and :515-644 (renderCUDA, F == 0).
// One tile at a time; each "thread" is a pixel of the tile.  The batch structure of the
// reference (collective fetch of BLOCK_SIZE entries, tile-wide early exit) does not
// change any per-pixel value, so pixels simply walk the tile's list.
static bool g_record_contrib = false;
static void render_forward(const olsr_scene& s, State& st, float* out_color, float* out_lang,
                           float* out_depth, float* out_opacity, int* n_touched) {
  if (g_record_contrib) st.contrib_mask.assign((size_t)st.R * 8, 0u); else st.contrib_mask.clear();
  const int W = s.width, H = s.height, F = s.F, tile = s.tile;
  const float* features = s.colors_precomp ? s.colors_precomp : st.rgb.data();
  const float* lang = s.language_precomp;
  const int ntiles = st.gx * st.gy;
  const size_t HW = (size_t)H * W;
#pragma omp parallel for schedule(dynamic, 1)
  for (int t = 0; t < ntiles; ++t) {
    const int bx = t % st.gx, by = t / st.gx;
    const uint32_t r0 = st.ranges[2 * t], r1 = st.ranges[2 * t + 1];
    std::vector<float> L(F > 0 ? F : 1);
    for (int ty = 0; ty < tile; ++ty)
      for (int tx = 0; tx < tile; ++tx) {
        const int px = bx * tile + tx, py = by * tile + ty;
        if (!(px < W && py < H)) continue;
        const size_t pix_id = (size_t)W * py + px;
        const float pixfx = (float)px, pixfy = (float)py;
        float T = 1.0f;
        uint32_t contributor = 0, last_contributor = 0;
        float C[3] = {0, 0, 0};
        std::fill(L.begin(), L.end(), 0.0f);
        float D = 0.0f;
        for (uint32_t k = r0; k < r1; ++k) {
          contributor++;
          const uint32_t id = st.point_list[k];
          const float xyx = st.means2D[2 * (size_t)id], xyy = st.means2D[2 * (size_t)id + 1];
          const float dx = xyx - pixfx, dy = xyy - pixfy;
          const float* co = &st.conic_opacity[4 * (size_t)id];
          const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
          if (power > 0.0f) continue;
          const float alpha = std::min(0.99f, co[3] * oracle_expf(power));
          if (alpha < 1.0f / 255.0f) continue;
          const float test_T = T * (1 - alpha);
          if (test_T < 0.0001f) break;  // done = true
          // `C += f * alpha * T` (CR/forward.cu:479-484): the reference is built by nvcc with its default
          // --fmad=true, which contracts the last product into the sum, fma(f * alpha, T, C).  Written
          // out because this file is built with -ffp-contract=off.
          for (int ch = 0; ch < 3; ch++) C[ch] = std::fmaf(features[(size_t)id * 3 + ch] * alpha, T, C[ch]);
          for (int ch = 0; ch < F; ch++) L[ch] = std::fmaf(lang[(size_t)id * F + ch] * alpha, T, L[ch]);
          D = std::fmaf(st.depths[id] * alpha, T, D);
          if (g_record_contrib) {  // (each list position belongs to one tile, each tile to one OpenMP thread)
            const int rank = ty * tile + tx;
            st.contrib_mask[(size_t)k * 8 + (rank >> 5)] |= 1u << (rank & 31);
          }
          if (test_T > 0.5f) {
#pragma omp atomic
            n_touched[id] += 1;
          }
          T = test_T;
          last_contributor = contributor;
        }
        st.final_T[pix_id] = T;
        st.n_contrib[pix_id] = last_contributor;
        for (int ch = 0; ch < 3; ch++) out_color[ch * HW + pix_id] = C[ch] + T * s.background[ch];
        for (int ch = 0; ch < F; ch++) out_lang[ch * HW + pix_id] = L[ch];
        out_depth[pix_id] = D;
        out_opacity[pix_id] = 1 - T;
      }
  }
}

// ------------------------------------------------------------------ backward: composite
// CR/backward.cu:932-1201 (language_render_cuda) and :706-930 (renderCUDA, F == 0).
// Tile-structured on purpose: in REFERENCE mode the per-Gaussian reduction keeps 128 of the
// 225 ranks (render_cuda_reduce_sum, :684-702), language gradients come from rank 0
// (:1137,:1194-1197) and the language recursion is only suspended when the WHOLE tile skips
// a Gaussian (:1087-1093 vs :1127-1139).
struct LaneState {
  bool inside;
  float pixfx, pixfy;
  float T_final, T;
  int last_contributor;
  float accum_rec[3], dL_dpixel[3], last_color[3];
  float accum_rec_depth, dL_dpixel_depth, last_depth, last_alpha;
};

static void render_backward(const olsr_scene& s, State& st, int mode, const float* dL_dpixels,
                            const float* dL_dpixels_lang, const float* dL_dpixels_depth, float* dL_dmean2D,
                            float* dL_dconic2D, float* dL_dopacity, float* dL_dcolors, float* dL_dlanguage,
                            float* dL_ddepths) {
  const int W = s.width, H = s.height, F = s.F, tile = s.tile;
  const int BS = tile * tile;
  const float* colors = s.colors_precomp ? s.colors_precomp : st.rgb.data();
  const float* lang = s.language_precomp;
  const int ntiles = st.gx * st.gy;
  const size_t HW = (size_t)H * W;
  const int NV = 10 + F;  // partial gradient row: mean2D(2) conic(3) opacity colour(3) depth lang(F)
  const size_t R = (size_t)st.R;
  // Per-instance tile sums, later added per Gaussian in tile order (stands in for the
  // reference's order-nondeterministic atomicAdd of thread 0, :1176-1198).
  std::vector<float> inst(R * NV, 0.0f);
  std::vector<uint8_t> inst_used(R, 0);
  const float ddelx_dx = 0.5f * W;
  const float ddely_dy = 0.5f * H;

#pragma omp parallel
  {
    std::vector<LaneState> lanes(BS);
    std::vector<float> lane_accF((size_t)BS * std::max(F, 1)), lane_lastF((size_t)BS * std::max(F, 1)),
        lane_dLF((size_t)BS * std::max(F, 1));
    std::vector<float> part((size_t)NV * BS);  // part[v*BS + rank]
    std::vector<uint8_t> skipv(BS);
    std::vector<float> Gv(BS), alphav(BS), dxv(BS), dyv(BS);
#pragma omp for schedule(dynamic, 1)
    for (int t = 0; t < ntiles; ++t) {
      const int bx = t % st.gx, by = t / st.gx;
      const uint32_t r0 = st.ranges[2 * t], r1 = st.ranges[2 * t + 1];
      if (r1 <= r0) continue;
      for (int rank = 0; rank < BS; ++rank) {
        LaneState& l = lanes[rank];
        const int tx = rank % tile, ty = rank / tile;  // cg thread_rank = ty*BLOCK_X + tx
        const int px = bx * tile + tx, py = by * tile + ty;
        l.inside = px < W && py < H;
        l.pixfx = (float)px;
        l.pixfy = (float)py;
        const size_t pix_id = (size_t)W * py + px;
        l.T_final = l.inside ? st.final_T[pix_id] : 0;
        l.T = l.T_final;
        l.last_contributor = l.inside ? (int)st.n_contrib[pix_id] : 0;
        for (int i = 0; i < 3; ++i) {
          l.accum_rec[i] = 0;
          l.last_color[i] = 0;
          l.dL_dpixel[i] = l.inside ? dL_dpixels[i * HW + pix_id] : 0;
        }
        l.accum_rec_depth = 0;
        l.last_depth = 0;
        l.dL_dpixel_depth = l.inside ? dL_dpixels_depth[pix_id] : 0;
        l.last_alpha = 0.f;
        for (int i = 0; i < F; ++i) {
          lane_accF[(size_t)rank * F + i] = 0;
          lane_lastF[(size_t)rank * F + i] = 0;
          lane_dLF[(size_t)rank * F + i] = l.inside ? dL_dpixels_lang[i * HW + pix_id] : 0;
        }
      }
      const uint32_t toDo = r1 - r0;
      // walk the tile's list back to front; `contributor` after the decrement equals the
      // 0-based list position k (CR/backward.cu:999,1073)
      for (uint32_t k = toDo; k-- > 0;) {
        const uint32_t sorted_pos = r0 + k;
        const uint32_t gid = st.point_list[sorted_pos];
        const float xyx = st.means2D[2 * (size_t)gid], xyy = st.means2D[2 * (size_t)gid + 1];
        const float* co = &st.conic_opacity[4 * (size_t)gid];
        int skip_counter = 0;
        for (int rank = 0; rank < BS; ++rank) {
          LaneState& l = lanes[rank];
          bool skip = !l.inside;  // done = !inside, never changes (:972)
          skip |= ((int)k >= l.last_contributor);
          const float dx = xyx - l.pixfx, dy = xyy - l.pixfy;
          const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
          skip |= power > 0.0f;
          const float G = oracle_expf(power);
          const float alpha = std::min(0.99f, co[3] * G);
          skip |= alpha < 1.0f / 255.0f;
          skipv[rank] = skip;
          Gv[rank] = G;
          alphav[rank] = alpha;
          dxv[rank] = dx;
          dyv[rank] = dy;
          skip_counter += skip;
        }
        if (skip_counter == BS) continue;  // :1091-1093
        inst_used[sorted_pos] = 1;
        const float depth = st.depths[gid];
        for (int rank = 0; rank < BS; ++rank) {
          LaneState& l = lanes[rank];
          const bool skip = skipv[rank];
          const float alpha = alphav[rank], G = Gv[rank], dx = dxv[rank], dy = dyv[rank];
          l.T = skip ? l.T : l.T / (1.f - alpha);
          const float dchannel_dcolor = alpha * l.T;
          float dL_dalpha = 0.0f;
          for (int ch = 0; ch < 3; ch++) {
            const float c = colors[(size_t)gid * 3 + ch];
            l.accum_rec[ch] =
                skip ? l.accum_rec[ch] : l.last_alpha * l.last_color[ch] + (1.f - l.last_alpha) * l.accum_rec[ch];
            l.last_color[ch] = skip ? l.last_color[ch] : c;
            const float dL_dchannel = l.dL_dpixel[ch];
            dL_dalpha += (c - l.accum_rec[ch]) * dL_dchannel;
            part[(size_t)(6 + ch) * BS + rank] = skip ? 0.0f : dchannel_dcolor * dL_dchannel;
          }
          l.accum_rec_depth =
              skip ? l.accum_rec_depth : l.last_alpha * l.last_depth + (1.f - l.last_alpha) * l.accum_rec_depth;
          l.last_depth = skip ? l.last_depth : depth;
          dL_dalpha += (depth - l.accum_rec_depth) * l.dL_dpixel_depth;
          part[(size_t)9 * BS + rank] = skip ? 0.f : dchannel_dcolor * l.dL_dpixel_depth;
          for (int ch = 0; ch < F; ch++) {
            const float f = lang[(size_t)gid * F + ch];
            float& acc = lane_accF[(size_t)rank * F + ch];
            float& lastf = lane_lastF[(size_t)rank * F + ch];
            if (mode == OLSR_BWD_REFERENCE) {  // unguarded, :1132-1133
              acc = l.last_alpha * lastf + (1.f - l.last_alpha) * acc;
              lastf = f;
            } else {
              acc = skip ? acc : l.last_alpha * lastf + (1.f - l.last_alpha) * acc;
              lastf = skip ? lastf : f;
            }
            const float dL_dchannel_F = lane_dLF[(size_t)rank * F + ch];
            dL_dalpha += (f - acc) * dL_dchannel_F;
            part[(size_t)(10 + ch) * BS + rank] = skip ? 0.0f : dchannel_dcolor * dL_dchannel_F;
          }
          dL_dalpha *= l.T;
          l.last_alpha = skip ? l.last_alpha : alpha;
          float bg_dot_dpixel = 0.f;
          for (int i = 0; i < 3; i++) bg_dot_dpixel += s.background[i] * l.dL_dpixel[i];
          dL_dalpha += (-l.T_final / (1.f - alpha)) * bg_dot_dpixel;
          const float dL_dG = co[3] * dL_dalpha;
          const float gdx = G * dx;
          const float gdy = G * dy;
          const float dG_ddelx = -gdx * co[0] - gdy * co[1];
          const float dG_ddely = -gdy * co[2] - gdx * co[1];
          part[(size_t)0 * BS + rank] = skip ? 0.f : dL_dG * dG_ddelx * ddelx_dx;
          part[(size_t)1 * BS + rank] = skip ? 0.f : dL_dG * dG_ddely * ddely_dy;
          part[(size_t)2 * BS + rank] = skip ? 0.f : -0.5f * gdx * dx * dL_dG;
          part[(size_t)3 * BS + rank] = skip ? 0.f : -0.5f * gdx * dy * dL_dG;
          part[(size_t)4 * BS + rank] = skip ? 0.f : -0.5f * gdy * dy * dL_dG;
          part[(size_t)5 * BS + rank] = skip ? 0.f : G * dL_dalpha;
        }
        float* row = &inst[(size_t)sorted_pos * NV];
        if (mode == OLSR_BWD_REFERENCE) {
          // render_cuda_reduce_sum over g.size() == BS lanes with integer halving (:696)
          for (int v = 0; v < 10; ++v) {
            float* a = &part[(size_t)v * BS];
            for (int i = BS / 2; i > 0; i /= 2)
              for (int lane = 0; lane < i; ++lane) a[lane] += a[lane + i];
            row[v] = a[0];
          }
          for (int ch = 0; ch < F; ++ch) row[10 + ch] = part[(size_t)(10 + ch) * BS];  // rank 0 only
        } else {
          for (int v = 0; v < NV; ++v) {
            double acc = 0.0;
            const float* a = &part[(size_t)v * BS];
            for (int lane = 0; lane < BS; ++lane) acc += a[lane];
            row[v] = (float)acc;
          }
        }
      }
    }
  }
  // "atomicAdd" of every tile's thread 0, in sorted (tile-major, then depth) order.  A Gaussian's sum only depends on the
  // order of ITS OWN rows, so the Gaussians are dealt to threads by index range and every thread walks the whole sorted
  // list, adding the rows of its Gaussians as it meets them: the same additions in the same order as the serial loop.
  {
    const int nthr = std::max(1, omp_get_max_threads());
    const int P = s.P;
#pragma omp parallel for schedule(static, 1)
    for (int t = 0; t < nthr; ++t) {
      const uint32_t g0 = (uint32_t)((long long)P * t / nthr), g1 = (uint32_t)((long long)P * (t + 1) / nthr);
      if (g0 == g1) continue;
      for (size_t sp = 0; sp < R; ++sp) {
        const uint32_t gid = st.point_list[sp];
        if (gid < g0 || gid >= g1 || !inst_used[sp]) continue;
        const float* row = &inst[sp * NV];
        dL_dmean2D[3 * (size_t)gid + 0] += row[0];
        dL_dmean2D[3 * (size_t)gid + 1] += row[1];
        dL_dconic2D[4 * (size_t)gid + 0] += row[2];
        dL_dconic2D[4 * (size_t)gid + 1] += row[3];
        dL_dconic2D[4 * (size_t)gid + 3] += row[4];
        dL_dopacity[gid] += row[5];
        dL_dcolors[(size_t)gid * 3 + 0] += row[6];
        dL_dcolors[(size_t)gid * 3 + 1] += row[7];
        dL_dcolors[(size_t)gid * 3 + 2] += row[8];
        dL_ddepths[gid] += row[9];
        for (int ch = 0; ch < F; ++ch) dL_dlanguage[(size_t)gid * F + ch] += row[10 + ch];
      }
    }
  }
}

// ------------------------------------------------------------------ backward: preprocess
// CR/backward.cu:150-346 (computeCov2DCUDA)
static void computeCov2D_backward(const olsr_scene& s, const State& st, const int* radii, const float* cov3Ds,
                                  const float* dL_dconics, float* dL_dmeans, float* dL_dcov, float* dL_dtau) {
  (void)st;
  const float h_y = s.height / (2.0f * s.tan_fovy);
  const float h_x = s.width / (2.0f * s.tan_fovx);
  const float* view = s.viewmatrix;
#pragma omp parallel for schedule(static)
  for (int idx = 0; idx < s.P; ++idx) {
    if (!(radii[idx] > 0)) continue;
    const float* cov3D = cov3Ds + 6 * (size_t)idx;
    f3 mean = {s.means3D[3 * idx], s.means3D[3 * idx + 1], s.means3D[3 * idx + 2]};
    f3 dL_dconic = {dL_dconics[4 * (size_t)idx], dL_dconics[4 * (size_t)idx + 1], dL_dconics[4 * (size_t)idx + 3]};
    Cov2DIntermediates ci;
    cov2D_common(mean, h_x, h_y, s.tan_fovx, s.tan_fovy, cov3D, view, ci);
    const f3 t = ci.t;
    const float limx = 1.3f * s.tan_fovx, limy = 1.3f * s.tan_fovy;
    const float x_grad_mul = ci.txtz < -limx || ci.txtz > limx ? 0 : 1;
    const float y_grad_mul = ci.tytz < -limy || ci.tytz > limy ? 0 : 1;
    const m3& J = ci.J;
    const m3& Wm = ci.Wm;
    const m3& T = ci.T;
    const m3& Vrk = ci.Vrk;
    float a = ci.cov2D.c[0][0] + 0.3f;
    float b = ci.cov2D.c[0][1];
    float c = ci.cov2D.c[1][1] + 0.3f;
    float denom = a * c - b * b;
    float dL_da = 0, dL_db = 0, dL_dc = 0;
    float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    float* oc = dL_dcov + 6 * (size_t)idx;
    if (denom2inv != 0) {
      dL_da = denom2inv * (-c * c * dL_dconic.x + 2 * b * c * dL_dconic.y + (denom - a * c) * dL_dconic.z);
      dL_dc = denom2inv * (-a * a * dL_dconic.z + 2 * a * b * dL_dconic.y + (denom - a * c) * dL_dconic.x);
      dL_db = denom2inv * 2 * (b * c * dL_dconic.x - (denom + 2 * b * b) * dL_dconic.y + a * b * dL_dconic.z);
      oc[0] = (T.c[0][0] * T.c[0][0] * dL_da + T.c[0][0] * T.c[1][0] * dL_db + T.c[1][0] * T.c[1][0] * dL_dc);
      oc[3] = (T.c[0][1] * T.c[0][1] * dL_da + T.c[0][1] * T.c[1][1] * dL_db + T.c[1][1] * T.c[1][1] * dL_dc);
      oc[5] = (T.c[0][2] * T.c[0][2] * dL_da + T.c[0][2] * T.c[1][2] * dL_db + T.c[1][2] * T.c[1][2] * dL_dc);
      oc[1] = 2 * T.c[0][0] * T.c[0][1] * dL_da + (T.c[0][0] * T.c[1][1] + T.c[0][1] * T.c[1][0]) * dL_db +
              2 * T.c[1][0] * T.c[1][1] * dL_dc;
      oc[2] = 2 * T.c[0][0] * T.c[0][2] * dL_da + (T.c[0][0] * T.c[1][2] + T.c[0][2] * T.c[1][0]) * dL_db +
              2 * T.c[1][0] * T.c[1][2] * dL_dc;
      oc[4] = 2 * T.c[0][2] * T.c[0][1] * dL_da + (T.c[0][1] * T.c[1][2] + T.c[0][2] * T.c[1][1]) * dL_db +
              2 * T.c[1][1] * T.c[1][2] * dL_dc;
    } else {
      for (int i = 0; i < 6; i++) oc[i] = 0;
    }
    float dL_dT00 = 2 * (T.c[0][0] * Vrk.c[0][0] + T.c[0][1] * Vrk.c[0][1] + T.c[0][2] * Vrk.c[0][2]) * dL_da +
                    (T.c[1][0] * Vrk.c[0][0] + T.c[1][1] * Vrk.c[0][1] + T.c[1][2] * Vrk.c[0][2]) * dL_db;
    float dL_dT01 = 2 * (T.c[0][0] * Vrk.c[1][0] + T.c[0][1] * Vrk.c[1][1] + T.c[0][2] * Vrk.c[1][2]) * dL_da +
                    (T.c[1][0] * Vrk.c[1][0] + T.c[1][1] * Vrk.c[1][1] + T.c[1][2] * Vrk.c[1][2]) * dL_db;
    float dL_dT02 = 2 * (T.c[0][0] * Vrk.c[2][0] + T.c[0][1] * Vrk.c[2][1] + T.c[0][2] * Vrk.c[2][2]) * dL_da +
                    (T.c[1][0] * Vrk.c[2][0] + T.c[1][1] * Vrk.c[2][1] + T.c[1][2] * Vrk.c[2][2]) * dL_db;
    float dL_dT10 = 2 * (T.c[1][0] * Vrk.c[0][0] + T.c[1][1] * Vrk.c[0][1] + T.c[1][2] * Vrk.c[0][2]) * dL_dc +
                    (T.c[0][0] * Vrk.c[0][0] + T.c[0][1] * Vrk.c[0][1] + T.c[0][2] * Vrk.c[0][2]) * dL_db;
    float dL_dT11 = 2 * (T.c[1][0] * Vrk.c[1][0] + T.c[1][1] * Vrk.c[1][1] + T.c[1][2] * Vrk.c[1][2]) * dL_dc +
                    (T.c[0][0] * Vrk.c[1][0] + T.c[0][1] * Vrk.c[1][1] + T.c[0][2] * Vrk.c[1][2]) * dL_db;
    float dL_dT12 = 2 * (T.c[1][0] * Vrk.c[2][0] + T.c[1][1] * Vrk.c[2][1] + T.c[1][2] * Vrk.c[2][2]) * dL_dc +
                    (T.c[0][0] * Vrk.c[2][0] + T.c[0][1] * Vrk.c[2][1] + T.c[0][2] * Vrk.c[2][2]) * dL_db;
    float dL_dJ00 = Wm.c[0][0] * dL_dT00 + Wm.c[0][1] * dL_dT01 + Wm.c[0][2] * dL_dT02;
    float dL_dJ02 = Wm.c[2][0] * dL_dT00 + Wm.c[2][1] * dL_dT01 + Wm.c[2][2] * dL_dT02;
    float dL_dJ11 = Wm.c[1][0] * dL_dT10 + Wm.c[1][1] * dL_dT11 + Wm.c[1][2] * dL_dT12;
    float dL_dJ12 = Wm.c[2][0] * dL_dT10 + Wm.c[2][1] * dL_dT11 + Wm.c[2][2] * dL_dT12;
    float tz = 1.f / t.z;
    float tz2 = tz * tz;
    float tz3 = tz2 * tz;
    float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
    float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
    float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t.x) * tz3 * dL_dJ02 +
                   (2 * h_y * t.y) * tz3 * dL_dJ12;
    // pose part 1 (:273-288): dpC_drho = I, dpC_dtheta = -skew(t), cols of skew(v):
    // (0,v.z,-v.y), (-v.z,0,v.x), (v.y,-v.x,0)  (CR/math.h:27-31)
    const f3 rho_cols[3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    const f3 th_cols[3] = {{-0.f, -t.z, t.y}, {t.z, -0.f, -t.x}, {-t.y, t.x, -0.f}};
    float* tau = dL_dtau + 6 * (size_t)idx;
    float dL_dt[6];
    for (int i = 0; i < 3; i++) {
      dL_dt[i] = dL_dtx * rho_cols[i].x + dL_dty * rho_cols[i].y + dL_dtz * rho_cols[i].z;
      dL_dt[i + 3] = dL_dtx * th_cols[i].x + dL_dty * th_cols[i].y + dL_dtz * th_cols[i].z;
    }
    for (int i = 0; i < 6; i++) tau[i] += dL_dt[i];
    f3 dL_dmean = transformVec4x3Transpose({dL_dtx, dL_dty, dL_dtz}, view);
    dL_dmeans[3 * (size_t)idx + 0] = dL_dmean.x;  // assignment (:297)
    dL_dmeans[3 * (size_t)idx + 1] = dL_dmean.y;
    dL_dmeans[3 * (size_t)idx + 2] = dL_dmean.z;
    float dL_dW00 = J.c[0][0] * dL_dT00;
    float dL_dW01 = J.c[0][0] * dL_dT01;
    float dL_dW02 = J.c[0][0] * dL_dT02;
    float dL_dW10 = J.c[1][1] * dL_dT10;
    float dL_dW11 = J.c[1][1] * dL_dT11;
    float dL_dW12 = J.c[1][1] * dL_dT12;
    float dL_dW20 = J.c[0][2] * dL_dT00 + J.c[1][2] * dL_dT10;
    float dL_dW21 = J.c[0][2] * dL_dT01 + J.c[1][2] * dL_dT11;
    float dL_dW22 = J.c[0][2] * dL_dT02 + J.c[1][2] * dL_dT12;
    // SE3(view).R() columns (CR/math.h:275-281): col i = (view[4i], view[4i+1], view[4i+2])
    f3 c1 = {view[0], view[1], view[2]}, c2 = {view[4], view[5], view[6]}, c3 = {view[8], view[9], view[10]};
    // dL_dW (mat33 from data[9], column-major): cols
    f3 dW1 = {dL_dW00, dL_dW10, dL_dW20}, dW2 = {dL_dW01, dL_dW11, dL_dW21}, dW3 = {dL_dW02, dL_dW12, dL_dW22};
    auto nskew_col = [](const f3& v, int i) -> f3 {  // column i of -skew(v)
      if (i == 0) return {-0.f, -v.z, v.y};
      if (i == 1) return {v.z, -0.f, -v.x};
      return {-v.y, v.x, -0.f};
    };
    auto dot = [](const f3& a, const f3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; };
    float dth[3];
    for (int i = 0; i < 3; ++i)
      dth[i] = dot(dW1, nskew_col(c1, i)) + dot(dW2, nskew_col(c2, i)) + dot(dW3, nskew_col(c3, i));
    tau[3] += dth[0];
    tau[4] += dth[1];
    tau[5] += dth[2];
  }
}

// CR/backward.cu:21-145
static void computeColorFromSH_backward(int idx, int deg, int max_coeffs, const float* means, const float* campos,
                                        const float* shs, const uint8_t* clamped, const float* dL_dcolor,
                                        float* dL_dmeans, float* dL_dshs, float* dL_dtau) {
  f3 pos = {means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]};
  f3 dir_orig = {pos.x - campos[0], pos.y - campos[1], pos.z - campos[2]};
  float len = std::sqrt(dir_orig.x * dir_orig.x + dir_orig.y * dir_orig.y + dir_orig.z * dir_orig.z);
  f3 dir = {dir_orig.x / len, dir_orig.y / len, dir_orig.z / len};
  const float* sh = shs + (size_t)idx * max_coeffs * 3;
  float dL_dRGB[3];
  for (int ch = 0; ch < 3; ++ch) dL_dRGB[ch] = dL_dcolor[3 * (size_t)idx + ch] * (clamped[3 * idx + ch] ? 0 : 1);
  float dRGBdx[3] = {0, 0, 0}, dRGBdy[3] = {0, 0, 0}, dRGBdz[3] = {0, 0, 0};
  float x = dir.x, y = dir.y, z = dir.z;
  float* dL_dsh = dL_dshs + (size_t)idx * max_coeffs * 3;
  auto S = [&](int k, int ch) { return sh[3 * k + ch]; };
  auto setsh = [&](int k, float w) {
    for (int ch = 0; ch < 3; ++ch) dL_dsh[3 * k + ch] = w * dL_dRGB[ch];
  };
  setsh(0, SH_C0);
  if (deg > 0) {
    setsh(1, -SH_C1 * y);
    setsh(2, SH_C1 * z);
    setsh(3, -SH_C1 * x);
    for (int ch = 0; ch < 3; ++ch) {
      dRGBdx[ch] = -SH_C1 * S(3, ch);
      dRGBdy[ch] = -SH_C1 * S(1, ch);
      dRGBdz[ch] = SH_C1 * S(2, ch);
    }
    if (deg > 1) {
      float xx = x * x, yy = y * y, zz = z * z;
      float xy = x * y, yz = y * z, xz = x * z;
      setsh(4, SH_C2[0] * xy);
      setsh(5, SH_C2[1] * yz);
      setsh(6, SH_C2[2] * (2.f * zz - xx - yy));
      setsh(7, SH_C2[3] * xz);
      setsh(8, SH_C2[4] * (xx - yy));
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
  auto dot3 = [&](const float* a) { return a[0] * dL_dRGB[0] + a[1] * dL_dRGB[1] + a[2] * dL_dRGB[2]; };
  f3 dL_ddir = {dot3(dRGBdx), dot3(dRGBdy), dot3(dRGBdz)};
  f3 dL_dmean = dnormvdv(dir_orig, dL_ddir);
  dL_dmeans[3 * (size_t)idx + 0] += dL_dmean.x;
  dL_dmeans[3 * (size_t)idx + 1] += dL_dmean.y;
  dL_dmeans[3 * (size_t)idx + 2] += dL_dmean.z;
  dL_dtau[6 * (size_t)idx + 0] += -dL_dmean.x;
  dL_dtau[6 * (size_t)idx + 1] += -dL_dmean.y;
  dL_dtau[6 * (size_t)idx + 2] += -dL_dmean.z;
}

// CR/backward.cu:350-413
static void computeCov3D_backward(int idx, const float* scale, float mod, const float* rot, const float* dL_dcov3Ds,
                                  float* dL_dscales, float* dL_drots) {
  float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
  m3 R = {{{1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
           {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
           {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}}};
  m3 S = {{{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}};
  f3 sv = {mod * scale[0], mod * scale[1], mod * scale[2]};
  S.c[0][0] = sv.x;
  S.c[1][1] = sv.y;
  S.c[2][2] = sv.z;
  m3 M = mul(S, R);
  const float* d = dL_dcov3Ds + 6 * (size_t)idx;
  m3 dL_dSigma = {{{d[0], 0.5f * d[1], 0.5f * d[2]}, {0.5f * d[1], d[3], 0.5f * d[4]}, {0.5f * d[2], 0.5f * d[4], d[5]}}};
  // dL_dM = 2.0f * M * dL_dSigma  (glm: scalar*mat first, then mat*mat)
  m3 M2;
  for (int c = 0; c < 3; ++c)
    for (int rr = 0; rr < 3; ++rr) M2.c[c][rr] = M.c[c][rr] * 2.0f;
  m3 dL_dM = mul(M2, dL_dSigma);
  m3 Rt = transpose(R);
  m3 dL_dMt = transpose(dL_dM);
  auto dotc = [](const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
  float* ds = dL_dscales + 3 * (size_t)idx;
  ds[0] = dotc(Rt.c[0], dL_dMt.c[0]);
  ds[1] = dotc(Rt.c[1], dL_dMt.c[1]);
  ds[2] = dotc(Rt.c[2], dL_dMt.c[2]);
  for (int k = 0; k < 3; ++k) {
    dL_dMt.c[0][k] *= sv.x;
    dL_dMt.c[1][k] *= sv.y;
    dL_dMt.c[2][k] *= sv.z;
  }
  auto m = [&](int c, int rr) { return dL_dMt.c[c][rr]; };
  float* dq = dL_drots + 4 * (size_t)idx;
  dq[0] = 2 * z * (m(0, 1) - m(1, 0)) + 2 * y * (m(2, 0) - m(0, 2)) + 2 * x * (m(1, 2) - m(2, 1));
  dq[1] = 2 * y * (m(1, 0) + m(0, 1)) + 2 * z * (m(2, 0) + m(0, 2)) + 2 * r * (m(1, 2) - m(2, 1)) -
          4 * x * (m(2, 2) + m(1, 1));
  dq[2] = 2 * x * (m(1, 0) + m(0, 1)) + 2 * r * (m(2, 0) - m(0, 2)) + 2 * z * (m(1, 2) + m(2, 1)) -
          4 * y * (m(2, 2) + m(0, 0));
  dq[3] = 2 * r * (m(0, 1) - m(1, 0)) + 2 * x * (m(2, 0) + m(0, 2)) + 2 * y * (m(1, 2) + m(2, 1)) -
          4 * z * (m(1, 1) + m(0, 0));
}

// CR/backward.cu:541-682 (language_preprocessCUDA) == :418-539 (preprocessCUDA)
static void preprocess_backward(const olsr_scene& s, const State& st, const int* radii, const float* dL_dmean2D,
                                float* dL_dmeans, float* dL_dcolor, const float* dL_ddepth, const float* dL_dcov3D,
                                float* dL_dsh, float* dL_dscale, float* dL_drot, float* dL_dtau) {
  const float* proj = s.projmatrix;
  const float* view = s.viewmatrix;
  const float* proj_raw = s.projmatrix_raw;
#pragma omp parallel for schedule(static)
  for (int idx = 0; idx < s.P; ++idx) {
    if (!(radii[idx] > 0)) continue;
    f3 m = {s.means3D[3 * idx], s.means3D[3 * idx + 1], s.means3D[3 * idx + 2]};
    f4 m_hom = transformPoint4x4(m, proj);
    float m_w = 1.0f / (m_hom.w + 0.0000001f);
    const float g2x = dL_dmean2D[3 * (size_t)idx], g2y = dL_dmean2D[3 * (size_t)idx + 1];
    float mul1 = (proj[0] * m.x + proj[4] * m.y + proj[8] * m.z + proj[12]) * m_w * m_w;
    float mul2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * m_w * m_w;
    f3 dL_dmean;
    dL_dmean.x = (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
    dL_dmean.y = (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
    dL_dmean.z = (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;
    float* dm = dL_dmeans + 3 * (size_t)idx;
    dm[0] += dL_dmean.x;
    dm[1] += dL_dmean.y;
    dm[2] += dL_dmean.z;
    float alpha = 1.0f * m_w;
    float beta = -m_hom.x * m_w * m_w;
    float gamma = -m_hom.y * m_w * m_w;
    float a = proj_raw[0];
    float b = proj_raw[5];
    float e = proj_raw[11];
    // SE3 T_CW(viewmatrix); p_C = R*m + t (CR/math.h:322-324, mat33*float3 :84-91)
    f3 c0 = {view[0], view[1], view[2]}, c1 = {view[4], view[5], view[6]}, c2 = {view[8], view[9], view[10]};
    f3 tt = {view[12], view[13], view[14]};
    f3 Rm = {c0.x * m.x + c1.x * m.y + c2.x * m.z, c0.y * m.x + c1.y * m.y + c2.y * m.z,
             c0.z * m.x + c1.z * m.y + c2.z * m.z};
    f3 p_C = {Rm.x + tt.x, Rm.y + tt.y, Rm.z + tt.z};
    // dp_C_d_theta = -skew(p_C); its transpose times v: rows of transpose = columns of -skew
    const f3 th_cols[3] = {{-0.f, -p_C.z, p_C.y}, {p_C.z, -0.f, -p_C.x}, {-p_C.y, p_C.x, -0.f}};
    f3 d1 = {alpha * a, 0.f, beta * e};
    f3 d2 = {0.f, alpha * b, gamma * e};
    // mat33::transpose() * v (CR/math.h:33-40,84-91): out.k = col_k(original) . v, summed x,y,z
    auto tmul = [](const f3 cols[3], const f3& v) -> f3 {
      // transpose has cols' = rows; (M^T v).x = c0'.x*v.x + c1'.x*v.y + c2'.x*v.z with c_j'.x = cols[0].{x,y,z}[j]
      return {cols[0].x * v.x + cols[0].y * v.y + cols[0].z * v.z, cols[1].x * v.x + cols[1].y * v.y + cols[1].z * v.z,
              cols[2].x * v.x + cols[2].y * v.y + cols[2].z * v.z};
    };
    const f3 I_cols[3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    f3 d1_rho = tmul(I_cols, d1), d2_rho = tmul(I_cols, d2);
    f3 d1_th = tmul(th_cols, d1), d2_th = tmul(th_cols, d2);
    float dmx[6] = {d1_rho.x, d1_rho.y, d1_rho.z, d1_th.x, d1_th.y, d1_th.z};
    float dmy[6] = {d2_rho.x, d2_rho.y, d2_rho.z, d2_th.x, d2_th.y, d2_th.z};
    float* tau = dL_dtau + 6 * (size_t)idx;
    float dL_dt[6];
    for (int i = 0; i < 6; i++) dL_dt[i] = g2x * dmx[i] + g2y * dmy[i];
    for (int i = 0; i < 6; i++) tau[i] += dL_dt[i];
    float dL_dpCz = dL_ddepth[idx];
    dm[0] += dL_dpCz * view[2];
    dm[1] += dL_dpCz * view[6];
    dm[2] += dL_dpCz * view[10];
    for (int i = 0; i < 3; i++) {
      tau[i] += dL_dpCz * I_cols[i].z;
      tau[i + 3] += dL_dpCz * th_cols[i].z;
    }
    if (s.shs)
      computeColorFromSH_backward(idx, s.D, s.M, s.means3D, s.cam_pos, s.shs, st.clamped.data(), dL_dcolor, dL_dmeans,
                                  dL_dsh, dL_dtau);
    if (s.scales)
      computeCov3D_backward(idx, s.scales + 3 * (size_t)idx, s.scale_modifier, s.rotations + 4 * (size_t)idx,
                            dL_dcov3D, dL_dscale, dL_drot);
  }
}

static bool check_scene(const olsr_scene* s) {
  if (!s || s->P < 0 || s->width <= 0 || s->height <= 0 || s->tile <= 0) return false;
  if (s->P == 0) return true;  // nothing is dereferenced (DGR/rasterize_points.cu:187,400)
  if ((s->shs == nullptr) == (s->colors_precomp == nullptr)) return false;
  const bool has_sr = s->scales != nullptr && s->rotations != nullptr;
  if (has_sr == (s->cov3D_precomp != nullptr)) return false;
  if (s->F > 0 && s->language_precomp == nullptr) return false;
  return true;
}

}  // namespace

extern "C" {

void* oracle_create() { return new State(); }
void oracle_destroy(void* h) { delete (State*)h; }

// DGR/rasterize_points.cu:135-241 + CR/rasterizer_impl.cu:364-525 (F > 0) / :216-362 (F == 0).
// All pointers are host pointers.  Outputs are fully overwritten.
int oracle_forward(void* h, const olsr_scene* s, float* out_color, float* out_language, float* out_depth,
                   float* out_opacity, int32_t* radii, int32_t* n_touched, int32_t* num_rendered) {
  if (!h || !check_scene(s)) return OLSR_ERR_ARG;
  const double t_in = omp_get_wtime();
  State& st = *(State*)h;
  const int P = s->P, W = s->width, H = s->height, F = s->F;
  st.P = P; st.F = F; st.W = W; st.H = H; st.tile = s->tile;
  st.gx = (W + s->tile - 1) / s->tile;
  st.gy = (H + s->tile - 1) / s->tile;
  const size_t N = (size_t)W * H;
  st.depths.assign(P, 0.f);
  st.means2D.assign((size_t)2 * P, 0.f);
  st.cov3D.assign((size_t)6 * P, 0.f);
  st.conic_opacity.assign((size_t)4 * P, 0.f);
  st.rgb.assign((size_t)3 * P, 0.f);
  st.clamped.assign((size_t)3 * P, 0);
  st.tiles_touched.assign(P, 0);
  st.point_offsets.assign(P, 0);
  st.ranges.assign((size_t)2 * st.gx * st.gy, 0);
  st.final_T.assign(N, 0.f);
  st.n_contrib.assign(N, 0);
  std::fill(out_color, out_color + 3 * N, 0.f);  // torch::full(..., 0.0) DGR/rasterize_points.cu:170-175
  if (F > 0) std::fill(out_language, out_language + (size_t)F * N, 0.f);
  std::fill(out_depth, out_depth + N, 0.f);
  std::fill(out_opacity, out_opacity + N, 0.f);
  std::fill(radii, radii + P, 0);
  std::fill(n_touched, n_touched + P, 0);
  st.R = 0;
  st.keys.clear();
  st.point_list.clear();
  if (P != 0) {
    const bool tm = std::getenv("ORACLE_TIMING") != nullptr;  // phase times to stderr (profiling the CPU baseline)
    const double t0 = omp_get_wtime();
    preprocess(*s, st, radii);
    st.radii.assign(radii, radii + P);
    const double t1 = omp_get_wtime();
    bin_and_sort(*s, st, radii);
    const double t2 = omp_get_wtime();
    render_forward(*s, st, out_color, out_language, out_depth, out_opacity, n_touched);
    const double t3 = omp_get_wtime();
    if (tm) std::fprintf(stderr, "[oracle fwd] alloc+fill %.3f preprocess %.3f bin_and_sort %.3f render %.3f s (%d threads)\n",
                         t0 - t_in, t1 - t0, t2 - t1, t3 - t2, omp_get_max_threads());
  }
  *num_rendered = st.R;
  return OLSR_OK;
}

// DGR/rasterize_points.cu:344-455 + CR/rasterizer_impl.cu:638-756 (F > 0) / :529-636 (F == 0).
int oracle_backward(void* h, const olsr_scene* s, const int32_t* radii, const float* dL_dout_color,
                    const float* dL_dout_language, const float* dL_dout_depth, float* dL_dmeans2D, float* dL_dconic,
                    float* dL_dopacity, float* dL_dcolors, float* dL_dlanguage, float* dL_ddepths, float* dL_dmeans3D,
                    float* dL_dcov3D, float* dL_dsh, float* dL_dscales, float* dL_drotations, float* dL_dtau) {
  if (!h || !check_scene(s) || !s->projmatrix_raw) return OLSR_ERR_ARG;
  State& st = *(State*)h;
  const int P = s->P, F = s->F, M = s->M;
  if (st.P != P || st.F != F || st.W != s->width || st.H != s->height) return OLSR_ERR_ARG;
  std::vector<float> conic_local, depth_local;
  if (!dL_dconic) { conic_local.assign((size_t)4 * P, 0.f); dL_dconic = conic_local.data(); }
  if (!dL_ddepths) { depth_local.assign(P, 0.f); dL_ddepths = depth_local.data(); }
  // torch::zeros for every gradient, DGR/rasterize_points.cu:386-398
  std::fill(dL_dmeans2D, dL_dmeans2D + (size_t)3 * P, 0.f);
  std::fill(dL_dconic, dL_dconic + (size_t)4 * P, 0.f);
  std::fill(dL_dopacity, dL_dopacity + P, 0.f);
  std::fill(dL_dcolors, dL_dcolors + (size_t)3 * P, 0.f);
  if (F > 0) std::fill(dL_dlanguage, dL_dlanguage + (size_t)F * P, 0.f);
  std::fill(dL_ddepths, dL_ddepths + P, 0.f);
  std::fill(dL_dmeans3D, dL_dmeans3D + (size_t)3 * P, 0.f);
  std::fill(dL_dcov3D, dL_dcov3D + (size_t)6 * P, 0.f);
  if (M > 0) std::fill(dL_dsh, dL_dsh + (size_t)3 * M * P, 0.f);
  std::fill(dL_dscales, dL_dscales + (size_t)3 * P, 0.f);
  std::fill(dL_drotations, dL_drotations + (size_t)4 * P, 0.f);
  std::fill(dL_dtau, dL_dtau + (size_t)6 * P, 0.f);
  if (P == 0) return OLSR_OK;
  const double t0 = omp_get_wtime();
  render_backward(*s, st, s->bwd_mode, dL_dout_color, dL_dout_language, dL_dout_depth, dL_dmeans2D, dL_dconic,
                  dL_dopacity, dL_dcolors, dL_dlanguage, dL_ddepths);
  const double t1 = omp_get_wtime();
  const float* cov3D_ptr = s->cov3D_precomp ? s->cov3D_precomp : st.cov3D.data();
  computeCov2D_backward(*s, st, radii, cov3D_ptr, dL_dconic, dL_dmeans3D, dL_dcov3D, dL_dtau);
  preprocess_backward(*s, st, radii, dL_dmeans2D, dL_dmeans3D, dL_dcolors, dL_ddepths, dL_dcov3D, dL_dsh, dL_dscales,
                      dL_drotations, dL_dtau);
  if (std::getenv("ORACLE_TIMING"))
    std::fprintf(stderr, "[oracle bwd] render %.3f per-Gaussian %.3f s\n", t1 - t0, omp_get_wtime() - t1);
  return OLSR_OK;
}

// The per-Gaussian half of the backward alone (CR/rasterizer_impl.cu:702-756: computeCov2DCUDA, then
// language_preprocessCUDA / preprocessCUDA), fed with composite-level gradients the CALLER provides instead of the ones
// oracle_backward's own composite produced.  Test infrastructure for one question: with identical inputs, does the
// product's fused per-Gaussian kernel agree with the reference's chain?  (Fed with the product's own composite-level
// gradients it separates the chain's arithmetic from the chain's sensitivity to its inputs.)
int oracle_backward_chain(void* h, const olsr_scene* s, const int32_t* radii, const float* dL_dmeans2D_in,
                          const float* dL_dconic_in, const float* dL_dcolors_in, const float* dL_ddepths_in,
                          float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscales, float* dL_drotations,
                          float* dL_dtau) {
  if (!h || !check_scene(s) || !s->projmatrix_raw) return OLSR_ERR_ARG;
  State& st = *(State*)h;
  const int P = s->P, M = s->M;
  if (st.P != P || st.F != s->F || st.W != s->width || st.H != s->height) return OLSR_ERR_ARG;
  std::fill(dL_dmeans3D, dL_dmeans3D + (size_t)3 * P, 0.f);
  std::fill(dL_dcov3D, dL_dcov3D + (size_t)6 * P, 0.f);
  if (M > 0) std::fill(dL_dsh, dL_dsh + (size_t)3 * M * P, 0.f);
  std::fill(dL_dscales, dL_dscales + (size_t)3 * P, 0.f);
  std::fill(dL_drotations, dL_drotations + (size_t)4 * P, 0.f);
  std::fill(dL_dtau, dL_dtau + (size_t)6 * P, 0.f);
  if (P == 0) return OLSR_OK;
  std::vector<float> colors(dL_dcolors_in, dL_dcolors_in + (size_t)3 * P);  // (the chain may write into it)
  const float* cov3D_ptr = s->cov3D_precomp ? s->cov3D_precomp : st.cov3D.data();
  computeCov2D_backward(*s, st, radii, cov3D_ptr, dL_dconic_in, dL_dmeans3D, dL_dcov3D, dL_dtau);
  preprocess_backward(*s, st, radii, dL_dmeans2D_in, dL_dmeans3D, colors.data(), dL_ddepths_in, dL_dcov3D, dL_dsh,
                      dL_dscales, dL_drotations, dL_dtau);
  return OLSR_OK;
}

// DGR/rasterize_points.cu:457-476 + CR/rasterizer_impl.cu:54-66
int oracle_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                        uint8_t* present) {
  (void)projmatrix;
  for (int idx = 0; idx < P; ++idx) {
    f3 p = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
    f3 pv = transformPoint4x3(p, viewmatrix);
    present[idx] = !(pv.z <= 0.2f);
  }
  return OLSR_OK;
}

// State inspection for stage-by-stage parity.  Returns element count, copies into dst if non-NULL.
int64_t oracle_get_field(void* h, const char* name, void* dst) {
  State& st = *(State*)h;
  auto cp = [&](const void* src, size_t n, size_t esz) -> int64_t {
    if (dst) std::memcpy(dst, src, n * esz);
    return (int64_t)n;
  };
  if (!std::strcmp(name, "depths")) return cp(st.depths.data(), st.depths.size(), 4);
  if (!std::strcmp(name, "means2D")) return cp(st.means2D.data(), st.means2D.size(), 4);
  if (!std::strcmp(name, "cov3D")) return cp(st.cov3D.data(), st.cov3D.size(), 4);
  if (!std::strcmp(name, "conic_opacity")) return cp(st.conic_opacity.data(), st.conic_opacity.size(), 4);
  if (!std::strcmp(name, "rgb")) return cp(st.rgb.data(), st.rgb.size(), 4);
  if (!std::strcmp(name, "clamped")) return cp(st.clamped.data(), st.clamped.size(), 1);
  if (!std::strcmp(name, "tiles_touched")) return cp(st.tiles_touched.data(), st.tiles_touched.size(), 4);
  if (!std::strcmp(name, "point_offsets")) return cp(st.point_offsets.data(), st.point_offsets.size(), 4);
  if (!std::strcmp(name, "point_list")) return cp(st.point_list.data(), st.point_list.size(), 4);
  if (!std::strcmp(name, "keys")) return cp(st.keys.data(), st.keys.size(), 8);
  if (!std::strcmp(name, "ranges")) return cp(st.ranges.data(), st.ranges.size(), 4);
  if (!std::strcmp(name, "final_T")) return cp(st.final_T.data(), st.final_T.size(), 4);
  if (!std::strcmp(name, "n_contrib")) return cp(st.n_contrib.data(), st.n_contrib.size(), 4);
  if (!std::strcmp(name, "contrib_mask")) return cp(st.contrib_mask.data(), st.contrib_mask.size(), 4);
  return -1;
}

float oracle_expf_probe(float x) { return oracle_expf(x); }

// computeCov3D's backward (CR/backward.cu:350-413) on its own, for the gradient goldens generated from the reference's
// build_covariance_from_scaling_rotation (tests/golden/make_golden_gradients.py): the same function preprocess_backward calls.
int oracle_cov3d_backward(int32_t P, const float* scales, float scale_modifier, const float* rotations,
                          const float* dL_dcov3D, float* dL_dscales, float* dL_drotations) {
  if (P < 0 || !scales || !rotations || !dL_dcov3D || !dL_dscales || !dL_drotations) return OLSR_ERR_ARG;
  for (int idx = 0; idx < P; ++idx)
    computeCov3D_backward(idx, scales + 3 * (size_t)idx, scale_modifier, rotations + 4 * (size_t)idx, dL_dcov3D,
                          dL_dscales, dL_drotations);
  return OLSR_OK;
}

// OpenMP team size of the following calls (bench.py's cpu_baseline: all host cores, and one thread)
// record, per sorted list position, the 256-bit mask of tile thread ranks that blended it (sensitivity studies)
void oracle_set_record(int on) { g_record_contrib = on != 0; }
const char* oracle_variant() {
#if defined(ORACLE_LIBM_EXP)
  return "libm_exp";
#elif defined(ORACLE_CONTRACT_FAST)
  return "contract_fast";
#else
  return "default";
#endif
}
void oracle_set_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }
int oracle_get_threads() { return omp_get_max_threads(); }

// distCUDA2 / SimpleKNN::knn (/root/reference/submodules/simple-knn/spatial.cu:15-26,
// simple_knn.cu:131-145,185-221): mean of the squared distances to the 3 nearest neighbours.  The
// reference searches Morton-sorted boxes with conservative pruning, i.e. it returns the EXACT three
// smallest distances; restated here as the brute-force definition (O(P^2), OpenMP over points) with
// the same update rule (updateKBest<3>) and the distance expression as nvcc contracts it
// (d.x*d.x + d.y*d.y + d.z*d.z -> fma(d.z, d.z, fma(d.y, d.y, d.x*d.x))).  Missing neighbours (P < 4) stay
// FLT_MAX as in the reference.  Parity unpinned by the reference (no tests, CUDA only).
int oracle_knn_mean_dist2(int32_t P, const float* points, float* out) {
  if (P < 0 || (P > 0 && (!points || !out))) return OLSR_ERR_ARG;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < P; ++i) {
    const float qx = points[3 * (size_t)i], qy = points[3 * (size_t)i + 1], qz = points[3 * (size_t)i + 2];
    float best[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};
    for (int j = 0; j < P; ++j) {
      if (j == i) continue;
      const float dx = points[3 * (size_t)j] - qx, dy = points[3 * (size_t)j + 1] - qy,
                  dz = points[3 * (size_t)j + 2] - qz;
      float dist = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
      for (int k = 0; k < 3; ++k) {
        if (best[k] > dist) {
          const float t = best[k];
          best[k] = dist;
          dist = t;
        }
      }
    }
    out[i] = ((best[0] + best[1]) + best[2]) / 3.0f;
  }
  return OLSR_OK;
}

}  // extern "C"
