// k_preprocess.hip — per-Gaussian forward preprocess and the near-plane visibility test.
//
// Replaces preprocessCUDA / languagePreprocessCUDA (CR/forward.cu:158-259, 262-371) and
// checkFrustum (CR/rasterizer_impl.cu:54-66).  One lane per Gaussian; the 4x4 matrices are
// wave-uniform and arrive through the scalar cache.  Besides the reference's outputs the
// kernel seeds the depth sort: key = depth bits (for every Gaussian, culled or not: see
// preprocess_one), value = Gaussian index.
#include "olsr_device.h"
#include "olsr_kernels.h"

namespace olsr {

// computeColorFromSH, CR/forward.cu:23-74
__device__ __forceinline__ f3 color_from_sh(int idx, int deg, int max_coeffs, const f3& pos, const float* campos,
                                            const float* __restrict__ shs, uint8_t* __restrict__ clamped) {
  f3 dir = {pos.x - campos[0], pos.y - campos[1], pos.z - campos[2]};
  const float len = sqrtf(dir.x * dir.x + dir.y * dir.y + dir.z * dir.z);
  dir = {dir.x / len, dir.y / len, dir.z / len};
  const float* sh = shs + (size_t)idx * max_coeffs * 3;
  float res[3];
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    float result = SH_C0 * sh[ch];
    if (deg > 0) {
      const float x = dir.x, y = dir.y, z = dir.z;
      result = result - SH_C1 * y * sh[3 * 1 + ch] + SH_C1 * z * sh[3 * 2 + ch] - SH_C1 * x * sh[3 * 3 + ch];
      if (deg > 1) {
        const float xx = x * x, yy = y * y, zz = z * z;
        const float xy = x * y, yz = y * z, xz = x * z;
        result = result + SH_C2[0] * xy * sh[3 * 4 + ch] + SH_C2[1] * yz * sh[3 * 5 + ch] +
                 SH_C2[2] * (2.0f * zz - xx - yy) * sh[3 * 6 + ch] + SH_C2[3] * xz * sh[3 * 7 + ch] +
                 SH_C2[4] * (xx - yy) * sh[3 * 8 + ch];
        if (deg > 2) {
          result = result + SH_C3[0] * y * (3.0f * xx - yy) * sh[3 * 9 + ch] + SH_C3[1] * xy * z * sh[3 * 10 + ch] +
                   SH_C3[2] * y * (4.0f * zz - xx - yy) * sh[3 * 11 + ch] +
                   SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[3 * 12 + ch] +
                   SH_C3[4] * x * (4.0f * zz - xx - yy) * sh[3 * 13 + ch] + SH_C3[5] * z * (xx - yy) * sh[3 * 14 + ch] +
                   SH_C3[6] * x * (xx - 3.0f * yy) * sh[3 * 15 + ch];
        }
      }
    }
    result += 0.5f;
    clamped[3 * (size_t)idx + ch] = (result < 0);
    res[ch] = fmaxf_ref(result, 0.0f);
  }
  return {res[0], res[1], res[2]};
}

// computeCov3D, CR/forward.cu:121-155
__device__ __forceinline__ void cov3d_from_scale_rot(const float* scale, float mod, const float* rot, float* cov3D) {
  m3 S = {{{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}}};
  S.c[0][0] = mod * scale[0];
  S.c[1][1] = mod * scale[1];
  S.c[2][2] = mod * scale[2];
  const float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
  const m3 R = {{{1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
                 {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
                 {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}}};
  const m3 M = mul(S, R);
  const m3 Sigma = mul(transpose(M), M);
  cov3D[0] = Sigma.c[0][0];
  cov3D[1] = Sigma.c[0][1];
  cov3D[2] = Sigma.c[0][2];
  cov3D[3] = Sigma.c[1][1];
  cov3D[4] = Sigma.c[1][2];
  cov3D[5] = Sigma.c[2][2];
}

// One Gaussian; returns the area of the reference's tile rectangle (0 when culled).
// Exact binning counts, per tile row of a Gaussian's rect, the tile columns its alpha-floor ellipse reaches.  A lane that
// walked its own rows made the wave wait for its tallest member (2.5 rows on average, ~10 for the tallest of 64, ~100
// instructions per row): the rows of the wave's 64 Gaussians are instead dealt one per lane — a row -> owner table in
// LDS, the owner's ellipse read back from LDS, one LDS atomic per row for the sum.  Same function, same inputs as the
// emission (cull_row_span), so the counts and the emitted instances agree by construction.
struct PendingRows {
  CullEllipse e;
  int ya, nrows, x0, x1;
  float depth;  // view-space depth of the Gaussian (per-tile depth cut-offs, include/olsr.h)
};
constexpr int PRE_ROWCAP = 512;  // rows per fill of a wave's row -> owner table
struct PreWaveLds {
  float4 e0[64];  // px py b inv_a
  float4 e1[64];  // A det_lo xstar ystar
  int4 m[64];     // exact, x0 | x1 << 16, first row, first row's index among the wave's rows
  float dz[64];   // depth (only read with depth cut-offs)
  u32 keep[64];   // depth cut-offs: some tile of the footprint has a cut-off behind the Gaussian
  u32 cnt[64];
  uint8_t owner[PRE_ROWCAP];
};

// cut (may be null): per-tile depth cut-offs — a Gaussian that lies behind the cut-off of EVERY tile its footprint reaches
// counts no tiles at all (the emission then skips it like a culled one); any other keeps its whole footprint
template <int TILE>
__device__ __forceinline__ u32 count_pending_rows(const PendingRows& pd, PreWaveLds& L, int W, int H, int gx,
                                                  const float* __restrict__ cut) {
  const int lane = threadIdx.x & 63;
  const u32 nrows = (u32)pd.nrows;
  u32 incl = nrows;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const u32 o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  const u32 total = (u32)__shfl((int)incl, 63);
  if (total == 0) return 0;  // (wave-uniform)
  const u32 rowoff = incl - nrows;
  L.cnt[lane] = 0;
  L.keep[lane] = 0;
  if (nrows) {
    L.e0[lane] = make_float4(pd.e.px, pd.e.py, pd.e.b, pd.e.inv_a);
    L.e1[lane] = make_float4(pd.e.A, pd.e.det_lo, pd.e.xstar, pd.e.ystar);
    L.m[lane] = make_int4(pd.e.exact ? 1 : 0, pd.x0 | (pd.x1 << 16), pd.ya, (int)rowoff);
    L.dz[lane] = pd.depth;
  }
  for (u32 sbase = 0; sbase < total; sbase += PRE_ROWCAP) {
    const u32 k0 = max(rowoff, sbase), k1 = min(rowoff + nrows, sbase + (u32)PRE_ROWCAP);
    for (u32 k = k0; k < k1; ++k) L.owner[k - sbase] = (uint8_t)lane;
    // one wave, one LDS queue: its own writes are visible to its later reads; only the compiler must not reorder
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const u32 send = min(total, sbase + (u32)PRE_ROWCAP);
    for (u32 rbase = sbase; rbase < send; rbase += 64) {
      const u32 idx = rbase + (u32)lane;
      if (idx < send) {
        const int o = (int)L.owner[idx - sbase];
        const float4 q0 = L.e0[o], q1 = L.e1[o];
        const int4 m = L.m[o];
        CullEllipse e;
        e.px = q0.x; e.py = q0.y; e.b = q0.z; e.inv_a = q0.w;
        e.A = q1.x; e.det_lo = q1.y; e.xstar = q1.z; e.ystar = q1.w;
        e.ymax = 0.f;
        e.exact = m.x != 0;
        const int y = m.z + (int)(idx - (u32)m.w);
        int xa = m.y & 0xFFFF, xb = (int)((u32)m.y >> 16);
        cull_row_span<TILE>(e, xa, xb, y, W, H, xa, xb);
        if (xb > xa) {
          atomicAdd(&L.cnt[o], (u32)(xb - xa));
          if (cut != nullptr && L.keep[o] == 0u) {  // (stale reads only repeat the search)
            const float dzo = L.dz[o];
            for (int x = xa; x < xb; ++x)
              if (depth_cut_keeps(dzo, cut[y * gx + x])) {
                L.keep[o] = 1u;
                break;
              }
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  return (cut != nullptr && L.keep[lane] == 0u) ? 0u : L.cnt[lane];
}

template <int TILE>
__device__ __forceinline__ u32 preprocess_one(
    int idx, int D, int M, const float* __restrict__ orig_points, const float* __restrict__ scales, float scale_modifier,
    const float* __restrict__ rotations, const float* __restrict__ opacities, const float* __restrict__ shs,
    uint8_t* __restrict__ clamped, const float* __restrict__ cov3D_precomp, const float* __restrict__ colors_precomp,
    const float* __restrict__ viewmatrix, const float* __restrict__ projmatrix, const float* __restrict__ cam_pos, int W,
    int H, float tan_fovx, float tan_fovy, float focal_x, float focal_y, int32_t* __restrict__ radii,
    float* __restrict__ means2D, float* __restrict__ depths, float* __restrict__ cov3Ds, float* __restrict__ rgb,
    float* __restrict__ conic_opacity, int gx, int gy, u32* __restrict__ tiles_touched, float4* __restrict__ emit_rec,
    u32* __restrict__ sort_key, u32* __restrict__ sort_val, int32_t* __restrict__ n_touched, int prefiltered,
    int ellipse, int act, u32& count_out, PendingRows& pend) {
  count_out = 0;
  pend.nrows = 0;
  n_touched[idx] = 0;  // the forward composite counts into it with integer atomics
  radii[idx] = 0;
  // (the emission record of a Gaussian without instances is never read — the emission goes by tiles_touched — so culled
  //  Gaussians leave theirs untouched)

  // in_frustum, CR/auxiliary.h:139-164
  const f3 p_orig = {orig_points[3 * (size_t)idx], orig_points[3 * (size_t)idx + 1], orig_points[3 * (size_t)idx + 2]};
  const f3 p_view = transformPoint4x3(p_orig, viewmatrix);
  // The depth-sort key is defined for EVERY Gaussian (round 6): the view-space depth's bits whenever z > 0.2 — also for a
  // Gaussian the tests below cull —, and a monotone function of z below bits(0.2) behind the near plane.  Only Gaussians that
  // emit instances matter to the lists, and those have z > 0.2 and the reference's key (CR/forward.cu:247, duplicateWithKeys);
  // the others may stand anywhere.  Ordering them by depth as well makes the order a function of the pose alone, not of what
  // is visible, so that the order of one frame is nearly the order of the next (k_order_carry.hip: a Gaussian that enters the
  // frustum must not jump from the tail of the order into its middle).  Keys are never 0 (the repair's padding lies below them).
  sort_key[idx] = (p_view.z <= 0.2f) ? 0x3E4CCCCCu - min(f2bits(0.2f - p_view.z) >> 1, 0x3E4CCCCBu) : f2bits(p_view.z);
  if (p_view.z <= 0.2f) {
    if (prefiltered) {
      printf("Point is filtered although prefiltered is set. This shouldn't happen!");
      __builtin_trap();
    }
    return 0;
  }
  const f4 p_hom = transformPoint4x4(p_orig, projmatrix);
  const float p_w = 1.0f / (p_hom.w + 0.0000001f);
  const f3 p_proj = {p_hom.x * p_w, p_hom.y * p_w, p_hom.z * p_w};

  float cov3D[6];
  if (cov3D_precomp != nullptr) {
#pragma unroll
    for (int i = 0; i < 6; ++i) cov3D[i] = cov3D_precomp[6 * (size_t)idx + i];
  } else {
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
    cov3d_from_scale_rot(sc3, scale_modifier, q4, cov3D);
#pragma unroll
    for (int i = 0; i < 6; ++i) cov3Ds[6 * (size_t)idx + i] = cov3D[i];
  }

  Cov2D ci;
  cov2d_common(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, viewmatrix, ci);
  const float cx = ci.cov.c[0][0] + 0.3f, cy = ci.cov.c[0][1], cz = ci.cov.c[1][1] + 0.3f;
  const float det = (cx * cz - cy * cy);
  if (det == 0.0f) return 0;
  const float det_inv = 1.f / det;
  const f3 conic = {cz * det_inv, -cy * det_inv, cx * det_inv};

  const float mid = 0.5f * (cx + cz);
  const float lambda1 = mid + sqrtf(fmaxf_ref(0.1f, mid * mid - det));
  const float lambda2 = mid - sqrtf(fmaxf_ref(0.1f, mid * mid - det));
  const float my_radius = ceilf(3.f * sqrtf(fmaxf_ref(lambda1, lambda2)));
  const float pix_x = ndc2Pix(p_proj.x, W), pix_y = ndc2Pix(p_proj.y, H);
  const int irad = f2i_sat(my_radius);
  const Rect rc = get_rect<TILE>(pix_x, pix_y, irad, gx, gy);
  const u32 area = (u32)((rc.y1 - rc.y0) * (rc.x1 - rc.x0));
  if (area == 0) {
    // (OLSR_FLAG_SIGNED_EMPTY_RADII, carried in the upper half of `act`: the radius of a square that covers no tile)
    if (act & (OLSR_FLAG_SIGNED_EMPTY_RADII << 16)) radii[idx] = -irad;
    return 0;
  }

  if (colors_precomp == nullptr) {
    const f3 c = color_from_sh(idx, D, M, p_orig, cam_pos, shs, clamped);
    rgb[3 * (size_t)idx + 0] = c.x;
    rgb[3 * (size_t)idx + 1] = c.y;
    rgb[3 * (size_t)idx + 2] = c.z;
  }
  const float opacity = (act & OLSR_ACT_OPACITY_SIGMOID) ? act_sigmoid(opacities[idx]) : opacities[idx];
  depths[idx] = p_view.z;
  radii[idx] = irad;
  means2D[2 * (size_t)idx + 0] = pix_x;
  means2D[2 * (size_t)idx + 1] = pix_y;
  float4 co = make_float4(conic.x, conic.y, conic.z, opacity);
  reinterpret_cast<float4*>(conic_opacity)[idx] = co;

  u32 count = area;
  float t2 = 0.f;
  int ya = rc.y0, yb = rc.y1;  // tile rows that hold instances
  if (ellipse) {
    // exact binning: count, per tile row of the rect, the tile columns the alpha-floor ellipse reaches
    t2 = cull_threshold(conic.x, conic.y, conic.z, opacity, irad, TILE);
    count = 0;
    if (t2 >= 0.0f) {
      const CullEllipse e = cull_setup(pix_x, pix_y, conic.x, conic.y, conic.z, t2, irad);
      cull_rows<TILE>(e, rc.y0, rc.y1, ya, yb);
      // the row spans are evaluated by the whole wave, one row per lane (count_pending_rows): the caller adds them up
      pend.e = e;
      pend.ya = ya;
      pend.nrows = yb - ya;
      pend.x0 = rc.x0;
      pend.x1 = rc.x1;
      pend.depth = p_view.z;
      // the emission re-evaluates the row spans from this record without the radius: a radius beyond cull_setup's
      // range (full spans) is handed on as a threshold beyond its range (full spans as well)
      if (!(irad < (1 << 20))) t2 = 2e6f;
    }
  }
  count_out = count;
  // {mean x, mean y, conic a, conic b}, {conic c, cull t2, first row | rows << 16, first column | end column << 16}
  emit_rec[2 * (size_t)idx] = make_float4(pix_x, pix_y, conic.x, conic.y);
  emit_rec[2 * (size_t)idx + 1] = make_float4(conic.z, t2, __uint_as_float((u32)ya | ((u32)(yb - ya) << 16)),
                                              __uint_as_float((u32)rc.x0 | ((u32)rc.x1 << 16)));
  return area;
}

template <int TILE>
__global__ __launch_bounds__(256) void preprocess_kernel(
    int P, int D, int M, const float* __restrict__ orig_points, const float* __restrict__ scales, float scale_modifier,
    const float* __restrict__ rotations, const float* __restrict__ opacities, const float* __restrict__ shs,
    uint8_t* __restrict__ clamped, const float* __restrict__ cov3D_precomp, const float* __restrict__ colors_precomp,
    const float* __restrict__ viewmatrix, const float* __restrict__ projmatrix, const float* __restrict__ cam_pos, int W,
    int H, float tan_fovx, float tan_fovy, float focal_x, float focal_y, int32_t* __restrict__ radii,
    float* __restrict__ means2D, float* __restrict__ depths, float* __restrict__ cov3Ds, float* __restrict__ rgb,
    float* __restrict__ conic_opacity, int gx, int gy, u32* __restrict__ tiles_touched, float4* __restrict__ emit_rec,
    u32* __restrict__ sort_key, u32* __restrict__ sort_val, int32_t* __restrict__ n_touched, int prefiltered,
    int ellipse, int act, u32* __restrict__ rect_partials, u32* __restrict__ count_partials,
    uint4* __restrict__ sync_words, int sync_quads, const float* __restrict__ depth_cut, uint8_t* __restrict__ blended,
    u32* __restrict__ vis_partials) {
  __shared__ u32 s_area[4], s_cnt[4], s_vis[4];
  __shared__ PreWaveLds s_rows[4];
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  // the words the frame's fused kernels synchronise through (tickets, digit histograms, published block counts):
  // zeroed here, at the head of the frame, by as many threads as there are 16-byte pieces
  if (idx < sync_quads) sync_words[idx] = make_uint4(0u, 0u, 0u, 0u);
  if (idx < P) blended[idx] = 0;  // set by the forward composite for a Gaussian that blends anywhere (olsr_state.h)
  u32 area = 0, count = 0;
  PendingRows pend;
  pend.nrows = 0;
  if (idx < P)
    area = preprocess_one<TILE>(idx, D, M, orig_points, scales, scale_modifier, rotations, opacities, shs, clamped,
                                cov3D_precomp, colors_precomp, viewmatrix, projmatrix, cam_pos, W, H, tan_fovx,
                                tan_fovy, focal_x, focal_y, radii, means2D, depths, cov3Ds, rgb, conic_opacity, gx, gy,
                                tiles_touched, emit_rec, sort_key, sort_val, n_touched, prefiltered, ellipse, act,
                                count, pend);
  if (ellipse) count += count_pending_rows<TILE>(pend, s_rows[threadIdx.x >> 6], W, H, gx, depth_cut);
  if (idx < P) tiles_touched[idx] = count;
  const u32 nvis = (u32)__popcll(ballot(idx < P && count > 0u));  // Gaussians of this wave that emit instances
  // instances of the reference's rect binning (its num_rendered) and instances this frame emits: one partial per
  // block each, summed by the next kernel (7.8 k same-address atomics would cost more than the whole kernel)
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    area += __shfl_xor(area, m);
    count += __shfl_xor(count, m);
  }
  if ((threadIdx.x & 63) == 0) {
    s_area[threadIdx.x >> 6] = area;
    s_cnt[threadIdx.x >> 6] = count;
    s_vis[threadIdx.x >> 6] = nvis;
  }
  __syncthreads();
  if (threadIdx.x == 0 && (int)(blockIdx.x * blockDim.x) < P) {  // (blocks beyond P only zero sync words)
    rect_partials[blockIdx.x] = s_area[0] + s_area[1] + s_area[2] + s_area[3];
    count_partials[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    vis_partials[blockIdx.x] = s_vis[0] + s_vis[1] + s_vis[2] + s_vis[3];
  }
}

void launch_preprocess(const olsr_scene& s, const FrameDims& d, const GeometryState& g, int32_t* radii,
                       int32_t* n_touched, hipStream_t st) {
  if (s.P <= 0) return;
  const int sync_quads = (int)(g.sync_count / 4);
  const int nb = (max(s.P, sync_quads) + 255) / 256;  // (the sync words are far fewer than the Gaussians)
#define OLSR_PRE_ARGS                                                                                                 \
  s.P, s.D, s.M, s.means3D, s.scales, s.scale_modifier, s.rotations, s.opacities, s.shs, g.clamped, s.cov3D_precomp,  \
      s.colors_precomp, s.viewmatrix, s.projmatrix, s.cam_pos, d.W, d.H, s.tan_fovx, s.tan_fovy, d.focal_x,           \
      d.focal_y, radii, g.means2D, g.depths, g.cov3D, g.rgb, g.conic_opacity, d.gx, d.gy, g.tiles_touched,            \
      g.emit_rec, g.key_a, g.val_a, n_touched, s.prefiltered, (int)(s.binning == OLSR_BINNING_ELLIPSE),      \
      s.activations | (s.flags << 16), g.part_rect, g.part_count, reinterpret_cast<uint4*>(g.sync_words), sync_quads,      \
      ((s.binning == OLSR_BINNING_ELLIPSE) ? s.tile_depth_cut : nullptr), g.blended, g.part_vis
  if (d.tile == 15)
    preprocess_kernel<15><<<nb, 256, 0, st>>>(OLSR_PRE_ARGS);
  else
    preprocess_kernel<16><<<nb, 256, 0, st>>>(OLSR_PRE_ARGS);
#undef OLSR_PRE_ARGS
}

__global__ __launch_bounds__(256) void mark_visible_kernel(int P, const float* __restrict__ pts,
                                                           const float* __restrict__ view,
                                                           uint8_t* __restrict__ present) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= P) return;
  const f3 p = {pts[3 * (size_t)idx], pts[3 * (size_t)idx + 1], pts[3 * (size_t)idx + 2]};
  const f3 pv = transformPoint4x3(p, view);
  present[idx] = !(pv.z <= 0.2f);
}

void launch_mark_visible(int P, const float* means3D, const float* view, uint8_t* present, hipStream_t st) {
  if (P <= 0) return;
  mark_visible_kernel<<<(P + 255) / 256, 256, 0, st>>>(P, means3D, view, present);
}

}  // namespace olsr
