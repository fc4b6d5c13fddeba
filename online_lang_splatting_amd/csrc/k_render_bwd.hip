// k_render_bwd.hip — backward of the alpha compositing: per-pixel gradients -> per-(tile, splat)
// partial-gradient rows.
//
// Replaces renderCUDA / language_render_cuda (backward) with render_cuda_reduce_sum
// (CR/backward.cu:706-930, 932-1201, 684-702).
//
// The reference spends, per splat per tile, 3 block barriers + a shared-memory atomic per
// skipping thread + an 8-barrier shared-memory tree over 225 threads + up to 25 global float
// atomics.  MI355X mapping (same fold as the forward: one wave64 per tile, 4 pixel slots per
// lane, rank = slot*64 + lane):
//   * no barriers: the tile-wide "does any pixel use this splat" predicate was recorded by the
//     forward composite (flags[]), so unused instances are skipped by a uniform branch;
//   * each lane first adds its (up to) 4 pixels in registers, then ONE multi-value wave
//     butterfly (N values in ~N+log2 shuffle-adds instead of 6N) leaves value k in lane
//     k*(64/N); the row is written to HBM with a single coalesced 64..256-byte store;
//   * no float atomics: rows are indexed by the instance's emission position, so the
//     per-Gaussian reduction (k_preprocess_bwd.hip) reads a contiguous run of rows and the
//     gradients are bit-reproducible from run to run.
//
// MODE = OLSR_BWD_REFERENCE reproduces the shipped reference bit-compatibly in structure:
//   - only the ranks that survive the 225-lane integer-halving tree contribute to
//     mean2D / conic / opacity / colour / depth (ref_survives<15>),
//   - language gradients are taken from rank 0 only,
//   - the language recursion (accum_rec_F, last_language_feature) runs for every pixel of the
//     tile whenever the tile as a whole does not skip the splat.
// MODE = OLSR_BWD_EXACT sums all pixels and guards the language recursion like colour.
#include "olsr_device.h"
#include "olsr_kernels.h"

namespace olsr {

// Which thread ranks reach element 0 of render_cuda_reduce_sum's tree when g.size() == TILE*TILE
// (steps size/2, /2, ... with integer division).  256 lanes: all of them.  225 lanes: the tree
// 112,56,28,14,7,3,1 drops rank 224 and every rank whose residue mod 7 is 2, 5 or 6.
template <int TILE>
__device__ __forceinline__ bool ref_survives(int rank) {
  if constexpr (TILE == 16) {
    return true;
  } else {
    static_assert(TILE == 15, "closed form derived for 15x15 and 16x16 tiles only");
    const int m = rank % 7;
    return rank < 224 && (m == 0 || m == 1 || m == 3 || m == 4);
  }
}

constexpr int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

// Sum N per-lane values across the wave.  On return lane l holds (in v[0]) the wave total of
// value index l / (64 / N); all 64/N lanes of a group hold the same total.
template <int N>
__device__ __forceinline__ void wave_reduce_multi(float (&v)[N], int lane) {
  static_assert(N >= 1 && N <= 64 && (N & (N - 1)) == 0, "N must be a power of two <= 64");
  int m = 32;
#pragma unroll
  for (int h = N / 2; h >= 1; h >>= 1, m >>= 1) {
    const bool upper = (lane & m) != 0;
#pragma unroll
    for (int i = 0; i < h; ++i) {
      const float send = upper ? v[i] : v[i + h];
      const float keep = upper ? v[i + h] : v[i];
      v[i] = keep + __shfl_xor(send, m);
    }
  }
#pragma unroll
  for (; m >= 1; m >>= 1) v[0] += __shfl_xor(v[0], m);
}

template <int TILE, int F, int MODE>
__global__ __launch_bounds__(64) void render_bwd_kernel(
    const u32* __restrict__ ranges, const u32* __restrict__ point_list, const u32* __restrict__ src,
    const uint8_t* __restrict__ flags, int W, int H, int gx, int ntiles, const float* __restrict__ bg,
    const float* __restrict__ means2D, const float* __restrict__ conic_opacity, const float* __restrict__ colors,
    const float* __restrict__ lang, const float* __restrict__ depths, const float* __restrict__ final_Ts,
    const u32* __restrict__ n_contrib, const float* __restrict__ dL_dpixels, const float* __restrict__ dL_dpixels_lang,
    const float* __restrict__ dL_dpixels_depth, float* __restrict__ rows) {
  constexpr int BS = TILE * TILE;
  constexpr int SLOTS = (BS + 63) / 64;
  constexpr int FR = feat_row(F);
  constexpr int ROW = grad_row(F);
  constexpr bool REF = (MODE == OLSR_BWD_REFERENCE);
  constexpr int NV = REF ? 10 : 10 + F;  // values that go through the wave reduction
  constexpr int NP = next_pow2(NV);
  constexpr int FX = (F > 0) ? F : 1;

  __shared__ float2 s_xy[64];
  __shared__ float4 s_co[64];
  __shared__ __attribute__((aligned(16))) float s_feat[64 * FR];
  __shared__ u32 s_src[64];
  __shared__ u32 s_flag[64];
  __shared__ __attribute__((aligned(16))) float s_row[ROW];

  const int tile_id = xcd_remap((int)blockIdx.x, ntiles);
  const int lane = threadIdx.x;
  const int bx = tile_id % gx, by = tile_id / gx;
  const u32 r0 = ranges[2 * tile_id], r1 = ranges[2 * tile_id + 1];
  if (r1 <= r0) return;
  const size_t HW = (size_t)H * W;

  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
  const bool has_bg = (bg0 != 0.0f) || (bg1 != 0.0f) || (bg2 != 0.0f);
  const float ddelx_dx = 0.5f * W;
  const float ddely_dy = 0.5f * H;

  bool inside[SLOTS], surv[SLOTS];
  float pixfx[SLOTS], pixfy[SLOTS], T_final[SLOTS], T[SLOTS], last_alpha[SLOTS], bg_dot[SLOTS];
  int last_contributor[SLOTS];
  float accum_c[SLOTS][3], last_c[SLOTS][3], dLc[SLOTS][3];
  float accum_d[SLOTS], last_d[SLOTS], dLd[SLOTS];
  float accum_f[SLOTS][FX], dLf[SLOTS][FX];
  float last_f[REF ? 1 : SLOTS][FX];  // REF: one wave-uniform copy (the recursion is unguarded)
  int kmax = 0;
#pragma unroll
  for (int q = 0; q < SLOTS; ++q) {
    const int rank = q * 64 + lane;
    const int px = bx * TILE + rank % TILE, py = by * TILE + rank / TILE;
    inside[q] = (rank < BS) && (px < W) && (py < H);
    surv[q] = REF ? ref_survives<TILE>(rank) : true;
    pixfx[q] = (float)px;
    pixfy[q] = (float)py;
    const size_t pix = (size_t)W * py + px;
    T_final[q] = inside[q] ? final_Ts[pix] : 0.f;
    T[q] = T_final[q];
    last_contributor[q] = inside[q] ? (int)n_contrib[pix] : 0;
    kmax = max(kmax, last_contributor[q]);
    last_alpha[q] = 0.f;
    float bd = 0.f;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      accum_c[q][ch] = 0.f;
      last_c[q][ch] = 0.f;
      dLc[q][ch] = inside[q] ? dL_dpixels[ch * HW + pix] : 0.f;
    }
    bd += bg0 * dLc[q][0];
    bd += bg1 * dLc[q][1];
    bd += bg2 * dLc[q][2];
    bg_dot[q] = bd;
    accum_d[q] = 0.f;
    last_d[q] = 0.f;
    dLd[q] = inside[q] ? dL_dpixels_depth[pix] : 0.f;
#pragma unroll
    for (int ch = 0; ch < FX; ++ch) {
      accum_f[q][ch] = 0.f;
      dLf[q][ch] = (F > 0 && inside[q]) ? dL_dpixels_lang[ch * HW + pix] : 0.f;
    }
  }
#pragma unroll
  for (int q = 0; q < (REF ? 1 : SLOTS); ++q)
#pragma unroll
    for (int ch = 0; ch < FX; ++ch) last_f[q][ch] = 0.f;
  // entries at list positions >= max(last_contributor) are skipped by every pixel of the tile
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) kmax = max(kmax, __shfl_xor(kmax, m));
  if (lane < ROW) s_row[lane] = 0.f;
  if (ROW > 64 && lane + 64 < ROW) s_row[lane + 64] = 0.f;

  for (int kstart = kmax - 1; kstart >= 0; kstart -= 64) {
    const int cnt = min(64, kstart + 1);
    __syncthreads();
    if (lane < cnt) {
      const u32 sp = r0 + (u32)(kstart - lane);
      const u32 gid = point_list[sp];
      const u32 u = src[sp];
      s_src[lane] = u;
      s_flag[lane] = flags[u];
      s_xy[lane] = reinterpret_cast<const float2*>(means2D)[gid];
      s_co[lane] = reinterpret_cast<const float4*>(conic_opacity)[gid];
      float* fr = &s_feat[lane * FR];
      fr[0] = colors[3 * (size_t)gid + 0];
      fr[1] = colors[3 * (size_t)gid + 1];
      fr[2] = colors[3 * (size_t)gid + 2];
      fr[3] = depths[gid];
#pragma unroll
      for (int ch = 0; ch < F; ++ch) fr[4 + ch] = lang[(size_t)gid * F + ch];
    }
    __syncthreads();

    for (int i = 0; i < cnt; ++i) {
      if (s_flag[i] == 0) continue;  // whole tile skips this splat: no state changes (CR/backward.cu:1091-1093)
      const int k = kstart - i;      // == `contributor` after its decrement (CR/backward.cu:999,1073)
      const float2 xy = s_xy[i];
      const float4 co = s_co[i];
      const float* fr = &s_feat[i * FR];
      float sum[NP];
#pragma unroll
      for (int v = 0; v < NP; ++v) sum[v] = 0.f;
      float lang0[FX];  // REF: rank 0's language partials (slot 0 of lane 0)
#pragma unroll
      for (int ch = 0; ch < FX; ++ch) lang0[ch] = 0.f;

#pragma unroll
      for (int q = 0; q < SLOTS; ++q) {
        bool skip = !inside[q];
        skip |= (k >= last_contributor[q]);
        const float dx = xy.x - pixfx[q], dy = xy.y - pixfy[q];
        const float power = -0.5f * (co.x * dx * dx + co.z * dy * dy) - co.y * dx * dy;
        skip |= power > 0.0f;
        const float G = pinned_expf(power);
        const float alpha = fminf_ref(0.99f, co.w * G);
        skip |= alpha < 1.0f / 255.0f;
        const bool slot_live = wave_any(!skip);
        if (!REF && !slot_live) continue;  // nothing in this slot changes

        if (REF && !slot_live) {
          // every pixel of this slot skips, but the tile does not: only the unguarded language
          // recursion advances (CR/backward.cu:1127-1139)
          if constexpr (F > 0) {
#pragma unroll
            for (int ch = 0; ch < F; ++ch)
              accum_f[q][ch] = last_alpha[q] * last_f[0][ch] + (1.f - last_alpha[q]) * accum_f[q][ch];
          }
          continue;
        }

        T[q] = skip ? T[q] : T[q] / (1.f - alpha);
        const float dchannel_dcolor = alpha * T[q];
        const bool count = !skip && surv[q];
        float dL_dalpha = 0.0f;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
          const float c = fr[ch];
          accum_c[q][ch] = skip ? accum_c[q][ch] : last_alpha[q] * last_c[q][ch] + (1.f - last_alpha[q]) * accum_c[q][ch];
          last_c[q][ch] = skip ? last_c[q][ch] : c;
          const float dL_dchannel = dLc[q][ch];
          dL_dalpha += (c - accum_c[q][ch]) * dL_dchannel;
          sum[6 + ch] += count ? dchannel_dcolor * dL_dchannel : 0.0f;
        }
        const float depth = fr[3];
        accum_d[q] = skip ? accum_d[q] : last_alpha[q] * last_d[q] + (1.f - last_alpha[q]) * accum_d[q];
        last_d[q] = skip ? last_d[q] : depth;
        dL_dalpha += (depth - accum_d[q]) * dLd[q];
        sum[9] += count ? dchannel_dcolor * dLd[q] : 0.f;
        if constexpr (F > 0) {
#pragma unroll
          for (int ch = 0; ch < F; ++ch) {
            const float f = fr[4 + ch];
            if constexpr (REF) {
              accum_f[q][ch] = last_alpha[q] * last_f[0][ch] + (1.f - last_alpha[q]) * accum_f[q][ch];
            } else {
              accum_f[q][ch] =
                  skip ? accum_f[q][ch] : last_alpha[q] * last_f[q][ch] + (1.f - last_alpha[q]) * accum_f[q][ch];
              last_f[q][ch] = skip ? last_f[q][ch] : f;
            }
            const float dL_dchannel_F = dLf[q][ch];
            dL_dalpha += (f - accum_f[q][ch]) * dL_dchannel_F;
            const float part = skip ? 0.0f : dchannel_dcolor * dL_dchannel_F;
            if constexpr (REF) {
              if (q == 0) lang0[ch] = part;
            } else {
              sum[10 + ch] += part;
            }
          }
        }
        dL_dalpha *= T[q];
        last_alpha[q] = skip ? last_alpha[q] : alpha;
        if (has_bg) dL_dalpha += (-T_final[q] / (1.f - alpha)) * bg_dot[q];

        const float dL_dG = co.w * dL_dalpha;
        const float gdx = G * dx;
        const float gdy = G * dy;
        const float dG_ddelx = -gdx * co.x - gdy * co.y;
        const float dG_ddely = -gdy * co.z - gdx * co.y;
        sum[0] += count ? dL_dG * dG_ddelx * ddelx_dx : 0.f;
        sum[1] += count ? dL_dG * dG_ddely * ddely_dy : 0.f;
        sum[2] += count ? -0.5f * gdx * dx * dL_dG : 0.f;
        sum[3] += count ? -0.5f * gdx * dy * dL_dG : 0.f;
        sum[4] += count ? -0.5f * gdy * dy * dL_dG : 0.f;
        sum[5] += count ? G * dL_dalpha : 0.f;
      }
      if constexpr (REF && F > 0) {
        // last_language_feature = f for every thread of a non-skipping tile (CR/backward.cu:1133)
#pragma unroll
        for (int ch = 0; ch < F; ++ch) last_f[0][ch] = fr[4 + ch];
      }

      wave_reduce_multi<NP>(sum, lane);
      constexpr int G_LANES = 64 / NP;  // lanes per value group after the butterfly
      const int vi = lane / G_LANES;
      if ((lane % G_LANES) == 0 && vi < NV) s_row[vi] = sum[0];
      if constexpr (REF && F > 0) {
        if (lane == 0) {
#pragma unroll
          for (int ch = 0; ch < F; ++ch) s_row[10 + ch] = lang0[ch];
        }
      }
      __syncthreads();
      float* dst = rows + (size_t)s_src[i] * ROW;
      if (lane < ROW) dst[lane] = s_row[lane];
      if (ROW > 64 && lane + 64 < ROW) dst[lane + 64] = s_row[lane + 64];
      __syncthreads();
    }
  }
}

template <int TILE, int F, int MODE>
static void launch_bwd_t(const olsr_scene& s, const FrameDims& d, const GeometryState& g, const BinningState& b,
                         const ImageState& im, const float* dc, const float* dl, const float* dd, hipStream_t st) {
  const float* colors = s.colors_precomp ? s.colors_precomp : g.rgb;
  render_bwd_kernel<TILE, F, MODE><<<d.ntiles, 64, 0, st>>>(im.ranges, b.point_list, b.src, b.flags, d.W, d.H, d.gx,
                                                            d.ntiles, s.background, g.means2D, g.conic_opacity, colors,
                                                            s.language_precomp, g.depths, im.final_T, im.n_contrib, dc,
                                                            dl, dd, b.rows);
}

template <int TILE, int MODE>
static void launch_bwd_f(const olsr_scene& s, const FrameDims& d, const GeometryState& g, const BinningState& b,
                         const ImageState& im, const float* dc, const float* dl, const float* dd, hipStream_t st) {
  switch (s.F) {
    case 0: launch_bwd_t<TILE, 0, MODE>(s, d, g, b, im, dc, dl, dd, st); break;
    case 3: launch_bwd_t<TILE, 3, MODE>(s, d, g, b, im, dc, dl, dd, st); break;
    case 15: launch_bwd_t<TILE, 15, MODE>(s, d, g, b, im, dc, dl, dd, st); break;
    case 16: launch_bwd_t<TILE, 16, MODE>(s, d, g, b, im, dc, dl, dd, st); break;
    case 32: launch_bwd_t<TILE, 32, MODE>(s, d, g, b, im, dc, dl, dd, st); break;
    default: break;
  }
}

#ifndef OLSR_BWD_TU_MODE
#error "compile with -DOLSR_BWD_TU_MODE=0 (reference) or 1 (exact)"
#endif

#if OLSR_BWD_TU_MODE == 0
void launch_render_backward_reference(const olsr_scene& s, const FrameDims& d, const GeometryState& g,
                                      const BinningState& b, const ImageState& im, const float* dc, const float* dl,
                                      const float* dd, hipStream_t st) {
  if (d.tile == 15)
    launch_bwd_f<15, OLSR_BWD_REFERENCE>(s, d, g, b, im, dc, dl, dd, st);
  else
    launch_bwd_f<16, OLSR_BWD_REFERENCE>(s, d, g, b, im, dc, dl, dd, st);
}
#else
void launch_render_backward_exact(const olsr_scene& s, const FrameDims& d, const GeometryState& g,
                                  const BinningState& b, const ImageState& im, const float* dc, const float* dl,
                                  const float* dd, hipStream_t st) {
  if (d.tile == 15)
    launch_bwd_f<15, OLSR_BWD_EXACT>(s, d, g, b, im, dc, dl, dd, st);
  else
    launch_bwd_f<16, OLSR_BWD_EXACT>(s, d, g, b, im, dc, dl, dd, st);
}
#endif

}  // namespace olsr
