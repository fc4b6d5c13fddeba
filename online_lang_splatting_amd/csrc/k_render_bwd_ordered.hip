// k_render_bwd_ordered.hip — the composite backward in the REFERENCE'S OWN ASSOCIATION.  A test instrument
// (olsr_debug_backward_ordered, include/olsr.h), never on a product path: slow on purpose.
//
// The product's composite backward (k_render_bwd.hip) is a tolerance-level restatement of CR/backward.cu:932-1201 / 706-930:
// running form of the behind-colour recursion, dot form of the language recursion, a Newton-refined reciprocal, a wave-level
// reduction instead of the 225-lane tree, rows summed per (instance, slot).  What it is held against is the CPU oracle, which
// follows the reference's source expression by expression (oracle/oracle.cpp: render_backward).  This file is the same
// restatement ON THE GPU, bit for bit:
//   * one workgroup per tile, thread rank = ty * TILE + tx as cg::thread_rank() in the reference (not the product's quadrants);
//   * the tile's list walked back to front, EVERY entry, with the reference's per-lane state (T, accum_rec, last_color,
//     last_alpha, ... CR/backward.cu:1062-1139) and its expressions in source order — T / (1 - alpha), the three-term
//     recursion, dL_dG * dG_ddelx * ddelx_dx — in a translation unit built with -ffp-contract=off like every other;
//   * the tile-wide skip (skip_counter == BLOCK_SIZE, :1087-1093) as a block vote;
//   * OLSR_BWD_REFERENCE: render_cuda_reduce_sum (:684-702) as written — for (i = BS / 2; i > 0; i /= 2) a[lane] += a[lane + i]
//     through LDS, a barrier per step, which keeps 128 of the 225 ranks of a 15 x 15 tile; the language row from rank 0 (:1137,
//     :1194-1197), its recursion unguarded (:1127-1139).  OLSR_BWD_EXACT: every lane, summed in lane order in double like the
//     oracle's exact mode, the recursion guarded;
//   * ONE row per instance the tile does not skip, and a second kernel that adds a Gaussian's rows one after the other in
//     sorted-list order (its instances in emission order = ascending tile), the order in which the oracle replays the
//     reference's atomicAdd of thread 0 (:1176-1198).
// Its outputs — dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolors, dL_ddepths, dL_dlanguage — EQUAL the oracle's
// (tests/test_gpu_bwd_ordered.py), at sizes the CPU oracle needs minutes for; the product's fast kernel is then compared with
// THIS kernel on the GPU, which is what replaces the suite's end-to-end bound.
#include "olsr_device.h"
#include "olsr_kernels.h"

namespace olsr {

// ABS = true: not the gradient but its CONDITION — every product replaced by the product of magnitudes and every difference by
// the sum of magnitudes (|c| + |accum_rec| for c - accum_rec, sum |.| for dL_dalpha, |.| for every partial), pushed through the
// same tree and the same per-Gaussian sums.  The result A bounds, to first order, what ANY re-association and any few-ulp
// variation of the value path can change: |g' - g| <= K eps A with K of the order of the operations a term went through (the
// transmittance recursion T / (1 - alpha) makes that grow with the list depth).  The tests hold the fast kernel to that
// bound element by element, which is the statement "it differs from the reference's association by rounding only".
template <int TILE, int F, int MODE, bool ABS>
__global__ __launch_bounds__(256) void render_bwd_ordered_kernel(
    const u32* __restrict__ ranges, const u32* __restrict__ inst_gid, const u32* __restrict__ src, int W, int H, int gx,
    const float* __restrict__ means2D, const float* __restrict__ conic_opacity, const float* __restrict__ depths,
    const float* __restrict__ colors, const float* __restrict__ lang, const float* __restrict__ bg,
    const float* __restrict__ final_T, const u32* __restrict__ n_contrib, const float* __restrict__ dL_dpixels,
    const float* __restrict__ dL_dpixels_lang, const float* __restrict__ dL_dpixels_depth, float* __restrict__ rows,
    uint8_t* __restrict__ used, const int32_t* __restrict__ counters) {
  constexpr int BS = TILE * TILE;
  constexpr int NV = 10 + F;
  constexpr int NVP = grad_row(F);  // row stride
  constexpr int FA = F > 0 ? F : 1;
  __shared__ float part[NV][BS];
  if (frame_unusable(counters) || counters[2] != 0) return;  // (no lists to walk: olsr_state.h)
  const int t = (int)blockIdx.x;
  const int rank = (int)threadIdx.x;
  const bool lane_ok = rank < BS;  // (blockDim = 256: 31 idle threads on a 15 x 15 tile)
  const int bx = t % gx, by = t / gx;
  const u32 r0 = ranges[2 * t], r1 = ranges[2 * t + 1];
  if (r1 <= r0) return;
  const size_t HW = (size_t)H * W;
  const int tx = rank % TILE, ty = rank / TILE;
  const int px = bx * TILE + tx, py = by * TILE + ty;
  const bool inside = lane_ok && px < W && py < H;
  const float pixfx = (float)px, pixfy = (float)py;
  const size_t pix_id = (size_t)W * py + px;
  const float T_final = inside ? final_T[pix_id] : 0.f;
  float T = T_final;
  const int last_contributor = inside ? (int)n_contrib[pix_id] : 0;
  float accum_rec[3] = {0.f, 0.f, 0.f}, last_color[3] = {0.f, 0.f, 0.f}, dL_dpixel[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) dL_dpixel[i] = inside ? dL_dpixels[i * HW + pix_id] : 0.f;
  float accum_rec_depth = 0.f, last_depth = 0.f;
  const float dL_dpixel_depth = (inside && dL_dpixels_depth != nullptr) ? dL_dpixels_depth[pix_id] : 0.f;
  float last_alpha = 0.f;
  float accF[FA], lastF[FA], dLF[FA];
#pragma unroll
  for (int i = 0; i < FA; ++i) {
    accF[i] = 0.f;
    lastF[i] = 0.f;
    dLF[i] = (F > 0 && inside && dL_dpixels_lang != nullptr) ? dL_dpixels_lang[i * HW + pix_id] : 0.f;
  }
  const float ddelx_dx = 0.5f * W;
  const float ddely_dy = 0.5f * H;
  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];

  const u32 toDo = r1 - r0;
  for (u32 k = toDo; k-- > 0;) {
    const u32 sorted_pos = r0 + k;
    const u32 u = src[sorted_pos];  // emission index of the instance
    const u32 gid = inst_gid[u];
    const float xyx = means2D[2 * (size_t)gid], xyy = means2D[2 * (size_t)gid + 1];
    const float co0 = conic_opacity[4 * (size_t)gid], co1 = conic_opacity[4 * (size_t)gid + 1],
                co2 = conic_opacity[4 * (size_t)gid + 2], co3 = conic_opacity[4 * (size_t)gid + 3];
    bool skip = !inside;  // done = !inside, never changes (CR/backward.cu:972)
    skip |= ((int)k >= last_contributor);
    const float dx = xyx - pixfx, dy = xyy - pixfy;
    const float power = -0.5f * (co0 * dx * dx + co2 * dy * dy) - co1 * dx * dy;
    skip |= power > 0.0f;
    const float G = pinned_expf(power);
    const float alpha = fminf_ref(0.99f, co3 * G);
    skip |= alpha < 1.0f / 255.0f;
    // (idle threads of the block vote "skip" too: the count is compared with the 256 of the launch)
    if (__syncthreads_count(skip ? 1 : 0) == (int)blockDim.x) continue;  // :1091-1093
    const float depth = depths[gid];
    if (lane_ok) {
      T = skip ? T : T / (1.f - alpha);
      const float dchannel_dcolor = alpha * T;
      float dL_dalpha = 0.0f;
#pragma unroll
      for (int ch = 0; ch < 3; ch++) {
        const float c = colors[(size_t)gid * 3 + ch];
        accum_rec[ch] = skip ? accum_rec[ch] : last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
        last_color[ch] = skip ? last_color[ch] : c;
        const float dL_dchannel = dL_dpixel[ch];
        if constexpr (ABS) dL_dalpha += (fabsf(c) + fabsf(accum_rec[ch])) * fabsf(dL_dchannel);
        else dL_dalpha += (c - accum_rec[ch]) * dL_dchannel;
        part[6 + ch][rank] = skip ? 0.0f : (ABS ? fabsf(dchannel_dcolor * dL_dchannel) : dchannel_dcolor * dL_dchannel);
      }
      accum_rec_depth = skip ? accum_rec_depth : last_alpha * last_depth + (1.f - last_alpha) * accum_rec_depth;
      last_depth = skip ? last_depth : depth;
      if constexpr (ABS) dL_dalpha += (fabsf(depth) + fabsf(accum_rec_depth)) * fabsf(dL_dpixel_depth);
      else dL_dalpha += (depth - accum_rec_depth) * dL_dpixel_depth;
      part[9][rank] = skip ? 0.f : (ABS ? fabsf(dchannel_dcolor * dL_dpixel_depth) : dchannel_dcolor * dL_dpixel_depth);
#pragma unroll
      for (int ch = 0; ch < F; ch++) {
        const float f = lang[(size_t)gid * F + ch];
        if constexpr (MODE == OLSR_BWD_REFERENCE) {  // unguarded, :1132-1133
          accF[ch] = last_alpha * lastF[ch] + (1.f - last_alpha) * accF[ch];
          lastF[ch] = f;
        } else {
          accF[ch] = skip ? accF[ch] : last_alpha * lastF[ch] + (1.f - last_alpha) * accF[ch];
          lastF[ch] = skip ? lastF[ch] : f;
        }
        const float dL_dchannel_F = dLF[ch];
        if constexpr (ABS) dL_dalpha += (fabsf(f) + fabsf(accF[ch])) * fabsf(dL_dchannel_F);
        else dL_dalpha += (f - accF[ch]) * dL_dchannel_F;
        part[10 + ch][rank] = skip ? 0.0f : (ABS ? fabsf(dchannel_dcolor * dL_dchannel_F) : dchannel_dcolor * dL_dchannel_F);
      }
      dL_dalpha *= T;
      last_alpha = skip ? last_alpha : alpha;
      float bg_dot_dpixel = 0.f;
      if constexpr (ABS) {
        bg_dot_dpixel += fabsf(bg0 * dL_dpixel[0]);
        bg_dot_dpixel += fabsf(bg1 * dL_dpixel[1]);
        bg_dot_dpixel += fabsf(bg2 * dL_dpixel[2]);
        dL_dalpha += fabsf(T_final / (1.f - alpha)) * bg_dot_dpixel;
      } else {
        bg_dot_dpixel += bg0 * dL_dpixel[0];
        bg_dot_dpixel += bg1 * dL_dpixel[1];
        bg_dot_dpixel += bg2 * dL_dpixel[2];
        dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;
      }
      const float dL_dG = ABS ? fabsf(co3) * dL_dalpha : co3 * dL_dalpha;
      const float gdx = G * dx;
      const float gdy = G * dy;
      if constexpr (ABS) {
        const float dG_ddelx = fabsf(gdx * co0) + fabsf(gdy * co1);
        const float dG_ddely = fabsf(gdy * co2) + fabsf(gdx * co1);
        part[0][rank] = skip ? 0.f : dL_dG * dG_ddelx * ddelx_dx;
        part[1][rank] = skip ? 0.f : dL_dG * dG_ddely * ddely_dy;
        part[2][rank] = skip ? 0.f : fabsf(0.5f * gdx * dx) * dL_dG;
        part[3][rank] = skip ? 0.f : fabsf(0.5f * gdx * dy) * dL_dG;
        part[4][rank] = skip ? 0.f : fabsf(0.5f * gdy * dy) * dL_dG;
        part[5][rank] = skip ? 0.f : G * dL_dalpha;
      } else {
        const float dG_ddelx = -gdx * co0 - gdy * co1;
        const float dG_ddely = -gdy * co2 - gdx * co1;
        part[0][rank] = skip ? 0.f : dL_dG * dG_ddelx * ddelx_dx;
        part[1][rank] = skip ? 0.f : dL_dG * dG_ddely * ddely_dy;
        part[2][rank] = skip ? 0.f : -0.5f * gdx * dx * dL_dG;
        part[3][rank] = skip ? 0.f : -0.5f * gdx * dy * dL_dG;
        part[4][rank] = skip ? 0.f : -0.5f * gdy * dy * dL_dG;
        part[5][rank] = skip ? 0.f : G * dL_dalpha;
      }
    }
    __syncthreads();
    float* row = rows + (size_t)u * NVP;
    if constexpr (MODE == OLSR_BWD_REFERENCE) {
      // render_cuda_reduce_sum over g.size() == BS lanes with integer halving (:696)
      for (int i = BS / 2; i > 0; i /= 2) {
        if (rank < i) {
#pragma unroll
          for (int v = 0; v < 10; ++v) part[v][rank] += part[v][rank + i];
        }
        __syncthreads();
      }
      if (rank < 10) row[rank] = part[rank][0];
      else if (rank < NV) row[rank] = part[rank][0];  // language: rank 0's own value (never reduced)
    } else {
      if (rank < NV) {
        double acc = 0.0;
        for (int lane = 0; lane < BS; ++lane) acc += (double)part[rank][lane];
        row[rank] = (float)acc;
      }
    }
    if (rank == 0) used[u] = 1;
    __syncthreads();  // (part is rewritten by the next entry)
  }
}

// one thread per Gaussian: its rows, in emission order (= ascending tile = the order of the sorted list), one after the other
__global__ __launch_bounds__(256) void gauss_rows_ordered_kernel(int P, int F, int NVP, const u32* __restrict__ inst_start,
                                                                 const u32* __restrict__ inst_count,
                                                                 const uint8_t* __restrict__ used,
                                                                 const float* __restrict__ rows, float* dL_dmean2D,
                                                                 float* dL_dconic, float* dL_dopacity, float* dL_dcolors,
                                                                 float* dL_dlanguage, float* dL_ddepths,
                                                                 const int32_t* __restrict__ counters) {
  const int g = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (g >= P) return;
  float acc[10 + 32];
  const int NV = 10 + F;
  for (int v = 0; v < NV; ++v) acc[v] = 0.f;
  const bool ok = !(frame_unusable(counters) || counters[2] != 0);
  const u32 n = ok ? inst_count[g] : 0u;
  if (n > 0) {
    const u32 u0 = inst_start[g];
    for (u32 u = u0; u < u0 + n; ++u) {
      if (!used[u]) continue;
      const float* row = rows + (size_t)u * NVP;
      for (int v = 0; v < NV; ++v) acc[v] += row[v];
    }
  }
  dL_dmean2D[3 * (size_t)g + 0] = acc[0];
  dL_dmean2D[3 * (size_t)g + 1] = acc[1];
  dL_dmean2D[3 * (size_t)g + 2] = 0.f;
  dL_dconic[4 * (size_t)g + 0] = acc[2];
  dL_dconic[4 * (size_t)g + 1] = acc[3];
  dL_dconic[4 * (size_t)g + 2] = 0.f;
  dL_dconic[4 * (size_t)g + 3] = acc[4];
  dL_dopacity[g] = acc[5];
  dL_dcolors[3 * (size_t)g + 0] = acc[6];
  dL_dcolors[3 * (size_t)g + 1] = acc[7];
  dL_dcolors[3 * (size_t)g + 2] = acc[8];
  dL_ddepths[g] = acc[9];
  for (int ch = 0; ch < F; ++ch) dL_dlanguage[(size_t)g * F + ch] = acc[10 + ch];
}

template <int TILE, int MODE, bool ABS>
static void launch_ordered_f(const olsr_scene& s, int F_rows, const FrameDims& d, const GeometryState& g, const BinningState& b,
                             const ImageState& im, const float* dL_dcolor, const float* dL_dlanguage, const float* dL_ddepth,
                             float* rows, uint8_t* used, hipStream_t st) {
  const float* colors = s.colors_precomp ? s.colors_precomp : g.rgb;
#define OLSR_ORD_ARGS                                                                                                  \
  im.ranges, b.inst_gid, b.src, d.W, d.H, d.gx, g.means2D, g.conic_opacity, g.depths, colors, s.language_precomp,      \
      s.background, im.final_T, im.n_contrib, dL_dcolor, dL_dlanguage, dL_ddepth, rows, used, g.counters
  switch (F_rows) {
    case 0: render_bwd_ordered_kernel<TILE, 0, MODE, ABS><<<d.ntiles, 256, 0, st>>>(OLSR_ORD_ARGS); break;
    case 3: render_bwd_ordered_kernel<TILE, 3, MODE, ABS><<<d.ntiles, 256, 0, st>>>(OLSR_ORD_ARGS); break;
    case 15: render_bwd_ordered_kernel<TILE, 15, MODE, ABS><<<d.ntiles, 256, 0, st>>>(OLSR_ORD_ARGS); break;
    case 16: render_bwd_ordered_kernel<TILE, 16, MODE, ABS><<<d.ntiles, 256, 0, st>>>(OLSR_ORD_ARGS); break;
    default: render_bwd_ordered_kernel<TILE, 32, MODE, ABS><<<d.ntiles, 256, 0, st>>>(OLSR_ORD_ARGS); break;
  }
#undef OLSR_ORD_ARGS
}

// rows: [num_rendered][grad_row(F)] floats, used: [num_rendered] bytes (zeroed here)
void launch_render_backward_ordered(const olsr_scene& s, const FrameDims& d, const GeometryState& g, const BinningState& b,
                                    const ImageState& im, int64_t num_rendered, const float* dL_dcolor,
                                    const float* dL_dlanguage, const float* dL_ddepth, float* rows, uint8_t* used,
                                    float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolors,
                                    float* dL_dlanguage_out, float* dL_ddepths, bool condition, hipStream_t st) {
  (void)hipMemsetAsync(used, 0, (size_t)num_rendered, st);
  const int F = s.F;
#define OLSR_ORD_GO(T_, M_)                                                                                               \
  do {                                                                                                                    \
    if (condition) launch_ordered_f<T_, M_, true>(s, F, d, g, b, im, dL_dcolor, dL_dlanguage, dL_ddepth, rows, used, st); \
    else launch_ordered_f<T_, M_, false>(s, F, d, g, b, im, dL_dcolor, dL_dlanguage, dL_ddepth, rows, used, st);          \
  } while (0)
  if (d.tile == 15) {
    if (s.bwd_mode == OLSR_BWD_REFERENCE) OLSR_ORD_GO(15, OLSR_BWD_REFERENCE);
    else OLSR_ORD_GO(15, OLSR_BWD_EXACT);
  } else {
    if (s.bwd_mode == OLSR_BWD_REFERENCE) OLSR_ORD_GO(16, OLSR_BWD_REFERENCE);
    else OLSR_ORD_GO(16, OLSR_BWD_EXACT);
  }
#undef OLSR_ORD_GO
  gauss_rows_ordered_kernel<<<(s.P + 255) / 256, 256, 0, st>>>(s.P, F, grad_row(F), g.inst_start, g.tiles_touched, used, rows,
                                                               dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolors,
                                                               dL_dlanguage_out, dL_ddepths, g.counters);
}

}  // namespace olsr
