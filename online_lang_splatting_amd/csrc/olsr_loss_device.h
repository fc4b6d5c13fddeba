// olsr_loss_device.h — the per-pixel arithmetic of the mapping / tracking loss and of its image cotangents, shared by the
// stand-alone loss kernel (k_loss.hip) and by the forward composite's fused epilogue (k_render_fwd.hip, olsr_forward_async_loss)
// so that both produce the same cotangents bit for bit.
//
// Replaces get_loss_mapping / get_loss_mapping_rgbd (utils/slam_utils.py:124-165), get_loss_tracking* (:92-121), the bilinear
// resize of the language target and its L1 (utils/slam_backend.py:579-597) and the autograd backward of all of them.
#pragma once
#include "olsr_device.h"
#include "olsr_kernels.h"

namespace olsr {

constexpr int LOSS_SUMS = 5;  // |rgb|, |depth|, |language|, dL/da, dL/db (unweighted sums)

__device__ __forceinline__ float loss_sgn(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }

// upsample_bilinear2d, align_corners=False (ATen UpSample.h: area_pixel_compute_source_index):
// src = scale * (dst + 0.5) - 0.5, clamped at 0; i1 = i0 + (i0 < in - 1)
__device__ __forceinline__ void bilinear_index(int dst, float scale, int in_size, int& i0, int& i1, float& l0,
                                               float& l1) {
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  src = (src < 0.f) ? 0.f : src;
  i0 = min((int)src, in_size - 1);
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.f - l1;
}

// One colour channel of one pixel: |m (e^a x + b) - m gt| (utils/slam_utils.py:125-127,143-146); TRACK: weighted by the
// rendered opacity (:103).  Adds to s[0] (loss), s[3], s[4] (exposure gradients); returns d loss / d x.
template <bool TRACK>
__device__ __forceinline__ float loss_rgb_term(float x, float gt, float m, float op, int use_exposure, float ea, float eb,
                                               float wrgb, float* s) {
  const float ab = use_exposure ? ea * x + eb : x;
  const float v = ab * m - gt * m;
  s[0] += TRACK ? op * fabsf(v) : fabsf(v);
  const float dab = TRACK ? op * (loss_sgn(v) * m) : loss_sgn(v) * m;  // d loss term / d(image_ab)
  s[3] += dab * (ea * x);                                             // d(image_ab)/da = e^a image
  s[4] += dab;
  return wrgb * dab * ea;
}

// Depth of one pixel: |m_d depth - m_d gt_depth| (:144,147); TRACK: masked by opacity > 0.95 as well (:113-117).
template <bool TRACK>
__device__ __forceinline__ float loss_depth_term(float x, float gd, float op, float wd, float* s) {
  float md = (gd > 0.01f) ? 1.f : 0.f;
  if constexpr (TRACK) md *= (op > 0.95f) ? 1.f : 0.f;
  const float v = x * md - gd * md;
  s[1] += fabsf(v);
  return wd * loss_sgn(v) * md;
}

// One language channel of one pixel against the bilinearly resized target (utils/slam_backend.py:579-590).
__device__ __forceinline__ float loss_lang_term(float l, const float* g0, const float* g1, int x0, int x1, float lx0,
                                                float lx1, float ly0, float ly1, float wl, float* s) {
  const float t = ly0 * (lx0 * g0[x0] + lx1 * g0[x1]) + ly1 * (lx0 * g1[x0] + lx1 * g1[x1]);
  const float v = l - t;
  s[2] += fabsf(v);
  return wl * loss_sgn(v);
}

// What the forward composite needs to evaluate the loss in its epilogue (k_render_fwd.hip); mode 0: not fused.
struct FusedLossArgs {
  const float *gt_image, *gt_depth, *gt_lang, *exposure, *grad_mask;
  float *d_image, *d_depth, *d_lang;
  float* partials;  // [tiles][LOSS_SUMS]
  int lw, lh, use_exposure, write_images;
  float alpha, thr, lamda;
};

// The final reduction of a loss's partial sums (per block of the stand-alone loss kernel, per tile of the forward composite's
// fused epilogue) -> loss[4] = {total, rgb, depth, language}, dL_dexposure[2]: ONE 256-thread block, doubles, fixed order.
// Called by mapping_loss_final_kernel (k_loss.hip) and, for the fused epilogue, by one block of the tile-order kernel that
// follows the composite anyway (k_binning.hip: one launch less on the dependent chain of a tracking iteration).
__device__ __forceinline__ void loss_final_block(const LossFinalArgs& a, double (*red)[LOSS_SUMS]) {
  double acc[LOSS_SUMS] = {0, 0, 0, 0, 0};
  for (int b = threadIdx.x; b < a.nb; b += 256)
#pragma unroll
    for (int k = 0; k < LOSS_SUMS; ++k) acc[k] += (double)a.partials[(size_t)b * LOSS_SUMS + k];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < LOSS_SUMS; ++k) {
    double v = acc[k];
#pragma unroll
    for (int mm = 32; mm >= 1; mm >>= 1) v += __shfl_xor(v, mm);
    if (lane == 0) red[w][k] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t[LOSS_SUMS];
    for (int k = 0; k < LOSS_SUMS; ++k) t[k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
    const double HW = (double)a.H * (double)a.W;
    const double l_rgb = (double)a.alpha * t[0] / (3.0 * HW);
    const double l_depth = (1.0 - (double)a.alpha) * t[1] / HW;
    const double l_lang = (a.has_lang && a.F > 0) ? (double)a.lamda * t[2] / ((double)a.F * HW) : 0.0;
    a.loss[0] = (float)(l_rgb + l_depth + l_lang);
    a.loss[1] = (float)l_rgb;
    a.loss[2] = (float)l_depth;
    a.loss[3] = (float)l_lang;
    if (a.d_exposure != nullptr && a.use_exposure) {
      a.d_exposure[0] = (float)((double)a.alpha * t[3] / (3.0 * HW));
      a.d_exposure[1] = (float)((double)a.alpha * t[4] / (3.0 * HW));
    } else if (a.d_exposure != nullptr && a.zero_exposure) {
      a.d_exposure[0] = 0.f;
      a.d_exposure[1] = 0.f;
    }
  }
}

}  // namespace olsr
