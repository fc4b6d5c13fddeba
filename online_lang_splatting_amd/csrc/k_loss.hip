// k_loss.hip — mapping loss + its gradient w.r.t. the rendered images, one pass over the pixels.
//
// Caller side of the rasterizer path.  Replaces get_loss_mapping /
// get_loss_mapping_rgbd (utils/slam_utils.py:124-165), the bilinear resize of the language target
// and its L1 (utils/slam_backend.py:579-597, gaussian_splatting/utils/loss_utils.py:21-22) and
// the autograd backward of all of them.  HBM-bound elementwise work: one thread per pixel, every
// image plane is read once (coalesced: planes are [C][H][W], consecutive lanes = consecutive x) and
// every cotangent plane written once; the 192x192 language target (2 MB) stays in L2.
// Sums are deterministic: fixed-order wave/block trees into per-block partials, then one block
// adds the partials in double.
#include "olsr_device.h"
#include "olsr_kernels.h"

namespace olsr {

constexpr int LOSS_THREADS = 256;
constexpr int LOSS_SUMS = 5;  // |rgb|, |depth|, |language|, dL/da, dL/db (unweighted sums)

__device__ __forceinline__ float sgn(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }

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

template <int F>
__global__ __launch_bounds__(LOSS_THREADS) void mapping_loss_kernel(
    int W, int H, int lw, int lh, int use_exposure, float alpha, float thr, float lamda,
    const float* __restrict__ image, const float* __restrict__ depth, const float* __restrict__ lang,
    const float* __restrict__ gt_image, const float* __restrict__ gt_depth, const float* __restrict__ gt_lang,
    const float* __restrict__ exposure, float* __restrict__ d_image, float* __restrict__ d_depth,
    float* __restrict__ d_lang, float* __restrict__ partials) {
  const size_t HW = (size_t)H * W;
  const size_t p = (size_t)blockIdx.x * LOSS_THREADS + threadIdx.x;
  float s[LOSS_SUMS] = {0.f, 0.f, 0.f, 0.f, 0.f};
  if (p < HW) {
    const float ea = use_exposure ? expf(exposure[0]) : 1.f;
    const float eb = use_exposure ? exposure[1] : 0.f;
    // ---- RGB: |m * (e^a image + b) - m * gt|, utils/slam_utils.py:125-127,143-146
    const float g0 = gt_image[p], g1 = gt_image[HW + p], g2 = gt_image[2 * HW + p];
    const float m = ((g0 + g1) + g2 > thr) ? 1.f : 0.f;
    const float wrgb = alpha / (3.0f * (float)HW);
    const float gts[3] = {g0, g1, g2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float x = image[c * HW + p];
      const float ab = use_exposure ? ea * x + eb : x;
      const float v = ab * m - gts[c] * m;
      s[0] += fabsf(v);
      const float dab = sgn(v) * m;  // d|v| / d(image_ab)
      d_image[c * HW + p] = wrgb * dab * ea;
      s[3] += dab * (ea * x);        // d(image_ab)/da = e^a image
      s[4] += dab;
    }
    // ---- depth: |m_d * depth - m_d * gt_depth|, :144,147
    {
      const float gd = gt_depth[p];
      const float md = (gd > 0.01f) ? 1.f : 0.f;
      const float v = depth[p] * md - gd * md;
      s[1] += fabsf(v);
      d_depth[p] = (1.f - alpha) / (float)HW * sgn(v) * md;
    }
    // ---- language: |language - bilinear(gt_language)|, utils/slam_backend.py:579-590
    if constexpr (F > 0) {
      const int x = (int)(p % (size_t)W), y = (int)(p / (size_t)W);
      if (gt_lang != nullptr) {
        int x0, x1, y0, y1;
        float lx0, lx1, ly0, ly1;
        bilinear_index(x, (float)lw / (float)W, lw, x0, x1, lx0, lx1);
        bilinear_index(y, (float)lh / (float)H, lh, y0, y1, ly0, ly1);
        const float wl = lamda / ((float)F * (float)HW);
        const size_t plane = (size_t)lh * lw;
#pragma unroll 5
        for (int c = 0; c < F; ++c) {
          const float* g = gt_lang + c * plane;
          const float t = ly0 * (lx0 * g[(size_t)y0 * lw + x0] + lx1 * g[(size_t)y0 * lw + x1]) +
                          ly1 * (lx0 * g[(size_t)y1 * lw + x0] + lx1 * g[(size_t)y1 * lw + x1]);
          const float v = lang[c * HW + p] - t;
          s[2] += fabsf(v);
          d_lang[c * HW + p] = wl * sgn(v);
        }
      } else {
#pragma unroll 5
        for (int c = 0; c < F; ++c) d_lang[c * HW + p] = 0.f;
      }
    }
  }
  // fixed-order block reduction of the five sums
  __shared__ float red[LOSS_THREADS / 64][LOSS_SUMS];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < LOSS_SUMS; ++k) {
    float v = s[k];
#pragma unroll
    for (int mm = 32; mm >= 1; mm >>= 1) v += __shfl_xor(v, mm);
    if (lane == 0) red[w][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < LOSS_SUMS) {
    float v = 0.f;
    for (int k = 0; k < LOSS_THREADS / 64; ++k) v += red[k][threadIdx.x];
    partials[(size_t)blockIdx.x * LOSS_SUMS + threadIdx.x] = v;
  }
}

__global__ __launch_bounds__(256) void mapping_loss_final_kernel(const float* __restrict__ partials, int nb, int W, int H,
                                                                 int F, int has_lang, float alpha, float lamda,
                                                                 float* __restrict__ loss,
                                                                 float* __restrict__ d_exposure) {
  __shared__ double red[4][LOSS_SUMS];
  double acc[LOSS_SUMS] = {0, 0, 0, 0, 0};
  for (int b = threadIdx.x; b < nb; b += 256)
#pragma unroll
    for (int k = 0; k < LOSS_SUMS; ++k) acc[k] += (double)partials[(size_t)b * LOSS_SUMS + k];
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
    const double HW = (double)H * (double)W;
    const double l_rgb = (double)alpha * t[0] / (3.0 * HW);
    const double l_depth = (1.0 - (double)alpha) * t[1] / HW;
    const double l_lang = (has_lang && F > 0) ? (double)lamda * t[2] / ((double)F * HW) : 0.0;
    loss[0] = (float)(l_rgb + l_depth + l_lang);
    loss[1] = (float)l_rgb;
    loss[2] = (float)l_depth;
    loss[3] = (float)l_lang;
    if (d_exposure) {
      d_exposure[0] = (float)((double)alpha * t[3] / (3.0 * HW));
      d_exposure[1] = (float)((double)alpha * t[4] / (3.0 * HW));
    }
  }
}

int loss_blocks(int W, int H) { return (int)(((size_t)W * H + LOSS_THREADS - 1) / LOSS_THREADS); }

void launch_mapping_loss(const olsr_loss_params& p, const float* image, const float* depth, const float* language,
                         const float* gt_image, const float* gt_depth, const float* gt_language, const float* exposure,
                         float* dL_dimage, float* dL_ddepth, float* dL_dlanguage, float* loss, float* dL_dexposure,
                         float* partials, hipStream_t st) {
  const int nb = loss_blocks(p.width, p.height);
  const int use_exposure = (exposure != nullptr && !p.initialization) ? 1 : 0;
#define OLSR_LOSS(FV)                                                                                                  \
  case FV:                                                                                                             \
    mapping_loss_kernel<FV><<<nb, LOSS_THREADS, 0, st>>>(p.width, p.height, p.lang_width, p.lang_height, use_exposure, \
                                                         p.alpha, p.rgb_boundary_threshold, p.lamda_lang, image, depth, \
                                                         language, gt_image, gt_depth, gt_language, exposure, dL_dimage, \
                                                         dL_ddepth, dL_dlanguage, partials);                           \
    break;
  switch (p.F) {
    OLSR_LOSS(0) OLSR_LOSS(3) OLSR_LOSS(15) OLSR_LOSS(16) OLSR_LOSS(32)
    default: break;
  }
#undef OLSR_LOSS
  mapping_loss_final_kernel<<<1, 256, 0, st>>>(partials, nb, p.width, p.height, p.F, gt_language != nullptr ? 1 : 0,
                                               p.alpha, p.lamda_lang, loss, use_exposure ? dL_dexposure : nullptr);
  if (!use_exposure && dL_dexposure) (void)hipMemsetAsync(dL_dexposure, 0, 2 * sizeof(float), st);
}

}  // namespace olsr
