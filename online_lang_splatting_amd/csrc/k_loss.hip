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
#include "olsr_loss_device.h"

namespace olsr {

constexpr int LOSS_THREADS = 256;

// VEC consecutive pixels of one row per thread (VEC = 4 when W % 4 == 0: 16-byte loads and stores on
// every plane; VEC = 1 otherwise).
template <int VEC>
struct PixVec {
  float v[VEC];
};
template <int VEC>
__device__ __forceinline__ PixVec<VEC> load_px(const float* __restrict__ plane, size_t p) {
  PixVec<VEC> r;
  if constexpr (VEC == 4) {
    const float4 q = *reinterpret_cast<const float4*>(plane + p);
    r.v[0] = q.x; r.v[1] = q.y; r.v[2] = q.z; r.v[3] = q.w;
  } else {
    r.v[0] = plane[p];
  }
  return r;
}
template <int VEC>
__device__ __forceinline__ void store_px(float* __restrict__ plane, size_t p, const PixVec<VEC>& r) {
  if constexpr (VEC == 4)
    *reinterpret_cast<float4*>(plane + p) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
  else
    plane[p] = r.v[0];
}

// TRACK: the tracking loss (get_loss_tracking / _rgb / _rgbd, utils/slam_utils.py:92-121): the RGB term is weighted
// by the rendered opacity and masked by the keyframe's image-gradient mask, the depth term is masked by
// opacity > 0.95; no language term.  (The gradient THROUGH the opacity is not produced: the rasterizer's autograd
// function ignores the cotangent of its opacity output, DGR/diff_gaussian_rasterization/__init__.py:333-343.)
template <int F, int VEC, bool TRACK>
__global__ __launch_bounds__(LOSS_THREADS) void mapping_loss_kernel(
    int W, int H, int lw, int lh, int use_exposure, float alpha, float thr, float lamda,
    const float* __restrict__ image, const float* __restrict__ depth, const float* __restrict__ lang,
    const float* __restrict__ gt_image, const float* __restrict__ gt_depth, const float* __restrict__ gt_lang,
    const float* __restrict__ exposure, const float* __restrict__ opacity, const float* __restrict__ grad_mask,
    float* __restrict__ d_image, float* __restrict__ d_depth, float* __restrict__ d_lang,
    float* __restrict__ partials) {
  const size_t HW = (size_t)H * W;
  const size_t p = ((size_t)blockIdx.x * LOSS_THREADS + threadIdx.x) * VEC;  // first pixel of this thread
  float s[LOSS_SUMS] = {0.f, 0.f, 0.f, 0.f, 0.f};
  if (p < HW) {
    const float ea = use_exposure ? expf(exposure[0]) : 1.f;
    const float eb = use_exposure ? exposure[1] : 0.f;
    // ---- RGB: |m * (e^a image + b) - m * gt|, utils/slam_utils.py:125-127,143-146
    PixVec<VEC> gt[3], m;
#pragma unroll
    for (int c = 0; c < 3; ++c) gt[c] = load_px<VEC>(gt_image + c * HW, p);
#pragma unroll
    for (int i = 0; i < VEC; ++i) m.v[i] = ((gt[0].v[i] + gt[1].v[i]) + gt[2].v[i] > thr) ? 1.f : 0.f;
    PixVec<VEC> op;
#pragma unroll
    for (int i = 0; i < VEC; ++i) op.v[i] = 1.f;
    if constexpr (TRACK) {
      op = load_px<VEC>(opacity, p);
      if (grad_mask != nullptr) {
        const PixVec<VEC> gm = load_px<VEC>(grad_mask, p);
#pragma unroll
        for (int i = 0; i < VEC; ++i) m.v[i] *= gm.v[i];  // rgb_pixel_mask * viewpoint.grad_mask, :102
      }
    }
    const float wrgb = alpha / (3.0f * (float)HW);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const PixVec<VEC> x = load_px<VEC>(image + c * HW, p);
      PixVec<VEC> d;
#pragma unroll
      for (int i = 0; i < VEC; ++i)
        d.v[i] = loss_rgb_term<TRACK>(x.v[i], gt[c].v[i], m.v[i], op.v[i], use_exposure, ea, eb, wrgb, s);
      store_px<VEC>(d_image + c * HW, p, d);
    }
    // ---- depth: |m_d * depth - m_d * gt_depth|, :144,147
    {
      const PixVec<VEC> gd = load_px<VEC>(gt_depth, p), x = load_px<VEC>(depth, p);
      PixVec<VEC> d;
      const float wd = (1.f - alpha) / (float)HW;
#pragma unroll
      for (int i = 0; i < VEC; ++i) d.v[i] = loss_depth_term<TRACK>(x.v[i], gd.v[i], op.v[i], wd, s);
      store_px<VEC>(d_depth, p, d);
    }
    // ---- language: |language - bilinear(gt_language)|, utils/slam_backend.py:579-590
    if constexpr (F > 0) {
      const int x = (int)(p % (size_t)W), y = (int)(p / (size_t)W);
      if (gt_lang != nullptr) {
        int x0[VEC], x1[VEC], y0, y1;
        float lx0[VEC], lx1[VEC], ly0, ly1;
#pragma unroll
        for (int i = 0; i < VEC; ++i) bilinear_index(x + i, (float)lw / (float)W, lw, x0[i], x1[i], lx0[i], lx1[i]);
        bilinear_index(y, (float)lh / (float)H, lh, y0, y1, ly0, ly1);
        const float wl = lamda / ((float)F * (float)HW);
        const size_t plane = (size_t)lh * lw;
#pragma unroll 3
        for (int c = 0; c < F; ++c) {
          const float* g0 = gt_lang + c * plane + (size_t)y0 * lw;
          const float* g1 = gt_lang + c * plane + (size_t)y1 * lw;
          const PixVec<VEC> l = load_px<VEC>(lang + c * HW, p);
          PixVec<VEC> d;
#pragma unroll
          for (int i = 0; i < VEC; ++i)
            d.v[i] = loss_lang_term(l.v[i], g0, g1, x0[i], x1[i], lx0[i], lx1[i], ly0, ly1, wl, s);
          store_px<VEC>(d_lang + c * HW, p, d);
        }
      } else {
        PixVec<VEC> z;
#pragma unroll
        for (int i = 0; i < VEC; ++i) z.v[i] = 0.f;
#pragma unroll 3
        for (int c = 0; c < F; ++c) store_px<VEC>(d_lang + c * HW, p, z);
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

__global__ __launch_bounds__(256) void mapping_loss_final_kernel(const LossFinalArgs a) {
  __shared__ double red[4][LOSS_SUMS];
  loss_final_block(a, red);
}

// the per-tile partials of the forward composite's fused epilogue -> loss[4], dL_dexposure[2] (same final reduction)
LossFinalArgs loss_final_args(const float* partials, int nb, const olsr_loss_params& p, bool tracking, bool has_lang,
                              bool use_exposure, float* loss, float* dL_dexposure) {
  LossFinalArgs a{};
  a.partials = partials;
  a.nb = nb;
  a.W = p.width;
  a.H = p.height;
  a.F = tracking ? 0 : p.F;
  a.has_lang = (!tracking && has_lang) ? 1 : 0;
  a.alpha = p.alpha;
  a.lamda = p.lamda_lang;
  a.loss = loss;
  a.d_exposure = dL_dexposure;
  a.use_exposure = use_exposure ? 1 : 0;
  a.zero_exposure = use_exposure ? 0 : 1;
  return a;
}

int loss_blocks(int W, int H) { return (int)(((size_t)W * H + LOSS_THREADS - 1) / LOSS_THREADS); }

void launch_mapping_loss(const olsr_loss_params& p, const float* image, const float* depth, const float* language,
                         const float* gt_image, const float* gt_depth, const float* gt_language, const float* exposure,
                         const float* opacity, const float* grad_mask, bool tracking, float* dL_dimage,
                         float* dL_ddepth, float* dL_dlanguage, float* loss, float* dL_dexposure, float* partials,
                         hipStream_t st) {
  // 16-byte path: planes are [C][H][W], so with W % 4 == 0 every plane and row stays 16-byte aligned if the
  // base pointers are
  auto al16 = [](const void* q) { return q == nullptr || ((uintptr_t)q & 15u) == 0; };
  const bool vec4 = (p.width % 4) == 0 && al16(image) && al16(depth) && al16(language) && al16(gt_image) &&
                    al16(gt_depth) && al16(dL_dimage) && al16(dL_ddepth) && al16(dL_dlanguage) && al16(opacity) &&
                    al16(grad_mask);
  const size_t threads = ((size_t)p.width * p.height + (vec4 ? 3 : 0)) / (vec4 ? 4 : 1);
  const int nb = (int)((threads + LOSS_THREADS - 1) / LOSS_THREADS);  // <= loss_blocks(): the scratch is sized for VEC = 1
  const int use_exposure = (exposure != nullptr && !p.initialization) ? 1 : 0;
#define OLSR_LOSS_ARGS                                                                                           \
  p.width, p.height, p.lang_width, p.lang_height, use_exposure, p.alpha, p.rgb_boundary_threshold, p.lamda_lang, \
      image, depth, language, gt_image, gt_depth, gt_language, exposure, opacity, grad_mask, dL_dimage, dL_ddepth, \
      dL_dlanguage, partials
#define OLSR_LOSS(FV)                                                                       \
  case FV:                                                                                  \
    if (vec4)                                                                               \
      mapping_loss_kernel<FV, 4, false><<<nb, LOSS_THREADS, 0, st>>>(OLSR_LOSS_ARGS);       \
    else                                                                                    \
      mapping_loss_kernel<FV, 1, false><<<nb, LOSS_THREADS, 0, st>>>(OLSR_LOSS_ARGS);       \
    break;
  if (tracking) {
    if (vec4)
      mapping_loss_kernel<0, 4, true><<<nb, LOSS_THREADS, 0, st>>>(OLSR_LOSS_ARGS);
    else
      mapping_loss_kernel<0, 1, true><<<nb, LOSS_THREADS, 0, st>>>(OLSR_LOSS_ARGS);
  } else {
    switch (p.F) {
      OLSR_LOSS(0) OLSR_LOSS(3) OLSR_LOSS(15) OLSR_LOSS(16) OLSR_LOSS(32)
      default: break;
    }
  }
#undef OLSR_LOSS
#undef OLSR_LOSS_ARGS
  mapping_loss_final_kernel<<<1, 256, 0, st>>>(loss_final_args(partials, nb, p, tracking, gt_language != nullptr,
                                                               use_exposure != 0, loss, dL_dexposure));
}

}  // namespace olsr
