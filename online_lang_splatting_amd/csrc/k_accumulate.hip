// k_accumulate.hip — accumulate one view's per-Gaussian gradients into the flat buffer that the
// frame-sharded step all-reduces, plus the densification statistics.
//
// Counterpart of what autograd's `.grad +=` over the 12 views of a mapping iteration and
// GaussianModel.add_densification_stats (gaussian_splatting/scene/gaussian_model.py:965-969) do
// with ~10 separate elementwise PyTorch kernels; here it is one pass: every gradient array is
// read once, the flat row [3 xyz | 3M sh | 1 opacity | 3 scale | 4 rot | F lang] is read-modify-
// written once.  One wave handles 64 / WIDTH-chunks... kept simple: one thread per (Gaussian, column).
#include "olsr_device.h"
#include "olsr_kernels.h"

namespace olsr {

__global__ __launch_bounds__(256) void accumulate_kernel(
    int P, int M, int F, int width, const float* __restrict__ dmeans3D, const float* __restrict__ dsh,
    const float* __restrict__ dopacity, const float* __restrict__ dscales, const float* __restrict__ drot,
    const float* __restrict__ dlang, const float* __restrict__ dmeans2D, const int32_t* __restrict__ radii,
    float* __restrict__ flat, float* __restrict__ densify, int32_t* __restrict__ max_radii) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)P * width;
  if (i >= total) return;
  const int g = (int)(i / width), c = (int)(i % width);
  const int sh_w = 3 * M;
  float v;
  if (c < 3) v = dmeans3D[3 * (size_t)g + c];
  else if (c < 3 + sh_w) v = dsh[(size_t)g * sh_w + (c - 3)];
  else if (c < 4 + sh_w) v = dopacity[g];
  else if (c < 7 + sh_w) v = dscales[3 * (size_t)g + (c - 4 - sh_w)];
  else if (c < 11 + sh_w) v = drot[4 * (size_t)g + (c - 7 - sh_w)];
  else v = dlang[(size_t)g * F + (c - 11 - sh_w)];
  flat[i] += v;
  if (c == 0) {
    const int r = radii[g];
    const bool vis = r > 0;
    const float gx = dmeans2D[3 * (size_t)g], gy = dmeans2D[3 * (size_t)g + 1];
    densify[2 * (size_t)g] += vis ? sqrtf(gx * gx + gy * gy) : 0.f;  // ||viewspace grad||, per view
    densify[2 * (size_t)g + 1] += vis ? 1.f : 0.f;
    max_radii[g] = max(max_radii[g], r);
  }
}

void launch_accumulate(int P, int M, int F, const float* dmeans3D, const float* dsh, const float* dopacity,
                       const float* dscales, const float* drot, const float* dlang, const float* dmeans2D,
                       const int32_t* radii, float* flat, float* densify, int32_t* max_radii, hipStream_t st) {
  if (P <= 0) return;
  const int width = 11 + 3 * M + F;
  const int64_t total = (int64_t)P * width;
  accumulate_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(P, M, F, width, dmeans3D, dsh, dopacity, dscales,
                                                                   drot, dlang, dmeans2D, radii, flat, densify,
                                                                   max_radii);
}

}  // namespace olsr
