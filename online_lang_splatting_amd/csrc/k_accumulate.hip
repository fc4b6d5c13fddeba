// k_accumulate.hip — accumulate one view's per-Gaussian gradients into the flat buffer that the
// frame-sharded step all-reduces, plus the densification statistics.
//
// Counterpart of what autograd's `.grad +=` over the 12 views of a mapping iteration and
// GaussianModel.add_densification_stats (gaussian_splatting/scene/gaussian_model.py:965-969) do
// with ~10 separate elementwise PyTorch kernels; here it is one pass: every gradient array is
// read once, the flat row [3 xyz | 3M sh | 1 opacity | 3 scale | 4 rot | F lang] is read-modify-
// written once (or just written, for the first view of a step: `assign`).
#include "olsr_device.h"
#include "olsr_kernels.h"

namespace olsr {

// A block covers ACC_G consecutive Gaussians = ACC_G * width consecutive floats of `flat`, so the
// read-modify-write of `flat` is perfectly coalesced; (gaussian, column) of an element come from a
// small-range division done in fp32 (exact for e < 2^22).
constexpr int ACC_G = 64;

template <bool ASSIGN>
__global__ __launch_bounds__(256) void accumulate_kernel(
    int P, int M, int F, int width, const float* __restrict__ dmeans3D, const float* __restrict__ dsh,
    const float* __restrict__ dopacity, const float* __restrict__ dscales, const float* __restrict__ drot,
    const float* __restrict__ dlang, const float* __restrict__ dmeans2D, const int32_t* __restrict__ radii,
    float* __restrict__ flat, float* __restrict__ densify, int32_t* __restrict__ max_radii) {
  const int g0 = blockIdx.x * ACC_G;
  const int ng = min(ACC_G, P - g0);
  const int count = ng * width;
  const int sh_w = 3 * M;
  const float inv_w = 1.0f / (float)width;
  float* out = flat + (size_t)g0 * width;
  for (int e = threadIdx.x; e < count; e += 256) {
    const int gl = (int)(((float)e + 0.5f) * inv_w);
    const int c = e - gl * width;
    const int g = g0 + gl;
    float v;
    if (c < 3) v = dmeans3D[3 * (size_t)g + c];
    else if (c < 3 + sh_w) v = dsh[(size_t)g * sh_w + (c - 3)];
    else if (c < 4 + sh_w) v = dopacity[g];
    else if (c < 7 + sh_w) v = dscales[3 * (size_t)g + (c - 4 - sh_w)];
    else if (c < 11 + sh_w) v = drot[4 * (size_t)g + (c - 7 - sh_w)];
    else v = dlang[(size_t)g * F + (c - 11 - sh_w)];
    out[e] = ASSIGN ? v : out[e] + v;
  }
  if (threadIdx.x < ng) {
    const int g = g0 + threadIdx.x;
    const int r = radii[g];
    const bool vis = r > 0;
    const float gx = dmeans2D[3 * (size_t)g], gy = dmeans2D[3 * (size_t)g + 1];
    const float nrm = vis ? sqrtf(gx * gx + gy * gy) : 0.f;  // ||viewspace grad||, taken per view
    const float cnt = vis ? 1.f : 0.f;
    float2* dz = reinterpret_cast<float2*>(densify) + g;
    if (ASSIGN) {
      *dz = make_float2(nrm, cnt);
      max_radii[g] = r;
    } else {
      const float2 o = *dz;
      *dz = make_float2(o.x + nrm, o.y + cnt);
      max_radii[g] = max(max_radii[g], r);
    }
  }
}

void launch_accumulate(int P, int M, int F, bool assign, const float* dmeans3D, const float* dsh,
                       const float* dopacity, const float* dscales, const float* drot, const float* dlang,
                       const float* dmeans2D, const int32_t* radii, float* flat, float* densify, int32_t* max_radii,
                       hipStream_t st) {
  if (P <= 0) return;
  const int width = 11 + 3 * M + F;
  const unsigned nb = (unsigned)((P + ACC_G - 1) / ACC_G);
  if (assign)
    accumulate_kernel<true><<<nb, 256, 0, st>>>(P, M, F, width, dmeans3D, dsh, dopacity, dscales, drot, dlang, dmeans2D,
                                                radii, flat, densify, max_radii);
  else
    accumulate_kernel<false><<<nb, 256, 0, st>>>(P, M, F, width, dmeans3D, dsh, dopacity, dscales, drot, dlang,
                                                 dmeans2D, radii, flat, densify, max_radii);
}

}  // namespace olsr
