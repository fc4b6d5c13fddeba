// k_adam.hip — one fused Adam step over every Gaussian parameter, fed by the flat gradient bucket.
//
// Caller side of the path.  The reference steps `torch.optim.Adam(param_groups, lr=0.0,
// eps=1e-15)` over seven parameter tensors (gaussian_splatting/scene/gaussian_model.py:393-440,
// utils/slam_backend.py:747-749): per tensor a chain of elementwise PyTorch kernels.  Here the bucket that
// the frame-sharded step all-reduces IS the gradient of all seven (row = [3 xyz | 3M sh | 1 opacity |
// 3 scale | 4 rotation | F language]), so one pass reads a row of gradient, parameters and both moments and
// writes parameters and moments back.  The arithmetic is torch.optim.Adam's single-tensor path, operation for
// operation (torch/optim/adam.py: lerp, mul/addcmul, sqrt / bias_correction2_sqrt + eps, addcdiv), dense:
// a Gaussian with zero gradient still decays its moments and moves, exactly as in the reference — a
// "visible rows only" step would be cheaper but is a different optimiser.
//
// HBM-bound: 5 reads + 3 writes of P x width floats, all coalesced (consecutive threads = consecutive floats
// of the flat arrays; the parameter arrays are [P, k] slices addressed per (row, column)).
#include "olsr_device.h"
#include "olsr_kernels.h"

namespace olsr {

constexpr int ADAM_G = 64;  // Gaussians per block, as in k_accumulate.hip

// every scalar of the update, formed in double on the host and rounded to fp32 once (torch passes Python floats
// to lerp_ / mul_ / addcmul_ / addcdiv_, which round them to the tensor's dtype)
struct AdamScalars {
  float one_minus_beta1, beta2, one_minus_beta2, bias_correction2_sqrt, eps;
  float neg_step_xyz, neg_step_sh_dc, neg_step_sh_rest, neg_step_opacity, neg_step_scale, neg_step_rotation,
      neg_step_language;  // -(lr / bias_correction1)
};

// up to OLSR_ADAM_MAX_BUCKETS gradient buckets summed on the fly, in order: ((f0 + f1) + f2) + ... — what a sum of the lane
// buckets (frame_shard.FrameLanes) leaves, bit for bit, without writing it anywhere
struct AdamBuckets {
  const float* more[OLSR_ADAM_MAX_BUCKETS - 1];
  int n_more;
  // row masks (olsr_grad_bucket.row_mask: bit g clear = row g of that bucket is zero), or null = every row is read.
  // A block covers ADAM_G = 64 Gaussians = one mask word: rows a mask proves zero are not read at all (their gradient is the
  // +0.0 the row holds), the update itself stays dense — parameters and moments are those of torch.optim.Adam bit for bit.
  const unsigned long long* mask0;
  const unsigned long long* mask_more[OLSR_ADAM_MAX_BUCKETS - 1];
};

__global__ __launch_bounds__(256) void adam_step_kernel(int P, int M, int F, int width, const float* __restrict__ flat,
                                                        AdamBuckets extra,
                                                        float* __restrict__ means3D, float* __restrict__ shs,
                                                        float* __restrict__ opacities, float* __restrict__ scales,
                                                        float* __restrict__ rotations, float* __restrict__ language,
                                                        float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq,
                                                        AdamScalars hp) {
  const int g0 = blockIdx.x * ADAM_G;
  const int ng = min(ADAM_G, P - g0);
  const int count = ng * width;
  const int sh_w = 3 * M;
  const float inv_w = 1.0f / (float)width;
  const size_t base = (size_t)g0 * width;
  static_assert(ADAM_G == 64, "one row-mask word per block");
  const unsigned long long w0 = extra.mask0 ? extra.mask0[blockIdx.x] : ~0ull;
  unsigned long long wm[OLSR_ADAM_MAX_BUCKETS - 1];
#pragma unroll
  for (int b = 0; b < OLSR_ADAM_MAX_BUCKETS - 1; ++b)
    wm[b] = (b < extra.n_more) ? (extra.mask_more[b] ? extra.mask_more[b][blockIdx.x] : ~0ull) : 0ull;
  for (int e = threadIdx.x; e < count; e += 256) {
    const int gl = (int)(((float)e + 0.5f) * inv_w);
    const int c = e - gl * width;
    const size_t g = (size_t)(g0 + gl);
    float* p;
    float neg_step;
    if (c < 3) { p = means3D + 3 * g + c; neg_step = hp.neg_step_xyz; }
    else if (c < 3 + sh_w) { p = shs + g * sh_w + (c - 3); neg_step = (c < 6) ? hp.neg_step_sh_dc : hp.neg_step_sh_rest; }
    else if (c < 4 + sh_w) { p = opacities + g; neg_step = hp.neg_step_opacity; }
    else if (c < 7 + sh_w) { p = scales + 3 * g + (c - 4 - sh_w); neg_step = hp.neg_step_scale; }
    else if (c < 11 + sh_w) { p = rotations + 4 * g + (c - 7 - sh_w); neg_step = hp.neg_step_rotation; }
    else { p = language + g * F + (c - 11 - sh_w); neg_step = hp.neg_step_language; }
    float grad = ((w0 >> gl) & 1ull) ? flat[base + e] : 0.0f;
#pragma unroll
    for (int b = 0; b < OLSR_ADAM_MAX_BUCKETS - 1; ++b)
      if ((wm[b] >> gl) & 1ull) grad += extra.more[b][base + e];
    float m = exp_avg[base + e], v = exp_avg_sq[base + e];
    m = m + (grad - m) * hp.one_minus_beta1;              // exp_avg.lerp_(grad, 1 - beta1)
    v = v * hp.beta2 + hp.one_minus_beta2 * grad * grad;  // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    const float denom = sqrtf(v) / hp.bias_correction2_sqrt + hp.eps;
    *p = *p + neg_step * (m / denom);                     // param.addcdiv_(exp_avg, denom, value=-step_size)
    exp_avg[base + e] = m;
    exp_avg_sq[base + e] = v;
  }
}

void launch_adam_step(int P, int M, int F, const olsr_adam_params& hp, const float* const* flats,
                      const unsigned long long* const* masks, int n_flats, float* means3D, float* shs, float* opacities, float* scales, float* rotations, float* language,
                      float* exp_avg, float* exp_avg_sq, hipStream_t st) {
  if (P <= 0) return;
  const float* flat = flats[0];
  AdamBuckets extra{};
  extra.n_more = n_flats - 1;
  extra.mask0 = masks ? masks[0] : nullptr;
  for (int b = 1; b < n_flats; ++b) {
    extra.more[b - 1] = flats[b];
    extra.mask_more[b - 1] = masks ? masks[b] : nullptr;
  }
  const int width = 11 + 3 * M + F;
  // torch/optim/adam.py, _single_tensor_adam: Python-float (double) arithmetic for every scalar
  const double bc1 = 1.0 - pow(hp.beta1, (double)hp.step);
  const double bc2 = 1.0 - pow(hp.beta2, (double)hp.step);
  AdamScalars k;
  k.one_minus_beta1 = (float)(1.0 - hp.beta1);
  k.beta2 = (float)hp.beta2;
  k.one_minus_beta2 = (float)(1.0 - hp.beta2);
  k.bias_correction2_sqrt = (float)sqrt(bc2);
  k.eps = (float)hp.eps;
  k.neg_step_xyz = (float)(-(hp.lr_xyz / bc1));
  k.neg_step_sh_dc = (float)(-(hp.lr_sh_dc / bc1));
  k.neg_step_sh_rest = (float)(-(hp.lr_sh_rest / bc1));
  k.neg_step_opacity = (float)(-(hp.lr_opacity / bc1));
  k.neg_step_scale = (float)(-(hp.lr_scale / bc1));
  k.neg_step_rotation = (float)(-(hp.lr_rotation / bc1));
  k.neg_step_language = (float)(-(hp.lr_language / bc1));
  adam_step_kernel<<<(P + ADAM_G - 1) / ADAM_G, 256, 0, st>>>(P, M, F, width, flat, extra, means3D, shs, opacities, scales,
                                                             rotations, language, exp_avg, exp_avg_sq, k);
}

}  // namespace olsr
