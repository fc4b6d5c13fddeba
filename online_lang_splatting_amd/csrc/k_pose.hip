// k_pose.hip — one tracking iteration's pose update on the device.
//
// Caller side of the path (the front end).  After every backward of its tracking loop the
// reference runs, as separate PyTorch calls with a dozen tiny kernels and a host read-back each
// (utils/slam_frontend.py:216-243):
//     pose_optimizer.step()                      torch.optim.Adam over cam_rot_delta (lr 0.003), cam_trans_delta
//                                                (lr 0.001), exposure_a, exposure_b (lr 0.01)
//     converged = update_pose(viewpoint)         utils/pose_utils.py:79-97: tau = [trans_delta | rot_delta],
//                                                new_w2c = SE3_exp(tau) @ T_w2c, |tau| < 1e-4, deltas zeroed
// and, on the next render, the camera properties (utils/camera_utils.py:103-117)
//     world_view_transform = getWorld2View2(R, T)^T,  full_proj_transform = world_view_transform @ projection_matrix,
//     camera_center = world_view_transform.inverse()[3, :3].
// Here that is ONE launch of one wave: the tracking loop is a chain of dependent iterations, so what counts is latency,
// and ~40 dependent launches of one-element kernels cost more than the 0.6 ms of rasterizer work they separate.
//
// Arithmetic: torch's single-tensor Adam operation for operation (lerp, mul/addcmul, sqrt / bias_correction2_sqrt +
// eps, addcdiv on a parameter that update_pose reset to zero), scalars formed in double on the host; SE3_exp with the
// reference's small-angle branches (SO3_exp / V, utils/pose_utils.py:26-58); fp32 matrix products in torch's
// row-times-column order.  Pinned by golden vectors generated from the reference's own SE3_exp / update_pose / Camera
// (tests/golden/make_golden_pose.py).
#include "olsr_device.h"
#include "olsr_kernels.h"

namespace olsr {

struct PoseScalars {
  float one_minus_beta1, beta2, one_minus_beta2, bias_correction2_sqrt, eps;
  float neg_step_rot, neg_step_trans, neg_step_exposure, converged_threshold;
  int has_grad, has_exposure;
  // step_on_device: the Adam step count is status[1] + 1 (params->step <= 0), so that the same launch can be replayed
  // from a HIP graph; the bias corrections are then formed here, in double like launch_pose_step forms them on the host
  int step_on_device;
  double beta1_d, beta2_d, lr_rot_d, lr_trans_d, lr_exposure_d;
};

// state (floats): [0,16) T_w2c row-major | [16,32) world_view_transform = W2C^T | [32,48) full_proj_transform |
// [48,52) camera_center + pad | [52,58) exp_avg of tau = [trans | rot] | [58,64) exp_avg_sq | [64,70) tau applied by
// the last step | [70,72) exposure a, b | [72,74) their exp_avg | [74,76) exp_avg_sq | [76,80) pad
__global__ __launch_bounds__(64) void pose_step_kernel(PoseScalars hp, const float* __restrict__ dL_dtau_sum,
                                                       const float* __restrict__ dL_dexposure,
                                                       const float* __restrict__ proj, float* __restrict__ state,
                                                       int32_t* __restrict__ status,
                                                       const int32_t* __restrict__ frame_status) {
  if (threadIdx.x != 0) return;
  // olsr_pose_step_gated: the frame the gradient came from was not usable (overflow, synchronisation error, depth cut-off
  // miss) — no optimiser step, nothing of the state changes, the matrices are re-derived from the pose as it is
  if (frame_status != nullptr && frame_status[1] != 0) {
    hp.has_grad = 0;
    status[0] = 0;
  }
  float T[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) T[i] = state[i];
  float tau[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (hp.has_grad && hp.step_on_device) {
    const double step = (double)(status[1] + 1);
    const double bc1 = 1.0 - pow(hp.beta1_d, step);
    const double bc2 = 1.0 - pow(hp.beta2_d, step);
    hp.bias_correction2_sqrt = (float)sqrt(bc2);
    hp.neg_step_rot = (float)(-(hp.lr_rot_d / bc1));
    hp.neg_step_trans = (float)(-(hp.lr_trans_d / bc1));
    hp.neg_step_exposure = (float)(-(hp.lr_exposure_d / bc1));
  }
  if (hp.has_grad) {
    // the rasterizer's dL_dtau is [rho | theta] (DGR/diff_gaussian_rasterization/__init__.py:383-385): rho is the
    // gradient of cam_trans_delta, theta of cam_rot_delta
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const float grad = dL_dtau_sum[i];
      float m = state[52 + i], v = state[58 + i];
      m = m + (grad - m) * hp.one_minus_beta1;
      v = v * hp.beta2 + hp.one_minus_beta2 * grad * grad;
      const float denom = sqrtf(v) / hp.bias_correction2_sqrt + hp.eps;
      tau[i] = 0.0f + (i < 3 ? hp.neg_step_trans : hp.neg_step_rot) * (m / denom);
      state[52 + i] = m;
      state[58 + i] = v;
    }
    if (hp.has_exposure) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float grad = dL_dexposure[i];
        float m = state[72 + i], v = state[74 + i];
        m = m + (grad - m) * hp.one_minus_beta1;
        v = v * hp.beta2 + hp.one_minus_beta2 * grad * grad;
        const float denom = sqrtf(v) / hp.bias_correction2_sqrt + hp.eps;
        state[70 + i] = state[70 + i] + hp.neg_step_exposure * (m / denom);
        state[72 + i] = m;
        state[74 + i] = v;
      }
    }
    // SE3_exp(tau): rho = tau[:3], theta = tau[3:]
    const float tx = tau[3], ty = tau[4], tz = tau[5];
    const float W[9] = {0.f, -tz, ty, tz, 0.f, -tx, -ty, tx, 0.f};  // skew_sym_mat
    float W2[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) W2[3 * r + c] = W[3 * r] * W[c] + W[3 * r + 1] * W[3 + c] + W[3 * r + 2] * W[6 + c];
    const float angle = sqrtf(tx * tx + ty * ty + tz * tz);
    float a, b, cV;  // R = I + a W + b W2,  V = I + b' W + c W2
    float bV;
    if (angle < 1e-5f) {
      a = 1.0f;
      b = 0.5f;
      bV = 0.5f;
      cV = 1.0f / 6.0f;
    } else {
      const float s = sinf(angle), co = cosf(angle);
      a = s / angle;
      b = (1.0f - co) / (angle * angle);
      bV = b;
      cV = (angle - s) / (angle * angle * angle);
    }
    float Rm[9], Vm[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const float eye = (i % 4 == 0) ? 1.0f : 0.0f;
      Rm[i] = (eye + a * W[i]) + b * W2[i];
      Vm[i] = (eye + W[i] * bV) + W2[i] * cV;
    }
    const float t3[3] = {Vm[0] * tau[0] + Vm[1] * tau[1] + Vm[2] * tau[2], Vm[3] * tau[0] + Vm[4] * tau[1] + Vm[5] * tau[2],
                         Vm[6] * tau[0] + Vm[7] * tau[1] + Vm[8] * tau[2]};
    // new_w2c = E @ T_w2c with E = [Rm t3; 0 1]
    float N[16];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        N[4 * r + c] = Rm[3 * r] * T[c] + Rm[3 * r + 1] * T[4 + c] + Rm[3 * r + 2] * T[8 + c] + t3[r] * T[12 + c];
    N[12] = T[12];
    N[13] = T[13];
    N[14] = T[14];
    N[15] = T[15];
    // update_RT keeps only R and T; the next T_w2c is rebuilt from them with a [0 0 0 1] last row
    N[12] = 0.f;
    N[13] = 0.f;
    N[14] = 0.f;
    N[15] = 1.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      T[i] = N[i];
      state[i] = N[i];
    }
    const float nrm = sqrtf(tau[0] * tau[0] + tau[1] * tau[1] + tau[2] * tau[2] + tau[3] * tau[3] + tau[4] * tau[4] +
                            tau[5] * tau[5]);
    status[0] = (nrm < hp.converged_threshold) ? 1 : 0;
    status[1] = status[1] + 1;
#pragma unroll
    for (int i = 0; i < 6; ++i) state[64 + i] = tau[i];
  }
  // world_view_transform = W2C^T
  float Vw[16];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) Vw[4 * r + c] = T[4 * c + r];
#pragma unroll
  for (int i = 0; i < 16; ++i) state[16 + i] = Vw[i];
  // full_proj_transform = world_view_transform @ projection_matrix (both as the callers hold them: transposes)
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c)
      state[32 + 4 * r + c] =
          Vw[4 * r] * proj[c] + Vw[4 * r + 1] * proj[4 + c] + Vw[4 * r + 2] * proj[8 + c] + Vw[4 * r + 3] * proj[12 + c];
  // camera_center = world_view_transform.inverse()[3, :3] = -R^-1 t, R^-1 by cofactors (a general inverse, as the
  // reference's: R is a product of rounded rotations, not exactly orthonormal)
  const float r00 = T[0], r01 = T[1], r02 = T[2], r10 = T[4], r11 = T[5], r12 = T[6], r20 = T[8], r21 = T[9], r22 = T[10];
  const float c00 = r11 * r22 - r12 * r21, c01 = r12 * r20 - r10 * r22, c02 = r10 * r21 - r11 * r20;
  const float det = r00 * c00 + r01 * c01 + r02 * c02;
  const float id = 1.0f / det;
  const float i00 = c00 * id, i01 = (r02 * r21 - r01 * r22) * id, i02 = (r01 * r12 - r02 * r11) * id;
  const float i10 = c01 * id, i11 = (r00 * r22 - r02 * r20) * id, i12 = (r02 * r10 - r00 * r12) * id;
  const float i20 = c02 * id, i21 = (r01 * r20 - r00 * r21) * id, i22 = (r00 * r11 - r01 * r10) * id;
  const float t0 = T[3], t1 = T[7], t2 = T[11];
  state[48] = -(i00 * t0 + i01 * t1 + i02 * t2);
  state[49] = -(i10 * t0 + i11 * t1 + i12 * t2);
  state[50] = -(i20 * t0 + i21 * t1 + i22 * t2);
  state[51] = 0.f;
}

void launch_pose_step(const olsr_pose_params& p, const float* dL_dtau_sum, const float* dL_dexposure, const float* proj,
                      float* state, int32_t* status, const int32_t* frame_status, hipStream_t st) {
  PoseScalars k{};
  const int step = p.step > 0 ? p.step : 1;
  const double bc1 = 1.0 - pow(p.beta1, (double)step);
  const double bc2 = 1.0 - pow(p.beta2, (double)step);
  k.one_minus_beta1 = (float)(1.0 - p.beta1);
  k.beta2 = (float)p.beta2;
  k.one_minus_beta2 = (float)(1.0 - p.beta2);
  k.bias_correction2_sqrt = (float)sqrt(bc2);
  k.eps = (float)p.eps;
  k.neg_step_rot = (float)(-(p.lr_rot / bc1));
  k.neg_step_trans = (float)(-(p.lr_trans / bc1));
  k.neg_step_exposure = (float)(-(p.lr_exposure / bc1));
  k.converged_threshold = (float)p.converged_threshold;
  k.has_grad = dL_dtau_sum != nullptr;
  k.has_exposure = dL_dexposure != nullptr;
  k.step_on_device = (p.step <= 0) ? 1 : 0;
  k.beta1_d = p.beta1;
  k.beta2_d = p.beta2;
  k.lr_rot_d = p.lr_rot;
  k.lr_trans_d = p.lr_trans;
  k.lr_exposure_d = p.lr_exposure;
  pose_step_kernel<<<1, 64, 0, st>>>(k, dL_dtau_sum, dL_dexposure, proj, state, status, frame_status);
}

}  // namespace olsr
