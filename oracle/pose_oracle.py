"""CPU restatement of one tracking iteration's pose update — TEST INFRASTRUCTURE ONLY (the checker of olsr_pose_step;
only tests/ may import it).  float32 numpy, operation for operation what the reference runs between two renders of its
tracking loop:

  torch.optim.Adam.step over (cam_rot_delta, cam_trans_delta, exposure_a, exposure_b)   utils/slam_frontend.py:180-233,
                                                                                          torch/optim/adam.py (single tensor)
  update_pose: tau = [trans | rot], new_w2c = SE3_exp(tau) @ T_w2c, |tau| < thr          utils/pose_utils.py:61-97
  SO3_exp / V with their small-angle branches                                            utils/pose_utils.py:26-58
  Camera.world_view_transform / full_proj_transform / camera_center                      utils/camera_utils.py:103-117

Pinned by tests/golden/pose.npz, generated from the reference's own functions (tests/golden/make_golden_pose.py,
tests/test_pose_oracle_golden.py)."""
import math

import numpy as np

f32 = np.float32


def skew(x):
    z = f32(0)
    return np.array([[z, -x[2], x[1]], [x[2], z, -x[0]], [-x[1], x[0], z]], dtype=f32)


def se3_exp(tau):
    """utils/pose_utils.py:61-76 (tau = [rho | theta])."""
    tau = np.asarray(tau, dtype=f32)
    rho, theta = tau[:3], tau[3:]
    W = skew(theta)
    W2 = (W @ W).astype(f32)
    angle = f32(np.sqrt(f32(theta[0] * theta[0] + theta[1] * theta[1] + theta[2] * theta[2])))
    eye = np.eye(3, dtype=f32)
    if angle < 1e-5:
        R = eye + W + f32(0.5) * W2
        V = eye + f32(0.5) * W + f32(1.0 / 6.0) * W2
    else:
        s, c = f32(np.sin(angle)), f32(np.cos(angle))
        R = eye + (s / angle) * W + ((f32(1) - c) / (angle * angle)) * W2
        V = eye + W * ((f32(1) - c) / (angle * angle)) + W2 * ((angle - s) / (angle * angle * angle))
    T = np.eye(4, dtype=f32)
    T[:3, :3] = R
    T[:3, 3] = (V @ rho).astype(f32)
    return T


class PoseOracle:
    """State of one tracked frame: pose, exposure, Adam moments."""

    def __init__(self, R, T, proj, lr_rot=0.003, lr_trans=0.001, lr_exposure=0.01, betas=(0.9, 0.999), eps=1e-8,
                 converged_threshold=1e-4):
        self.T_w2c = np.eye(4, dtype=f32)
        self.T_w2c[:3, :3] = np.asarray(R, dtype=f32)
        self.T_w2c[:3, 3] = np.asarray(T, dtype=f32)
        self.proj = np.asarray(proj, dtype=f32)
        self.lr = (lr_rot, lr_trans, lr_exposure)
        self.betas, self.eps, self.thr = betas, eps, converged_threshold
        self.m = np.zeros(8, dtype=f32)   # [trans 3 | rot 3 | exposure 2]
        self.v = np.zeros(8, dtype=f32)
        self.exposure = np.zeros(2, dtype=f32)
        self.steps = 0
        self.tau = np.zeros(6, dtype=f32)
        self.converged = False

    def step(self, grad_tau, grad_exposure=None):
        """grad_tau = [rho | theta] (gradient of cam_trans_delta | cam_rot_delta), as the rasterizer's dL_dtau sums."""
        self.steps += 1
        b1, b2 = self.betas
        bc1 = 1.0 - b1 ** self.steps
        bc2 = 1.0 - b2 ** self.steps
        one_m_b1, b2f, one_m_b2 = f32(1.0 - b1), f32(b2), f32(1.0 - b2)
        bc2_sqrt, eps = f32(math.sqrt(bc2)), f32(self.eps)
        g = np.zeros(8, dtype=f32)
        g[:6] = np.asarray(grad_tau, dtype=f32)
        n = 6
        if grad_exposure is not None:
            g[6:] = np.asarray(grad_exposure, dtype=f32)
            n = 8
        lr = [self.lr[1]] * 3 + [self.lr[0]] * 3 + [self.lr[2]] * 2
        delta = np.zeros(8, dtype=f32)
        for i in range(n):
            m = f32(self.m[i] + f32(g[i] - self.m[i]) * one_m_b1)            # exp_avg.lerp_(grad, 1 - beta1)
            v = f32(f32(self.v[i] * b2f) + f32(f32(one_m_b2 * g[i]) * g[i]))  # mul_(beta2).addcmul_(grad, grad, 1 - beta2)
            denom = f32(f32(np.sqrt(v)) / bc2_sqrt + eps)
            delta[i] = f32(f32(-(lr[i] / bc1)) * f32(m / denom))             # param (= 0) .addcdiv_(exp_avg, denom, -step)
            self.m[i], self.v[i] = m, v
        self.tau = delta[:6].copy()
        self.exposure = (self.exposure + delta[6:]).astype(f32)
        self.T_w2c = (se3_exp(self.tau) @ self.T_w2c).astype(f32)
        self.T_w2c[3] = (0, 0, 0, 1)
        t = self.tau
        self.converged = bool(f32(np.sqrt(f32(t[0] * t[0] + t[1] * t[1] + t[2] * t[2] + t[3] * t[3] + t[4] * t[4] + t[5] * t[5])))
                              < self.thr)
        return self.converged

    @property
    def viewmatrix(self):       # world_view_transform = getWorld2View2(R, T)^T
        return self.T_w2c.T.copy()

    @property
    def projmatrix(self):       # full_proj_transform = world_view_transform @ projection_matrix
        return (self.viewmatrix @ self.proj).astype(f32)

    @property
    def campos(self):           # world_view_transform.inverse()[3, :3]
        return np.linalg.inv(self.viewmatrix.astype(np.float64))[3, :3].astype(f32)
