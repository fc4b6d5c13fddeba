"""Generates tests/golden/gradients.npz: GRADIENTS from autograd through the reference's importable Python (runs ONLY in the
authoring container: needs /root/reference; the committed .npz is data — inputs, cotangents and expected gradients).

The forward goldens (conventions.npz, geometry.npz) pin values; this one pins the three analytic backward routines of the
CUDA reference that have a Python counterpart autograd can differentiate (missing #4):

  * computeCov3D backward (CR/backward.cu:350-413)  <-  autograd through
    GaussianModel.build_covariance_from_scaling_rotation (gaussian_splatting/scene/gaussian_model.py:119-124):
    L = build_scaling_rotation(modifier * scaling, rotation); strip_symmetric(L @ L^T)
    (gaussian_splatting/utils/general_utils.py:109-148), loss = sum(w_cov * cov3D).
    build_rotation NORMALISES the quaternion, the CUDA kernel does not (CR/forward.cu:130): for a unit quaternion autograd's
    gradient is the kernel's projected on the tangent space, (I - q q^T) g — the test applies that projection to the
    oracle's gradient; the scale gradient needs no correction.
  * computeColorFromSH backward (CR/backward.cu:21-145), incl. the clamp mask and the gradient through the view direction
    <-  autograd through the `convert_SHs_python` branch of gaussian_renderer.render
    (gaussian_splatting/gaussian_renderer/__init__.py:274-284: dir = (xyz - camera_center) normalised,
    clamp_min(eval_sh(deg, shs, dir) + 0.5, 0); eval_sh = gaussian_splatting/utils/sh_utils.py:55-126), degrees 0-3,
    loss = sum(w_col * colour).  Coefficients are drawn so that a good share of the channels clamps.
  * the projection Jacobian of the means (CR/backward.cu:571-590: dL_dmean from dL_dmean2D; :640-646: the depth row)
    <-  autograd through p_hom = [m, 1] @ full_proj_transform, ndc = p_hom.xy / (p_hom.w + 1e-7),
    depth = ([m, 1] @ world_view_transform).z with the matrices of utils/camera_utils.Camera (float64),
    loss = sum(w_ndc * ndc) + sum(w_depth * depth).  (The kernel's dL_dmean2D is the cotangent of the NDC coordinates:
    the composite backward multiplies its pixel-space gradient by 0.5 W / 0.5 H, CR/backward.cu:1150-1151.)
The helpers hard-code device="cuda"; torch.zeros is wrapped for the duration of the calls to drop it.
"""
import math
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
sys.path.insert(0, REF)
import gaussian_splatting.utils.general_utils as GU  # noqa: E402
from gaussian_splatting.utils.graphics_utils import focal2fov, getProjectionMatrix2  # noqa: E402
from gaussian_splatting.utils.sh_utils import eval_sh  # noqa: E402
from utils.camera_utils import Camera  # noqa: E402

_zeros = torch.zeros


def _cpu_zeros(*a, **k):
    k.pop("device", None)
    return _zeros(*a, **k)


g = torch.Generator().manual_seed(20260929)
out = {}

# ---- 3D covariance: d sum(w * cov3D) / d (scaling, rotation) ---------------------------------------------------------
P = 192
scales0 = torch.exp(torch.randn(P, 3, generator=g) * 0.8 - 2.5)
q0 = torch.randn(P, 4, generator=g)
rot0 = q0 / q0.norm(dim=1, keepdim=True)
w_cov = torch.randn(P, 6, generator=g)
out["cov_scales"], out["cov_rotations"], out["cov_cotangent"] = scales0.numpy(), rot0.numpy(), w_cov.numpy()
for i, mod in enumerate((1.0, 0.37, 2.5)):
    s = scales0.clone().requires_grad_(True)
    r = rot0.clone().requires_grad_(True)
    GU.torch.zeros = _cpu_zeros
    try:
        L = GU.build_scaling_rotation(mod * s, r)
        cov = GU.strip_symmetric(L @ L.transpose(1, 2))
    finally:
        GU.torch.zeros = _zeros
    (cov * w_cov).sum().backward()
    out[f"cov_modifier{i}"] = np.array(mod)
    out[f"cov_dL_dscales{i}"] = s.grad.numpy()
    out[f"cov_dL_drotations_tangent{i}"] = r.grad.numpy()   # = (I - q q^T) x the kernel's gradient, |q| = 1
out["cov_num"] = np.array(3)

# ---- camera of the SH and projection cases -----------------------------------------------------------------------------
W, H = 200, 150
fx = fy = W / 2.0
cx, cy = (W - 1) / 2.0, (H - 1) / 2.0
a, b = math.radians(7.0), math.radians(-4.0)
Ry = torch.tensor([[math.cos(a), 0.0, math.sin(a)], [0.0, 1.0, 0.0], [-math.sin(a), 0.0, math.cos(a)]])
Rx = torch.tensor([[1.0, 0.0, 0.0], [0.0, math.cos(b), -math.sin(b)], [0.0, math.sin(b), math.cos(b)]])
R = Rx @ Ry
T = torch.tensor([0.12, -0.05, 0.08])
proj = getProjectionMatrix2(znear=0.01, zfar=100.0, fx=fx, fy=fy, cx=cx, cy=cy, W=W, H=H).transpose(0, 1)
cam = Camera(0, None, None, torch.eye(4), proj, fx, fy, cx, cy, focal2fov(fx, W), focal2fov(fy, H), H, W, device="cpu")
cam.update_RT(R, T)
out["cam_spec"] = np.array([W, H, fx, fy, cx, cy], dtype=np.float64)
out["cam_R"], out["cam_T"] = R.numpy(), T.numpy()
out["cam_viewmatrix"] = cam.world_view_transform.numpy()
out["cam_projmatrix"] = cam.full_proj_transform.numpy()
out["cam_projmatrix_raw"] = cam.projection_matrix.numpy()
out["cam_center"] = cam.camera_center.numpy()

# points well inside the frustum (every one must be visible to the rasterizer: radii > 0)
N = 160
z = torch.rand(N, generator=g) * 3.0 + 0.8
xy = (torch.rand(N, 2, generator=g) * 1.6 - 0.8) * z[:, None] * torch.tensor([1.0, H / W])
pts = (torch.cat([xy, z[:, None]], dim=1) - T) @ R    # world points: R^T (p_cam - T) as row vectors
out["points"] = pts.numpy()

# ---- SH colour: d sum(w * clamp_min(eval_sh + 0.5, 0)) / d (shs, xyz) --------------------------------------------------
sh0 = torch.randn(N, 16, 3, generator=g) * 0.6
sh0[:, 0, :] -= 0.4   # (a good share of the channels goes below zero and clamps)
w_col = torch.randn(N, 3, generator=g)
out["sh_coeffs"], out["sh_cotangent"] = sh0.numpy(), w_col.numpy()
for deg in range(4):
    M = (deg + 1) ** 2
    feats = sh0[:, :M, :].clone().requires_grad_(True)      # pc.get_features: [P, M, 3]
    xyz = pts.clone().requires_grad_(True)
    shs_view = feats.transpose(1, 2).view(-1, 3, M)
    dir_pp = xyz - cam.camera_center.repeat(feats.shape[0], 1)
    dir_pp_normalized = dir_pp / dir_pp.norm(dim=1, keepdim=True)
    sh2rgb = eval_sh(deg, shs_view, dir_pp_normalized)
    colors = torch.clamp_min(sh2rgb + 0.5, 0.0)
    (colors * w_col).sum().backward()
    out[f"sh_colors_deg{deg}"] = colors.detach().numpy()
    out[f"sh_dL_dsh_deg{deg}"] = feats.grad.numpy()
    out[f"sh_dL_dmeans_deg{deg}"] = (xyz.grad if xyz.grad is not None else torch.zeros_like(xyz)).numpy()  # (degree 0: none)
    out[f"sh_clamped_fraction_deg{deg}"] = np.array(float((sh2rgb + 0.5 < 0).double().mean()))

# ---- projection: d (sum(w_ndc * ndc) + sum(w_depth * depth)) / d xyz ---------------------------------------------------
w_ndc = torch.randn(N, 2, generator=g).double()
w_depth = torch.randn(N, generator=g).double()
xyz = pts.double().clone().requires_grad_(True)
ph = torch.cat([xyz, torch.ones(N, 1, dtype=torch.float64)], dim=1)
hom = ph @ cam.full_proj_transform.double()
ndc = hom[:, :2] / (hom[:, 3:4] + 1e-7)
depth = (ph @ cam.world_view_transform.double())[:, 2]
((ndc * w_ndc).sum() + (depth * w_depth).sum()).backward()
out["proj_cotangent_ndc"], out["proj_cotangent_depth"] = w_ndc.numpy(), w_depth.numpy()
out["proj_ndc"], out["proj_depth"] = ndc.detach().numpy(), depth.detach().numpy()
out["proj_dL_dmeans"] = xyz.grad.numpy()

np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "gradients.npz"), **out)
print("wrote gradients.npz with", len(out), "arrays; clamped fractions:",
      [float(out[f"sh_clamped_fraction_deg{d}"]) for d in range(4)])
