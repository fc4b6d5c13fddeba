"""Generates tests/golden/conventions.npz from the reference's importable Python helpers.

Runs ONLY in the authoring container (needs /root/reference); the committed .npz is what the
tests read.  Fixtures are data (inputs + expected outputs), no reference source travels.

Pins:
  * spherical-harmonics evaluation: gaussian_splatting/utils/sh_utils.py eval_sh, degrees 0..3,
    plus the `+0.5, clamp at 0` of gaussian_renderer/__init__.py:274-284;
  * camera matrices the caller hands to the rasterizer: getProjectionMatrix2, getWorld2View2
    (gaussian_splatting/utils/graphics_utils.py) and utils/camera_utils.Camera's
    world_view_transform / full_proj_transform / camera_center / projection_matrix;
  * pose parametrisation: utils/pose_utils.SE3_exp (tau = [rho | theta]).
"""
import math
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
sys.path.insert(0, REF)
# utils/camera_utils imports utils/slam_utils, which imports cv2-free code but hard-codes nothing at import
from gaussian_splatting.utils.graphics_utils import getProjectionMatrix2, getWorld2View2, focal2fov  # noqa: E402
from gaussian_splatting.utils.sh_utils import eval_sh  # noqa: E402
from utils.pose_utils import SE3_exp  # noqa: E402
from utils.camera_utils import Camera  # noqa: E402

g = torch.Generator().manual_seed(20250614)
out = {}

# ---- SH
P = 64
dirs = torch.randn(P, 3, generator=g)
dirs = dirs / dirs.norm(dim=1, keepdim=True)
sh = torch.randn(P, 16, 3, generator=g) * 0.5       # [P, M, 3] as the rasterizer receives it
out["sh_dirs"] = dirs.numpy()
out["sh_coeffs"] = sh.numpy()
for deg in range(4):
    # eval_sh wants [..., C, (deg+1)^2]
    res = eval_sh(deg, sh.transpose(1, 2), dirs)
    out[f"sh_rgb_deg{deg}"] = torch.clamp_min(res + 0.5, 0.0).numpy()
    out[f"sh_raw_deg{deg}"] = res.numpy()

# ---- cameras
cams = []
specs = [(256, 256), (640, 480), (1200, 680), (1920, 1080), (45, 30)]
for i, (W, H) in enumerate(specs):
    fx = fy = W / 2.0
    cx, cy = (W - 1) / 2.0, (H - 1) / 2.0
    a = math.radians(5.0 * i - 7.0)
    R = torch.tensor([[math.cos(a), 0.0, math.sin(a)], [0.0, 1.0, 0.0], [-math.sin(a), 0.0, math.cos(a)]])
    b = math.radians(3.0 * i)
    Rx = torch.tensor([[1.0, 0.0, 0.0], [0.0, math.cos(b), -math.sin(b)], [0.0, math.sin(b), math.cos(b)]])
    R = Rx @ R
    T = torch.tensor([0.1 * i, -0.05 * i, 0.02 * i])
    proj = getProjectionMatrix2(znear=0.01, zfar=100.0, fx=fx, fy=fy, cx=cx, cy=cy, W=W, H=H).transpose(0, 1)
    gt_T = torch.eye(4)
    cam = Camera(i, None, None, gt_T, proj, fx, fy, cx, cy, focal2fov(fx, W), focal2fov(fy, H), H, W, device="cpu")
    cam.update_RT(R, T)
    out[f"cam{i}_spec"] = np.array([W, H, fx, fy, cx, cy], dtype=np.float64)
    out[f"cam{i}_R"] = R.numpy()
    out[f"cam{i}_T"] = T.numpy()
    out[f"cam{i}_viewmatrix"] = cam.world_view_transform.numpy()
    out[f"cam{i}_projmatrix"] = cam.full_proj_transform.numpy()
    out[f"cam{i}_projmatrix_raw"] = cam.projection_matrix.numpy()
    out[f"cam{i}_campos"] = cam.camera_center.numpy()
    out[f"cam{i}_tanfov"] = np.array([math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)])
    out[f"cam{i}_w2v2"] = getWorld2View2(R, T).numpy()
out["num_cams"] = np.array(len(specs))

# ---- SE3
taus = torch.randn(8, 6, generator=g) * 0.05
out["se3_tau"] = taus.numpy()
out["se3_exp"] = torch.stack([SE3_exp(t) for t in taus]).numpy()

np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "conventions.npz"), **out)
print("wrote conventions.npz with", len(out), "arrays")
