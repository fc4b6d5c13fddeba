"""Generates tests/golden/pose.npz: the reference's own pose update of the tracking loop, run here on its importable
Python (utils/pose_utils.update_pose, utils/camera_utils.Camera, torch.optim.Adam set up as in
utils/slam_frontend.py:180-213) for sequences of pose / exposure gradients.  Runs ONLY in the authoring container
(needs /root/reference); the committed .npz is data: inputs (start pose, projection matrix, the gradient fed at every
iteration) and expected outputs (R, T, world_view_transform, full_proj_transform, camera_center, exposure and the
convergence flag after every iteration).  Pins oracle/pose_oracle.py and, through it, olsr_pose_step."""
import math
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
sys.path.insert(0, REF)
from gaussian_splatting.utils.graphics_utils import getProjectionMatrix2, focal2fov  # noqa: E402
from utils.camera_utils import Camera  # noqa: E402
from utils.pose_utils import update_pose  # noqa: E402

g = torch.Generator().manual_seed(20260929)
out = {}
NSEQ, NIT = 4, 12
specs = [(1200, 680), (640, 480), (256, 256), (1920, 1080)]
for s in range(NSEQ):
    W, H = specs[s]
    fx = fy = W / 2.0
    cx, cy = (W - 1) / 2.0, (H - 1) / 2.0
    proj = getProjectionMatrix2(znear=0.01, zfar=100.0, fx=fx, fy=fy, cx=cx, cy=cy, W=W, H=H).transpose(0, 1)
    cam = Camera(s, None, None, torch.eye(4), proj, fx, fy, cx, cy, focal2fov(fx, W), focal2fov(fy, H), H, W, device="cpu")
    a, b = math.radians(4.0 * s - 5.0), math.radians(2.5 * s)
    Ry = torch.tensor([[math.cos(a), 0.0, math.sin(a)], [0.0, 1.0, 0.0], [-math.sin(a), 0.0, math.cos(a)]])
    Rx = torch.tensor([[1.0, 0.0, 0.0], [0.0, math.cos(b), -math.sin(b)], [0.0, math.sin(b), math.cos(b)]])
    R0, T0 = (Rx @ Ry).contiguous(), torch.tensor([0.12 * s, -0.07 * s, 0.03 * s + 0.01])
    cam.update_RT(R0, T0)
    # the optimiser of slam_frontend.tracking(): four parameter groups
    # (sequence 3: learning rates small enough for the small-angle branch of SO3_exp / V and for `converged`)
    lr_rot, lr_trans = (3e-6, 1e-5) if s == 3 else (0.003, 0.001)
    out[f"seq{s}_lr"] = np.array([lr_rot, lr_trans, 0.01])
    opt = torch.optim.Adam([
        {"params": [cam.cam_rot_delta], "lr": lr_rot},
        {"params": [cam.cam_trans_delta], "lr": lr_trans},
        {"params": [cam.exposure_a], "lr": 0.01},
        {"params": [cam.exposure_b], "lr": 0.01},
    ])
    # gradients of decaying magnitude
    scale = torch.tensor([10.0 ** (-0.45 * i) for i in range(NIT)]).view(NIT, 1)
    gtau = torch.randn(NIT, 6, generator=g) * scale   # [rho | theta]
    gexp = torch.randn(NIT, 2, generator=g) * scale
    if s == 2:
        gtau[5:] = 0.0  # exactly zero gradients: Adam still moves on its momentum
    rec = {k: [] for k in ("R", "T", "view", "full", "campos", "exposure", "converged", "tau")}
    for i in range(NIT):
        opt.zero_grad()
        cam.cam_trans_delta.grad = gtau[i, :3].clone()
        cam.cam_rot_delta.grad = gtau[i, 3:].clone()
        cam.exposure_a.grad = gexp[i, 0:1].clone()
        cam.exposure_b.grad = gexp[i, 1:2].clone()
        with torch.no_grad():
            opt.step()
            rec["tau"].append(torch.cat([cam.cam_trans_delta, cam.cam_rot_delta]).detach().clone().numpy())
            conv = update_pose(cam)
        rec["converged"].append(bool(conv))
        rec["R"].append(cam.R.clone().numpy())
        rec["T"].append(cam.T.clone().numpy())
        rec["view"].append(cam.world_view_transform.clone().numpy())
        rec["full"].append(cam.full_proj_transform.clone().numpy())
        rec["campos"].append(cam.camera_center.clone().numpy())
        rec["exposure"].append(np.array([float(cam.exposure_a.detach()), float(cam.exposure_b.detach())], dtype=np.float32))
    out[f"seq{s}_proj"] = proj.numpy()
    out[f"seq{s}_R0"] = R0.numpy()
    out[f"seq{s}_T0"] = T0.numpy()
    out[f"seq{s}_grad_tau"] = gtau.numpy()
    out[f"seq{s}_grad_exposure"] = gexp.numpy()
    for k, v in rec.items():
        out[f"seq{s}_{k}"] = np.stack(v)
out["num_seq"] = np.array(NSEQ)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pose.npz"), **out)
print("wrote pose.npz with", len(out), "arrays; converged flags:", [out[f"seq{s}_converged"].tolist() for s in range(NSEQ)])
