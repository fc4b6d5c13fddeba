"""Second half of the BASELINE.json configs[3] substitute: the front end of the Replica loop
spends its GPU time in the TRACKING iteration (utils/slam_frontend.py: tracking(), up to 100 per frame) — render the
fixed map from the current pose estimate, photometric + depth loss, back-propagate to the camera pose only, Adam step on
(cam_rot_delta, cam_trans_delta), update_pose.  Iterations depend on each other through the pose, so this is the
single-frame LATENCY of the path, not its throughput.  Entirely on this library, synthetic data of config 3's shape:

    render (olsr_forward_async)                                   gaussian_renderer.render
    tracking loss + image cotangents (olsr_tracking_loss)         get_loss_tracking (utils/slam_utils.py:92-121)
    pose-only backward (olsr_backward, dL_dtau_sum alone)         loss.backward()
    Adam step on the 6 pose increments, update_pose on device     pose_optimizer.step(), utils/pose_utils.py:update_pose
    [convergence test: one 4-byte read-back per iteration, like the reference's `converged = update_pose(...)`]

Prints one JSON line: iterations/s with and without the per-iteration convergence read-back."""
import argparse, json, math, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from online_lang_splatting_amd import losses
from online_lang_splatting_amd.frame_shard import RasterWorkspace
from online_lang_splatting_amd.scene import CONFIGS, default_camera, make_scene

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=100)
ap.add_argument("--config", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = CONFIGS[a.config]
P, W, H, F = cfg["P"], cfg["W"], cfg["H"], cfg["F"]
sc = make_scene(P, W, H, F, seed=a.config, max_sh_degree=cfg["max_sh_degree"])
M = sc.shs.shape[1]
params = dict(means3D=sc.means3D.to(dev), shs=sc.shs.to(dev), opacities=sc.opacities.to(dev), scales=sc.scales.to(dev),
              rotations=sc.rotations.to(dev), language=None if sc.language is None else sc.language.to(dev))
bg = sc.bg.to(dev)
cam_gt = default_camera(W, H)
proj = cam_gt.projection_matrix.to(dev)  # P^T, what the callers hold


def skew(v):
    z = torch.zeros((), device=v.device)
    return torch.stack([torch.stack([z, -v[2], v[1]]), torch.stack([v[2], z, -v[0]]), torch.stack([-v[1], v[0], z])])


def se3_exp(tau):
    """utils/pose_utils.py:SE3_exp — tau = [rho | theta] -> 4x4."""
    rho, theta = tau[:3], tau[3:]
    ang = theta.norm()
    Wm = skew(theta)
    W2 = Wm @ Wm
    small = ang < 1e-5
    a_ = torch.where(small, 1.0 - ang * ang / 6.0, torch.sin(ang) / ang.clamp_min(1e-12))
    b_ = torch.where(small, 0.5 - ang * ang / 24.0, (1.0 - torch.cos(ang)) / (ang * ang).clamp_min(1e-24))
    c_ = torch.where(small, 1.0 / 6.0 - ang * ang / 120.0, (ang - torch.sin(ang)) / (ang ** 3).clamp_min(1e-36))
    eye = torch.eye(3, device=tau.device)
    R = eye + a_ * Wm + b_ * W2
    V = eye + b_ * Wm + c_ * W2
    T = torch.eye(4, device=tau.device)
    T[:3, :3] = R
    T[:3, 3] = V @ rho
    return T


def matrices(T_w2c):
    view = T_w2c.t().contiguous()  # world_view_transform = W2C^T
    full = (view.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0).contiguous()
    campos = torch.linalg.inv(view)[3, :3].contiguous()
    return dict(viewmatrix=view, projmatrix=full, projmatrix_raw=proj, campos=campos, tanfovx=cam_gt.tanfovx,
                tanfovy=cam_gt.tanfovy)


# capacity from one synchronous render of the ground-truth view
from online_lang_splatting_amd import _C
e = torch.empty(0, device=dev)
T_gt = torch.eye(4, device=dev)
m0 = matrices(T_gt)
r = (_C.rasterize_language_gaussians(bg, params["means3D"], e, params["language"], params["opacities"], params["scales"],
                                     params["rotations"], 1.0, e, m0["viewmatrix"], m0["projmatrix"], proj, m0["tanfovx"],
                                     m0["tanfovy"], H, W, params["shs"], sc.sh_degree, m0["campos"], False, False)
     if F > 0 else
     _C.rasterize_gaussians(bg, params["means3D"], e, params["opacities"], params["scales"], params["rotations"], 1.0, e,
                            m0["viewmatrix"], m0["projmatrix"], proj, m0["tanfovx"], m0["tanfovy"], H, W, params["shs"],
                            sc.sh_degree, m0["campos"], False, False))
R0 = int(r[0])
gt_image = r[1].clone()
gt_depth = (r[7] if F > 0 else r[6])[0].clone()
del r
ws = RasterWorkspace(P, W, H, F, M, int(1.4 * R0) + (1 << 16), dev)

# start from a perturbed pose (a few cm / a degree off), like the constant-velocity prior of the front end
T_cur = se3_exp(torch.tensor([0.02, -0.015, 0.01, 0.004, -0.006, 0.003], device=dev)) @ T_gt
rot_delta = torch.zeros(3, device=dev, requires_grad=True)
trans_delta = torch.zeros(3, device=dev, requires_grad=True)
opt = torch.optim.Adam([dict(params=[rot_delta], lr=0.003), dict(params=[trans_delta], lr=0.001)])
exposure = torch.zeros(2, device=dev)
zero_lang = torch.zeros(max(F, 1), H, W, device=dev) if F > 0 else None  # the tracking loss has no language term


def iteration(check_convergence):
    global T_cur
    ws.set_scene(bg=bg, sh_degree=sc.sh_degree, **matrices(T_cur), **params)
    out = ws.forward()
    lo = losses.tracking_loss(out["color"], out["depth"], out["opacity"], gt_image, gt_depth, None, exposure)
    g = ws.backward(lo["dL_dimage"], zero_lang, lo["dL_ddepth"], pose_only=True)
    tau = g["dL_dtau_sum"]  # [rho | theta]
    trans_delta.grad = tau[:3].clone()
    rot_delta.grad = tau[3:].clone()
    opt.step()
    with torch.no_grad():
        step = torch.cat([trans_delta, rot_delta])
        T_cur = se3_exp(step) @ T_cur
        trans_delta.zero_()
        rot_delta.zero_()
    if check_convergence:
        return bool(step.norm() < 1e-4)  # the reference's `converged` (one small read-back)
    return False


res = {}
for name, chk in (("with_convergence_readback", True), ("without_readback", False)):
    for _ in range(5):
        iteration(chk)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        iteration(chk)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    res[name] = {"iterations_per_s": round(a.iters / el, 1), "ms_per_iteration": round(1e3 * el / a.iters, 4)}
err = float((T_cur - T_gt).abs().max())
print(json.dumps({"metric": "tracking iteration: render + tracking loss + pose-only backward + pose update, iterations/s",
                  "config": {"workload": f"BASELINE.json configs[{a.config - 1}] Gaussians and image size, one fixed map, "
                                         "pose optimised from a perturbed start", "P": P, "width": W, "height": H, "F": F,
                             "instances": R0},
                  **res, "pose_error_after": err, "iters": a.iters, "n_gpus": 1, "data": "synthetic"}))
