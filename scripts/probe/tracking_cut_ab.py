"""Tracking iteration at config 3: plain loop with host / device step count, and the depth cut-off loop (wall clock per
iteration, 3 repeats of 60)."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from online_lang_splatting_amd.frame_shard import RasterWorkspace
from online_lang_splatting_amd.scene import default_camera, make_config_scene
from online_lang_splatting_amd.slam_iterations import PoseState, TrackingLoop
dev = torch.device("cuda:0")
sc = make_config_scene(3)
W, H, F, P, M = sc.camera.width, sc.camera.height, sc.F, sc.P, sc.shs.shape[1]
g = dict(bg=sc.bg.to(dev), means3D=sc.means3D.to(dev), opacities=sc.opacities.to(dev), scales=sc.scales.to(dev),
         rotations=sc.rotations.to(dev), shs=sc.shs.to(dev), language=sc.language.to(dev))
cam = default_camera(W, H)
proj = cam.projection_matrix.to(dev)
T_gt = torch.eye(4, device=dev)
tau0 = torch.tensor([0.02, -0.015, 0.01, 0.004, -0.006, 0.003])
th = tau0[3:]
Wm = torch.tensor([[0.0, -th[2], th[1]], [th[2], 0.0, -th[0]], [-th[1], th[0], 0.0]])
T0 = torch.eye(4); T0[:3, :3] = torch.eye(3) + Wm + 0.5 * Wm @ Wm; T0[:3, 3] = tau0[:3]; T0 = T0.to(dev)
CAP = 4_000_000
ws0 = RasterWorkspace(P, W, H, F, M, CAP, dev)
ps = PoseState(T_gt, proj, cam.tanfovx, cam.tanfovy)
ws0.set_scene(sh_degree=sc.sh_degree, **ps.camera(), **g)
o = ws0.forward(); gt_image, gt_depth = o["color"].clone(), o["depth"][0].clone()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
ONLY = sys.argv[2] if len(sys.argv) > 2 else None
for name, cut, devcount in (("plain host-count", False, False), ("plain device-count", False, True), ("cut device-count", True, True)):
    if ONLY and not name.startswith(ONLY):
        continue
    ws = RasterWorkspace(P, W, H, F, M, CAP, dev, depth_cut=cut)
    ps = PoseState(T_gt, proj, cam.tanfovx, cam.tanfovy, device_step_count=devcount)
    res = []
    for rep in range(3):
        ps.reset(T0); ws.reset_depth_cut()
        loop = TrackingLoop(ws, g, sc.sh_degree, ps, gt_image, gt_depth)
        for _ in range(5): loop.iteration()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(N): loop.iteration()
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        res.append((round(1e3 * el / N, 4), int(ps.status[1].item())))
    print(name, res, "R", ws.rendered()[0], "pose err", float((ps.T_w2c - T_gt).abs().max()))
