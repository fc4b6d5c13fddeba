"""Can a tracking iteration (render -> tracking loss -> pose-only backward -> pose step) be replayed from a HIP graph?
    python scripts/probe/graph_tracking.py [config=3] [iters=60]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from online_lang_splatting_amd import _abi
from online_lang_splatting_amd.frame_shard import RasterWorkspace
from online_lang_splatting_amd.scene import make_config_scene
from online_lang_splatting_amd.slam_iterations import PoseState, TrackingLoop

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device("cuda:0")
sc = make_config_scene(cfg)
cam = sc.camera
P, W, H, F, M = sc.P, cam.width, cam.height, sc.F, sc.shs.shape[1]
g_dev, c_gt = bench.device_inputs(sc, cam, dev)
R0 = bench._sized_capacity(F, g_dev, c_gt, H, W, sc.sh_degree, dev, (15, _abi.BWD_REFERENCE, _abi.BINNING_ELLIPSE))
ws = RasterWorkspace(P, W, H, F, M, int(1.4 * R0) + (1 << 16), dev)
ws.set_scene(sh_degree=sc.sh_degree, **c_gt, **g_dev)
o = ws.forward()
gt_image, gt_depth = o["color"].clone(), o["depth"][0].clone()
T0 = torch.eye(4, device=dev)
T0[:3, 3] = torch.tensor([0.02, -0.015, 0.01])
T_gt = torch.eye(4, device=dev)
proj = c_gt["projmatrix_raw"]


def run(mode):
    pose = PoseState(T0, proj, c_gt["tanfovx"], c_gt["tanfovy"], device_step_count=True)
    loop = TrackingLoop(ws, g_dev, sc.sh_degree, pose, gt_image, gt_depth)
    graph = loop.capture() if mode == "graph" else None
    pose.reset(T0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        if graph is not None:
            graph.replay()
        else:
            loop.iteration()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / iters, pose.state.clone(), pose.status.clone()


ms_e, st_e, status_e = run("eager")
ms_g, st_g, status_g = run("graph")
print(f"eager {ms_e:.4f} ms/iteration, graph replay {ms_g:.4f} ms/iteration")
print("same pose state bit for bit:", bool(torch.equal(st_e, st_g)), "steps", status_e.tolist(), status_g.tolist())
print("pose error start", float((T0 - T_gt).abs().max()), "after", float((st_g[:16].view(4, 4) - T_gt).abs().max()))
