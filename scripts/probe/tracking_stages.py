"""Stage times (library events) and wall time of the dependent tracking iteration on the room map or the volume.
usage: tracking_stages.py [room|volume]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from online_lang_splatting_amd import _abi, _lib  # noqa: E402
from online_lang_splatting_amd.frame_shard import RasterWorkspace  # noqa: E402
from online_lang_splatting_amd.scene import default_camera, make_room_scene, make_scene  # noqa: E402
from online_lang_splatting_amd.slam_iterations import PoseState, TrackingLoop  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "room"
dev = torch.device("cuda:0")
P, W, H, F = 500_000, 1200, 680, 15
if which == "room":
    rs = make_room_scene(P, W, H, F, views=10, random_views=2, seed=3)
    sc, cam0, tgt = rs.scene, rs.cameras[0], rs.targets[0]
else:
    sc = make_scene(P, W, H, F, seed=3)
    cam0 = default_camera(W, H)
    g = torch.Generator().manual_seed(1)
    tgt = (torch.rand(3, H, W, generator=g), torch.rand(H, W, generator=g) + 1.5, None)
M = sc.shs.shape[1]
g_dev, c0 = bench.device_inputs(sc, cam0, dev)
R0 = bench._sized_capacity(F, g_dev, c0, H, W, 0, dev, (15, _abi.BWD_REFERENCE, _abi.BINNING_ELLIPSE))
ws = RasterWorkspace(sc.P, W, H, F, M, int(1.5 * R0) + (1 << 16), dev)
T = torch.eye(4)
T[:3, :3], T[:3, 3] = cam0.R, cam0.T
pose = PoseState(T.to(dev), cam0.projection_matrix.to(dev), cam0.tanfovx, cam0.tanfovy)
loop = TrackingLoop(ws, g_dev, 0, pose, tgt[0].to(dev), tgt[1].to(dev))
for _ in range(10):
    loop.iteration()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(60):
    loop.iteration()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 60
_lib.set_profiling(True)
for _ in range(10):
    loop.iteration()
per = {}
for name, ms in _lib.stage_times():
    per.setdefault(name, []).append(ms)
_lib.set_profiling(False)
st = {k: round(sum(v) / len(v), 4) for k, v in per.items()}
print(json.dumps({"scene": which, "ms_per_iteration": round(1e3 * wall, 4), "stage_ms": st, "stage_sum": round(sum(st.values()), 4)}))
