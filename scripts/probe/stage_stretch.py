"""Stage durations (the library's own HIP events) with 1, 2 and 4 lanes in flight: how much every stage of a frame is
stretched by the other lanes' kernels.  usage: stage_stretch.py [lanes ...]"""
import sys

import torch

sys.path.insert(0, ".")
from online_lang_splatting_amd import _lib  # noqa: E402
from online_lang_splatting_amd.frame_shard import FrameLanes  # noqa: E402
from online_lang_splatting_amd.scene import make_config_scene  # noqa: E402

dev = torch.device("cuda:0")
sc = make_config_scene(3)
cam = sc.camera
P, W, H, F, M = sc.P, cam.width, cam.height, sc.F, sc.shs.shape[1]
g = dict(bg=sc.bg.to(dev), means3D=sc.means3D.to(dev), opacities=sc.opacities.to(dev), scales=sc.scales.to(dev),
         rotations=sc.rotations.to(dev), shs=sc.shs.to(dev), language=sc.language.to(dev))
c = dict(viewmatrix=cam.world_view_transform.to(dev), projmatrix=cam.full_proj_transform.to(dev),
         projmatrix_raw=cam.projection_matrix.to(dev), campos=cam.camera_center.to(dev), tanfovx=cam.tanfovx, tanfovy=cam.tanfovy)
cot = [t.to(dev) for t in sc.cotangents(3)]
for nl in [int(a) for a in sys.argv[1:]] or [1, 2, 4]:
    lanes = FrameLanes(nl, P, W, H, F, M, 3_600_000, dev)

    def step(lane):
        ws, b, st = lane
        with torch.cuda.stream(st):
            ws.set_scene(sh_degree=0, **c, **g)
            ws.forward()
            ws.backward(*cot, bucket=b, first=True, bucket_only=True)
    for _ in range(40):
        step(lanes.next_lane())
    torch.cuda.synchronize()
    _lib.set_profiling(True)
    for _ in range(48):
        step(lanes.next_lane())
    per = {}
    for name, ms in _lib.stage_times():
        per.setdefault(name, []).append(ms)
    _lib.set_profiling(False)
    med = {k: round(1e3 * sorted(v)[len(v) // 2], 1) for k, v in per.items()}
    print(nl, "lanes, stage medians (us):", med, "sum", round(sum(med.values()), 1))
    del lanes
