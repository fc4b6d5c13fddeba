"""Workload statistics of the room map on the GPU (what bench.py reports as config4_substitute.room_scene), quick form."""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from online_lang_splatting_amd import _C, _abi, _lib  # noqa: E402
from online_lang_splatting_amd.frame_shard import GradientBucket, GradLayout, RasterWorkspace  # noqa: E402
from online_lang_splatting_amd.scene import make_room_scene  # noqa: E402

dev = torch.device("cuda:0")
t0 = time.time()
rs = make_room_scene(500_000, 1200, 680, 15, views=10, seed=3)
print("built in", time.time() - t0, "s; P", rs.scene.P, "keyframes", rs.keyframes)
sc = rs.scene
P, W, H, F, M = sc.P, 1200, 680, 15, 1
g = dict(bg=sc.bg.to(dev), means3D=sc.means3D.to(dev), opacities=sc.opacities.to(dev), scales=sc.scales.to(dev),
         rotations=sc.rotations.to(dev), shs=sc.shs.to(dev), language=sc.language.to(dev))
out = {}
for v in (0, 3, 7):
    cam = rs.cameras[v]
    c = dict(viewmatrix=cam.world_view_transform.to(dev), projmatrix=cam.full_proj_transform.to(dev),
             projmatrix_raw=cam.projection_matrix.to(dev), campos=cam.camera_center.to(dev), tanfovx=cam.tanfovx, tanfovy=cam.tanfovy)
    ws = RasterWorkspace(P, W, H, F, M, 2_000_000, dev)
    bucket = GradientBucket(P, GradLayout(M, F), dev, track_rows=True)
    dc, dl, dd = [t.to(dev) for t in sc.cotangents(3)]

    def one():
        ws.set_scene(sh_degree=0, **c, **g)
        ws.forward()
        ws.backward(dc, dl, dd, bucket=bucket, first=True, bucket_only=True)
    for _ in range(5):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        one()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / 50
    _lib.set_profiling(True)
    for _ in range(10):
        one()
    per = {}
    for name, ms in _lib.stage_times():
        per.setdefault(name, []).append(ms)
    _lib.set_profiling(False)
    R, ovf = ws.rendered()
    L, _ = ws.backward_status()
    cnt = _C.state_field("geometry", ws.geom, "counters", P=P, F=F, dtype=torch.int32, count=8).cpu()
    live = int((bucket.flat != 0).any(1).sum())
    vis = int((ws.out["radii"] > 0).sum())
    out[v] = dict(ms_per_frame=round(1e3 * el, 4), fps=round(1 / el, 1), R_binned=R, R_rect=int(cnt[3]), visible=vis, live_rows=live,
                  live_of_visible=round(live / max(vis, 1), 4), gradient_rows_L=L, opacity_mean=float(ws.out["opacity"].mean()),
                  stage_ms={k: round(sum(x) / len(x), 4) for k, x in per.items()})
    print(v, json.dumps(out[v]))
    del ws, bucket
