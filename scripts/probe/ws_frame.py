"""One frame in flight through the sync-free workspace path (what bench.py's `isolated` leg times), for kernel traces."""
import sys
import time

import torch

sys.path.insert(0, ".")
from online_lang_splatting_amd.frame_shard import GradientBucket, GradLayout, RasterWorkspace  # noqa: E402
from online_lang_splatting_amd.scene import make_config_scene  # noqa: E402

dev = torch.device("cuda:0")
sc = make_config_scene(3)
cam = sc.camera
P, W, H, F, M = sc.P, cam.width, cam.height, sc.F, sc.shs.shape[1]
g = dict(bg=sc.bg.to(dev), means3D=sc.means3D.to(dev), opacities=sc.opacities.to(dev), scales=sc.scales.to(dev),
         rotations=sc.rotations.to(dev), shs=sc.shs.to(dev), language=sc.language.to(dev))
c = dict(viewmatrix=cam.world_view_transform.to(dev), projmatrix=cam.full_proj_transform.to(dev),
         projmatrix_raw=cam.projection_matrix.to(dev), campos=cam.camera_center.to(dev), tanfovx=cam.tanfovx, tanfovy=cam.tanfovy)
ws = RasterWorkspace(P, W, H, F, M, 3_500_000, dev)
b = GradientBucket(P, GradLayout(M, F), dev, track_rows=True)
cot = [t.to(dev) for t in sc.cotangents(3)]


def frame():
    ws.set_scene(sh_degree=0, **c, **g)
    ws.forward()
    ws.backward(*cot, bucket=b, first=True, bucket_only=True)


for _ in range(20):
    frame()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(60):
    frame()
torch.cuda.synchronize()
print("workspace path ms per frame", round((time.perf_counter() - t0) / 60 * 1e3, 4))
