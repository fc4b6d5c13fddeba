"""What the capacity-bound sparse exchange costs on ONE lane, piece by piece (HIP events, 1 rank over RCCL): the three library
launches, the two collectives as torch.distributed issues them, and the whole sequence — on a bucket left by a view of the
room map (20 % of the rows live) or of the config-3 volume (2 %)."""
import os
import socket
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, ".")
from online_lang_splatting_amd.frame_shard import GradientBucket, GradLayout, RasterWorkspace  # noqa: E402
from online_lang_splatting_amd.scene import make_room_scene, make_scene  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "room"
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
dist.init_process_group("nccl", rank=0, world_size=1)
GradientBucket.exchange_single_rank = True
P, W, H, F, M = 500_000, 1200, 680, 15, 1
sc = make_room_scene(P, W, H, F, views=10, seed=3).scene if kind == "room" else make_scene(P, W, H, F, seed=3)
cam = sc.camera
g = dict(bg=sc.bg.to(dev), means3D=sc.means3D.to(dev), opacities=sc.opacities.to(dev), scales=sc.scales.to(dev),
         rotations=sc.rotations.to(dev), shs=sc.shs.to(dev), language=sc.language.to(dev))
c = dict(viewmatrix=cam.world_view_transform.to(dev), projmatrix=cam.full_proj_transform.to(dev),
         projmatrix_raw=cam.projection_matrix.to(dev), campos=cam.camera_center.to(dev), tanfovx=cam.tanfovx, tanfovy=cam.tanfovy)
ws = RasterWorkspace(P, W, H, F, M, 4_000_000, dev)
b = GradientBucket(P, GradLayout(M, F), dev, track_rows=True)
cot = [t.to(dev) for t in sc.cotangents(3)]


def frame():
    ws.set_scene(sh_degree=0, **c, **g)
    ws.forward()
    ws.backward(*cot, bucket=b, first=True, bucket_only=True)


frame()
live = int((b.flat != 0).any(1).sum())
cap = min(P, int(1.25 * live) + 4096)
print(kind, "live rows", live, "capacity", cap)


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n * 1e3, 1), round((time.perf_counter() - t0) / n * 1e6, 1)


print("frame alone (us gpu, us wall)", timed(frame))
print("frame + sparse exchange", timed(lambda: (frame(), b.sparse_all_reduce_capped(cap))))
print("frame + dense all_reduce", timed(lambda: (frame(), b.all_reduce())))
print("frame + two-phase", timed(lambda: (frame(), b.reduce_scatter_all_gather(0, 1))))
x4 = torch.zeros(2 * P, dtype=torch.int32, device=dev)
print("all_reduce MAX int32[2P] alone", timed(lambda: dist.all_reduce(x4, op=dist.ReduceOp.MAX)))
xs = torch.zeros(cap * 29 + 2 * P, device=dev)
print("all_reduce SUM packed alone", timed(lambda: dist.all_reduce(xs)))
GradientBucket.exchange_single_rank = False   # the library launches without the collectives
print("frame + sparse local work only", timed(lambda: (frame(), b.sparse_all_reduce_capped(cap))))
dist.destroy_process_group()
