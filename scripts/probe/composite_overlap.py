"""How the composite kernels of several frames in flight share the GPU, read from the library's own clock stamps
(olsr_debug_composite_stamps) instead of a profiler: config 3 (or --room), N lanes, steady state."""
import sys

import torch

sys.path.insert(0, ".")
from online_lang_splatting_amd import _lib  # noqa: E402
from online_lang_splatting_amd.frame_shard import FrameLanes  # noqa: E402
from online_lang_splatting_amd.scene import make_config_scene, make_room_scene  # noqa: E402

room = "--room" in sys.argv
nl = int([a for a in sys.argv[1:] if a.isdigit()][0]) if any(a.isdigit() for a in sys.argv[1:]) else 4
dev = torch.device("cuda:0")
sc = make_room_scene(500_000, 1200, 680, 15, views=10, seed=3).scene if room else make_config_scene(3)
cam = sc.camera
P, W, H, F, M = sc.P, cam.width, cam.height, sc.F, sc.shs.shape[1]
g = dict(bg=sc.bg.to(dev), means3D=sc.means3D.to(dev), opacities=sc.opacities.to(dev), scales=sc.scales.to(dev),
         rotations=sc.rotations.to(dev), shs=sc.shs.to(dev), language=sc.language.to(dev))
c = dict(viewmatrix=cam.world_view_transform.to(dev), projmatrix=cam.full_proj_transform.to(dev),
         projmatrix_raw=cam.projection_matrix.to(dev), campos=cam.camera_center.to(dev), tanfovx=cam.tanfovx, tanfovy=cam.tanfovy)
lanes = FrameLanes(nl, P, W, H, F, M, 3_600_000, dev)
cot = [t.to(dev) for t in sc.cotangents(3)]


def step(lane):
    ws, b, st = lane
    with torch.cuda.stream(st):
        ws.set_scene(sh_degree=0, **c, **g)
        ws.forward()
        ws.backward(*cot, bucket=b, first=True, bucket_only=True)


for _ in range(60):
    step(lanes.next_lane())
torch.cuda.synchronize()
N = 120
buf = torch.zeros(2 * 4 * (N + 8), dtype=torch.int64, device=dev)
_lib.lib().olsr_debug_composite_stamps(buf.data_ptr(), 4 * (N + 8))
for _ in range(N):
    step(lanes.next_lane())
torch.cuda.synchronize()
_lib.lib().olsr_debug_composite_stamps(None, 0)
t = buf.cpu().view(-1, 2)
t = t[t[:, 0] > 0]
ticks, tag = t[:, 0].double() / 100.0, t[:, 1]        # 100 MHz -> microseconds
kind, stream = tag & 0xFF, tag >> 8
# composite windows: (kind 0 -> 1) and (2 -> 3) of the same stream, in issue order
wins = []
open_ = {}
for i in range(len(t)):
    k, s = int(kind[i]), int(stream[i])
    if k in (0, 2):
        open_[(s, k)] = float(ticks[i])
    elif (s, k - 1) in open_:
        wins.append((open_.pop((s, k - 1)), float(ticks[i]), k == 1))
wins.sort()
wins = wins[len(wins) // 4:]     # steady state
t0, t1 = wins[0][0], max(w[1] for w in wins)
ev = sorted([(a, 1) for a, b, f in wins] + [(b, -1) for a, b, f in wins])
cov = {}
n, prev = 0, ev[0][0]
for x, d in ev:
    cov[n] = cov.get(n, 0.0) + (x - prev)
    n += d
    prev = x
tot = sum(cov.values())
frames = len(wins) / 2
print(f"{'room' if room else 'volume'}, {nl} lanes: {frames:.0f} frames in {t1 - t0:.0f} us = {(t1 - t0) / frames:.1f} us per frame")
print("composite windows open (eligible or executing): " + ", ".join(f"{k}: {100 * v / tot:.1f} %" for k, v in sorted(cov.items())))
fw = [b - a for a, b, f in wins if f]
bw = [b - a for a, b, f in wins if not f]
print(f"window length, forward composite + tile order: median {sorted(fw)[len(fw) // 2]:.1f} us; backward composite: median {sorted(bw)[len(bw) // 2]:.1f} us")
