"""ms per 12-view mapping iteration (MappingStep, fused loss, four views in flight) on the room map or the volume — the
timing leg of variant A/Bs (OLSR_LIB=... selects the library).  usage: mapping_time.py [room|volume] [iterations] [carry 0|1] [lanes]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from online_lang_splatting_amd import _abi  # noqa: E402
from online_lang_splatting_amd.frame_shard import FrameLanes  # noqa: E402
from online_lang_splatting_amd.scene import arc_cameras, make_room_scene, make_scene  # noqa: E402
from online_lang_splatting_amd.slam_iterations import MappingStep  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "room"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 12
carry = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
nlanes = int(sys.argv[4]) if len(sys.argv) > 4 else 4
dev = torch.device("cuda:0")
P, W, H, F = 500_000, 1200, 680, 15
if which == "room":
    rs = make_room_scene(P, W, H, F, views=10, random_views=2, seed=3)
    sc, cams, targets = rs.scene, rs.cameras, rs.targets
else:
    sc = make_scene(P, W, H, F, seed=3)
    cams = arc_cameras(W, H, n=12)
    gen = torch.Generator().manual_seed(1)
    targets = [(torch.rand(3, H, W, generator=gen), torch.rand(H, W, generator=gen) + 1.5, torch.rand(F, 192, 192, generator=gen))
               for _ in cams]
M = sc.shs.shape[1]
g_dev, _ = bench.device_inputs(sc, cams[0], dev)
camd = [bench.device_inputs(sc, c, dev)[1] for c in cams]
R0 = max(bench._sized_capacity(F, g_dev, c, H, W, 0, dev, (15, _abi.BWD_REFERENCE, _abi.BINNING_ELLIPSE)) for c in camd)
lanes = FrameLanes(nlanes, sc.P, W, H, F, M, int(1.5 * R0) + (1 << 16), dev)
params = dict(means3D=g_dev["means3D"].clone(), shs=g_dev["shs"].clone(),
              opacities=torch.logit(g_dev["opacities"].clamp(1e-4, 1 - 1e-4)).contiguous(),
              scales=torch.log(g_dev["scales"]).contiguous(), rotations=g_dev["rotations"].clone(), language=g_dev["language"].clone())
lrs = dict(xyz=1.6e-4, sh_dc=2.5e-3, sh_rest=1.25e-4, opacity=0.05, scale=1e-3, rotation=1e-3, language=2.5e-3)
stp = MappingStep(lanes, params, g_dev["bg"], 0, camd, targets, lrs, exposure=torch.zeros(2, device=dev), fused_loss=True,
                  carry_order=carry)
for _ in range(4):
    stp.iteration()
torch.cuda.synchronize(dev)
t0 = time.perf_counter()
for _ in range(iters):
    stp.iteration()
torch.cuda.synchronize(dev)
print(f"{which} lanes={nlanes} carry={int(carry)} lib={os.environ.get('OLSR_LIB', 'base')[-24:]} ms_per_iteration {1e3 * (time.perf_counter() - t0) / iters:.4f}")
