#!/usr/bin/env python3
"""Per-iteration wall times of the 12-view mapping iteration (4 lanes): are there stalls?"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from online_lang_splatting_amd import _abi
from online_lang_splatting_amd.frame_shard import FrameLanes
from online_lang_splatting_amd.scene import CONFIGS, arc_cameras, make_scene
from online_lang_splatting_amd.slam_iterations import MappingStep
dev = torch.device("cuda:0")
cfg = CONFIGS[3]
P, W, H, F = cfg["P"], cfg["W"], cfg["H"], cfg["F"]
sc = make_scene(P, W, H, F, seed=3, max_sh_degree=cfg["max_sh_degree"])
M = sc.shs.shape[1]
g = dict(bg=sc.bg.to(dev), means3D=sc.means3D.to(dev), opacities=sc.opacities.to(dev), scales=sc.scales.to(dev),
         rotations=sc.rotations.to(dev), shs=sc.shs.to(dev), language=sc.language.to(dev))
cams = [dict(viewmatrix=c.world_view_transform.to(dev), projmatrix=c.full_proj_transform.to(dev),
             projmatrix_raw=c.projection_matrix.to(dev), campos=c.camera_center.to(dev), tanfovx=c.tanfovx,
             tanfovy=c.tanfovy) for c in arc_cameras(W, H, n=12)]
lanes = FrameLanes(int(sys.argv[1]) if len(sys.argv) > 1 else 4, P, W, H, F, M, 4_500_000, dev)
lrs = dict(xyz=1.6e-4, sh_dc=2.5e-3, sh_rest=1.25e-4, opacity=0.05, scale=1e-3, rotation=1e-3, language=2.5e-3)
raw = dict(means3D=g["means3D"].clone(), shs=g["shs"].clone(), opacities=torch.logit(g["opacities"].clamp(1e-4, 1 - 1e-4)).contiguous(),
           scales=torch.log(g["scales"]).contiguous(), rotations=g["rotations"].clone(), language=g["language"].clone())
gen = torch.Generator().manual_seed(0)
st = MappingStep(lanes, raw, g["bg"], sc.sh_degree, cams, None, lrs, exposure=torch.zeros(2, device=dev))
ws0 = lanes.lanes[0][0]
tg = []
for c in cams:
    o = st.render(ws0, c)
    tg.append((torch.clamp(o["color"] + 0.05 * torch.randn(3, H, W, generator=gen).to(dev), 0, 1).contiguous(),
               (o["depth"][0] * (1 + 0.02 * torch.randn(H, W, generator=gen).to(dev))).contiguous(),
               torch.nn.functional.normalize(torch.randn(F, 192, 192, generator=gen), dim=0).to(dev)))
st.targets = tg
ts = []
for i in range(60):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st.iteration()
    torch.cuda.synchronize()
    ts.append(1e3 * (time.perf_counter() - t0))
print("iterations (ms):", " ".join(f"{t:.2f}" for t in ts))
s = sorted(ts[2:])
print(f"median {s[len(s)//2]:.3f}  p10 {s[len(s)//10]:.3f}  p90 {s[9*len(s)//10]:.3f}  max {s[-1]:.3f}")
