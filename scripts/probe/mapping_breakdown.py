#!/usr/bin/env python3
"""Where does the 12-view mapping iteration lose against the four-in-flight headline rate?  Variants of the iteration of
bench.py's config-4 substitute, each timed the same way (4 lanes, config 3):
  base            12 arc views, raw parameters (activations in the kernels), fused loss, bucket add for later views, Adam
  same_view       the identity view 12 times (what the headline renders)
  no_activations  activated parameters handed over (the reference's calling convention)
  assign_only     every view OVERWRITES its lane's bucket (wrong sums; isolates the read-modify-write of the add)
  no_loss         fixed cotangents instead of the loss
  no_adam / no_lane_sum
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from online_lang_splatting_amd import _abi  # noqa: E402
from online_lang_splatting_amd.frame_shard import FrameLanes  # noqa: E402
from online_lang_splatting_amd.scene import CONFIGS, arc_cameras, make_scene  # noqa: E402
from online_lang_splatting_amd.slam_iterations import MappingStep  # noqa: E402

dev = torch.device("cuda:0")
cfg = CONFIGS[3]
P, W, H, F = cfg["P"], cfg["W"], cfg["H"], cfg["F"]
sc = make_scene(P, W, H, F, seed=3, max_sh_degree=cfg["max_sh_degree"])
M = sc.shs.shape[1]
g = dict(bg=sc.bg.to(dev), means3D=sc.means3D.to(dev), opacities=sc.opacities.to(dev), scales=sc.scales.to(dev),
         rotations=sc.rotations.to(dev), shs=sc.shs.to(dev), language=sc.language.to(dev))
views = 12


def cams_of(kind):
    cams = arc_cameras(W, H, n=views) if kind == "arc" else [arc_cameras(W, H, n=1)[0]] * views
    return [dict(viewmatrix=c.world_view_transform.to(dev), projmatrix=c.full_proj_transform.to(dev),
                 projmatrix_raw=c.projection_matrix.to(dev), campos=c.camera_center.to(dev), tanfovx=c.tanfovx,
                 tanfovy=c.tanfovy) for c in cams]


lanes = FrameLanes(4, P, W, H, F, M, 4_500_000, dev)
lrs = dict(xyz=1.6e-4, sh_dc=2.5e-3, sh_rest=1.25e-4, opacity=0.05, scale=1e-3, rotation=1e-3, language=2.5e-3)
raw = dict(means3D=g["means3D"].clone(), shs=g["shs"].clone(),
           opacities=torch.logit(g["opacities"].clamp(1e-4, 1 - 1e-4)).contiguous(),
           scales=torch.log(g["scales"]).contiguous(), rotations=g["rotations"].clone(), language=g["language"].clone())
act = {k: g[k].clone() for k in raw}
gen = torch.Generator().manual_seed(0)


def targets_for(camd, params, a):
    st = MappingStep(lanes, params, g["bg"], sc.sh_degree, camd, None, lrs, exposure=torch.zeros(2, device=dev), activations=a)
    ws0 = lanes.lanes[0][0]
    out = []
    for c in camd:
        o = st.render(ws0, c)
        out.append((torch.clamp(o["color"] + 0.05 * torch.randn(3, H, W, generator=gen).to(dev), 0, 1).contiguous(),
                    (o["depth"][0] * (1 + 0.02 * torch.randn(H, W, generator=gen).to(dev))).contiguous(),
                    torch.nn.functional.normalize(torch.randn(F, 192, 192, generator=gen), dim=0).to(dev)))
    return out


def run(name, cam_kind="arc", activations=_abi.ACT_ALL, patch=None, iters=4):
    params = {k: v.clone() for k, v in (raw if activations else act).items()}
    camd = cams_of(cam_kind)
    st = MappingStep(lanes, params, g["bg"], sc.sh_degree, camd, targets_for(camd, params, activations), lrs,
                     exposure=torch.zeros(2, device=dev), activations=activations)
    if patch:
        patch(st)
    for _ in range(2):
        st.iteration()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        st.iteration()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / iters
    Rs = [int(ws.num_rendered.cpu()[0]) for ws, _, _ in lanes.lanes]
    print(f"{name:16s} {ms:7.3f} ms / iteration = {ms / views:.4f} ms / view   R of the lanes' last views {Rs}", flush=True)
    return ms


res = {}
res["base"] = run("base")
res["same_view"] = run("same_view", cam_kind="same")
# (no_activations removed: Adam on already-activated opacities / scales leaves their domain; not a valid variant)


def assign_only(st):
    for ws, _, _ in lanes.lanes:
        orig = ws.backward
        ws.backward = (lambda o: (lambda *a, **k: o(*a, **{**k, "first": True})))(orig)


res["assign_only"] = run("assign_only", patch=assign_only)
for ws, _, _ in lanes.lanes:
    if "backward" in ws.__dict__:
        del ws.__dict__["backward"]


def no_loss(st):
    cot = [t.to(dev) for t in sc.cotangents(3)]
    for ws, _, _ in lanes.lanes:
        ws.forward_loss = (lambda w: (lambda *a, **k: (w.forward(), dict(loss=torch.zeros(4, device=dev), dL_dimage=cot[0],
                                                                            dL_dlanguage=cot[1], dL_ddepth=cot[2]))[1]))(ws)


res["no_loss"] = run("no_loss", patch=no_loss)
for ws, _, _ in lanes.lanes:
    if "forward_loss" in ws.__dict__:
        del ws.__dict__["forward_loss"]


def no_adam(st):
    st.adam.step = lambda *a, **k: None


res["no_adam"] = run("no_adam", patch=no_adam)
res["two_kernel_loss"] = run("two_kernel_loss", patch=lambda st: setattr(st, "fused", False))
out = os.path.join(ROOT, "gpurun_out", "mapping_breakdown.json")
if os.path.isdir(os.path.dirname(out)):
    json.dump(res, open(out, "w"), indent=1)
