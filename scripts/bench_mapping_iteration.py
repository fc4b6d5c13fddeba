"""BASELINE.json configs[3] substitute: the Replica slam.py loop cannot run here (no data, no
SED / auto-encoder checkpoints, front-end dependencies absent), so this times the MAPPING ITERATION it spends its
GPU time in (utils/slam_backend.py:510-760) on synthetic data of the same shape, entirely on this library:

  for each of 12 views (10 window keyframes + 2 random, one shared set of Gaussians):
      render (olsr_forward_async, raw parameters: OLSR_ACT_*)            gaussian_renderer.render
      mapping loss + image cotangents (olsr_mapping_loss)                get_loss_mapping + language L1
      backward into the gradient bucket (olsr_backward, bucket)          loss.backward(), densification stats
  [all-reduce of the bucket when frame-sharded]
  fused Adam step on the bucket (olsr_adam_step)                         gaussians.optimizer.step()

500 k Gaussians, 1200x680, RGB + depth + 15 language channels, 192x192 language target, ground truth = renders of a
perturbed copy of the scene.  `--lanes N` keeps N views in flight (each lane has its own bucket; the buckets are summed
before the step).  Prints one JSON line."""
import argparse, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from online_lang_splatting_amd import _abi, losses
from online_lang_splatting_amd.frame_shard import FrameLanes, FusedAdam, GradLayout
from online_lang_splatting_amd.scene import CONFIGS, arc_cameras, make_scene

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--views", type=int, default=12)
ap.add_argument("--lanes", type=int, default=4)
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = CONFIGS[3]
P, W, H, F = cfg["P"], cfg["W"], cfg["H"], cfg["F"]
sc = make_scene(P, W, H, F, seed=3, max_sh_degree=0)
M = sc.shs.shape[1]
cams = arc_cameras(W, H, n=a.views)
# raw (pre-activation) parameters, as GaussianModel stores them
params = dict(means3D=sc.means3D.to(dev).contiguous(), shs=sc.shs.to(dev).contiguous(),
              opacities=torch.logit(sc.opacities.clamp(1e-4, 1 - 1e-4)).to(dev).contiguous(),
              scales=torch.log(sc.scales).to(dev).contiguous(), rotations=sc.rotations.to(dev).contiguous(),
              language=sc.language.to(dev).contiguous())
bg = sc.bg.to(dev)
camd = [dict(viewmatrix=c.world_view_transform.to(dev), projmatrix=c.full_proj_transform.to(dev),
             projmatrix_raw=c.projection_matrix.to(dev), campos=c.camera_center.to(dev), tanfovx=c.tanfovx,
             tanfovy=c.tanfovy) for c in cams]
lanes = FrameLanes(a.lanes, P, W, H, F, M, 4_000_000, dev)
layout = GradLayout(M, F)
adam = FusedAdam(P, layout, dev)
lrs = dict(xyz=1.6e-4, sh_dc=2.5e-3, sh_rest=1.25e-4, opacity=0.05, scale=1e-3, rotation=1e-3, language=2.5e-3)
exposure = torch.zeros(2, device=dev)


def render(ws, cam):
    ws.set_scene(bg=bg, sh_degree=0, activations=_abi.ACT_ALL, **cam, **params)
    return ws.forward()


# ground truth: renders of the scene itself, perturbed so that every loss term has a gradient
g = torch.Generator().manual_seed(0)
ws0 = lanes.lanes[0][0]
gts = []
for cam in camd:
    o = render(ws0, cam)
    gts.append((torch.clamp(o["color"] + 0.05 * torch.randn(3, H, W, generator=g).to(dev), 0, 1).contiguous(),
                (o["depth"][0] * (1 + 0.02 * torch.randn(H, W, generator=g).to(dev))).contiguous(),
                torch.nn.functional.normalize(torch.randn(F, 192, 192, generator=g), dim=0).to(dev).contiguous()))
torch.cuda.synchronize()


def iteration():
    used = []
    for v, cam in enumerate(camd):
        ws, bucket, stream = lanes.next_lane()
        first = bucket not in used
        if first:
            used.append(bucket)
        with torch.cuda.stream(stream):
            out = render(ws, cam)
            lo = losses.mapping_loss(out["color"], out["depth"], out["language"], *gts[v], exposure)
            ws.backward(lo["dL_dimage"], lo["dL_dlanguage"], lo["dL_ddepth"], bucket=bucket, first=first, bucket_only=True)
    main = torch.cuda.current_stream(dev)
    for _, _, st in lanes.lanes:
        main.wait_stream(st)
    total = used[0]
    for b in used[1:]:
        total.sum_storage.add_(b.sum_storage)
        torch.maximum(total.max_radii, b.max_radii, out=total.max_radii)
    total.all_reduce()
    adam.step(total, params, lrs)
    for _, _, st in lanes.lanes:
        st.wait_stream(main)
    return lo


lo0 = iteration()
first_loss = float(lo0["loss"][0])
for _ in range(2):
    iteration()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    lo = iteration()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.iters
print(json.dumps({"workload": f"mapping iteration: {a.views} views x (render + mapping loss + backward) + fused Adam, "
                              f"{P} Gaussians, {W}x{H}, F={F}, synthetic (BASELINE configs[3] substitute)",
                  "views_in_flight": len(lanes), "ms_per_iteration": round(1e3 * dt, 3),
                  "iterations_per_s": round(1 / dt, 2), "views_per_s": round(a.views / dt, 1),
                  "loss_of_last_view_first_iteration": round(first_loss, 6),
                  "loss_of_last_view_final_iteration": round(float(lo["loss"][0]), 6)}))
