#!/usr/bin/env python3
"""Per-view stage times of the 12 arc views of the mapping substitute (one frame in flight, profiled), with the tile-order
hint coherent (same view repeated) and non-coherent (hint from another view), and without a hint."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from online_lang_splatting_amd import _lib  # noqa: E402
from online_lang_splatting_amd.frame_shard import GradientBucket, GradLayout, RasterWorkspace  # noqa: E402
from online_lang_splatting_amd.scene import CONFIGS, arc_cameras, make_scene  # noqa: E402

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
ws = RasterWorkspace(P, W, H, F, M, 4_500_000, dev)
bucket = GradientBucket(P, GradLayout(M, F), dev)
cot = [t.to(dev) for t in sc.cotangents(3)]


def frame(c):
    ws.set_scene(sh_degree=sc.sh_degree, **c, **g)
    ws.forward()
    ws.backward(*cot, bucket=bucket, first=True, bucket_only=True)


def stages(c, n=5, before=None):
    tot = {}
    for _ in range(n):
        if before is not None:
            before()
        torch.cuda.synchronize()
        _lib.set_profiling(True)
        frame(c)
        for name, ms in _lib.stage_times():
            tot[name] = tot.get(name, 0.0) + ms / n
        _lib.set_profiling(False)
    return tot


ident = torch.arange(ws.tile_order.numel(), dtype=torch.int32, device=dev)
for v, c in enumerate(cams):
    for _ in range(3):
        frame(c)
    coh = stages(c)
    R = int(ws.num_rendered.cpu()[0])
    other = cams[(v + 4) % 12]
    non = stages(c, before=lambda: frame(other))
    nohint = stages(c, before=lambda: ws.tile_order.copy_(ident))
    f = lambda d: f"fwd {d['render_forward']:.3f} bwd {d['render_backward']:.3f} sum {sum(d.values()):.3f}"  # noqa: E731
    print(f"view {v:2d} R={R:8d} L={ws.backward_status()[0]:7d} | coherent {f(coh)} | hint of view {(v + 4) % 12}: {f(non)} | "
          f"identity order: {f(nohint)} | pre {coh['preprocess']:.3f} dsort {coh['depth_sort']:.3f} emit {coh['emit']:.3f} "
          f"tsort {coh['tile_sort']:.3f} rc {coh['row_compaction']:.3f} pb {coh['preprocess_backward']:.3f}", flush=True)

# ---- could the order be derived from THIS frame's list lengths (known after the tile sort, before the composite)? ----------
from online_lang_splatting_amd import _C  # noqa: E402
nt = ws.tile_order.numel()
q, r = nt >> 3, nt & 7
starts = [(x * (q + 1) if x < r else r * (q + 1) + (x - r) * q) for x in range(8)] + [nt]


def order_from(weights):
    out = torch.empty(nt, dtype=torch.int32, device=dev)
    for x in range(8):
        a, b = starts[x], starts[x + 1]
        idx = torch.argsort(weights[a:b], descending=True, stable=True)
        out[a:b] = (idx + a).to(torch.int32)
    return out


print("order from this frame's list lengths:")
for v in (0, 3, 5, 8, 11):
    c = cams[v]
    for _ in range(3):
        frame(c)
    coh = stages(c)
    rg = _C.state_field("image", ws.img, "ranges", W=W, H=H, dtype=torch.int32, count=2 * nt).view(-1, 2).long()
    lens = (rg[:, 1] - rg[:, 0]).clamp(min=0)
    work = _C.state_field("image", ws.img, "tile_work", W=W, H=H, dtype=torch.int32, count=2 * nt)[:nt].long()
    res = {}
    for name, wgt in (("len", lens), ("min(len,128)", lens.clamp(max=128)), ("min(len,256)", lens.clamp(max=256)),
                      ("measured_work", work)):
        o = order_from(wgt.float())
        res[name] = stages(c, before=lambda: ws.tile_order.copy_(o))["render_forward"]
    cc = torch.corrcoef(torch.stack([lens.float(), work.float()]))[0, 1].item()
    print(f"view {v:2d}: coherent {coh['render_forward']:.3f} | " + " ".join(f"{k} {x:.3f}" for k, x in res.items()) +
          f" | corr(len, work) {cc:.2f}; len p50/p99/max {int(lens.median())}/{int(lens.float().quantile(0.99))}/{int(lens.max())}")
