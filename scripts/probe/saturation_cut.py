#!/usr/bin/env python3
"""What would a per-tile depth cut-off keep?  Config 3 (and one arc view): per tile the list position at which the forward
composite stops (every pixel saturated) or the list end; instances kept if only entries up to that depth x margin were
emitted.  Offline estimate from the state of a finished frame (RECT lists are not needed: the default exact binning)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from online_lang_splatting_amd import _C
from online_lang_splatting_amd.frame_shard import RasterWorkspace
from online_lang_splatting_amd.scene import CONFIGS, arc_cameras, make_scene
dev = torch.device("cuda:0")
cfg = CONFIGS[3]
P, W, H, F = cfg["P"], cfg["W"], cfg["H"], cfg["F"]
sc = make_scene(P, W, H, F, seed=3, max_sh_degree=cfg["max_sh_degree"])
M = sc.shs.shape[1]
g = dict(bg=sc.bg.to(dev), means3D=sc.means3D.to(dev), opacities=sc.opacities.to(dev), scales=sc.scales.to(dev),
         rotations=sc.rotations.to(dev), shs=sc.shs.to(dev), language=sc.language.to(dev))
ws = RasterWorkspace(P, W, H, F, M, 4_500_000, dev)
tile = 15
gx, gy = (W + tile - 1) // tile, (H + tile - 1) // tile
nt = gx * gy
for name, cam in (("identity", arc_cameras(W, H, n=1)[0]), ("arc view 0 of 12", arc_cameras(W, H, n=12)[0])):
    c = dict(viewmatrix=cam.world_view_transform.to(dev), projmatrix=cam.full_proj_transform.to(dev),
             projmatrix_raw=cam.projection_matrix.to(dev), campos=cam.camera_center.to(dev), tanfovx=cam.tanfovx, tanfovy=cam.tanfovy)
    ws.set_scene(sh_degree=sc.sh_degree, **c, **g)
    ws.forward()
    torch.cuda.synchronize()
    R = int(ws.num_rendered.cpu()[0])
    cap = ws.capacity
    src = _C.state_field("binning", ws.binning, "src", R=cap, F=F, dtype=torch.int32, count=R).long()
    gid = _C.state_field("binning", ws.binning, "inst_gid", R=cap, F=F, dtype=torch.int32, count=R).long()
    pl = gid[src]
    rg = _C.state_field("image", ws.img, "ranges", W=W, H=H, dtype=torch.int32, count=2 * nt).view(-1, 2).long()
    depths = _C.state_field("geometry", ws.geom, "depths", P=P, F=F, dtype=torch.float32, count=P)
    nc = _C.state_field("image", ws.img, "n_contrib", W=W, H=H, dtype=torch.int32, count=W * H).view(H, W).long()
    fT = _C.state_field("image", ws.img, "final_T", W=W, H=H, dtype=torch.float32, count=W * H).view(H, W)
    lens = (rg[:, 1] - rg[:, 0]).clamp(min=0)
    # per tile: does every pixel saturate (T < 1e-4 reached => the pixel is `done`)?  a pixel is done iff its walk ended by the
    # test_T < 1e-4 rule; final_T of such a pixel is the T BEFORE the terminating entry, which can be anything >= 1e-4 / (1 - 0.99).
    # Proxy that errs on the safe side: the forward's stop position = max n_contrib of the tile if all its pixels have
    # n_contrib < list length (they stopped before the end), else the list end.
    pad_h, pad_w = gy * tile - H, gx * tile - W
    ncp = torch.nn.functional.pad(nc, (0, pad_w, 0, pad_h), value=0).view(gy, tile, gx, tile).permute(0, 2, 1, 3).reshape(nt, -1)
    kmax = ncp.max(dim=1).values
    # a pixel that walked to the end of the list has n_contrib possibly < len too (trailing entries skipped) — we cannot tell
    # from n_contrib alone; use final_T: a saturated pixel has final_T * (1 - alpha_last) < 1e-4, so final_T < 1e-2 is necessary
    fTp = torch.nn.functional.pad(fT, (0, pad_w, 0, pad_h), value=0.0).view(gy, tile, gx, tile).permute(0, 2, 1, 3).reshape(nt, -1)
    saturated = (fTp.max(dim=1).values < 1e-2) & (lens > 0)
    stop = torch.where(saturated, torch.minimum(kmax + 1, lens), lens)   # (+1: the terminating entry is read as well)
    starts = rg[:, 0]
    idx = (starts + (stop - 1).clamp(min=0)).clamp(max=R - 1)
    dcut = torch.where(saturated & (stop > 0), depths[pl[idx]], torch.full((nt,), float("inf"), device=dev))
    tile_of = torch.repeat_interleave(torch.arange(nt, device=dev), lens)
    d_inst = depths[pl]
    print(f"{name}: R = {R}, tiles {nt}, saturating tiles {int(saturated.sum())} ({100.0 * float(saturated.float().mean()):.1f} %), "
          f"entries read (stop positions) {int(stop.sum())} = {100.0 * int(stop.sum()) / R:.1f} % of R")
    for margin in (1.0, 1.05, 1.1, 1.25, 1.5):
        keep = d_inst <= (dcut * margin)[tile_of]
        print(f"   margin x{margin}: kept {int(keep.sum())} = {100.0 * int(keep.sum()) / R:.1f} % of R")
