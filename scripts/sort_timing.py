"""Phase timing of the fused radix-sort passes of one config-3 frame (olsr_debug_sort_timing): runs ON THE GPU BOX.
Prints, per pass, when (us after the earliest block's start, shader clock ~2.1 GHz) the blocks reach each phase."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from online_lang_splatting_amd import _lib  # noqa: E402
from online_lang_splatting_amd.frame_shard import RasterWorkspace  # noqa: E402
from online_lang_splatting_amd.scene import make_config_scene  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
GHZ = float(os.environ.get("OLSR_CLOCK_GHZ", "2.1"))
dev = torch.device("cuda:0")
sc = make_config_scene(cfg)
cam = sc.camera
W, H = cam.width, cam.height
ws = RasterWorkspace(sc.P, W, H, sc.F, sc.shs.shape[1], 4_000_000 if cfg == 3 else 12_000_000, dev)
kw = dict(bg=sc.bg.to(dev), means3D=sc.means3D.to(dev), opacities=sc.opacities.to(dev), scales=sc.scales.to(dev),
          rotations=sc.rotations.to(dev), shs=sc.shs.to(dev), language=None if sc.language is None else sc.language.to(dev),
          viewmatrix=cam.world_view_transform.to(dev), projmatrix=cam.full_proj_transform.to(dev),
          projmatrix_raw=cam.projection_matrix.to(dev), campos=cam.camera_center.to(dev), tanfovx=cam.tanfovx,
          tanfovy=cam.tanfovy, sh_degree=sc.sh_degree)
ws.set_scene(**kw)
for _ in range(3):
    ws.forward()
torch.cuda.synchronize()
MAXB, MAXL = 2048, 8
buf = torch.zeros(MAXL * MAXB * 8, dtype=torch.int64, device=dev)
L = _lib.lib()
L.olsr_debug_sort_timing.argtypes = [C.c_void_p, C.c_int, C.c_int]
L.olsr_debug_sort_timing.restype = None
L.olsr_debug_sort_timing(buf.data_ptr(), MAXB, MAXL)
ws.forward()
torch.cuda.synchronize()
L.olsr_debug_sort_timing(None, 0, 0)
t = buf.cpu().view(MAXL, MAXB, 8)
names = ["ticket", "counted", "published", "ranked", "summed", "written"]
for l in range(MAXL):
    x = t[l]
    live = x[:, 0] > 0
    if not bool(live.any()):
        continue
    x = x[live].double()
    t0 = x[:, 0].min()
    full = x[:, 5] > 0
    print(f"pass {l}: {int(live.sum())} blocks took a ticket, {int(full.sum())} had keys")
    for k, nm in enumerate(names):
        col = x[full][:, k] if k > 0 else x[:, k]
        us = (col - t0) / (GHZ * 1e3)
        print(f"   {nm:10s} first {us.min():7.2f}  median {us.median():7.2f}  last {us.max():7.2f} us")
    # the shader clocks of different XCDs are not synchronised: absolute times only within one XCD
    xf = x[full]
    for xcd in range(8):
        sel = xf[:, 6] == xcd
        if not bool(sel.any()):
            continue
        y = xf[sel]
        t0x = y[:, 0].min()
        st = ((y[:, 0] - t0x) / (GHZ * 1e3)).sort().values
        en = ((y[:, 5] - t0x) / (GHZ * 1e3)).sort().values
        print(f"   XCD {xcd}: {int(sel.sum())} blocks; start us (min/med/max) {st[0]:.1f}/{st[len(st)//2]:.1f}/{st[-1]:.1f}; "
              f"end {en[0]:.1f}/{en[len(en)//2]:.1f}/{en[-1]:.1f}")
    d = (x[full][:, 1:6] - x[full][:, 0:5]) / (GHZ * 1e3)
    print("   per-block phase durations (median / max us): " +
          ", ".join(f"{names[k + 1]} {d[:, k].median():.2f}/{d[:, k].max():.2f}" for k in range(5)))
