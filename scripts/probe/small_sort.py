"""depth_sort stage time of small frames: the one-launch sort (k_sort.hip: sort_small_kernel) against histogram + four passes."""
import sys

import torch

sys.path.insert(0, ".")
from online_lang_splatting_amd import _lib  # noqa: E402
from online_lang_splatting_amd.frame_shard import RasterWorkspace  # noqa: E402
from online_lang_splatting_amd.scene import make_scene  # noqa: E402

dev = torch.device("cuda:0")
for P in (1000, 2000, 4000, 8000, 10000, 16000):
    sc = make_scene(P, 256, 256, 0, seed=1, max_sh_degree=0)
    cam = sc.camera
    g = dict(bg=sc.bg.to(dev), means3D=sc.means3D.to(dev), opacities=sc.opacities.to(dev), scales=sc.scales.to(dev),
             rotations=sc.rotations.to(dev), shs=sc.shs.to(dev), language=None)
    c = dict(viewmatrix=cam.world_view_transform.to(dev), projmatrix=cam.full_proj_transform.to(dev),
             projmatrix_raw=cam.projection_matrix.to(dev), campos=cam.camera_center.to(dev), tanfovx=cam.tanfovx, tanfovy=cam.tanfovy)
    ws = RasterWorkspace(P, 256, 256, 0, 1, 2_000_000, dev)
    res = {}
    for small in (1, 0):
        _lib.lib().olsr_debug_sort_small(small)
        ws.set_scene(sh_degree=0, **c, **g)
        for _ in range(5):
            ws.forward()
        torch.cuda.synchronize()
        _lib.set_profiling(True)
        for _ in range(20):
            ws.forward()
        per = {}
        for name, ms in _lib.stage_times():
            per.setdefault(name, []).append(ms)
        _lib.set_profiling(False)
        res[small] = round(1e3 * sorted(per["depth_sort"])[len(per["depth_sort"]) // 2], 1)
    _lib.lib().olsr_debug_sort_small(1)
    print(P, "one launch", res[1], "us   passes", res[0], "us")
