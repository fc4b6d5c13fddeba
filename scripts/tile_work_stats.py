"""How unevenly is the composites' work spread over the tiles (config 3 by default)?  Per tile: list length, entries the
forward reads before every pixel is saturated (kmax = max n_contrib), live (entry, slot) pairs.  If the heaviest tile's
serial walk is a large share of the kernel's duration, the kernel is bound by that critical path and not by throughput.
    python scripts/tile_work_stats.py [config=3]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from parity_common import run_backend
from online_lang_splatting_amd import _C as G
from online_lang_splatting_amd.scene import make_config_scene

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
sc = make_config_scene(cfg)
W, H = sc.camera.width, sc.camera.height
fg, gg = run_backend(G, sc, dev, cfg, 15, 0)
torch.cuda.synchronize()
T = 15
gx, gy = (W + T - 1) // T, (H + T - 1) // T
nt = gx * gy
G.TILE = 15
ranges = G.state_field("image", fg["img"], "ranges", W=W, H=H, dtype=torch.int32, count=2 * nt).cpu().view(nt, 2).long()
ncon = G.state_field("image", fg["img"], "n_contrib", W=W, H=H, dtype=torch.int32, count=W * H).cpu().view(H, W).long()
work = G.state_field("image", fg["img"], "tile_work", W=W, H=H, dtype=torch.int32, count=2 * nt).cpu().view(2, nt).long()
length = (ranges[:, 1] - ranges[:, 0]).clamp(min=0)
pad = torch.zeros(gy * T, gx * T, dtype=torch.long)
pad[:H, :W] = ncon
kmax = pad.view(gy, T, gx, T).permute(0, 2, 1, 3).reshape(nt, T * T).max(1).values


def q(t):
    t = t.double()
    return dict(mean=round(t.mean().item(), 1), p50=t.quantile(0.5).item(), p90=t.quantile(0.9).item(),
                p99=t.quantile(0.99).item(), max=t.max().item(), sum=int(t.sum().item()))


print(json.dumps(dict(config=cfg, tiles=nt, list_length=q(length), entries_read_kmax=q(kmax),
                      live_pairs_4slots=q(work[0]), live_pairs_packed_2waves=q(work[1]))))
