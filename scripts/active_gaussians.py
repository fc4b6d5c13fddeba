"""How many Gaussians receive any gradient in a frame (config 3 by default)?  The saturation that ends 87 % of every tile
list also leaves most visible Gaussians without a single gradient row.
    python scripts/active_gaussians.py [config=3]"""
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
fg, gg = run_backend(G, sc, dev, cfg, 15, 0)
torch.cuda.synchronize()
vis = (fg["radii"] > 0)
nz = (gg["dL_dmeans2D"].abs().sum(1) != 0) | (gg["dL_dopacity"].reshape(sc.P, -1).abs().sum(1) != 0) | \
     (gg["dL_dcolors"].abs().sum(1) != 0)
touched = fg["n_touched"] > 0
print(json.dumps(dict(config=cfg, P=sc.P, visible=int(vis.sum()), with_any_gradient=int(nz.sum()),
                      n_touched_positive=int(touched.sum()), R=fg["R"])))
