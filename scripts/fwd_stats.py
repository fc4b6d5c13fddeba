"""Experiment: what the forward composite's loop does with its (entry, wave) pairs at config 3 (needs the variant build
`bash scripts/build_variant.sh fwdstats k_render_fwd.hip k_render_fwd -DOLSR_FWD_STATS` and
OLSR_LIB=online_lang_splatting_amd/libolsr_fwdstats.so OLSR_BINDING=ctypes)."""
import ctypes as C, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from parity_common import run_backend
from online_lang_splatting_amd import _C as G, _lib
from online_lang_splatting_amd.scene import make_config_scene
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
sc = make_config_scene(cfg)
L = _lib.lib()
out = (C.c_ulonglong * 8)()
L.olsr_debug_fwd_stats(out, 1)
fg, gg = run_backend(G, sc, dev, cfg, 15, 0)
L.olsr_debug_fwd_stats(out, 1)
R = fg["R"]
d = dict(config=cfg, R=R, entry_wave_pairs_total=4 * out[5] // 4, pairs_looked_at=out[0], pairs_reaching=out[1],
         pairs_past_reach_evaluated=out[4], pairs_blending=out[2], lanes_blending=out[3])
d["list_entries_total"] = out[5] // 4
print(json.dumps(d))
