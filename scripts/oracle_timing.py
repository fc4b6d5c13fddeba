import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.environ["ORACLE_TIMING"] = "1"
import torch
from parity_common import run_backend
from oracle import oracle_C as O
from online_lang_splatting_amd.scene import make_config_scene
sc = make_config_scene(3)
for thr in (int(sys.argv[1]) if len(sys.argv) > 1 else os.cpu_count(), 64, 16):
    O.set_threads(thr)
    for rep in range(2):
        t0 = time.time()
        fo, go = run_backend(O, sc, None, 3, 15, 0)
        print(f"threads {thr}: frame {time.time() - t0:.2f} s", flush=True)
        O.release(fo["geom"])
