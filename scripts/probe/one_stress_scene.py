import os, sys
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from online_lang_splatting_amd import _C as hip
from oracle import oracle_C as oracle
import test_gpu_parity as T
from stress_scenes import random_scene
k, seed0, gen = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
sc, tile, mode, kw, desc = random_scene(k, seed0, gen)
print(desc)
for name, extra in (("strict", {}), ("elementwise", dict(elementwise=True, worst_bound=2e-2)),
                    ("composite+chain", dict(elementwise=True, worst_bound=2e-2, grad_keys=T.COMPOSITE_KEYS, chain=True))):
    try:
        T._check(hip, oracle, sc, seed=k, tile=tile, mode=mode, **extra, **kw)
        print(name, "OK")
    except AssertionError as e:
        print(name, "FAIL", str(e)[:300])
