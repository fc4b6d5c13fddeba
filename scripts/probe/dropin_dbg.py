import sys, time, torch
sys.path.insert(0, "/root/repo")
exec(open("/root/repo/scripts/bench_dropin.py").read().split("# per-step wall times")[0])
from online_lang_splatting_amd import _C
print("ratio, redone:", _C.debug_rows_ratio(True))
