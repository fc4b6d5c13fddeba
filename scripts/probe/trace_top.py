"""Kernels of a rocprofv3 --kernel-trace CSV by total time over the last `frac` of the trace, with full names."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * (1 - frac)):]
dur, cnt = defaultdict(float), defaultdict(int)
for r in rows:
    n = r["Kernel_Name"][:150]
    dur[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    cnt[n] += 1
for n in sorted(dur, key=lambda k: -dur[k])[:30]:
    print(f"{dur[n] / cnt[n] / 1e3:9.2f} us x {cnt[n]:5d}  {n}")
