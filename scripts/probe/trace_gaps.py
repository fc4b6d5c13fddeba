"""Idle gaps between consecutive kernels of a rocprofv3 --kernel-trace CSV (one stream): for every kernel name, the mean gap
that FOLLOWS it and its mean duration over the last `frac` of the trace.  usage: trace_gaps.py <kernel_trace.csv> [frac]"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * (1 - frac)):]
gap, dur, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
busy_until = int(rows[0]["End_Timestamp"])
tot_gap = 0.0
for a, b in zip(rows[:-1], rows[1:]):
    name = a["Kernel_Name"].split("(")[0][:60]
    busy_until = max(busy_until, int(a["End_Timestamp"]))
    g = max(0, int(b["Start_Timestamp"]) - busy_until)
    gap[name] += g
    tot_gap += g
    dur[name] += int(a["End_Timestamp"]) - int(a["Start_Timestamp"])
    cnt[name] += 1
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print(f"span {span / 1e3:.1f} us, idle {tot_gap / 1e3:.1f} us ({100 * tot_gap / span:.1f} %)")
for name in sorted(gap, key=lambda n: -gap[n])[:25]:
    print(f"{name:62s} n={cnt[name]:5d} dur={dur[name] / cnt[name] / 1e3:8.2f} us  gap after={gap[name] / cnt[name] / 1e3:7.2f} us")
