"""A window of the steady state of a rocprofv3 --kernel-trace CSV as a timeline: start, duration, queue, kernel."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print(list(rows[0].keys()))
n = len(rows)
w = rows[int(0.7 * n):int(0.7 * n) + int(sys.argv[2]) if len(sys.argv) > 2 else 90]
t0 = int(w[0]["Start_Timestamp"])
qs = sorted({r.get("Queue_Id", "?") for r in w})
for r in w:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    q = qs.index(r.get("Queue_Id", "?"))
    name = r["Kernel_Name"].replace("void olsr::", "").replace("olsr::", "").split("(")[0][:28]
    print(f"{s / 1e3:9.1f} {(e - s) / 1e3:7.1f}  q{q} " + "    " * q + name)
