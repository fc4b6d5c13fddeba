"""How the lanes' kernels overlap in a 4-frames-in-flight run (rocprofv3 --kernel-trace CSV): over the steady-state part of the
trace, the share of wall time during which 0, 1, 2, ... composite kernels (render_fwd / render_bwd) are executing, and the
time-weighted mean number of kernels of any kind in flight."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows)
rows = rows[int(0.55 * n):int(0.95 * n)]
ev = []
for r in rows:
    comp = 1 if ("render_fwd_kernel" in r["Kernel_Name"] or "render_bwd_kernel" in r["Kernel_Name"]) else 0
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ev.append((s, 1, comp))
    ev.append((e, -1, -comp))
ev.sort()
t_prev = ev[0][0]
alln = compn = 0
hist_c, hist_a = {}, {}
for t, da, dc in ev:
    dt = t - t_prev
    hist_c[compn] = hist_c.get(compn, 0) + dt
    hist_a[alln] = hist_a.get(alln, 0) + dt
    alln += da
    compn += dc
    t_prev = t
tot = sum(hist_c.values())
print("wall (ms)", tot / 1e6, "kernels", len(rows))
print("composite kernels executing: " + ", ".join(f"{k}: {100 * v / tot:.1f} %" for k, v in sorted(hist_c.items())))
print("kernels of any kind executing: " + ", ".join(f"{k}: {100 * v / tot:.1f} %" for k, v in sorted(hist_a.items())))
