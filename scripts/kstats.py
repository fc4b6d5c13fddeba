"""Print a rocprofv3 kernel_stats.csv compactly: python scripts/kstats.py <csv> [n]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    n = r["Name"].split("(")[0].replace("void olsr::", "").replace("olsr::", "")
    print(f"{n[:46]:46s} calls {r['Calls']:>4s} avg {float(r['AverageNs']) / 1e3:8.1f} us  {r['Percentage']:>6s}%")
