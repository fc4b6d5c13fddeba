"""Summarise rocprofv3 --pmc counter_collection.csv per kernel (mean over dispatches)."""
import csv, sys, collections, re
path = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "render_|preprocess_bwd|radix_scatter"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(path)):
    k = r["Kernel_Name"]
    if not re.search(pat, k): continue
    k = re.sub(r"\(.*", "", k).replace("void olsr::", "")
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:24s} {sum(v)/len(v):16.0f}  (n={len(v)})")
