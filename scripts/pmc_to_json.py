"""Turn gpurun_out/<tag>/{stats,fetch,write} into the summaries committed under profiles/.

HBM traffic per launch follows MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE come from
separate --pmc passes, are in KiB, and on gfx950 FETCH_SIZE tallies 128-byte read requests as
64 bytes, so the read side is doubled:  traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes.
"""
import collections, csv, json, os, re, shutil, sys

tag = sys.argv[1]
src = os.path.join("gpurun_out", tag)
dst = "profiles"
os.makedirs(dst, exist_ok=True)


def short(name):
    return re.sub(r"\(.*", "", name).replace("void olsr::", "").replace("olsr::", "")


def mean_counter(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch = mean_counter(os.path.join(src, "fetch", "p_counter_collection.csv"), "FETCH_SIZE")
write = mean_counter(os.path.join(src, "write", "p_counter_collection.csv"), "WRITE_SIZE")
stats = {}
for r in csv.DictReader(open(os.path.join(src, "stats", "s_kernel_stats.csv"))):
    stats[short(r["Name"])] = dict(calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3,
                                   pct=float(r["Percentage"]))
out = {}
for k in sorted(set(fetch) | set(write)):
    if not re.search(r"render_|row_reduce|preprocess|radix|emit|scan|tile_ranges|tau_final|finalize", k):
        continue
    f, w = fetch.get(k, 0.0), write.get(k, 0.0)
    out[k] = dict(FETCH_SIZE_KiB=round(f, 1), WRITE_SIZE_KiB=round(w, 1),
                  traffic_bytes=int((2 * f + w) * 1024), avg_us=round(stats.get(k, {}).get("avg_us", 0.0), 2))
json.dump(dict(tag=tag, formula="(2*FETCH_SIZE + WRITE_SIZE)*1024 bytes per launch (MI355X_MICROARCH.md HBM section)",
               kernels=out), open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)
shutil.copy(os.path.join(src, "stats", "s_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
if os.path.exists(os.path.join(src, "stats1", "s_kernel_stats.csv")):
    shutil.copy(os.path.join(src, "stats1", "s_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats_1_in_flight.csv"))
for name in ("bench", "bench_cfg1", "bench_cfg2", "bench_cfg5"):
    if os.path.exists(os.path.join(src, name + ".json")) and os.path.getsize(os.path.join(src, name + ".json")):
        shutil.copy(os.path.join(src, name + ".json"), os.path.join(dst, f"{tag}_{name}.json"))
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["avg_us"])[:12]:
    print(f"{k:40s} {v['avg_us']:9.2f} us  traffic {v['traffic_bytes']/1e6:9.1f} MB")
