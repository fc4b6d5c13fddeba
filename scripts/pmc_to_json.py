"""Turn gpurun_out/<tag>/{stats,fetch,write} into the summaries committed under profiles/.

HBM traffic per launch follows MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE come from
separate --pmc passes, are in KiB, and on gfx950 FETCH_SIZE tallies 128-byte read requests as
64 bytes, so the read side is doubled:  traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes.
"""
import collections, csv, json, os, re, shutil, sys

tag = sys.argv[1]
src = os.path.join("gpurun_out", tag)
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles"   # (on the GPU box: gpurun_out/<tag>/summary, copied into profiles/ afterwards)
# the workload the counters were collected on: bench.py attaches them only to lines of this scene and config
scene = sys.argv[3] if len(sys.argv) > 3 else "volume"
config = int(sys.argv[4]) if len(sys.argv) > 4 else 3
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
    if not re.search(r"render_|row_|preprocess|radix|sort_pass|sort_hist|order_repair|emit|scan|tile_|tau_final|forward_tail|finalize|depth_", k):
        continue
    f, w = fetch.get(k, 0.0), write.get(k, 0.0)
    out[k] = dict(FETCH_SIZE_KiB=round(f, 1), WRITE_SIZE_KiB=round(w, 1),
                  traffic_bytes=int((2 * f + w) * 1024), avg_us=round(stats.get(k, {}).get("avg_us", 0.0), 2))
json.dump(dict(tag=tag, scene=scene, config=config, formula="(2*FETCH_SIZE + WRITE_SIZE)*1024 bytes per launch (MI355X_MICROARCH.md HBM section)",
               kernels=out), open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)
shutil.copy(os.path.join(src, "stats", "s_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
if os.path.exists(os.path.join(src, "stats1", "s_kernel_stats.csv")):
    shutil.copy(os.path.join(src, "stats1", "s_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats_1_in_flight.csv"))
if os.path.exists(os.path.join(src, "stats1p", "s_kernel_stats.csv")):
    shutil.copy(os.path.join(src, "stats1p", "s_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats_1_in_flight_plain_sort.csv"))
for name in ("bench", "bench_cfg1", "bench_cfg2", "bench_cfg5", "bench_room"):
    if os.path.exists(os.path.join(src, name + ".json")) and os.path.getsize(os.path.join(src, name + ".json")):
        shutil.copy(os.path.join(src, name + ".json"), os.path.join(dst, f"{tag}_{name}.json"))
# the forced single-rank exchange legs (round 5): one summary
ex = {}
for sc in ("volume", "room"):
    for mode in ("none", "auto", "reduce_scatter"):
        pth = os.path.join(src, f"bench_exchange_{sc}_{mode}.json")
        if os.path.exists(pth) and os.path.getsize(pth):
            d = json.load(open(pth))
            c = d["config"]
            ex[f"{sc}:{mode}"] = dict(frames_per_s=d["value"], p10=c["value_runs"]["p10"], p90=c["value_runs"]["p90"],
                                      runs=c["value_runs"]["runs"], exchange=c.get("exchange"), detail=c.get("exchange_detail"))
if ex:
    for sc in ("volume", "room"):
        base = ex.get(f"{sc}:none", {}).get("frames_per_s")
        for mode in ("auto", "reduce_scatter"):
            if base and f"{sc}:{mode}" in ex:
                ex[f"{sc}:{mode}"]["against_no_exchange"] = round(ex[f"{sc}:{mode}"]["frames_per_s"] / base - 1.0, 4)
    json.dump(dict(tag=tag, what="bench.py weak-scaling step, four frames in flight, K = 60, on ONE GPU: no exchange against a group of "
                                  "one rank over RCCL with every collective of the exchange issued (OLSR_BENCH_FORCE_EXCHANGE=1) - "
                                  "everything of the exchange but the wire", legs=ex),
              open(os.path.join(dst, f"{tag}_exchange_overhead.json"), "w"), indent=1)
# ---- SQ passes: what bounds the kernels (VALU issue).  Durations from the one-frame-in-flight trace.
stats1 = {}
p1 = os.path.join(src, "stats1p", "s_kernel_stats.csv")
if not os.path.exists(p1):
    p1 = os.path.join(src, "stats1", "s_kernel_stats.csv")
if os.path.exists(p1):
    for r in csv.DictReader(open(p1)):
        stats1[short(r["Name"])] = float(r["AverageNs"]) / 1e3
sq = collections.defaultdict(dict)
for sub in ("sq_a", "sq_b"):
    path = os.path.join(src, sub, "p_counter_collection.csv")
    if not os.path.exists(path):
        continue
    names = {r["Counter_Name"] for r in csv.DictReader(open(path))}
    for c in sorted(names):
        for k, v in mean_counter(path, c).items():
            sq[k][c] = v
valu = {}
for k, d in sq.items():
    if not re.search(r"render_|preprocess|radix|sort_pass|sort_hist|order_repair|emit|scan|row_", k) or "SQ_INSTS_VALU" not in d:
        continue
    e = {c: int(v) for c, v in d.items()}
    us = stats1.get(k) or stats.get(k, {}).get("avg_us")
    if us:
        e["avg_us_1_in_flight"] = round(us, 2)
        # one wave64 VALU instruction per quad-cycle per SIMD (SQ_ACTIVE_INST_VALU == SQ_INSTS_VALU quad-cycles);
        # 1024 SIMDs at the 2.4 GHz peak clock
        e["valu_issue_frac_of_peak"] = round(d["SQ_INSTS_VALU"] / (us * 1e-6) / (1024 * 2.4e9 / 4), 4)
    if d.get("SQ_BUSY_CYCLES") and "SQ_ACTIVE_INST_VALU" in d:
        # SQ_BUSY_CYCLES is summed over the 32 shader engines -> kernel duration in clocks; VALU-active quad-cycles
        # x 4 over 1024 SIMDs -> fraction of the kernel during which a SIMD issues VALU work (clock-independent)
        e["kernel_clocks"] = int(d["SQ_BUSY_CYCLES"] / 32)
        e["valu_busy"] = round(d["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (d["SQ_BUSY_CYCLES"] / 32), 4)
        if us:
            e["effective_clock_GHz"] = round(d["SQ_BUSY_CYCLES"] / 32 / (us * 1e-6) / 1e9, 3)
    if d.get("SQ_WAVE_CYCLES"):
        e["resident_waves_per_simd"] = round(d["SQ_WAVE_CYCLES"] * 4 / 1024 / (d["SQ_BUSY_CYCLES"] / 32), 2) if d.get("SQ_BUSY_CYCLES") else None
        e["valu_active_over_wave_cycles"] = round(d.get("SQ_ACTIVE_INST_VALU", 0) / d["SQ_WAVE_CYCLES"], 4) if "SQ_ACTIVE_INST_VALU" in d else None
        e["wait_inst_lds_over_wave_cycles"] = round(d.get("SQ_WAIT_INST_LDS", 0) / d["SQ_WAVE_CYCLES"], 4)
    valu[k] = e
if valu:
    json.dump(dict(tag=tag, scene=scene, config=config, note="means per launch over the dispatches of one rocprofv3 --pmc run each (sq_a: SQ_INSTS_VALU "
                   "SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS; sq_b: SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY "
                   "SQ_WAIT_INST_LDS), one frame in flight; SQ_ACTIVE_* / SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles",
                   kernels=valu), open(os.path.join(dst, f"{tag}_pmc_valu.json"), "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["avg_us"])[:12]:
    print(f"{k:40s} {v['avg_us']:9.2f} us  traffic {v['traffic_bytes']/1e6:9.1f} MB")
