import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = d["config4_substitute"]
print(d["value"], d["ms_per_step"])
print(json.dumps(c["tracking"]["ms_per_iteration"]))
print(json.dumps(c["tracking"]["library_stage_ms"]), c["tracking"]["library_ms"])
t = dict(c["tracking_depth_cut"]); t.pop("what")
print(json.dumps(t))
