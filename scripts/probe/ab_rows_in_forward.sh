for r in 1 2; do for v in 1 0; do
out=$(OLSR_ROWS_IN_FORWARD=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1)
python - "$v" "$out" <<'PY'
import json,sys
d=json.loads(sys.argv[2]); c=d.get("config4_substitute",{})
print("rows_in_forward",sys.argv[1],"K20",round(d["value"]),"iso",d["isolated"]["value"],"dropin",d["dropin"]["value"],"track",c.get("tracking",{}).get("ms_per_iteration"),"map",c.get("mapping",{}).get("ms_per_iteration"), "room", c.get("room_scene",{}).get("four_in_flight",{}), "stages", d["isolated"]["stage_ms"])
PY
done; done
