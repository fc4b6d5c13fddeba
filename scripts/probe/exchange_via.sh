#!/bin/bash
# The forced single-rank exchange legs (everything of the multi-GPU step but the wire), RCCL's C API on the lane's stream
# against torch.distributed, both scenes, alternating.  -> gpurun_out/exchange_via/*.json
mkdir -p gpurun_out/exchange_via
A="--steps 60 --warmup 12 --no-cpu-baseline --no-extra-legs --isolated-steps 0"
for round in 1 2; do for sc in room volume; do
  timeout 120 python bench.py --scene $sc $A > gpurun_out/exchange_via/${sc}_none_$round.json 2>/dev/null
  for ex in auto reduce_scatter; do for via in rccl torch; do
    OLSR_BENCH_FORCE_EXCHANGE=1 timeout 120 python bench.py --scene $sc $A --exchange $ex --exchange-via $via > gpurun_out/exchange_via/${sc}_${ex}_${via}_$round.json 2>gpurun_out/exchange_via/${sc}_${ex}_${via}_$round.err
  done; done
done; done
python - <<'PY'
import glob, json, os
for f in sorted(glob.glob("gpurun_out/exchange_via/*.json")):
    try:
        d = json.load(open(f))
        c = d["config"]
        print(os.path.basename(f), round(d["value"]), c.get("exchange"), c.get("exchange_via"), (c.get("exchange_detail") or {}).get("check", {}).get("equals_dense"))
    except Exception as e:
        print(os.path.basename(f), "failed", e)
PY
