# A/B of a variant library against the default one, alternating in one call: bash scripts/probe/ab_lib.sh <variant name> [bench args]
V=$1; shift
B="python bench.py --steps 60 --warmup 12 --no-cpu-baseline --no-extra-legs $*"
for i in 1 2; do
$B > gpurun_out/ab_default_$i.json 2>/dev/null
OLSR_LIB=online_lang_splatting_amd/libolsr_$V.so $B > gpurun_out/ab_${V}_$i.json 2>/dev/null
done
python - <<P
import json
for n in ["default_1","${V}_1","default_2","${V}_2"]:
    d=json.loads(open(f"gpurun_out/ab_{n}.json").read().splitlines()[0])
    print(n, d["value"], d["isolated"]["value"], d["isolated"]["stage_ms"])
P
