#!/bin/bash
# headline (and isolated) rate of variant builds of the library, alternating in one call.  usage: ab_variants.sh "name[:ENV=V,...]" ...
# name "base" = the regular build
for round in 1 2; do
for spec in "$@"; do
  name=${spec%%:*}; envs=""; [[ "$spec" == *:* ]] && envs=${spec#*:}
  lib=""; [ "$name" != base ] && lib="OLSR_LIB=$PWD/online_lang_splatting_amd/_variants/libolsr_$name.so"
  out=$(env $lib ${envs//,/ } timeout 300 python bench.py --steps 60 --warmup 20 --repeats 3 --no-cpu-baseline --no-extra-legs ${BENCH_ARGS} 2>/dev/null | tail -1)
  python - "$spec" "$out" <<'PY'
import json, sys
try:
    d = json.loads(sys.argv[2])
    iso = d.get("isolated", {})
    print(sys.argv[1], "fps", round(d["value"]), "isolated", iso.get("value"))
except Exception as e:
    print(sys.argv[1], "failed", e, sys.argv[2][:200])
PY
done; done
