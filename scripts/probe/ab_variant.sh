#!/bin/bash
# A/B of kernel variants: bash scripts/probe/ab_variant.sh [lib ...]  ("" = the in-tree libolsr.so), config 3 and config 5
for lib in "$@"; do
  for cfg in 3 5; do
  OLSR_LIB=$lib python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --isolated-steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
b=d.get('bracket',{})
print('lib=[$lib] cfg $cfg: 4-in-flight', d['value'], 'isolated', d['isolated']['value'], 'bwd', d['isolated']['stage_ms']['render_backward'], 'fwd', d['isolated']['stage_ms']['render_forward'], 'exact', b.get('exact_mode',{}).get('value'), 'tile16', b.get('tile16',{}).get('value'), 'trk', d.get('config4_substitute',{}).get('tracking_iteration_ms'))"
  done
done
