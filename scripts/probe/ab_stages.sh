#!/bin/bash
# per-stage A/B of library variants at config 3: bash scripts/probe/ab_stages.sh [lib ...]   ("" = the in-tree libolsr.so)
for lib in "$@"; do
  for i in 1 2; do
  OLSR_LIB=$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --isolated-steps 40 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lib=[$lib]', d['value'], d['isolated']['value'], d['isolated']['stage_ms'])"
  done
done
