#!/bin/bash
# binning-chain stage times of library variants: bash scripts/probe/ab_chain.sh [lib ...]
for lib in "$@"; do
  OLSR_LIB=$lib python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs --isolated-steps 40 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['isolated']['stage_ms']
print('lib=[$lib]', 'depth_sort', s['depth_sort'], 'emit', s['emit'], 'tile_sort', s['tile_sort'], 'isolated', d['isolated']['value'])"
done
