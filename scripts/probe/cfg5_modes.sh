#!/bin/bash
# is config 5's composite backward time bimodal across runs / flags on one box?
for i in 1 2; do
  for flags in "--steps 20 --warmup 5 --isolated-steps 30" "--steps 40 --warmup 8"; do
    python bench.py --config 5 $flags --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$flags', '| 4-in-flight', d['value'], 'isolated', d['isolated']['value'], 'bwd', d['isolated']['stage_ms']['render_backward'], 'fwd', d['isolated']['stage_ms']['render_forward'])"
  done
done
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
