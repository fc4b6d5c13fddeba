for lib in "" online_lang_splatting_amd/libolsr_sc32.so "" online_lang_splatting_amd/libolsr_sc32.so; do
  OLSR_LIB=$lib python bench.py --config 5 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --isolated-steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lib=[$lib] cfg5', d['value'], d['isolated']['value'], d['isolated']['stage_ms']['render_backward'], d['isolated']['stage_ms']['render_forward'])"
done
