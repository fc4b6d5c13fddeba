for r in 1 2; do for q in 4 3 6 2; do
out=$(GPU_MAX_HW_QUEUES=$q timeout 200 python bench.py --steps 60 --warmup 20 --repeats 3 --no-cpu-baseline --no-extra-legs --isolated-steps 0 2>/dev/null | tail -1)
python -c "
import json,sys
d=json.loads(sys.argv[2]); print('hwq',sys.argv[1],round(d['value']))" $q "$out"
done; done
