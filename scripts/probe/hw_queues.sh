# how the frames-in-flight rate and the forced single-rank exchange react to the number of HIP hardware queues
B="python bench.py --no-extra-legs --isolated-steps 0 --no-cpu-baseline --steps 60"
run() { name=$1; shift; "$@" > gpurun_out/hq_$name.json 2> gpurun_out/hq_$name.err; python -c "
import json
try:
    d=json.loads(open('gpurun_out/hq_$name.json').read().splitlines()[0]); print('$name', d['value'], d['value_runs']['fps'])
except Exception as e: print('$name FAILED', e); print(open('gpurun_out/hq_$name.err').read()[-800:])
"; }
run none_s4 $B
run none_s4_q8 env GPU_MAX_HW_QUEUES=8 $B
run none_s6_q8 env GPU_MAX_HW_QUEUES=8 $B --streams 6
run none_s8_q8 env GPU_MAX_HW_QUEUES=8 $B --streams 8
run none_s6 $B --streams 6
run sparse_s4 env OLSR_BENCH_FORCE_EXCHANGE=1 $B
run sparse_s4_q8 env OLSR_BENCH_FORCE_EXCHANGE=1 GPU_MAX_HW_QUEUES=8 $B
run sparse_s3 env OLSR_BENCH_FORCE_EXCHANGE=1 $B --streams 3
run sparse_s6_q8 env OLSR_BENCH_FORCE_EXCHANGE=1 GPU_MAX_HW_QUEUES=8 $B --streams 6
