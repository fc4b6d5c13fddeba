set -x
python -m pytest tests/test_gpu_sparse_exchange.py tests/test_gpu_frame_shard.py -x -q 2>&1 | tail -5
B="python bench.py --no-extra-legs --isolated-steps 0 --no-cpu-baseline --steps 60"
$B > gpurun_out/ex_none.json 2>gpurun_out/ex_none.err
OLSR_BENCH_FORCE_EXCHANGE=1 $B --exchange sparse > gpurun_out/ex_sparse_fused.json 2>gpurun_out/ex_sparse_fused.err
OLSR_BENCH_FORCE_EXCHANGE=1 OLSR_BENCH_EXCHANGE_TORCH=1 $B --exchange sparse > gpurun_out/ex_sparse_torch.json 2>gpurun_out/ex_sparse_torch.err
OLSR_BENCH_FORCE_EXCHANGE=1 $B --exchange all_reduce > gpurun_out/ex_all_reduce.json 2>gpurun_out/ex_all_reduce.err
OLSR_BENCH_FORCE_EXCHANGE=1 $B --exchange reduce_scatter > gpurun_out/ex_reduce_scatter.json 2>gpurun_out/ex_reduce_scatter.err
for f in none sparse_fused sparse_torch all_reduce reduce_scatter; do python -c "
import json,sys
try:
    d=json.load(open('gpurun_out/ex_$f.json')); print('$f', d['value'], d['value_runs']['fps'], d['config'].get('exchange_detail'))
except Exception as e: print('$f', 'FAILED', e); print(open('gpurun_out/ex_$f.err').read()[-1500:])
"; done
