# per-rank cost of the weak-scaling points: what rank K of an N-GPU run renders, measured alone on one GPU
# (",arc": the rotated poses the weak-scaling mode used until round 4)
B="python bench.py --no-extra-legs --isolated-steps 20 --no-cpu-baseline --steps 40 --repeats 3"
for kn in 0,1 0,2 0,4 0,8 3,8 7,8 0,8,arc; do
$B --rank-view $kn > gpurun_out/av_$kn.json 2> gpurun_out/av_$kn.err; python -c "
import json
d=json.loads(open('gpurun_out/av_$kn.json').read().splitlines()[0]); print('$kn', d['value'], d['isolated']['value'], d['config']['R_binned'], d['config']['live_gradient_rows'])
"; done
