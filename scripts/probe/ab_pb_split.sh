python -m pytest tests/test_gpu_parity.py tests/test_gpu_row_mask.py tests/test_gpu_api.py tests/test_gpu_pose.py tests/test_gpu_loss.py -x -q 2>&1 | tail -4
B="python bench.py --steps 60 --warmup 12 --no-cpu-baseline --no-extra-legs"
for i in 1 2; do
$B > gpurun_out/ab_split_$i.json 2>/dev/null
OLSR_LIB=online_lang_splatting_amd/libolsr_nosplit.so $B > gpurun_out/ab_nosplit_$i.json 2>/dev/null
done
python - <<'P'
import json
for n in ["split_1","nosplit_1","split_2","nosplit_2"]:
    d=json.loads(open(f"gpurun_out/ab_{n}.json").read().splitlines()[0])
    print(n, d["value"], d["isolated"]["value"], d["isolated"]["stage_ms"]["preprocess_backward"])
P
