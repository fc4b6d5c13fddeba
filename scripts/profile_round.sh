#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + separate FETCH_SIZE / WRITE_SIZE
# PMC passes of the benchmark command.  Outputs under gpurun_out/<tag>/ ; summarise afterwards with
# scripts/pmc_to_json.py and copy the summaries into profiles/.
set -u
TAG=${1:-r1}
STEPS=${2:-20}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python bench.py --steps $STEPS --warmup 5 --no-cpu-baseline > $OUT/bench_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_write.log 2>&1
# the same kernels with ONE frame in flight (no co-scheduling): per-kernel durations of the `isolated` leg
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats1 -o s -- python bench.py --steps $STEPS --warmup 5 --no-cpu-baseline --streams 1 > $OUT/bench_stats1.log 2>&1
python bench.py --steps 60 --warmup 12 > $OUT/bench.json 2> $OUT/bench.err
for c in 1 2 5; do python bench.py --config $c --steps 40 --warmup 8 > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err; done
rm -f $OUT/*/*.db
ls -R $OUT | head -30
