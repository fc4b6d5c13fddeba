#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of the benchmark command
# (FETCH_SIZE, WRITE_SIZE, two SQ passes — counters are collected in their own runs, with --kernel-trace only).
# Outputs under gpurun_out/<tag>/ ; summarise afterwards with scripts/pmc_to_json.py <tag> and commit profiles/.
set -u
TAG=${1:-r2}
STEPS=${2:-20}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
# the bench lines first, on the box as the driver finds it (a minute of profiler runs before them cost config 5 about 3 %)
python bench.py --steps 60 --warmup 12 > $OUT/bench.json 2> $OUT/bench.err
for c in 1 2 5; do python bench.py --config $c --steps 40 --warmup 8 > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err; done
# the same legs on the surface-structured room map (round 5), and what one GPU can measure of the multi-GPU step on both
# scenes: a group of ONE rank over RCCL with every collective of the chosen exchange issued (never under rocprofv3: the
# profiler and RCCL's start-up hang together)
python bench.py --scene room --steps 60 --warmup 12 --no-cpu-baseline > $OUT/bench_room.json 2> $OUT/bench_room.err
for sc in volume room; do for ex in auto reduce_scatter; do
  OLSR_BENCH_FORCE_EXCHANGE=1 timeout 300 python bench.py --scene $sc --steps 60 --warmup 12 --no-cpu-baseline --no-extra-legs --isolated-steps 0 --exchange $ex > $OUT/bench_exchange_${sc}_$ex.json 2> $OUT/bench_exchange_${sc}_$ex.err
done; timeout 300 python bench.py --scene $sc --steps 60 --warmup 12 --no-cpu-baseline --no-extra-legs --isolated-steps 0 > $OUT/bench_exchange_${sc}_none.json 2> /dev/null; done
B="python bench.py --no-cpu-baseline --no-extra-legs --repeats 1"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $B --steps $STEPS --warmup 5 > $OUT/bench_stats.log 2>&1
# the same kernels with ONE frame in flight (no co-scheduling): per-kernel durations of the `isolated` leg
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats1 -o s -- $B --steps $STEPS --warmup 5 --streams 1 --isolated-steps 0 > $OUT/bench_stats1.log 2>&1
# ... once more with the plain five-launch depth sort (no carried order), where every launched kernel does its work; the
# counter passes below run that way too (with a carried order most radix launches return at once and would dilute the averages)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats1p -o s -- $B --steps $STEPS --warmup 5 --streams 1 --isolated-steps 0 --carry-order 0 > $OUT/bench_stats1p.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- $B --steps 3 --warmup 1 --streams 1 --isolated-steps 0 --carry-order 0 > $OUT/bench_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- $B --steps 3 --warmup 1 --streams 1 --isolated-steps 0 --carry-order 0 > $OUT/bench_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/sq_a -o p -- $B --steps 3 --warmup 1 --streams 1 --isolated-steps 0 --carry-order 0 > $OUT/bench_sq_a.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq_b -o p -- $B --steps 3 --warmup 1 --streams 1 --isolated-steps 0 --carry-order 0 > $OUT/bench_sq_b.log 2>&1
rm -f $OUT/*/*.db
# summarise ON the box (the raw traces exceed what gpurun carries back) into $OUT/summary/, then drop the raw directories
python scripts/pmc_to_json.py $TAG $OUT/summary volume 3
rm -rf $OUT/stats $OUT/stats1 $OUT/stats1p $OUT/fetch $OUT/write $OUT/sq_a $OUT/sq_b
ls -R $OUT | head -60
