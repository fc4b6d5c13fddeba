#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the rocprofv3 passes of scripts/profile_round.sh for ONE scene of bench.py
# (`--scene room` is the surface-structured map that stands in for BASELINE configs[3]); no bench legs besides the line
# of that scene.  Counters are collected in their own runs, with --kernel-trace only.
#   scripts/profile_scene.sh <tag> <scene> [steps]   ->  gpurun_out/<tag>/summary/<tag>_*.{csv,json}
set -u
TAG=${1:-r6_room}
SCENE=${2:-room}
STEPS=${3:-20}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
python bench.py --scene $SCENE --steps 60 --warmup 12 --no-cpu-baseline > $OUT/bench_$SCENE.json 2> $OUT/bench_$SCENE.err
B="python bench.py --scene $SCENE --no-cpu-baseline --no-extra-legs --repeats 1"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $B --steps $STEPS --warmup 5 > $OUT/bench_stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats1 -o s -- $B --steps $STEPS --warmup 5 --streams 1 --isolated-steps 0 > $OUT/bench_stats1.log 2>&1
# ... once more with the plain five-launch depth sort (no carried order), where every launched kernel does its work; the
# counter passes below run that way too (with a carried order most radix launches return at once and would dilute the averages)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats1p -o s -- $B --steps $STEPS --warmup 5 --streams 1 --isolated-steps 0 --carry-order 0 > $OUT/bench_stats1p.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- $B --steps 3 --warmup 1 --streams 1 --isolated-steps 0 --carry-order 0 > $OUT/bench_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- $B --steps 3 --warmup 1 --streams 1 --isolated-steps 0 --carry-order 0 > $OUT/bench_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/sq_a -o p -- $B --steps 3 --warmup 1 --streams 1 --isolated-steps 0 --carry-order 0 > $OUT/bench_sq_a.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq_b -o p -- $B --steps 3 --warmup 1 --streams 1 --isolated-steps 0 --carry-order 0 > $OUT/bench_sq_b.log 2>&1
rm -f $OUT/*/*.db
python scripts/pmc_to_json.py $TAG $OUT/summary $SCENE 3
rm -rf $OUT/stats $OUT/stats1 $OUT/stats1p $OUT/fetch $OUT/write $OUT/sq_a $OUT/sq_b
ls -R $OUT | head -40
