#!/bin/bash
# per-kernel durations of the tracking loop, plain vs depth cut-offs (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in plain\ host cut; do
  tag=$(echo $v | cut -d' ' -f1)
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trkprof_$tag -o s -- python scripts/probe/tracking_cut_ab.py 60 "$v" > gpurun_out/trkprof_$tag.log 2>&1
  rm -f gpurun_out/trkprof_$tag/*/*.db
  f=$(ls gpurun_out/trkprof_$tag/*/s_kernel_stats.csv | head -1)
  echo "== $v"; python - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.1f} us  min {float(r["MinNs"])/1e3:8.1f}')
P
done
