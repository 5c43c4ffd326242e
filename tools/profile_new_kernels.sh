#!/bin/bash
# rocprofv3 --kernel-trace --stats of the kernel families added in round 4, one short timing run each (tools/quick_time.py):
# per-kernel average durations for profiles/.  Usage on the GPU box: bash tools/profile_new_kernels.sh
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_r04_new
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for spec in "2e6 20 16" "2e6 20 16 gibbs" "2e6 24 32" "2e6 16 96" "2e6 28 48" "2e6 2 160" "2e6 2 256 gibbs" "2e6 1 256"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/run$i -- python3 $REPO/tools/quick_time.py $spec > $OUT/run$i.txt 2> $OUT/run$i.log
  echo "## quick_time.py $spec: $(grep '^N=' $OUT/run$i.txt)" >> $OUT/summary.txt
  f=$(find $OUT/run$i -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && python3 - "$f" >> $OUT/summary.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r.get("TotalDurationNs", r.get("total_ns", 0)) or 0))
for r in rows[:6]:
    name = r.get("Name", r.get("name", "?"))[:90]
    print("  %-90s calls %6s avg_ns %12s pct %s" % (name, r.get("Calls", "?"), r.get("AverageNs", r.get("avg_ns", "?")), r.get("Percentage", "?")))
PY
done
cd $REPO
cat $OUT/summary.txt
