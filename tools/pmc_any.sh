#!/bin/bash
# PMC counters of every kernel of one diagnostic command:  bash tools/pmc_any.sh <tag> <counters...> -- python3 tools/quick_time.py ...
TAG=$1; shift
CNT=()
while [ "$1" != "--" ]; do CNT+=("$1"); shift; done
shift
export TMPDIR=/tmp; REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --pmc "${CNT[@]}" --output-format csv -d $OUT -- "$@" > $OUT/cmd.log 2>&1
cd $REPO
python3 - "$OUT" <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for p in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, {c: "%.4e (n=%d)" % (sum(x) / len(x), len(x)) for c, x in v.items()})
PY
