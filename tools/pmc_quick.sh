export TMPDIR=/tmp; REPO=$(pwd); cd /tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $REPO/gpurun_out/pmc_quick -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $REPO/gpurun_out/pmc_quick_bench.json 2>/dev/null
cd $REPO; python3 - <<'PY'
import csv, glob, json
from collections import defaultdict
acc = defaultdict(list)
for p in glob.glob("gpurun_out/pmc_quick/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "fused_kernel<10, 1, 0" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k: sum(v)/len(v) for k, v in acc.items()})
d = json.loads(open("gpurun_out/pmc_quick_bench.json").read().strip().splitlines()[-1]); print(d["roofline"]["kernel_ms"], d["value"])
PY
