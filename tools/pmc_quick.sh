# LDS bank-conflict share of the fused kernel for one bench config:  bash tools/pmc_quick.sh [c2|c3|c4|c5]
CFG=${1:-c2}
export TMPDIR=/tmp; REPO=$(pwd); cd /tmp
rm -rf $REPO/gpurun_out/pmc_quick
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $REPO/gpurun_out/pmc_quick -- python3 $REPO/bench.py --config $CFG --steps 5 --warmup 2 --no-cpu-baseline > $REPO/gpurun_out/pmc_quick_bench.json 2>/dev/null
cd $REPO; python3 - <<'PY'
import csv, glob, json
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for p in glob.glob("gpurun_out/pmc_quick/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "fused_kernel" in r["Kernel_Name"] or "estep_chunked" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    print(k, {c: "%.3e" % x for c, x in m.items()}, "conflict/active = %.2f" % (m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 1), 1)),
          "lds_active/gui(8 XCD) = %.2f" % (m.get("SQ_LDS_IDX_ACTIVE", 0) / 256 / max(m.get("GRBM_GUI_ACTIVE", 1) / 8, 1)))
d = json.loads(open("gpurun_out/pmc_quick_bench.json").read().strip().splitlines()[-1]); print(d["roofline"]["kernel_ms"], d["value"])
PY
