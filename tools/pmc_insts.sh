# Instruction mix of the fused kernel for one bench config:  bash tools/pmc_insts.sh [c2|c3|c4|c5]
CFG=${1:-c2}
export TMPDIR=/tmp; REPO=$(pwd); cd /tmp
rm -rf $REPO/gpurun_out/pmc_insts
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d $REPO/gpurun_out/pmc_insts -- python3 $REPO/bench.py --config $CFG --steps 5 --warmup 2 --no-cpu-baseline > $REPO/gpurun_out/pmc_insts_bench.json 2>/dev/null
cd $REPO; python3 - <<'PY'
import csv, glob, json
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for p in glob.glob("gpurun_out/pmc_insts/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "fused_kernel" in r["Kernel_Name"] or "estep_chunked" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
d = json.loads(open("gpurun_out/pmc_insts_bench.json").read().strip().splitlines()[-1])
ntiles = d["config"]["rows_per_gpu"] / 32
for k, v in acc.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    print(k, {c: "%.3e" % x for c, x in m.items()})
    print("   per wave-tile:", {c: round(x / ntiles / 4, 1) for c, x in m.items() if c.startswith("SQ_INSTS") or c.startswith("SQ_ACTIVE")})
print(d["roofline"]["kernel_ms"], d["value"])
PY
