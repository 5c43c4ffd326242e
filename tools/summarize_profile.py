"""Condense a tools/profile.sh output directory into the text summary committed under profiles/."""
import csv, glob, os, sys
from collections import defaultdict

root = sys.argv[1]


def find(sub, suffix):
    hits = glob.glob(os.path.join(root, sub, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


def short(name):
    return name.split("(")[0].replace("void ", "")[:70]


p = find("trace", "kernel_stats.csv")
if p:
    print("== rocprofv3 --kernel-trace --stats (bench.py, default --steps 10 --warmup 2) ==")
    for row in csv.DictReader(open(p)):
        print(f"{short(row['Name']):70s} calls {row['Calls']:>6s} total_ns {row['TotalDurationNs']:>14s} "
              f"avg_ns {float(row['AverageNs']):14.1f} pct {row['Percentage']}")
p = find("trace", "kernel_trace.csv")
if p:
    d = defaultdict(list)
    for row in csv.DictReader(open(p)):
        d[short(row["Kernel_Name"])].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"]),
                                             row.get("VGPR_Count"), row.get("Accum_VGPR_Count"), row.get("SGPR_Count"),
                                             row.get("LDS_Block_Size"), row.get("Scratch_Size"), row.get("Grid_Size"),
                                             row.get("Workgroup_Size")))
    print("\n== per-kernel launch geometry / registers (kernel_trace.csv) ==")
    for k, v in d.items():
        if "fused" in k or "reduce" in k or "unpack" in k:
            durs = sorted(x[0] for x in v)
            print(f"{k:60s} n={len(v)} median_ns={durs[len(durs)//2]} min_ns={durs[0]} vgpr={v[0][1]} agpr={v[0][2]} "
                  f"sgpr={v[0][3]} lds={v[0][4]} scratch={v[0][5]} grid={v[0][6]} wg={v[0][7]}")

for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_lds"):
    p = find(sub, "counter_collection.csv")
    if not p:
        continue
    acc = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(p)):
        acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print(f"\n== PMC pass {sub} (bench.py --steps 5 --warmup 2; mean per launch of mimo::fused_kernel) ==")
    for k, cs in acc.items():
        if "fused" not in k:
            continue
        for c, vals in cs.items():
            print(f"{k:50s} {c:28s} mean {sum(vals)/len(vals):18.1f}  (n={len(vals)})")
