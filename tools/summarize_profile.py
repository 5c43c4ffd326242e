"""Condense a tools/profile.sh output directory into the text summary committed under profiles/, and (with a second
argument: config name) merge the measured HBM bytes per step into profiles/hbm_traffic.json."""
import csv, glob, json, os, sys
from collections import defaultdict

root = sys.argv[1]
cfg = sys.argv[2] if len(sys.argv) > 2 else None
STEPS = 7          # the PMC passes run bench.py --steps 5 --warmup 2


def find(sub, suffix):
    hits = glob.glob(os.path.join(root, sub, "**", "*" + suffix), recursive=True)
    # gpurun merges a call's files into gpurun_out/ without removing those of an earlier call: take the newest run
    return max(hits, key=os.path.getmtime) if hits else None


def short(name):
    return name.split("(")[0].replace("void ", "")[:70]


p = find("trace", "kernel_stats.csv")
if p:
    print("== rocprofv3 --kernel-trace --stats (bench.py, default --steps 20 --warmup 5, sustained leg included) ==")
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
        if k.startswith("mimo::"):
            durs = sorted(x[0] for x in v)
            print(f"{k:60s} n={len(v)} median_ns={durs[len(durs)//2]} min_ns={durs[0]} vgpr={v[0][1]} agpr={v[0][2]} "
                  f"sgpr={v[0][3]} lds={v[0][4]} scratch={v[0][5]} grid={v[0][6]} wg={v[0][7]}")

traffic = defaultdict(dict)
ordered = {}          # counter -> [(dispatch id, kernel, value)] of the mimo:: kernels, in launch order
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_lds"):
    p = find(sub, "counter_collection.csv")
    if not p:
        continue
    acc = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(p)):
        acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
        if row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE") and short(row["Kernel_Name"]).startswith("mimo::"):
            ordered.setdefault(row["Counter_Name"], []).append((int(row.get("Dispatch_Id", 0) or 0), short(row["Kernel_Name"]),
                                                                 float(row["Counter_Value"])))
    print(f"\n== PMC pass {sub} (bench.py --steps 5 --warmup 2 --no-sustained; mean per launch of the mimo:: kernels) ==")
    for k, cs in acc.items():
        if not k.startswith("mimo::"):
            continue
        for c, vals in cs.items():
            print(f"{k:50s} {c:28s} mean {sum(vals)/len(vals):18.1f}  (n={len(vals)})")


def per_step_launches(rows):
    """The bench runs its one-off launches (initial statistics) first and then STEPS identical steps: the launch sequence ends in
    STEPS repeats of one period.  Returns (period length, rows of the periodic tail) — launches in front of it are one-off and
    are NOT counted per step (round 2 counted the initial label statistics of c5 as per-step traffic)."""
    rows = sorted(rows)
    names = [r[1] for r in rows]
    for P in range(1, len(names) // STEPS + 1):
        tail = names[-STEPS * P:]
        if all(tail[i] == tail[i % P] for i in range(len(tail))):
            return P, rows[-STEPS * P:]
    return 0, []


if cfg and ordered:
    # bytes per step = (2 * FETCH_SIZE + WRITE_SIZE) KB summed over the launches of one step
    # (MI355X_MICROARCH.md: gfx950 FETCH_SIZE reports half of a coalesced streaming read; WRITE_SIZE is exact)
    per_step, detail, oneoff = 0.0, {}, {}
    for cname, factor in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
        rows = ordered.get(cname, [])
        P, tail = per_step_launches(rows)
        for _, k, v in tail:
            d = detail.setdefault(k, {"FETCH_SIZE_KB": 0.0, "WRITE_SIZE_KB": 0.0, "launches_per_step": 0, "hbm_bytes_per_step": 0.0})
            d[cname + "_KB"] += v / STEPS
            d["hbm_bytes_per_step"] += factor * v * 1024.0 / STEPS
            if cname == "FETCH_SIZE":
                d["launches_per_step"] += 1.0 / STEPS
            per_step += factor * v * 1024.0 / STEPS
        for _, k, v in sorted(rows)[:len(rows) - len(tail)]:
            oneoff[k] = oneoff.get(k, 0) + (1 if cname == "FETCH_SIZE" else 0)
    for d in detail.values():
        d["launches_per_step"] = round(d["launches_per_step"], 3)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "hbm_traffic.json")
    try:
        allc = json.load(open(path))
    except Exception:
        allc = {}
    allc["_comment"] = ("HBM bytes per bench step from rocprofv3 PMC passes (tools/profile.sh: separate --pmc FETCH_SIZE and "
                        "--pmc WRITE_SIZE runs of `bench.py --steps 5 --warmup 2`), corrected as MI355X_MICROARCH.md prescribes "
                        "for gfx950: bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024, summed over the mimo:: launches of ONE step (the "
                        "periodic tail of the launch sequence; one-off launches in front of the steps are listed, not counted)")
    tag = os.environ.get("MIMO_PROFILE_TAG", "r03")
    allc[cfg] = {"hbm_bytes_per_step": per_step, "kernels": detail, "one_off_launches_not_counted": oneoff,
                 "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, profiles/{tag}_{cfg}_rocprof_summary.txt"}
    json.dump(allc, open(path, "w"), indent=1)
    print(f"\n== HBM bytes per step ({cfg}): {per_step:.4e}   (one-off launches not counted: {oneoff})")
