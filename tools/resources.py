"""Register / LDS / scratch / occupancy of every kernel instantiation of a .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
    python tools/resources.py mimo_amd/csrc/mimo_small.hip [filter]"""
import re, subprocess, sys
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-honor-nans",
                      "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"], capture_output=True, text=True).stderr
rows, cur = [], None
for ln in out.splitlines():
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        cur = {"name": m.group(1)}; rows.append(cur); continue
    for key, pat in (("vgpr", r"\bVGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"\bSGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, ln)
        if m and cur is not None: cur[key] = int(m.group(1))
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, n in sorted(zip(rows, names), key=lambda t: t[1]):
    n = re.sub(r"^void mimo::", "", n).replace("(mimo::KernelArgs)", "")
    if flt in n:
        print(f"{n:48s} vgpr={r.get('vgpr')} agpr={r.get('agpr')} sgpr={r.get('sgpr')} scratch={r.get('scratch')} occ={r.get('occ')} lds={r.get('lds')}")
