"""Diagnostic: one line per kernel instantiation from hipcc's -Rpass-analysis=kernel-resource-usage output (stderr file).
    python tools/resources_summary.py /tmp/narrow_res.txt [substring]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
sel = sys.argv[2] if len(sys.argv) > 2 else ""
KEYS = (("vgpr", r"VGPRs"), ("agpr", r"AGPRs"), ("scratch", r"ScratchSize \[bytes/lane\]"), ("occ", r"Occupancy \[waves/SIMD\]"),
        ("lds", r"LDS Size \[bytes/block\]"))
names, rows = [], []
for blk in re.split(r'remark: [^\n]*Function Name: ', txt)[1:]:
    names.append(blk.split('\n')[0].strip())
    vals = []
    for _, k in KEYS:
        m = re.search(k + r': (\d+)', blk)
        vals.append(int(m.group(1)) if m else -1)
    rows.append(vals)
dem = subprocess.run(['c++filt'], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
for d, vals in zip(dem, rows):
    if sel in d:
        print("%-64s " % d[:64] + " ".join("%s=%d" % (k[0], v) for k, v in zip(KEYS, vals)))
