"""Static check of the gfx950 ISA hipcc emitted for mimo_kernels.hip: every s_barrier must be reached with no LDS
operation of the same wave still in flight (an `s_waitcnt lgkmcnt(0)` after the last ds_* instruction on EVERY path).
A workgroup barrier that a wave signals with its own LDS stores pending lets the other waves read stale data.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-honor-nans -S -o k.s mimo_amd/csrc/mimo_kernels.hip
    python tools/check_barrier_waits.py k.s

Forward dataflow over the basic blocks of each kernel (state: may an LDS op be pending?), to a fixed point."""
import re
import sys


def kernels_of(path):
    name, lines, out = None, [], {}
    for ln in open(path).read().split('\n'):
        m = re.match(r'^(_ZN4mimo\w+):\s*; @', ln)      # device definition (the host stub of the same name has no "; @")
        if m:
            name, lines = m.group(1), []
            out[name] = lines
        if name:
            lines.append(ln)
        if 's_endpgm' in ln:
            name = None
    return out


def is_instr(l):
    s = l.strip()
    return bool(s) and not s.startswith(';') and not s.startswith('.') and not re.match(r'^[\w.$]+:', s)


def check(L):
    # basic blocks: start at labels and after branch instructions
    starts = {0}
    label_at = {}
    for i, l in enumerate(L):
        m = re.match(r'^(\.LBB\w+):', l.strip())
        if m:
            starts.add(i); label_at[m.group(1)] = i
        if is_instr(l) and re.match(r's_(c?branch|endpgm|setpc)', l.strip()):
            starts.add(i + 1)
    starts = sorted(s for s in starts if s < len(L))
    block_of = {}
    blocks = []
    for bi, s in enumerate(starts):
        e = starts[bi + 1] if bi + 1 < len(starts) else len(L)
        blocks.append((s, e))
        block_of[s] = bi
    succ = [[] for _ in blocks]
    for bi, (s, e) in enumerate(blocks):
        last = None
        for i in range(e - 1, s - 1, -1):
            if is_instr(L[i]):
                last = L[i].strip(); break
        fall = True
        if last:
            m = re.match(r's_(c?branch)\w*\s+(\.LBB\w+)', last)
            if m:
                tgt = label_at.get(m.group(2))
                if tgt is not None: succ[bi].append(block_of[tgt])
                if m.group(1) == 'branch': fall = False
            if last.startswith('s_endpgm'): fall = False
        if fall and bi + 1 < len(blocks): succ[bi].append(bi + 1)
    state_in = [False] * len(blocks)
    bad = set()
    changed = True
    while changed:
        changed = False
        for bi, (s, e) in enumerate(blocks):
            st = state_in[bi]
            for i in range(s, e):
                if not is_instr(L[i]): continue
                t = L[i].strip()
                if t.startswith('ds_') and not re.match(r'ds_(b?permute|swizzle)', t): st = True   # (lane exchanges touch no LDS memory)
                elif t.startswith('s_waitcnt') and ('lgkmcnt(0)' in t or re.fullmatch(r's_waitcnt\s+0', t)): st = False
                elif t == 's_barrier':
                    if st: bad.add(i)
                    st = False       # (report each barrier once; what follows is checked from a clean state)
            for nb in succ[bi]:
                if st and not state_in[nb]:
                    state_in[nb] = True; changed = True
    return sorted(bad)


if __name__ == "__main__":
    total = 0
    for path in sys.argv[1:]:
        ks = kernels_of(path)
        nbad = 0
        for k, L in ks.items():
            b = check(L)
            if b:
                nbad += 1
                print(f"{path}: {k}: s_barrier reachable with LDS ops pending at lines {b[:6]}")
        print(f"{path}: {len(ks)} kernels, {nbad} with an unprotected barrier")
        total += nbad
    sys.exit(1 if total else 0)
