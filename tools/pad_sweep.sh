#!/bin/bash
# Diagnostic: C2 kernel time for LDS row paddings of the l / feature tiles (MIMO_LS_PAD, MIMO_RS_PAD)
for pad in 2 1 3 4 6 10; do
  echo -n "LS_PAD $pad: "; MIMO_LS_PAD=$pad python bench.py --config ${1:-c2} --no-cpu-baseline --no-sustained --steps 10 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms"], d["roofline"]["frac"])'
done
for pad in 1 0 2 3; do
  echo -n "RS_PAD $pad: "; MIMO_RS_PAD=$pad python bench.py --config ${1:-c2} --no-cpu-baseline --no-sustained --steps 10 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms"], d["roofline"]["frac"])'
done
