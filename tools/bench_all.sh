#!/bin/bash
# quick look at all four bench configs (no CPU baseline): prints kernel ms and roofline fraction per config
for c in c2 c3 c4 c5; do
  python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$c', 'ms/step %.3f' % d['ms_per_step'], 'kernel_ms %.3f' % r['kernel_ms'], 'frac %.4f' % r['frac'], 'evals/s %.4g' % d['value'])"
done
