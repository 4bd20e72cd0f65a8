#!/bin/bash
# same-box comparison of round 5's tree (git archive e63ffe4 unpacked and built under _r05/, untracked) and the current tree: one process per measurement
R=${GRAFT_REPO_ROOT:-/root/repo}
for img in 518 224; do
  for r in 1 2 3; do
    for t in _r05 .; do
      (cd $R/$t && python3 bench.py --img $img --no-cpu-baseline --no-roofline --no-fp32 --steps 40 --warmup 6 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$img', 'round5' if '$t' == '_r05' else 'round6', round(d['value'],1), 'pairs/s', round(d['ms_per_step'],3), 'ms/step')")
    done
  done
done
