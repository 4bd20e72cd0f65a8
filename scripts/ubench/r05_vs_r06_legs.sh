#!/bin/bash
# the fp32 leg and the training step of round 5's tree (_r05/) and the current tree on one box
R=${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2; do
  for t in _r05 .; do
    (cd $R/$t && python3 bench.py --dtype fp32 --no-cpu-baseline --no-roofline --steps 12 --warmup 3 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fp32 518', 'round5' if '$t' == '_r05' else 'round6', round(d['value'],1), 'pairs/s', round(d['ms_per_step'],3), 'ms/step')")
    (cd $R/$t && python3 bench.py --train --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('train 4096', 'round5' if '$t' == '_r05' else 'round6', round(d['ms_per_step'],2), 'ms/step')")
  done
done
