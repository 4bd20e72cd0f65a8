#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2 3; do
  python3 $R/bench.py --train --train-pts 2048 --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('train 2048', round(d['ms_per_step'],2), 'ms/step')"
done
