#!/bin/bash
# whole-forward pipelining (224 x 224) with the model's in-forward side streams on / off (probe hook UNOPOSE_IO_WHOLE in pipeline.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2 3; do
  for v in 0 1; do
    UNOPOSE_IO_WHOLE=$v python3 $R/bench.py --img 224 --no-cpu-baseline --no-extra --no-fp32 --no-roofline --steps 60 --warmup 6 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('internal overlap in whole mode = $v ', round(d['value'],1), 'pairs/s', round(d['ms_per_step'],3), 'ms')"
  done
done
