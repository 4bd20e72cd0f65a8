#!/bin/bash
# stage-mode stream priorities (features, matching): one process per measurement (needs the UNOPOSE_FPRIO / UNOPOSE_MPRIO probe hook in pipeline.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2 3; do
  for fp in 0 -1; do
    for mp in -1 0; do
      UNOPOSE_FPRIO=$fp UNOPOSE_MPRIO=$mp python3 $R/bench.py --no-cpu-baseline --no-extra --no-fp32 --no-roofline --steps 40 --warmup 6 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('features prio $fp  matching prio $mp ', round(d['value'],1), 'pairs/s', round(d['ms_per_step'],3), 'ms  latency', round(d['forward_latency_ms']['median'],1))"
    done
  done
done
