#!/bin/bash
# Same-box A/B of whole-step throughput between builds of the library: bash scripts/ab_bench.sh <rounds> <img> name1 name2 ...
# ("prod" = unopose_amd/libunopose_hip.so, otherwise unopose_amd/libunopose_hip_<name>.so from scripts/build_variant.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
rounds=$1; img=$2; shift 2
for r in $(seq $rounds); do
  for n in "$@"; do
    lib=$R/unopose_amd/libunopose_hip_$n.so; [ $n = prod ] && lib=$R/unopose_amd/libunopose_hip.so
    UNOPOSE_LIB=$lib python3 $R/bench.py --img $img --steps 30 --warmup 3 --no-cpu-baseline --no-roofline --no-fp32 --no-extra 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('$n', round(d['value'],1), 'pairs/s', round(d['ms_per_step'],2), 'ms', d['step_ms_hip_events']['median'])"
  done
done
