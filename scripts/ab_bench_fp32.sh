#!/bin/bash
# Same-box A/B of the fp32 step between builds of the library: bash scripts/ab_bench_fp32.sh <rounds> name1 name2 ...
# ("prod" = unopose_amd/libunopose_hip.so, otherwise unopose_amd/libunopose_hip_<name>.so from scripts/build_variant.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
rounds=$1; shift
for r in $(seq $rounds); do for n in "$@"; do
  lib=$R/unopose_amd/libunopose_hip_$n.so; [ $n = prod ] && lib=$R/unopose_amd/libunopose_hip.so
  UNOPOSE_LIB=$lib python3 $R/bench.py --dtype fp32 --steps 15 --warmup 3 --no-cpu-baseline --no-roofline --no-extra 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('$n fp32', round(d['value'],1), 'pairs/s', round(d['ms_per_step'],2), 'ms')"
done; done
