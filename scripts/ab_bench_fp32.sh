R=${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2; do for n in prod r03f32; do
  lib=$R/unopose_amd/libunopose_hip_$n.so; [ $n = prod ] && lib=$R/unopose_amd/libunopose_hip.so
  UNOPOSE_LIB=$lib python3 $R/bench.py --dtype fp32 --steps 15 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('$n fp32', round(d['value'],1), 'pairs/s', round(d['ms_per_step'],2), 'ms')"
done; done
