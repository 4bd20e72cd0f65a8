#!/bin/bash
# round 6, first GPU call: default bench line with the new bounded legs, then kernel stats of the training step
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r06_bench_first.json 2> $R/gpurun_out/r06_bench_first.log
tail -c 600 $R/gpurun_out/r06_bench_first.log
python3 - <<'P'
import json,os
d=json.load(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"),"gpurun_out/r06_bench_first.json")))
print(d["value"], d["ms_per_step"]); print(json.dumps({k:d.get(k) for k in ("ref_cached","contract_224","train","fp32")})[:3000])
P
bash $R/scripts/prof_train_quick.sh
head -40 $R/gpurun_out/quick_train_kernel_stats.csv
