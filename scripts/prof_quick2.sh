#!/bin/bash
# kernel stats of the pipelined bench step (inflight 1: isolated kernel durations), 518 and 224 crops -> gpurun_out/quick_kernel_stats_s{518,224}.csv
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
for img in ${1:-518 224}; do
  rm -rf /tmp/ks
  rocprofv3 --kernel-trace --stats -d /tmp/ks -o ks -- python3 $R/bench.py --img $img --steps 6 --warmup 2 --inflight 1 --no-cpu-baseline --no-roofline --no-fp32 --no-extra > /dev/null 2>&1
  python3 $R/scripts/rocpd_stats.py $(find /tmp/ks -name "*.db" | head -1) 60 > $R/gpurun_out/quick_kernel_stats_s$img.csv
done
