#!/bin/bash
# rocprofv3 kernel trace of the fp32 step (the reference's default precision) -> gpurun_out/<tag>_kernel_stats_fp32_b32_s518.csv
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kf
rocprofv3 --kernel-trace --stats -d /tmp/kf -o kf -- python3 $R/bench.py --dtype fp32 --img 518 --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/${TAG}_bench_under_rocprof_fp32_s518.json 2>/dev/null
python3 $R/scripts/rocpd_stats.py $(find /tmp/kf -name "*.db" | head -1) 40 > $R/gpurun_out/${TAG}_kernel_stats_fp32_b32_s518.csv
