#!/bin/bash
# kernel stats of the training step only (rocprofv3 --kernel-trace --stats) -> gpurun_out/quick_train_kernel_stats.csv
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python3 $R/bench.py --train --steps 3 --warmup 2 > /dev/null 2>&1
python3 $R/scripts/rocpd_stats.py $(find /tmp/kt -name "*.db" | head -1) 60 > $R/gpurun_out/quick_train_kernel_stats.csv
