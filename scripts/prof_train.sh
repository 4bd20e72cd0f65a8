#!/bin/bash
# rocprofv3 kernel trace of the training step (bench.py --train, BASELINE configs[3] shape) -> gpurun_out/<tag>_train_*
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --train --steps 8 --warmup 2 > $R/gpurun_out/${TAG}_train_n1_fp32_4096.json 2>/dev/null
python3 $R/bench.py --train --train-pts 2048 --steps 8 --warmup 2 > $R/gpurun_out/${TAG}_train_n1_fp32_2048.json 2>/dev/null
rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python3 $R/bench.py --train --steps 3 --warmup 2 > /dev/null 2>&1
python3 $R/scripts/rocpd_stats.py $(find /tmp/kt -name "*.db" | head -1) 40 > $R/gpurun_out/${TAG}_train_kernel_stats.csv
