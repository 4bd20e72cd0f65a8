#!/bin/bash
# critical-path composition of the training step: idle / solo / shared time per kernel (scripts/rocpd_timeline.py; marker = the once-per-step
# 6144 -> 4096 FPS) -> gpurun_out/<tag>_train_timeline.csv
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktl
rocprofv3 --kernel-trace -d /tmp/ktl -o kt -- python3 $R/bench.py --train --steps 6 --warmup 2 > /dev/null 2>&1
python3 $R/scripts/rocpd_timeline.py $(find /tmp/ktl -name "*.db" | head -1) 1 3 "fps_kernel<12" > $R/gpurun_out/${TAG}_train_timeline.csv
head -45 $R/gpurun_out/${TAG}_train_timeline.csv
