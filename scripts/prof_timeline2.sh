#!/bin/bash
# rocprofv3 kernel trace of the default (pipelined) bench command + critical-path composition (scripts/rocpd_timeline.py), 518 and 224
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp
for img in 518 224; do
  rm -rf /tmp/kt
  rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python3 $R/bench.py --img $img --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-fp32 --no-extra > $R/gpurun_out/${TAG}_bench_under_rocprof_timeline_s$img.json 2>/dev/null
  python3 $R/scripts/rocpd_timeline.py $(find /tmp/kt -name "*.db" | head -1) 12 4 > $R/gpurun_out/${TAG}_timeline_s$img.csv
done
