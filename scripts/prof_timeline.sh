#!/bin/bash
# rocprofv3 kernel trace of the default (pipelined) bench command + critical-path composition (scripts/rocpd_timeline.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python3 $R/bench.py --img 518 --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-fp32 --no-extra > $R/gpurun_out/${TAG}_bench_under_rocprof_timeline.json 2>/dev/null
python3 $R/scripts/rocpd_timeline.py $(find /tmp/kt -name "*.db" | head -1) 12 4 > $R/gpurun_out/${TAG}_timeline_s518.csv
