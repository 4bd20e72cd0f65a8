R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
for fl in 2 1; do
  sfx=$([ $fl = 1 ] && echo _isolated || echo "")
  rm -rf /tmp/ks
  rocprofv3 --kernel-trace --stats -d /tmp/ks -o ks -- python3 $R/bench.py --img 518 --steps 6 --warmup 2 --inflight $fl --no-cpu-baseline --no-roofline --no-fp32 --no-extra > $R/gpurun_out/${TAG}_bench_under_rocprof${sfx}_s518.json 2>/dev/null
  python3 $R/scripts/rocpd_stats.py $(find /tmp/ks -name "*.db" | head -1) 60 > $R/gpurun_out/${TAG}_kernel_stats${sfx}_b32_s518.csv
done
