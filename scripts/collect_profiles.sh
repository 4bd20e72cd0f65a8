#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 evidence for profiles/.  Kernel-trace statistics of the default
# bench command, then the HBM-traffic PMC passes (FETCH_SIZE and WRITE_SIZE in SEPARATE runs, kernel
# trace only) over the kernels bench.py prices.  Summaries land in gpurun_out/ for copying to profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
for img in 518 224; do
  rm -rf /tmp/ks$img
  rocprofv3 --kernel-trace --stats -d /tmp/ks$img -o ks -- python3 $R/bench.py --img $img --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-fp32 > $R/gpurun_out/${TAG}_bench_under_rocprof_s$img.json 2>/dev/null
  python3 $R/scripts/rocpd_stats.py $(find /tmp/ks$img -name "*.db" | head -1) 60 > $R/gpurun_out/${TAG}_kernel_stats_b32_s$img.csv
done
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o pmc -- python3 $R/scripts/pmc_kernels.py 32 > /dev/null 2>&1
  python3 $R/scripts/rocpd_pmc.py $(find /tmp/pmc_$c -name "*.db" | head -1) unopose > $R/gpurun_out/${TAG}_pmc_$(echo $c | tr A-Z a-z).csv
done
python3 $R/bench.py > $R/gpurun_out/${TAG}_bench_n1_bf16_s518.json 2>/dev/null
python3 $R/bench.py --img 224 > $R/gpurun_out/${TAG}_bench_n1_bf16_s224.json 2>/dev/null
tail -c 600 $R/gpurun_out/${TAG}_bench_n1_bf16_s518.json
