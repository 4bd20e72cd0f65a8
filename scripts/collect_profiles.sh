#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 evidence for profiles/.
#  1. kernel-trace statistics of the default bench command (two forwards in flight) and of the same command with
#     --inflight 1 (kernels isolated: no co-scheduled stream stretching their durations), at 518 and 224;
#  2. HBM-traffic PMC passes (FETCH_SIZE and WRITE_SIZE in SEPARATE runs, kernel trace only) over the kernels bench.py prices
#     (scripts/pmc_kernels.py), over the four ViT linears of the `roofline` object (scripts/pmc_gemm.py, per dispatch) and over
#     the FETCH_SIZE calibration reads (scripts/ubench/fetch_calib.py);
#  3. the bench lines themselves.
# Summaries land in gpurun_out/ for copying to profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
for img in 518 224; do
  for fl in 2 1; do
    sfx=$([ $fl = 1 ] && echo _isolated || echo "")
    rm -rf /tmp/ks
    rocprofv3 --kernel-trace --stats -d /tmp/ks -o ks -- python3 $R/bench.py --img $img --steps 6 --warmup 2 --inflight $fl --no-cpu-baseline --no-roofline --no-fp32 --no-extra > $R/gpurun_out/${TAG}_bench_under_rocprof${sfx}_s$img.json 2>/dev/null
    python3 $R/scripts/rocpd_stats.py $(find /tmp/ks -name "*.db" | head -1) 60 > $R/gpurun_out/${TAG}_kernel_stats${sfx}_b32_s$img.csv
  done
done
for c in FETCH_SIZE WRITE_SIZE; do
  lc=$(echo $c | tr A-Z a-z)
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o pmc -- python3 $R/scripts/pmc_kernels.py 32 > /dev/null 2>&1
  python3 $R/scripts/rocpd_pmc.py $(find /tmp/pmc_$c -name "*.db" | head -1) unopose > $R/gpurun_out/${TAG}_pmc_$lc.csv
  rm -rf /tmp/pmcg_$c
  rocprofv3 --pmc $c --kernel-trace -d /tmp/pmcg_$c -o pmc -- python3 $R/scripts/pmc_gemm.py 32 > /dev/null 2>&1
  python3 $R/scripts/rocpd_pmc.py $(find /tmp/pmcg_$c -name "*.db" | head -1) gemm --dispatches > $R/gpurun_out/${TAG}_pmc_gemm_$lc.csv
done
python3 $R/scripts/ubench/dma_rate.py build
rm -rf /tmp/pmc_cal
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc_cal -o pmc -- python3 $R/scripts/ubench/fetch_calib.py > /dev/null 2>&1
python3 $R/scripts/rocpd_pmc.py $(find /tmp/pmc_cal -name "*.db" | head -1) dma_rate > $R/gpurun_out/${TAG}_pmc_fetch_calibration.csv
python3 $R/scripts/pmc_summary.py $R/gpurun_out/${TAG}_pmc_fetch_size.csv $R/gpurun_out/${TAG}_pmc_write_size.csv $R/gpurun_out/${TAG}_pmc_summary.json \
  $R/gpurun_out/${TAG}_pmc_gemm_fetch_size.csv $R/gpurun_out/${TAG}_pmc_gemm_write_size.csv $R/gpurun_out/${TAG}_pmc_fetch_calibration.csv
mkdir -p $R/profiles && cp $R/gpurun_out/${TAG}_pmc_summary.json $R/profiles/   # bench.py reads the traffic from profiles/
python3 $R/bench.py > $R/gpurun_out/${TAG}_bench_n1_bf16_s518.json 2> $R/gpurun_out/${TAG}_bench_n1_bf16_s518.log
python3 $R/bench.py --img 224 > $R/gpurun_out/${TAG}_bench_n1_bf16_s224.json 2>/dev/null
tail -c 1500 $R/gpurun_out/${TAG}_bench_n1_bf16_s518.json
# round 6: the training step (kernel statistics + the bench line of `--train`)
rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python3 $R/bench.py --train --steps 3 --warmup 2 > /dev/null 2>&1
python3 $R/scripts/rocpd_stats.py $(find /tmp/kt -name "*.db" | head -1) 80 > $R/gpurun_out/${TAG}_train_kernel_stats.csv
python3 $R/bench.py --train --steps 10 --warmup 3 > $R/gpurun_out/${TAG}_train_n1_fp32_4096.json 2>/dev/null
python3 $R/bench.py --train --train-pts 2048 --steps 10 --warmup 3 > $R/gpurun_out/${TAG}_train_n1_fp32_2048.json 2>/dev/null
