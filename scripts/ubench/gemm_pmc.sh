#!/bin/bash
# L2 behaviour of the GEMM's LDS-DMA stream: TCC hit / miss / request counters per variant (separate rocprofv3 --pmc runs)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  for c in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
    rm -rf /tmp/gp; rocprofv3 --pmc $c --kernel-trace -d /tmp/gp -o gp -- python3 $R/scripts/ubench/gemm_pmc.py $v > /dev/null 2>&1
    python3 $R/scripts/rocpd_pmc.py $(find /tmp/gp -name '*.db' | head -1) gemm_bf16 | tail -n +2 | sed "s/.*\",/variant $v: /"
  done
done
