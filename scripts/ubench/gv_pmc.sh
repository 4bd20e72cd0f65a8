#!/bin/bash
# usage: gv_pmc.sh <shape> <variant>...   L2 counters of gemm_var.py builds (separate rocprofv3 --pmc passes, kernel trace only)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
SHAPE=$1; shift
for v in "$@"; do
  for c in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
    rm -rf /tmp/gp; rocprofv3 --pmc $c --kernel-trace -d /tmp/gp -o gp -- python3 $R/scripts/ubench/gv_pmc.py $v $SHAPE > /dev/null 2>&1
    python3 $R/scripts/rocpd_pmc.py $(find /tmp/gp -name '*.db' | head -1) gemm256 | tail -n +2 | sed "s/.*\",/$SHAPE $v: /"
  done
done
