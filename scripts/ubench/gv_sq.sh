#!/bin/bash
# usage: gv_sq.sh <shape> <variant>...   SQ issue / stall split of gemm_var.py builds (two rocprofv3 --pmc passes each)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
SHAPE=$1; shift
for v in "$@"; do
  for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA"; do
    rm -rf /tmp/gp; rocprofv3 --pmc $c --kernel-trace -d /tmp/gp -o gp -- python3 $R/scripts/ubench/gv_pmc.py $v $SHAPE > /tmp/gp.log 2>&1 || tail -3 /tmp/gp.log
    python3 $R/scripts/rocpd_pmc.py $(find /tmp/gp -name '*.db' | head -1) gemm_bf16 | tail -n +2 | sed "s/.*\",/$SHAPE $v: /"
  done
done
