#!/bin/bash
# usage: [VA_SHORT=1] va_pmc.sh <variant>...   SQ counters of vit_attn_var.py builds (separate rocprofv3 --pmc passes, kernel trace only)
# (SQ cycle counters tick once per 4 clocks; VALU_MFMA_BUSY_CYCLES is in clocks = 32 x MFMA count for the 32x32x16 form)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
GROUPS_=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC")
[ -z "$VA_SHORT" ] && GROUPS_+=("SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  "SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES SQ_INSTS_VALU_TRANS")
for v in "$@"; do
  for c in "${GROUPS_[@]}"; do
    rm -rf /tmp/vp; rocprofv3 --pmc $c --kernel-trace -d /tmp/vp -o vp -- python3 $R/scripts/ubench/vit_attn_var.py one $v > /tmp/vp.log 2>&1 || { echo "$v [$c]: failed: $(tail -2 /tmp/vp.log | tr '\n' ' ')"; continue; }
    python3 $R/scripts/rocpd_pmc.py $(find /tmp/vp -name '*.db' | head -1) vit_attn | tail -n +2 | sed "s/.*\",/$v: /"
  done
done
