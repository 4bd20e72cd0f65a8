#!/bin/bash
# PMC passes over scripts/vit_attn_bench.py (run on the GPU box): issue/stall split and LDS conflicts.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d /tmp/p1 -o p1 -- python3 $R/scripts/vit_attn_bench.py 1374 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAVES --kernel-trace -d /tmp/p2 -o p2 -- python3 $R/scripts/vit_attn_bench.py 1374 > /dev/null 2>&1
for d in /tmp/p1 /tmp/p2; do f=$(find $d -name "*.db" | head -1); python3 $R/scripts/rocpd_pmc.py $f vit_attn | cut -d, -f2-; done
