#!/bin/bash
# usage: pmc_any.sh <kernel-name-pattern> <python script> [args]   (run on the GPU box)
# SQ issue/stall split + LDS counters of one kernel, two rocprofv3 --pmc passes (kernel trace only).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
PAT=$1; shift
rm -rf /tmp/q1 /tmp/q2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d /tmp/q1 -o p -- python3 $R/$@ > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAVES --kernel-trace -d /tmp/q2 -o p -- python3 $R/$@ > /dev/null 2>&1
for d in /tmp/q1 /tmp/q2; do python3 $R/scripts/rocpd_pmc.py $(find $d -name "*.db" | head -1) "$PAT" | cut -d, -f2- | sed 's/^.*)",//'; done
