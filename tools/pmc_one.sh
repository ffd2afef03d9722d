#!/bin/bash
# Runs ON THE GPU BOX: SQ / LDS counters of one driver script.  usage: bash tools/pmc_one.sh <script.py> <kernel substring>
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc1 /tmp/pmc2
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU --output-format rocpd -d /tmp/pmc1 -o t -- python $REPO/$1 > /tmp/pmc1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA --output-format rocpd -d /tmp/pmc2 -o t -- python $REPO/$1 > /tmp/pmc2.log 2>&1
for d in /tmp/pmc1 /tmp/pmc2; do python $REPO/tools/rocpd_pmc.py $(find $d -name '*.db' | head -1) | grep -E "kernel|$2"; done
tail -3 /tmp/pmc2.log
