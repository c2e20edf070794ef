#!/bin/bash
# SQ counter passes over the decode kernels (tools/time_kernels.py): run on the GPU box through gpurun.
# usage: tools/pmc_sq.sh <tag> [frames]      -> gpurun_out/<tag>_pmc{1,2,3}/ ; summarise with tools/summarize_sq.py
set -e
tag=${1:-sq}
frames=${2:-160000}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA"; do
    i=$((i+1))
    rocprofv3 --pmc $set --kernel-trace -d $root/gpurun_out/${tag}_pmc$i -o run --output-format csv -- python3 $root/tools/time_kernels.py --frames $frames --steps 2 > $root/gpurun_out/${tag}_pmc$i.log 2>&1
done
