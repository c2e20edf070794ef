#!/bin/bash
# SQ counter passes over a bench.py workload: run on the GPU box through gpurun.
# usage: tools/pmc_sq_bench.sh <tag> <bench.py arguments...>   -> gpurun_out/<tag>_pmc{1,2,3}/ ; summarise with tools/summarize_sq.py
set -e
tag=$1
shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA"; do
    i=$((i+1))
    rocprofv3 --pmc $set --kernel-trace -d $root/gpurun_out/${tag}_pmc$i -o run --output-format csv -- python3 $root/bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $root/gpurun_out/${tag}_pmc$i.log 2>&1
done
