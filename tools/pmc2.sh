#!/bin/bash
# Extra PMC groups (latency levels, L2 / TA stalls) for one command; see tools/pmc.sh.   tools/pmc2.sh <tag> <command...>
set -u
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
cmd="cd $root && $*"
rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_LEVEL_WAVES SQ_INSTS_VMEM_RD -d $out/prof_${tag}_h1 -o pmc -- bash -c "$cmd" > $out/prof_${tag}_h1.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum -d $out/prof_${tag}_h2 -o pmc -- bash -c "$cmd" > $out/prof_${tag}_h2.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum -d $out/prof_${tag}_h3 -o pmc -- bash -c "$cmd" > $out/prof_${tag}_h3.log 2>&1
cd $root
find $out -name "*_results.db" | grep "_h[0-9]" | sort
