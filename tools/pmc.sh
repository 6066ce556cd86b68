#!/bin/bash
# rocprofv3 passes for one command (run on the GPU box, from the repo root): a kernel trace, then three PMC groups
# collected in their own runs (counters never share a run with the sys/hip/hsa trace domains).
#   tools/pmc.sh <tag> <command...>      -> gpurun_out/prof_<tag>_{trace,g1,g2,g3,mem}/
set -u
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
cmd="cd $root && $*"
rocprofv3 --kernel-trace --stats -d $out/prof_${tag}_trace -o t -- bash -c "$cmd" > $out/prof_${tag}_trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $out/prof_${tag}_g1 -o pmc -- bash -c "$cmd" > $out/prof_${tag}_g1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM -d $out/prof_${tag}_g2 -o pmc -- bash -c "$cmd" > $out/prof_${tag}_g2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_WAVE_CYCLES -d $out/prof_${tag}_g3 -o pmc -- bash -c "$cmd" > $out/prof_${tag}_g3.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/prof_${tag}_fetch -o pmc -- bash -c "$cmd" > $out/prof_${tag}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/prof_${tag}_write -o pmc -- bash -c "$cmd" > $out/prof_${tag}_write.log 2>&1
cd $root
find $out -name "*_results.db" | sort
