#!/bin/bash
# Development tool (GPU box): instruction counters of fbank_kernel for the library in $1 (or the product) -> stdout
root=${GRAFT_REPO_ROOT:-$PWD}; out=$root/gpurun_out; lib=${1:-}
cd /tmp && export TMPDIR=/tmp
rm -rf $out/prof_fbx
WEKWS_HIP_LIB=$lib rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES -d $out/prof_fbx -o pmc -- bash -c "cd $root && python tools/probe/run_fbank.py" > /dev/null 2>&1
cd $root; python tools/prof_summary.py $(find $out/prof_fbx -name "*_results.db") | grep -E "SQ_|fbank_kernel" | grep -v "^void" | head -12
rm -rf $out/prof_fbx
