#!/bin/bash
# The three rocprofv3 passes profiles/pmc_traffic.json is made from (tools/pmc.sh without the instruction-mix groups):
#   tools/pmc_min.sh <tag> <command...>      -> gpurun_out/prof_<tag>_{trace,fetch,write}/
set -u
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
cmd="cd $root && $*"
rocprofv3 --kernel-trace --stats -d $out/prof_${tag}_trace -o t -- bash -c "$cmd" > $out/prof_${tag}_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/prof_${tag}_fetch -o pmc -- bash -c "$cmd" > $out/prof_${tag}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/prof_${tag}_write -o pmc -- bash -c "$cmd" > $out/prof_${tag}_write.log 2>&1
cd $root
