#!/bin/bash
# round 5, GPU call 1: the new safety tests first (their own log), the whole GPU suite, the GRU hand-over A/B, the bench line
set -u
root=${GRAFT_REPO_ROOT:-$PWD}; out=$root/gpurun_out; mkdir -p $out; cd $root
timeout 600 python -m pytest tests/test_hip_gru_safety.py -m gpu -q --timeout 240 -s 2>&1 | tail -60 > $out/r05a_safety.txt
timeout 900 python -m pytest tests -m gpu -q --timeout 300 --deselect tests/test_hip_gru_safety.py 2>&1 | tail -40 > $out/r05a_pytest_gpu.txt
cp $out/parity_errors.json $out/r05a_parity_errors.json 2>/dev/null
for v in base gl1 gl2; do
  lib=build/var/lib$v.so; [ $v = base ] && lib=wekws_amd/lib/libwekws_hip.so
  WEKWS_HIP_LIB=$root/$lib timeout 300 python tools/probe/gru_l2_ab.py $v >> $out/r05a_gru_l2_ab.jsonl 2>> $out/r05a_gru_l2_ab.err
done
timeout 600 python bench.py > $out/r05a_bench.json 2> $out/r05a_bench.err
tail -5 $out/r05a_safety.txt; tail -3 $out/r05a_pytest_gpu.txt; cat $out/r05a_gru_l2_ab.jsonl; head -c 600 $out/r05a_bench.json
