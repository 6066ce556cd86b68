#!/bin/bash
# round 5, GPU call 2: safety tests with the calibrated bound, hand-over store variants of the headline kernel, FETCH / WRITE
# of the GRU wavefront with the L2-first-request build against the shipped one
set -u
root=${GRAFT_REPO_ROOT:-$PWD}; out=$root/gpurun_out; mkdir -p $out; cd $root
timeout 600 python -m pytest tests/test_hip_gru_safety.py -m gpu -q --timeout 240 -s 2>&1 | tail -80 > $out/r05b_safety.txt
for v in base h1 h2 base h2 h1; do
  lib=build/var/lib$v.so; [ $v = base ] && lib=wekws_amd/lib/libwekws_hip.so
  echo "== $v" >> $out/r05b_handover_ab.txt
  WEKWS_HIP_LIB=$root/$lib timeout 200 python tools/time_ds.py >> $out/r05b_handover_ab.txt 2>&1
done
cd /tmp && export TMPDIR=/tmp
for v in base gl1; do
  lib=build/var/lib$v.so; [ $v = base ] && lib=wekws_amd/lib/libwekws_hip.so
  for c in FETCH_SIZE WRITE_SIZE; do
    WEKWS_HIP_LIB=$root/$lib timeout 300 rocprofv3 --kernel-trace --pmc $c -d $out/prof_gru_${v}_$c -o pmc -- python $root/tools/probe/gru_one.py > $out/prof_gru_${v}_$c.log 2>&1
  done
done
cd $root
python tools/prof_summary.py $(find $out -path "*prof_gru_*" -name "*_results.db" | sort) > $out/r05b_gru_l2_traffic.txt 2>&1
rm -rf $out/prof_gru_*
tail -8 $out/r05b_safety.txt; cat $out/r05b_handover_ab.txt | grep -v amdgpu.ids | tail -30; grep -i "gru_pipe\|FETCH\|WRITE" $out/r05b_gru_l2_traffic.txt | head -20
