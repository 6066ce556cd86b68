#!/bin/bash
# Everything profiles/rNN* is made from, in one GPU session (run on the GPU box from the repo root, e.g. through gpurun):
#   tools/round_profile.sh r02b
# -> gpurun_out/<tag>_{pytest_gpu.txt,parity_errors.json,f16x3.txt,f32.txt,stream_kernels.txt,pmc_traffic.json,bench.json,
#    configs.jsonl}; the rocprofv3 databases are deleted at the end (gpurun merges at most 64 MiB back).
set -u
tag=${1:-r06a}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
mkdir -p $out
cd $root
python -m pytest tests -m gpu -q 2>&1 | grep -E "^E  |^FAILED|passed|failed" | cut -c1-700 | tail -40 > $out/${tag}_pytest_gpu.txt
cp $out/parity_errors.json $out/${tag}_parity_errors.json 2>/dev/null
# headline kernel, both precisions: kernel trace + PMC groups + FETCH_SIZE / WRITE_SIZE passes
tools/pmc.sh hl python bench.py --no-cpu-baseline --no-extras --steps 7 --warmup 2 > /dev/null
tools/pmc.sh hl32 python bench.py --no-cpu-baseline --no-extras --steps 7 --warmup 2 --precision f32 > /dev/null
tools/pmc.sh md python bench.py --model mdtc_h64 --no-cpu-baseline --no-extras --steps 7 --warmup 2 > /dev/null
tools/pmc.sh gru python bench.py --model gru_2x128 --no-cpu-baseline --no-extras --steps 7 --warmup 2 > /dev/null
tools/pmc.sh d64 python bench.py --model ds_tcn_h64 --no-cpu-baseline --no-extras --steps 7 --warmup 2 > /dev/null
tools/pmc.sh m32 python bench.py --model mdtc_small --no-cpu-baseline --no-extras --steps 7 --warmup 2 > /dev/null
tools/pmc.sh fb python tools/probe/run_fbank.py > /dev/null
python tools/prof_summary.py $(find $out -path "*prof_hl_*" -name "*_results.db" | sort) > $out/${tag}_ds_tcn_h256_f16x3.txt
python tools/prof_summary.py $(find $out -path "*prof_gru_*" -name "*_results.db" | sort) > $out/${tag}_gru_2x128_f16x3.txt
python tools/prof_summary.py $(find $out -path "*prof_md_*" -name "*_results.db" | sort) > $out/${tag}_mdtc_h64_f16x3.txt
python tools/prof_summary.py $(find $out -path "*prof_hl32_*" -name "*_results.db" | sort) > $out/${tag}_ds_tcn_h256_f32.txt
python tools/prof_summary.py $(find $out -path "*prof_d64_*" -name "*_results.db" | sort) > $out/${tag}_ds_tcn_h64_f16x3.txt
python tools/prof_summary.py $(find $out -path "*prof_m32_*" -name "*_results.db" | sort) > $out/${tag}_mdtc_small_f16x3.txt
python tools/prof_summary.py $(find $out -path "*prof_fb_*" -name "*_results.db" | sort) > $out/${tag}_fbank.txt
python tools/pmc_traffic.py ds_tcn_h256/B1024/f16x3=hl=profiles/${tag}_ds_tcn_h256_f16x3.txt ds_tcn_h256/B1024/f32=hl32=profiles/${tag}_ds_tcn_h256_f32.txt mdtc_h64/B1024/f16x3=md=profiles/${tag}_mdtc_h64_f16x3.txt gru_2x128/B1024/f16x3=gru=profiles/${tag}_gru_2x128_f16x3.txt ds_tcn_h64/B1024/f16x3=d64=profiles/${tag}_ds_tcn_h64_f16x3.txt mdtc_small/B1024/f16x3=m32=profiles/${tag}_mdtc_small_f16x3.txt > $out/${tag}_pmc_traffic.log 2>&1
cp profiles/pmc_traffic.json $out/${tag}_pmc_traffic.json
# streaming kernels: kernel trace of the many-streams sweep and of the GRU rows
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $out/prof_strm_a -o t -- bash -c "cd $root && python tools/bench_configs.py manystreams" > $out/prof_strm_a.log 2>&1)
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $out/prof_strm_b -o t -- bash -c "cd $root && python tools/bench_configs.py gru" > $out/prof_strm_b.log 2>&1)
python tools/prof_summary.py $(find $out -path "*prof_strm*" -name "*_results.db" | sort) > $out/${tag}_stream_kernels.txt
rm -rf $out/prof_*
# the bench line (reads the PMC traffic file written above: same library build) and the secondary configs
python bench.py --extras-out $out/${tag}_bench_extras.json 2> $out/${tag}_bench.err | tail -1 > $out/${tag}_bench.json
python tools/bench_configs.py > $out/${tag}_configs.jsonl 2> $out/${tag}_configs.err
ls -la $out | tail -20
