#!/bin/bash
# The round's LAST session when only host code / one side kernel changed after the full one (tools/round_profile.sh): the whole
# GPU suite, the kernel-trace + FETCH_SIZE / WRITE_SIZE passes of every bench row (profiles/pmc_traffic.json is keyed by the
# library hash, so all rows are taken again), the bench line.  The instruction-mix groups, the streaming-kernel traces and the
# secondary configs of the full session stay valid: those kernels are byte-identical.
#   tools/round_profile_final.sh r05k
set -u
tag=${1:-r06b}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
mkdir -p $out
cd $root
python -m pytest tests -m gpu -q 2>&1 | grep -E "^E  |^FAILED|passed|failed" | cut -c1-700 | tail -40 > $out/${tag}_pytest_gpu.txt
cp $out/parity_errors.json $out/${tag}_parity_errors.json 2>/dev/null
tools/pmc_min.sh hl python bench.py --no-cpu-baseline --no-extras --steps 7 --warmup 2 > /dev/null
tools/pmc_min.sh hl32 python bench.py --no-cpu-baseline --no-extras --steps 7 --warmup 2 --precision f32 > /dev/null
tools/pmc_min.sh md python bench.py --model mdtc_h64 --no-cpu-baseline --no-extras --steps 7 --warmup 2 > /dev/null
tools/pmc_min.sh gru python bench.py --model gru_2x128 --no-cpu-baseline --no-extras --steps 7 --warmup 2 > /dev/null
tools/pmc_min.sh d64 python bench.py --model ds_tcn_h64 --no-cpu-baseline --no-extras --steps 7 --warmup 2 > /dev/null
tools/pmc_min.sh m32 python bench.py --model mdtc_small --no-cpu-baseline --no-extras --steps 7 --warmup 2 > /dev/null
python tools/prof_summary.py $(find $out -path "*prof_*" -name "*_results.db" | sort) > $out/${tag}_traffic_passes.txt
p=profiles/${tag}_traffic_passes.txt
python tools/pmc_traffic.py ds_tcn_h256/B1024/f16x3=hl=$p ds_tcn_h256/B1024/f32=hl32=$p mdtc_h64/B1024/f16x3=md=$p gru_2x128/B1024/f16x3=gru=$p ds_tcn_h64/B1024/f16x3=d64=$p mdtc_small/B1024/f16x3=m32=$p > $out/${tag}_pmc_traffic.log 2>&1
cp profiles/pmc_traffic.json $out/${tag}_pmc_traffic.json
rm -rf $out/prof_*
python bench.py --extras-out $out/${tag}_bench_extras.json 2> $out/${tag}_bench.err | tail -1 > $out/${tag}_bench.json
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/${tag}_smoke.txt 2>&1
ls -la $out | tail -12
