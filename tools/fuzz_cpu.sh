#!/bin/bash
# The CPU-side fuzzers in one go (BUILD CONTAINER: they import the live reference from /root/reference; ~6 min):
#   oracle vs the live reference model on random configurations; the reference exporter's output through both model-file readers
#   (Python reader -> oracle forward vs the reference's outputs; C++ reader == Python reader bit for bit); the splice oracle against the
#   reference's own context_expansion / frame_skip on random shapes incl. utterances no longer than their context.
#   tools/fuzz_cpu.sh [seeds] [exports]
set -e
cd "$(dirname "$0")/.."
make -C runtime > /dev/null
export PYTHONPATH=/root/reference:$PWD
python tools/probe/fuzz_oracle_vs_reference.py "${1:-20}" 2>&1 | grep -v Warning | tail -3
python tools/probe/fuzz_onnx_reader.py "${2:-80}" 2>&1 | grep -v Warning | tail -3
python tools/probe/fuzz_splice_vs_reference.py "${3:-20000}" 2>&1 | grep -v Warning | tail -2
