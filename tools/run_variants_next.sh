#!/bin/bash
# Developer tool (GPU box): tools/bench_next_rows.py for every named variant library.  bash tools/run_variants_next.sh name...
cd "$(dirname "$0")/.."
for n in "$@"; do echo "== $n"; HVLC_LIB=$PWD/huffman-gpu_amd/variants/libhvlc_$n.so timeout -k 10 200 python3 tools/bench_next_rows.py 2>&1 | tail -1 | cut -c40-260; done
