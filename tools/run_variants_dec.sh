#!/bin/bash
# Developer tool (GPU box): time hvlc_decode / hvlc_merge_shard with every huffman-gpu_amd/variants/libhvlc_*.so.
cd "$(dirname "$0")/.."
names="$@"; [ -z "$names" ] && names=$(ls huffman-gpu_amd/variants/ | sed -n 's/^libhvlc_\(.*\)\.so$/\1/p')
for n in $names; do
  echo "== $n"
  HVLC_LIB=$PWD/huffman-gpu_amd/variants/libhvlc_$n.so timeout -k 10 120 python3 tools/bench_next_rows.py ${MIB:-4096} 2>&1 | grep -v amdgpu.ids | cut -c1-300
done
