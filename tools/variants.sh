#!/bin/bash
# Developer tool: build variants of libhvlc.so with extra -D flags (cross-compiles without a GPU).
#   tools/variants.sh name1:"-DX=1 -DY=2" name2:"-DZ" ...     -> huffman-gpu_amd/variants/libhvlc_<name>.so
# Run them on the GPU box with tools/run_variants.sh.
cd "$(dirname "$0")/.."
mkdir -p huffman-gpu_amd/variants
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"; [ "$flags" = "$spec" ] && flags=""
  python3 huffman-gpu_amd/hazard_pass.py -o huffman-gpu_amd/variants/libhvlc_$name.so huffman-gpu_amd/csrc/hvlc_api.hip $flags &
done
wait
ls -la huffman-gpu_amd/variants/
