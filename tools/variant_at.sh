#!/bin/bash
# Developer tool: build huffman-gpu_amd/variants/libhvlc_<name>.so from the device / host sources of a git revision
# (a reference to time new code against in the same run).   tools/variant_at.sh <name> <rev> [-D flags...]
cd "$(dirname "$0")/.."
name=$1; rev=$2; shift 2
d=build_tmp/$name; rm -rf $d; mkdir -p $d/huffman-gpu_amd/csrc $d/include huffman-gpu_amd/variants
git show $rev:huffman-gpu_amd/csrc/hvlc_api.hip > $d/huffman-gpu_amd/csrc/hvlc_api.hip
git show $rev:huffman-gpu_amd/csrc/hvlc_device.hip.h > $d/huffman-gpu_amd/csrc/hvlc_device.hip.h
git show $rev:include/hvlc.h > $d/include/hvlc.h
python3 huffman-gpu_amd/hazard_pass.py -o huffman-gpu_amd/variants/libhvlc_$name.so $d/huffman-gpu_amd/csrc/hvlc_api.hip "$@"
