#!/bin/bash
# Developer tool (GPU box): time every huffman-gpu_amd/variants/libhvlc_*.so with tools/perf_sweep.py.
#   MIB=4096 WL=H2.2,H4.0,H1.0 ITERS=10 bash tools/run_variants.sh [names...]
cd "$(dirname "$0")/.."
names="$@"; [ -z "$names" ] && names=$(ls huffman-gpu_amd/variants/ | sed -n 's/^libhvlc_\(.*\)\.so$/\1/p')
for n in $names; do
  echo "== $n"
  HVLC_LIB=$PWD/huffman-gpu_amd/variants/libhvlc_$n.so timeout -k 10 120 python3 tools/perf_sweep.py --mib ${MIB:-4096} --workloads ${WL:-H2.2} --iters ${ITERS:-10} 2>&1 | grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.rstrip()); continue
    print(' ', d['workload'], 'ms', d['ms_kernel'], 'min', d['ms_kernel_min'], 'in_GBs', d['in_GBs'], 'rc', d['rc'], 'ok', d['bits_ok'])
"
done
