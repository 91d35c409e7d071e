#!/bin/bash
# Developer tool: ablation of k_encode phases with the -DHVLC_EXPERIMENT build (wrong outputs by design,
# timing only).  HVLC_DBG bits: 1 no look-back, 2 no stream stores, 4 no LDS packing, 16 no offsets.
cd "$(dirname "$0")/.."
python3 huffman-gpu_amd/hazard_pass.py -o huffman-gpu_amd/libhvlc_exp.so huffman-gpu_amd/csrc/hvlc_api.hip -DHVLC_EXPERIMENT $XFLAGS || exit 1
export HVLC_LIB=$PWD/huffman-gpu_amd/libhvlc_exp.so
for dbg in ${DBGS:-0 1 2 4 16 7 23}; do
  echo "== HVLC_DBG=$dbg"
  HVLC_DBG=$dbg python tools/perf_sweep.py --mib ${MIB:-1024} --bpc ${BPC:-1,4} --workloads ${WL:-H2.2} 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.rstrip()); continue
    print(d['workload'], 'bpc', d['info']['blocks_per_cu'], 'ms', d['ms_kernel'], 'in_GBs', d['in_GBs'], 'rc', d['rc'])
"
done
