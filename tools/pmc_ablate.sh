#!/bin/bash
# Developer tool: SQ instruction counters of k_encode per ablation (experiment build, wrong outputs by design).
# usage (on the GPU box, repo root): bash tools/pmc_ablate.sh "0 1 5 7"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
python3 huffman-gpu_amd/hazard_pass.py -o huffman-gpu_amd/libhvlc_exp.so huffman-gpu_amd/csrc/hvlc_api.hip -DHVLC_EXPERIMENT -DHVLC_NO_CLOCK || exit 1
export HVLC_LIB=$PWD/huffman-gpu_amd/libhvlc_exp.so
OUT=gpurun_out/pmcab; mkdir -p $OUT
for dbg in ${1:-0 1 5 7}; do
  export HVLC_DBG=$dbg
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \
     --output-format csv -d $OUT/dbg$dbg -- python3 tools/perf_sweep.py --mib ${MIB:-1024} --bpc 0 --workloads ${WL:-H2.2} --iters 3 > $OUT/dbg$dbg.log 2>&1 || exit 1
done
python3 tools/pmc_summary.py $OUT
