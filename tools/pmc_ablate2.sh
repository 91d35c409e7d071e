#!/bin/bash
# Developer tool (GPU box): SQ instruction counters of k_encode per ablation of the experiment build (libhvlc_exp.so).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
export HVLC_LIB=$PWD/huffman-gpu_amd/libhvlc_exp.so
for dbg in ${DBGS:-0 2 4 6}; do
  OUT=gpurun_out/pmcabl/dbg$dbg; rm -rf $OUT; mkdir -p $OUT
  export HVLC_DBG=$dbg
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS \
     --output-format csv -d $OUT/p1 -- python3 tools/perf_sweep.py --mib ${MIB:-4096} --bpc 0 --workloads ${WL:-H2.2} --iters 3 > $OUT/p1.log 2>&1 || exit 1
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_BUSY_CYCLES \
     --output-format csv -d $OUT/p2 -- python3 tools/perf_sweep.py --mib ${MIB:-4096} --bpc 0 --workloads ${WL:-H2.2} --iters 3 > $OUT/p2.log 2>&1 || exit 1
  echo "== HVLC_DBG=$dbg"; python3 tools/pmc_summary.py $OUT
done
